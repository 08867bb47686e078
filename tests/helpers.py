"""Shared test helpers: golden fixtures and model construction from a fixture's recipe."""
from __future__ import annotations

import ast
import os
from typing import Dict

import numpy as np
import torch

import vf_nerf_amd
from vf_nerf_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# trained_256*: weights the reference's own trainer arrived at after 1200 optimizer steps (tests/golden/make_trained_golden.py);
# every other fixture holds weights of the synthetic init family (seed + default init x gain + recentred vector head)
FIXTURE_NAMES = ("c1_det", "c1_perturb", "odd_orbit", "w1_det", "shipped_sizes", "bench_sizes", "trained_256", "trained_256_shipped",
                 "trained_far")


def load_fixture(name: str):
    """-> (fx: dict of scalars, data: dict of torch tensors)."""
    raw = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    fx = ast.literal_eval(str(raw["fixture"]))
    fx["name"] = name
    data = {k: torch.from_numpy(raw[k]) for k in raw.files if raw[k].dtype.kind not in "US"}
    return fx, data


def build_model(fx: dict, data: Dict[str, torch.Tensor], device="cpu") -> "vf_nerf_amd.VectorFieldNerf":
    """Rebuild the fixture's weights with this repo's own modules (seed + gain + stored head rows) and check the
    fingerprint recorded when the reference produced the golden outputs."""
    torch.manual_seed(fx["seed"])
    cfg = vf_nerf_amd.shipped_config(torch.device("cpu"), n_samples=fx["n_samples"], n_importance=fx["n_importance"],
                                     perturb=fx["perturb"], near=fx["near"], far=fx["far"],
                                     fine_range=fx["fine_range"], dir_to_normal_th=fx["th"], n_window=fx["n_window"])
    cfg.numerical_jacobian = bool(fx.get("numjac", False))
    cfg.rendering_net_config.detach_normals = bool(fx.get("detach_normals", True))
    model = vf_nerf_amd.VectorFieldNerf(cfg)
    if fx.get("trained"):
        # trained state: the weights are stored (in this fixture or in the one it names)
        src = data if "weights_in" not in fx else load_fixture(fx["weights_in"])[1]
        for tag, mod in (("vf", model.vector_field_network), ("rn", model.rendering_network), ("density", model.density)):
            sd = {k[len(f"w.{tag}."):]: v for k, v in src.items() if k.startswith(f"w.{tag}.")}
            mod.load_state_dict(sd)
    else:
        synthetic.scale_hidden_weights(model.vector_field_network, model.rendering_network, fx["gain"])
        with torch.no_grad():
            last = model.vector_field_network.layers[8]
            last.weight[:3] = data["head_weight"]
            last.bias[:3] = data["head_bias"]
    chk = synthetic.weights_checksum({"vf": model.vector_field_network.state_dict(),
                                      "rn": model.rendering_network.state_dict(),
                                      "density": model.density.state_dict()})
    want = data["weights_checksum"].tolist()
    got = [chk["sum"], chk["abs_sum"], chk["count"]]
    assert got[2] == want[2] and abs(got[0] - want[0]) <= 1e-9 * abs(want[1]) and \
        abs(got[1] - want[1]) <= 1e-9 * abs(want[1]), f"weights rebuilt from the recipe differ from the fixture: {got} vs {want}"
    model.eval()
    if device != "cpu":
        dev = torch.device(device)
        model.to(dev)
        model.config.cuda_config.device = dev
        model.config.cos_sim_weights = model.config.cos_sim_weights.to(dev)
    if "far_per_ray" in data:
        model.ray_sampler.far = data["far_per_ray"].to(device)
        model.fine_sampler.far = data["far_per_ray"].to(device)
    return model


_SCALARS = {}


def density_scalars(fx: dict):
    """(beta, mean, scale) as stored parameters: the shipped initial values, or the trained fixture's."""
    if not fx.get("trained"):
        return 0.5, 0.7, 100.0
    src = fx.get("weights_in", fx.get("name", "trained_256"))       # a trained fixture holds its own weights unless it names another's
    if src not in _SCALARS:
        raw = np.load(os.path.join(GOLDEN_DIR, f"{src}.npz"))
        _SCALARS[src] = tuple(float(raw[f"w.density.{k}"].reshape(-1)[0]) for k in ("beta", "mean", "scale"))
    return _SCALARS[src]


def oracle_settings(fx: dict):
    from oracle import vfnerf_oracle as O
    beta0, mean0, scale0 = density_scalars(fx)
    return O.RenderSettings(detach_normals=bool(fx.get("detach_normals", True)), numerical_jacobian=bool(fx.get("numjac", False)), train_mode=bool(fx.get("train", False)), n_samples=fx["n_samples"], n_fine=fx["n_importance"], near=fx["near"], far=fx["far"],
                            fine_range=fx["fine_range"], perturb=fx["perturb"], n_window=fx["n_window"],
                            dir_to_normal_th=fx["th"], normalize=True,
                            density=O.DensityParams(beta=beta0, mean=mean0, scale=scale0, beta_bounds=(1e-4, 1e9),
                                                    mean_bounds=(0.6, 1.0), scale_min=1.0))


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max(1, max|b|): the '1e-4 rel fp32' yardstick, robust to near-zero entries."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


GRAD_KEYS = (("vf", "layers.0.0.weight"), ("vf", "layers.0.1.weight"), ("vf", "layers.3.0.bias"),
             ("vf", "layers.4.0.weight"), ("vf", "layers.4.1.bias"), ("vf", "layers.7.1.weight"),
             ("vf", "layers.8.weight"), ("vf", "layers.8.bias"), ("rn", "layers.0.0.weight"),
             ("rn", "layers.2.1.weight"), ("rn", "layers.4.weight"), ("rn", "layers.4.bias"))


def loss_coefficients(n, s_t):
    """The fixed linear functional of (rgb, depth, normals) used when the reference gradients were captured
    (tests/golden/make_golden.py:loss_coefficients)."""
    g = torch.Generator().manual_seed(4242)
    return (torch.randn(n, 3, generator=g), torch.randn(n, 1, generator=g), 0.05 * torch.randn(n, s_t, 3, generator=g))


def grad_rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b| (gradient tensors span orders of magnitude across layers; scale by the tensor's own max)."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def oracle_gradients(fx, d, model, masks=None):
    """Autograd of the CPU oracle for the same functional -> {name: grad} for every parameter + density scalars.
    ``masks``: ReLU masks of the fine pass (8 VF + 4 rendering layers, [M, width] bool) to pin, see oracle._relu."""
    from oracle import vfnerf_oracle as O
    vf_sd = {k: v.detach().cpu().clone() for k, v in model.vector_field_network.state_dict().items()}
    rn_sd = {k: v.detach().cpu().clone() for k, v in model.rendering_network.state_dict().items()}
    leaves = {}
    for tag, net, sd in (("vf", model.vector_field_network, vf_sd), ("rn", model.rendering_network, rn_sd)):
        for name, _ in net.named_parameters():
            sd[name].requires_grad_(True)
            leaves[f"{tag}.{name}"] = sd[name]
    beta0, mean0, scale0 = density_scalars(fx)
    beta = torch.tensor(beta0, requires_grad=True)
    mean = torch.tensor(mean0, requires_grad=True)
    scale = torch.tensor(scale0, requires_grad=True)
    hidden = []
    out = O.render(d["uv"], d["pose"], d["intrinsics"], vf_sd, rn_sd, oracle_settings(fx), u_coarse=d.get("u_coarse"),
                   u_fine=d.get("u_fine"), u_add=d["u_add"], far=d.get("far_per_ray"), beta=beta, mean=mean, scale=scale,
                   hidden=hidden, masks=None if masks is None else list(masks))
    a, b, c = loss_coefficients(*d["z_vals"].shape)
    loss = (out["rgb"] * a).sum() + (out["depth"] * b).sum() + (out["normals"] * c).sum()
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    grads.update({"density.beta": beta.grad, "density.mean": mean.grad, "density.scale": scale.grad})
    grads["_hidden"] = hidden     # 8 VF + 4 rendering post-ReLU activations [M, width]
    grads["_out"] = out           # every stage of the oracle's forward
    grads["_state"] = {"vf": vf_sd, "rn": rn_sd}      # train mode: running statistics advanced in place
    return float(loss), grads


# ------------------------------------------------------------------------------------------------
# trainer fixture (tests/golden/make_train_golden.py: outputs of the reference's own train_epoch / VFLoss / samplers)
# ------------------------------------------------------------------------------------------------
TRAINER_WATCH = (("vf", "layers.0.0.weight", None), ("vf", "layers.0.1.weight", None), ("vf", "layers.2.0.weight", 8),
                 ("vf", "layers.4.1.bias", None), ("vf", "layers.8.weight", "head"), ("vf", "layers.8.bias", None),
                 ("rn", "layers.0.0.weight", 8), ("rn", "layers.1.1.weight", None), ("rn", "layers.4.weight", None),
                 ("rn", "layers.4.bias", None))


def load_trainer_fixture():
    return load_fixture("trainer_steps")


def trainer_batches(fx, d, device="cpu"):
    out = []
    for t in range(fx["steps"]):
        b = {k: d[f"s{t}.{k}"] for k in ("uv", "pose", "intrinsics", "rgb_gt", "depth_gt", "u_coarse", "u_fine", "u_add",
                                        "border_draws", "center_draws", "border_u", "center_u")}
        out.append({k: v.to(device) for k, v in b.items()})
    return out


def watched_slice(p: torch.Tensor, how):
    if how == "head":
        return p[:3]
    if isinstance(how, int):
        return p[::how, ::how]
    return p


def trainer_loss_weights(fx):
    from oracle import vfnerf_oracle as O
    return O.LossWeights()       # the shipped weights / thresholds (confs/vf_nerf.conf:74-90), which the fixture ran with


def lr_gamma(fx) -> float:
    return 0.1 ** (1.0 / fx["lr_decay_steps"])     # vector_field_nerf.py:65-67


def narrow_checkpoint_model(d, device="cpu"):
    """The facade with the geometry of the reference-written checkpoint fixture (tests/golden/ref_checkpoint_latest.pth;
    recipe stored beside it by make_train_golden.py)."""
    import numpy as np
    raw = np.load(os.path.join(GOLDEN_DIR, "trainer_steps.npz"))
    c = ast.literal_eval(str(raw["ckpt.recipe"]))
    cfg = vf_nerf_amd.shipped_config(torch.device(device), n_samples=c["n_samples"], n_importance=c["n_importance"], perturb=True,
                                     near=c["near"], far=c["far"], fine_range=c["fine_range"], dir_to_normal_th=c["th"],
                                     n_window=c["n_window"])
    cfg.vf_net_config.dimensions = list(c["vf_dims"])
    cfg.vf_net_config.skip_connection_in = list(c["vf_skip"])
    cfg.vf_net_config.feature_vector_dims = c["feat"]
    cfg.rendering_net_config.dimensions = list(c["rn_dims"])
    cfg.rendering_net_config.feature_vector_dims = c["feat"]
    model = vf_nerf_amd.VectorFieldNerf(cfg)
    model.eval()
    return model, c


REF_CHECKPOINT = os.path.join(GOLDEN_DIR, "ref_checkpoint_latest.pth")
