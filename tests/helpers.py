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
FIXTURE_NAMES = ("c1_det", "c1_perturb", "odd_orbit", "w1_det", "shipped_sizes")


def load_fixture(name: str):
    """-> (fx: dict of scalars, data: dict of torch tensors)."""
    raw = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
    fx = ast.literal_eval(str(raw["fixture"]))
    data = {k: torch.from_numpy(raw[k]) for k in raw.files if k != "fixture"}
    return fx, data


def build_model(fx: dict, data: Dict[str, torch.Tensor], device="cpu") -> "vf_nerf_amd.VectorFieldNerf":
    """Rebuild the fixture's weights with this repo's own modules (seed + gain + stored head rows) and check the
    fingerprint recorded when the reference produced the golden outputs."""
    torch.manual_seed(fx["seed"])
    cfg = vf_nerf_amd.shipped_config(torch.device("cpu"), n_samples=fx["n_samples"], n_importance=fx["n_importance"],
                                     perturb=fx["perturb"], near=fx["near"], far=fx["far"],
                                     fine_range=fx["fine_range"], dir_to_normal_th=fx["th"], n_window=fx["n_window"])
    cfg.numerical_jacobian = bool(fx.get("numjac", False))
    model = vf_nerf_amd.VectorFieldNerf(cfg)
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


def oracle_settings(fx: dict):
    from oracle import vfnerf_oracle as O
    return O.RenderSettings(numerical_jacobian=bool(fx.get("numjac", False)), train_mode=bool(fx.get("train", False)), n_samples=fx["n_samples"], n_fine=fx["n_importance"], near=fx["near"], far=fx["far"],
                            fine_range=fx["fine_range"], perturb=fx["perturb"], n_window=fx["n_window"],
                            dir_to_normal_th=fx["th"], normalize=True,
                            density=O.DensityParams(beta=0.5, mean=0.7, scale=100.0, beta_bounds=(1e-4, 1e9),
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
    beta = torch.tensor(0.5, requires_grad=True)
    mean = torch.tensor(0.7, requires_grad=True)
    scale = torch.tensor(100.0, requires_grad=True)
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
