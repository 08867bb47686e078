#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference, read-only).  It imports the reference's
``VectorFieldNerf`` on CPU, records every ``torch.rand`` draw, captures per-stage tensors with hooks and
writes small ``.npz`` fixtures.  Nothing of the reference's source travels: the fixtures hold inputs and
expected outputs only, plus the seed/recipe needed to rebuild the weights with this repo's own modules
(verified here to be bit-identical to the reference's initialisation) and a checksum of those weights.

    python tests/golden/make_golden.py            # regenerates every fixture
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.path.insert(0, REPO)
from vf_nerf_amd import synthetic  # noqa: E402  (this repo's recipe code, shared with tests/bench)
import vf_nerf_amd  # noqa: E402

sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # utils/pinhole_model.py imports cv2 at top (Q14)
from config_parser import vf_nerf_config as rcfg  # noqa: E402
from models.nerf.vector_field_nerf import VectorFieldNerf as RefNerf  # noqa: E402
import models.helpers.functions as ref_functions  # noqa: E402
import utils.rendering as ref_rendering  # noqa: E402

CPU = torch.device("cpu")

FIXTURES = {
    # name: dict(...)
    "c1_det": dict(seed=0, gain=2.0, n_rays=48, n_samples=32, n_importance=32, perturb=False, th=-2.0, n_window=11,
                   near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=1, pose="identity",
                   skew=0.0, far_per_ray=False),
    "c1_perturb": dict(seed=0, gain=2.0, n_rays=48, n_samples=32, n_importance=32, perturb=True, th=-0.2, n_window=11,
                       near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=2,
                       pose="identity", skew=0.0, far_per_ray=False),
    "odd_orbit": dict(seed=3, gain=2.0, n_rays=24, n_samples=24, n_importance=17, perturb=True, th=-0.2, n_window=5,
                      near=0.1, far=1.5, fine_range=0.25, width=80, height=60, focal=70.0, cam_seed=5,
                      pose="orbit", skew=0.5, far_per_ray=True),
    # numerical_jacobian=True (vector_field_nerf.py:258-262,299-301,500-526): 6 extra VF forwards per pass
    "numjac_det": dict(seed=5, gain=2.0, n_rays=10, n_samples=14, n_importance=9, perturb=False, th=-0.2, n_window=5,
                       near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=9, pose="identity",
                       skew=0.0, far_per_ray=False, numjac=True),
    # networks in TRAIN mode (model.train(), vector_field_nerf.py:139-150): batch-statistics BatchNorm in both nets, the VF
    # forward appends the three autograd.grad rows, analytic directional derivatives (:260-261,303-305,476-498)
    "train_mode": dict(seed=6, gain=2.0, n_rays=12, n_samples=16, n_importance=10, perturb=True, th=-0.2, n_window=5,
                       near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=11, pose="identity",
                       skew=0.0, far_per_ray=False, train=True),
    # the shipped sampler sizes (confs/vf_nerf.conf:56-66: 100 proposal samples, n_importance 30 grown by 5 at epoch 0 -> 35, Q16)
    "shipped_sizes": dict(seed=8, gain=2.0, n_rays=10, n_samples=100, n_importance=35, perturb=True, th=-2.0, n_window=11,
                          near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=13, pose="orbit",
                          skew=0.0, far_per_ray=False),
    # detach_normals=False (rendering_network.py:76-77; not the shipped value): the colours' gradient also reaches the normals
    "attached_normals": dict(seed=9, gain=2.0, n_rays=12, n_samples=16, n_importance=10, perturb=True, th=-0.2, n_window=5,
                             near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=15, pose="identity",
                             skew=0.0, far_per_ray=False, detach_normals=False),
    # the sampler sizes of BASELINE.json's headline (64 proposal + 64 fine samples, stratified), a proposal block of whole groups of
    # 32 points (the training render then evaluates the VF net once per distinct sample, backward.StoredFinePass)
    "bench_sizes": dict(seed=11, gain=2.0, n_rays=96, n_samples=64, n_importance=64, perturb=True, th=-0.2, n_window=11,
                        near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=17, pose="orbit",
                        skew=0.0, far_per_ray=False),
    # an epoch past anneal_start with the shipped "hard" annealing: the reference then rewrites config.cos_sim_weights every call
    # (vector_field_nerf.py:232-234), which get_density ignores.  (white=True cannot be captured: the reference
    # raises UnboundLocalError in its coarse block, Q12.)
    "anneal_epoch": dict(seed=12, gain=2.0, n_rays=20, n_samples=24, n_importance=16, perturb=True, th=-0.2, n_window=11,
                         near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=19, pose="orbit",
                         skew=0.0, far_per_ray=False, epoch=1000),
    "w1_det": dict(seed=4, gain=2.0, n_rays=16, n_samples=20, n_importance=12, perturb=False, th=-2.0, n_window=1,
                   near=0.0, far=1.0, fine_range=0.3, width=64, height=64, focal=60.0, cam_seed=7, pose="identity",
                   skew=0.0, far_per_ray=False),
}


def ref_config(fx) -> "rcfg.VFNerfConfig":
    """confs/vf_nerf.conf network/density values, fixture-specific sampler sizes (pyhocon is absent)."""
    return rcfg.VFNerfConfig(
        vf_net_config=rcfg.VFNetConfig(input_dims=3, output_dims=3, dimensions=[256] * 8, feature_vector_dims=256,
                                       embedder_multires=6, weight_norm=False, batch_norm=True,
                                       skip_connection_in=[4], bias_init=0.0, dropout=False, dropout_probability=0.2,
                                       xavier_init=False, init=""),
        rendering_net_config=rcfg.RenderingNetConfig(output_dims=3, dimensions=[256] * 4, feature_vector_dims=256,
                                                     weight_norm=False, batch_norm=True, mode="idr",
                                                     embedder_multires=4, detach_normals=bool(fx.get("detach_normals", True))),
        ray_sampler_config=rcfg.RaySamplerConfig(n_samples=fx["n_samples"], n_importance=fx["n_importance"],
                                                 rays_per_batch=1024, perturb=fx["perturb"], near=fx["near"],
                                                 far=fx["far"], fine_range=fx["fine_range"], increase_every=50,
                                                 max_samples=100),
        cuda_config=rcfg.CudaConfig(device=CPU, num_gpus=0),
        scheduler_config=rcfg.SchedulerConfig(lr=5e-4, lr_decay_factor=0.1, clip_norm=0.5, weight_decay=0.0),
        density_config=rcfg.DensityConfig(beta_bounds=[1e-4, 1e9], mean_bounds=[0.6, 1.0], scale_min=1.0,
                                          params_init={'beta': 0.5, 'scale': 100.0, 'mean': 0.7}, cutoff=-2.0),
        cos_sim_weights=[0.09] * fx["n_window"], cos_sim_weights_anneal="hard", anneal_start=700, anneal_end=1400,
        rendering="volsdf", normalize_rendering=True, dir_to_normal_th=fx["th"],
        numerical_jacobian=bool(fx.get("numjac", False)),
        border_supervision=True, center_supervision=True)


def build_reference_model(fx):
    torch.manual_seed(fx["seed"])
    model = RefNerf(ref_config(fx))
    model.eval()
    synthetic.scale_hidden_weights(model.vector_field_network, model.rendering_network, fx["gain"])
    # pre-tanh statistics of the 3 vector columns over the frustum, via a hook on the last Linear
    pts = synthetic.frustum_points(20000, seed=1234, near=fx["near"], far=fx["far"])
    grabbed = {}
    h = model.vector_field_network.layers[8].register_forward_hook(lambda m, i, o: grabbed.__setitem__("pre", o.detach()))
    with torch.no_grad():
        model.vector_field_network(pts)
    h.remove()
    pre = grabbed["pre"][:, :3]
    synthetic.recentre_vector_head(model.vector_field_network, pre.mean(0), pre.std(0))
    return model


def own_model_matches(fx, ref_model) -> None:
    """This repo's modules, same seed + recipe, must hold exactly the reference's weights."""
    torch.manual_seed(fx["seed"])
    cfg = vf_nerf_amd.shipped_config(CPU, n_samples=fx["n_samples"], n_importance=fx["n_importance"])
    mine = vf_nerf_amd.VectorFieldNerf(cfg)
    synthetic.scale_hidden_weights(mine.vector_field_network, mine.rendering_network, fx["gain"])
    with torch.no_grad():
        last = mine.vector_field_network.layers[8]
        rl = ref_model.vector_field_network.layers[8]
        last.weight[:3] = rl.weight[:3]
        last.bias[:3] = rl.bias[:3]
    for a, b in ((mine.vector_field_network, ref_model.vector_field_network),
                 (mine.rendering_network, ref_model.rendering_network), (mine.density, ref_model.density)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys()), (list(sa.keys())[:5], list(sb.keys())[:5])
        for k in sa:
            assert torch.equal(sa[k], sb[k]), f"weights differ at {k}"


def capture(fx, model):
    uv, pose, K = synthetic.pinhole_batch(fx["n_rays"], fx["width"], fx["height"], fx["focal"], fx["cam_seed"],
                                          pose=(synthetic.orbit_pose(25.0, 10.0, 0.9) if fx["pose"] == "orbit" else None),
                                          skew=fx["skew"])
    if fx["far_per_ray"]:
        g = torch.Generator().manual_seed(99)
        far = fx["far"] * (0.8 + 0.4 * torch.rand(fx["n_rays"], 1, generator=g))
        model.ray_sampler.far = far
        model.fine_sampler.far = far
    rec = {"rand": [], "vf": [], "wcos": [], "vol_in": [], "vol_out": [], "colors": []}
    real_rand = torch.rand

    def rand_spy(*a, **k):
        out = real_rand(*a, **k)
        rec["rand"].append(out.clone())
        return out

    real_vol = ref_rendering.volsdf_volume_rendering
    real_wcos = ref_functions.window_cosine_similarity

    def vol_spy(z, d, normalize=True):
        w = real_vol(z, d, normalize)
        rec["vol_in"].append((z.detach().clone(), d.detach().clone()))
        rec["vol_out"].append(w.detach().clone())
        return w

    def wcos_spy(x, y, w):
        out = real_wcos(x, y, w)
        rec["wcos"].append(out.detach().clone())
        return out

    h1 = model.vector_field_network.register_forward_hook(lambda m, i, o: rec["vf"].append(o.detach().clone()))
    h2 = model.rendering_network.register_forward_hook(lambda m, i, o: rec["colors"].append(o.detach().clone()))
    torch.rand = rand_spy
    ref_rendering.volsdf_volume_rendering = vol_spy
    ref_functions.window_cosine_similarity = wcos_spy
    try:
        torch.manual_seed(1000 + fx["seed"])
        with torch.no_grad():
            out = model.render(pose, uv, K, fx.get("epoch", 0), fx.get("white", False))
        directions, ray_dirs, cam_loc = ref_rendering.get_ray_directions_and_cam_location(uv, pose, K, device=CPU)
    finally:
        torch.rand = real_rand
        ref_rendering.volsdf_volume_rendering = real_vol
        ref_functions.window_cosine_similarity = real_wcos
        h1.remove()
        h2.remove()

    n, s_c, n_f = fx["n_rays"], fx["n_samples"], fx["n_importance"]
    s_t = s_c + n_f
    draws = list(rec["rand"])
    d = {"uv": uv, "pose": pose, "intrinsics": K}
    if fx["perturb"]:
        d["u_coarse"], d["u_fine"], d["u_add"] = draws[0], draws[1], draws[2]
        assert len(draws) == 3
    else:
        d["u_add"] = draws[0]
        assert len(draws) == 1
    if fx["far_per_ray"]:
        d["far_per_ray"] = model.ray_sampler.far
    if fx.get("numjac"):      # each pass = the main VF forward + 6 offset forwards of the numerical Jacobian
        assert len(rec["vf"]) == 14
        rec["vf"] = [rec["vf"][0], rec["vf"][7]]
        d["directional_derivatives"] = out.directional_derivtives
    assert len(rec["vf"]) == 2 and len(rec["vol_out"]) == 2 and len(rec["wcos"]) == 2 and len(rec["colors"]) == 1
    (z_c, sigma_c), (z_f, sigma_f) = rec["vol_in"]
    d.update(directions=directions.reshape(-1, 3), ray_dirs=ray_dirs.reshape(-1, 3), cam_loc=cam_loc.reshape(-1, 3),
             z_coarse=z_c, normals_coarse=rec["vf"][0][:, :3].reshape(n, s_c, 3), window_cos_coarse=rec["wcos"][0],
             sigma_coarse=sigma_c, weights_coarse=rec["vol_out"][0],
             max_indices=torch.argmax(rec["vol_out"][0], dim=-1),
             z_vals=out.z_vals, points=out.points_coarse, normals=out.coarse_normals,
             feats_sub=rec["vf"][1][::8, 3:].contiguous(), window_cos=rec["wcos"][1], sigma=sigma_f,
             weights=rec["vol_out"][1], colors=out.coarse_colors, rgb=out.coarse_rgb_values,
             depth=out.coarse_depth_map)
    assert torch.equal(z_f, out.z_vals) and out.z_vals.shape == (n, s_t)
    assert torch.equal(rec["colors"][0], out.coarse_colors)
    return d, model


GRAD_KEYS = (("vf", "layers.0.0.weight"), ("vf", "layers.0.1.weight"), ("vf", "layers.3.0.bias"),
             ("vf", "layers.4.0.weight"), ("vf", "layers.4.1.bias"), ("vf", "layers.7.1.weight"),
             ("vf", "layers.8.weight"), ("vf", "layers.8.bias"), ("rn", "layers.0.0.weight"),
             ("rn", "layers.2.1.weight"), ("rn", "layers.4.weight"), ("rn", "layers.4.bias"))


def loss_coefficients(n, s_t):
    """Fixed linear functional of (rgb, depth, normals); shared with the tests via the seed."""
    g = torch.Generator().manual_seed(4242)
    return (torch.randn(n, 3, generator=g), torch.randn(n, 1, generator=g), 0.05 * torch.randn(n, s_t, 3, generator=g))


def dd_coefficients(m):
    g = torch.Generator().manual_seed(4343)
    return 1e-3 * torch.rand(m, generator=g)


def capture_grads(fx, model, data):
    """Reference gradients of the functional for the replayed random draws (the shipped training regime: networks
    in eval mode, autograd through the fine pass only)."""
    draws = [data[k] for k in ("u_coarse", "u_fine", "u_add") if k in data]
    it = iter(draws)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: next(it).clone()
    try:
        for p in model.parameters():
            p.grad = None
        out = model.render(data["pose"], data["uv"], data["intrinsics"], fx.get("epoch", 0), fx.get("white", False))
    finally:
        torch.rand = real_rand
    assert torch.equal(out.z_vals, data["z_vals"])
    a, b, c = loss_coefficients(*data["z_vals"].shape)
    loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
    if out.directional_derivtives is not None:     # fixed positive weights on the directional-derivative norms
        loss = loss + (out.directional_derivtives * dd_coefficients(out.directional_derivtives.shape[0])).sum()
    loss.backward()
    nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
    g = {}
    for net, key in GRAD_KEYS:
        g[f"grad.{net}.{key}"] = dict(nets[net].named_parameters())[key].grad.clone()
    for name, p in model.density.named_parameters():
        g[f"grad.density.{name}"] = p.grad.clone().reshape(1)
    g["loss"] = loss.detach().reshape(1)
    return g


BN_KEYS = (("vf", 0), ("vf", 3), ("vf", 7), ("rn", 0), ("rn", 3))


def capture_train(fx, model):
    """One render() of the reference with the networks in train mode, gradients enabled: draws, the two VF outputs
    ([M,268]), outputs, gradients of the fixed functional, and the BatchNorm running statistics afterwards (each VF
    BatchNorm has seen two batches, each rendering-net BatchNorm one)."""
    model.train()
    assert model.vector_field_network.training and model.rendering_network.training
    uv, pose, K = synthetic.pinhole_batch(fx["n_rays"], fx["width"], fx["height"], fx["focal"], fx["cam_seed"], skew=fx["skew"])
    rec = {"rand": [], "vf": [], "colors": []}
    real_rand = torch.rand

    def rand_spy(*a, **k):
        out = real_rand(*a, **k)
        rec["rand"].append(out.clone())
        return out

    h1 = model.vector_field_network.register_forward_hook(lambda m, i, o: rec["vf"].append(o.detach().clone()))
    h2 = model.rendering_network.register_forward_hook(lambda m, i, o: rec["colors"].append(o.detach().clone()))
    torch.rand = rand_spy
    try:
        torch.manual_seed(1000 + fx["seed"])
        for p in model.parameters():
            p.grad = None
        out = model.render(pose, uv, K, epoch=0)
    finally:
        torch.rand = real_rand
        h1.remove()
        h2.remove()
    assert len(rec["rand"]) == 3 and len(rec["vf"]) == 2 and rec["vf"][0].shape[1] == 268
    assert not out.directional_derivtives.requires_grad       # computed under no_grad: the loss term has no gradient
    d = {"uv": uv, "pose": pose, "intrinsics": K, "u_coarse": rec["rand"][0], "u_fine": rec["rand"][1], "u_add": rec["rand"][2],
         "vf_out_coarse": rec["vf"][0], "vf_out": rec["vf"][1], "z_vals": out.z_vals, "points": out.points_coarse,
         "normals": out.coarse_normals.detach(), "colors": out.coarse_colors.detach(), "rgb": out.coarse_rgb_values.detach(),
         "depth": out.coarse_depth_map.detach(), "directional_derivatives": out.directional_derivtives}
    a, b, c = loss_coefficients(*out.z_vals.shape)
    loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
    loss.backward()
    nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
    for net, key in GRAD_KEYS:
        d[f"grad.{net}.{key}"] = dict(nets[net].named_parameters())[key].grad.clone()
    for name, p in model.density.named_parameters():
        d[f"grad.density.{name}"] = p.grad.clone().reshape(1)
    d["loss"] = loss.detach().reshape(1)
    for net, i in BN_KEYS:
        bn = nets[net].layers[i][1]
        d[f"bn.{net}.{i}.running_mean"] = bn.running_mean.clone()
        d[f"bn.{net}.{i}.running_var"] = bn.running_var.clone()
        d[f"bn.{net}.{i}.num_batches_tracked"] = bn.num_batches_tracked.clone().reshape(1)
    return d


def main() -> None:
    torch.set_num_threads(8)
    only = set(sys.argv[1:])          # optional: fixture names to (re)generate
    for name, fx in FIXTURES.items():
        if only and name not in only:
            continue
        model = build_reference_model(fx)
        own_model_matches(fx, model)
        head = model.vector_field_network.layers[8]
        chk = synthetic.weights_checksum({"vf": model.vector_field_network.state_dict(),
                                          "rn": model.rendering_network.state_dict(),
                                          "density": model.density.state_dict()})
        if fx.get("train"):
            data = capture_train(fx, model)
        else:
            data, model = capture(fx, model)
            data.update(capture_grads(fx, model, data))
        arrays = {k: v.detach().cpu().numpy() for k, v in data.items()}
        arrays["head_weight"] = head.weight[:3].detach().numpy()
        arrays["head_bias"] = head.bias[:3].detach().numpy()
        arrays["weights_checksum"] = np.array([chk["sum"], chk["abs_sum"], chk["count"]], dtype=np.float64)
        arrays["fixture"] = np.array(repr(fx))
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **arrays)
        if fx.get("train"):
            print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); rays with colour: "
                  f"{float((data['rgb'].abs().sum(-1) > 1e-3).float().mean()):.2f}")
            continue
        nz = float((data["weights"].sum(-1) > 0.5).float().mean())
        print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); rays with surface hits: {nz:.2f}; "
              f"argmax>0: {float((data['max_indices'] > 0).float().mean()):.2f}")


if __name__ == "__main__":
    main()
