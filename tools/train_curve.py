"""BASELINE.json configs[2] / SURVEY.md §8d C3: loss-curve agreement of the training step on the 16-bit matrix cores
(precision "f16x3": f16-split forward, bf16-split dX chain and weight gradients) against the exact-fp32 MFMA kernels, from
the same initial weights, the same rays, targets and random draws, over >= 100 optimizer steps.

    python tools/train_curve.py [steps] > profiles/r02/train_curve.json

Both runs are chaotic in the usual sense (a ReLU unit or a proposal argmax landing on the other side flips a discrete
event and the trajectories part), so what is reported is the relative loss difference per step and its running
statistics, not a bitwise match."""
import json, sys, time
import torch
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import optim, supervision

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda:0")
n_rays, s_c, n_f = 1024, 64, 64


def run(precision, activations="fp32", gradients="fp32"):
    model, uv, pose, K = bench.build_scene(dev, n_rays, s_c, n_f, seed=0)
    model.precision = precision
    model.activation_storage = activations
    model.gradient_storage = gradients
    g = torch.Generator().manual_seed(7)
    rgb_gt = torch.rand(n_rays, 3, generator=g).to(dev)
    depth_gt = (0.2 + 0.6 * torch.rand(n_rays, 1, generator=g)).to(dev)
    centroid = torch.tensor([0.0, 0.0, 0.6], device=dev)
    supervision.manual_seed(3)
    model.rng_seed, model._rng_offset = 11, 0
    clip = model.config.scheduler_config.clip_norm
    losses = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(steps):
        out = model.render(pose, uv, K, epoch=0)
        bp, b_gt = supervision.sample_border_points(0.75, 1.0, 4096, centroid, dev)
        sup = model.vector_field_network(bp)[:, :3]
        normals = out.coarse_normals.reshape(-1, 3)
        loss = 2.0 * (out.coarse_rgb_values - rgb_gt).abs().mean() + 0.5 * torch.clamp((out.coarse_depth_map - depth_gt).abs(), max=0.5).mean() + \
            0.1 * ((normals.norm(dim=-1) - 1.0) ** 2).mean() + 1.0 * ((sup - b_gt) ** 2).mean()
        model.optimizer.zero_grad()
        loss.backward()
        optim.clip_grad_norm_(model.parameters(), clip)
        model.optimizer.step()
        model.scheduler.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    return [float(x) for x in losses], (time.perf_counter() - t0) / steps * 1e3


exact, ms_exact = run("fp32")
runs = {"fp32 activations, fp32 gradients": run("f16x3", "fp32", "fp32"),
        "f16 activations, fp32 gradients": run("f16x3", "f16", "fp32"),
        "f16 activations, scaled f16 gradients (default)": run("f16x3", "f16", "f16"),
        "f16 activations, bf16 gradients": run("f16x3", "f16", "bf16")}


def stats(losses):
    rel = [abs(a - b) / max(abs(b), 1e-9) for a, b in zip(losses, exact)]
    return {"loss_first_last": [losses[0], losses[-1]], "rel_diff_step0": rel[0], "rel_diff_max_first_10": max(rel[:10]),
            "rel_diff_median": sorted(rel)[len(rel) // 2], "rel_diff_max": max(rel),
            "first_step_with_rel_diff_above_1e-3": next((i for i, r in enumerate(rel) if r > 1e-3), None),
            "mean_loss_last_20": sum(losses[-20:]) / 20, "loss_every_10_steps": losses[::10]}


print(json.dumps({
    "workload": f"{steps} optimizer steps, {n_rays} rays x {s_c + n_f} samples + 4096 supervision points, same weights / rays / targets / draws; "
                "relative loss differences against the exact-fp32 kernels' run",
    "fp32_kernels": {"ms_per_step": round(ms_exact, 3), "loss_first_last": [exact[0], exact[-1]], "mean_loss_last_20": sum(exact[-20:]) / 20,
                     "loss_every_10_steps": exact[::10]},
    "f16x3": {k: dict(stats(v[0]), ms_per_step=round(v[1], 3)) for k, v in runs.items()}}, indent=1))
