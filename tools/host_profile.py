#!/usr/bin/env python3
"""Where the HOST time of a training step goes (no device synchronisation inside a step: the numbers are CPU time spent
issuing work), next to the device time of the same step — the launch-bound regime of the reference's 1 024-ray batches
(VERDICT r01, weak item 6).   python tools/host_profile.py [rays] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vf_nerf_amd import optim as voptim, supervision  # noqa: E402

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
model, uv, pose, K = bench.build_scene(dev, rays, 64, 64, seed=0)
s_t = 128
g = torch.Generator().manual_seed(7)
rgb_gt = torch.rand(rays, 3, generator=g).to(dev)
depth_gt = (0.2 + 0.6 * torch.rand(rays, 1, generator=g)).to(dev)
centroid = torch.tensor([0.0, 0.0, 0.6], device=dev)
n_sup = (rays * s_t) // 10
clip = model.config.scheduler_config.clip_norm
acc = {}


def lap(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def step(record):
    t = time.perf_counter()
    out = model.render(pose, uv, K, epoch=0)
    if record: t = lap("render (forward)", t)
    bp, b_gt = supervision.sample_border_points(0.75, 1.0, n_sup, centroid, dev)
    cp, c_gt = supervision.sample_center_points(centroid, 0.05, n_sup, dev)
    sup_n = model.vector_field_network(torch.cat([bp, cp]))[:, :3]
    sup_gt = torch.cat([b_gt, c_gt])
    if record: t = lap("supervision points + VF forward", t)
    normals = out.coarse_normals.reshape(-1, 3)
    loss = 2.0 * (out.coarse_rgb_values - rgb_gt).abs().mean() + \
        0.5 * torch.clamp((out.coarse_depth_map - depth_gt).abs(), max=0.5).mean() + \
        0.1 * ((normals.norm(dim=-1) - 1.0) ** 2).mean() + 1.0 * ((sup_n - sup_gt) ** 2).mean()
    if record: t = lap("loss", t)
    model.optimizer.zero_grad()
    if record: t = lap("zero_grad", t)
    loss.backward()
    if record: t = lap("backward", t)
    voptim.clip_grad_norm_(model.parameters(), clip)
    if record: t = lap("clip_grad_norm_", t)
    model.optimizer.step()
    model.scheduler.step()
    if record: t = lap("optimizer.step + scheduler.step", t)


for _ in range(5):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{rays} rays x {s_t}: wall {wall / steps * 1e3:.3f} ms/step, host issue time {host / steps * 1e3:.3f} ms/step "
      f"(optimizer: {type(model.optimizer).__name__})")
for k, v in acc.items():
    print(f"   {k:36s} {v / steps * 1e3:7.3f} ms")
