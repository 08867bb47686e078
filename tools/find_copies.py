#!/usr/bin/env python3
"""Which Python lines issue the device copies of a training step?  torch.profiler over a few one-call steps, every Memcpy / Memset with its stack.
python tools/find_copies.py [rays] [one_call|drop_in]"""
import os
import sys
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
import reference_sequence  # noqa: E402
from vf_nerf_amd import loss as vloss, supervision, trainer  # noqa: E402

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "one_call"
dev = torch.device("cuda", 0)
model, uv, pose, K, info = bench.build_trained_scene(dev, rays, 64, 64, seed=0)
model._bench_trained_weights = info
rgb_gt, depth_gt, centroid, radius = bench.training_targets(model, uv, pose, K, dev, 64, 64)
supervision.manual_seed(7)
if mode == "drop_in":
    crit = vloss.VFLoss(SimpleNamespace(**trainer.SHIPPED_LOSS_CONFIG), SimpleNamespace(**trainer.SHIPPED_LOSS_WEIGHTS))
    loop = reference_sequence.ReferenceLoop(model, crit, reference_sequence.StandInDataset(centroid, 1.0), radius, sync_each_step=False)
    data = {"uv": uv.unsqueeze(0), "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0), "rgb": rgb_gt.unsqueeze(0), "depth": depth_gt.unsqueeze(0)}
    fn = lambda: loop(data, 0)  # noqa: E731
else:
    step = trainer.TrainStep(model, centroid, border_radius=radius, far=1.0)
    fn = lambda: step(pose, uv, K, rgb_gt, depth_gt, epoch=0)  # noqa: E731
for _ in range(12):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    name = ev.name
    if ("emcpy" in name or "emset" in name or "copy_" in name or "fill_" in name or "zero_" in name) and ev.stack:
        frames = [s for s in ev.stack if "/root/repo" in s or "vf_nerf_amd" in s or "tools/" in s][:3]
        key = (name, tuple(frames))
        seen[key] = seen.get(key, 0) + 1
from collections import Counter
kinds = Counter(ev.name for ev in prof.events() if "emcpy" in ev.name or "emset" in ev.name or "hipMem" in ev.name)
print(dict(kinds))
# the runtime calls in issue order around each copy (CPU side): what precedes a hipMemcpy* call
evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU), key=lambda e: e.time_range.start)
for i, e in enumerate(evs):
    if "hipMemcpy" in e.name:
        print("   ...", [x.name for x in evs[max(0, i - 4):i + 3]])
for (name, frames), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d} x {name}")
    for f in frames:
        print("        ", f)
