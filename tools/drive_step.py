#!/usr/bin/env python3
"""N training steps issued one way, nothing else — the program to put behind `rocprofv3 --kernel-trace --stats --`:

    python tools/drive_step.py [rays] [steps] [one_call|drop_in|drop_in_item|python]

Prints one JSON line: wall ms per step."""
import json
import os
import sys
import time
from types import SimpleNamespace

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = sys.argv[3] if len(sys.argv) > 3 else "drop_in"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bench  # noqa: E402
import reference_sequence  # noqa: E402
from vf_nerf_amd import loss as vloss, supervision, trainer  # noqa: E402

dev = torch.device("cuda", 0)
model, uv, pose, K, info = bench.build_trained_scene(dev, rays, 64, 64, seed=0)
model._bench_trained_weights = info
rgb_gt, depth_gt, centroid, radius = bench.training_targets(model, uv, pose, K, dev, 64, 64)
supervision.manual_seed(7)
if mode.startswith("drop_in"):
    crit = vloss.VFLoss(SimpleNamespace(**trainer.SHIPPED_LOSS_CONFIG), SimpleNamespace(**trainer.SHIPPED_LOSS_WEIGHTS))
    loop = reference_sequence.ReferenceLoop(model, crit, reference_sequence.StandInDataset(centroid, 1.0), radius, sync_each_step=mode.endswith("item"))
    data = {"uv": uv.unsqueeze(0), "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0), "rgb": rgb_gt.unsqueeze(0), "depth": depth_gt.unsqueeze(0)}
    fn = lambda: loop(data, 0)  # noqa: E731
else:
    model.one_call_train_step = mode == "one_call"
    model.step_sessions = mode == "one_call"
    step = trainer.TrainStep(model, centroid, border_radius=radius, far=1.0)
    fn = lambda: step(pose, uv, K, rgb_gt, depth_gt, epoch=0)  # noqa: E731
for _ in range(10):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    fn()
torch.cuda.synchronize()
print(json.dumps({"mode": mode, "rays": rays, "steps": steps, "wall_ms_per_step": round((time.perf_counter() - t0) / steps * 1e3, 4)}))
