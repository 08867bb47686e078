#!/usr/bin/env python3
"""cProfile of the host side of a training step issued as the reference trainer's own call sequence (tools/reference_sequence.py) or as
trainer.TrainStep's one C call: where the Python-side time of a step goes.

    python tools/host_cprofile.py [rays] [steps] [drop_in|one_call] [top]"""
import cProfile
import io
import os
import pstats
import sys
from types import SimpleNamespace

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = sys.argv[3] if len(sys.argv) > 3 else "drop_in"
top = int(sys.argv[4]) if len(sys.argv) > 4 else 45
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bench  # noqa: E402
import reference_sequence  # noqa: E402
from vf_nerf_amd import loss as vloss, supervision, trainer  # noqa: E402

dev = torch.device("cuda", 0)
built = bench.build_trained_scene(dev, rays, 64, 64, seed=0)
model, uv, pose, K, info = built
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
for _ in range(10):
    fn()
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(steps):
    torch.cuda.synchronize()          # empty queue: the time below is issue time, not back-pressure
    fn()
prof.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(prof, stream=buf).sort_stats(key).print_stats(top)
    print(f"==== {mode}, {rays} rays, {steps} steps, sorted by {key} (divide by {steps} for per-step) ====")
    print(buf.getvalue())
