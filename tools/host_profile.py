#!/usr/bin/env python3
"""HOST time of a training step (CPU time spent issuing work: no device synchronisation inside the timed loop until its end) next to the
wall time of the same steps, for trainer.TrainStep's two paths: the whole step as ONE C call (vfn_train_step, the default) and the
launch-by-launch Python path (model.one_call_train_step = False).  The reference's batch size (1 024 rays) is the launch-bound regime
(VERDICT r01 weak 6, r03 next 3).

    python tools/host_profile.py [rays] [steps] [cores]

``cores``: restrict THIS process to that many host cores first (os.sched_setaffinity, before any GPU call) — what one of eight ranks on
a 32-core host gets."""
import json
import os
import sys
import time

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cores = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if cores > 0:
    os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:cores]))      # before torch / HIP are loaded

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vf_nerf_amd import supervision, trainer  # noqa: E402

if cores > 0:
    torch.set_num_threads(cores)
dev = torch.device("cuda", 0)
out = {"rays": rays, "samples": 128, "steps": steps, "host_cores": len(os.sched_getaffinity(0)), "train_step_streams": int(os.environ.get("VFN_TRAIN_STREAMS", "2"))}
for path in ("one_call", "python"):
    built = bench.build_trained_scene(dev, rays, 64, 64, seed=0)             # trained weights, targets = the model's own render (bench.training_targets)
    if built is not None:
        model, uv, pose, K, info = built
        model._bench_trained_weights = info
    else:
        model, uv, pose, K = bench.build_scene(dev, rays, 64, 64, seed=0)
    model.one_call_train_step = path == "one_call"
    model.sparse_colour_training = os.environ.get("VFN_SPARSE_COLOURS", "1") != "0"
    model.train_step_streams = int(os.environ.get("VFN_TRAIN_STREAMS", "2"))        # A/B of the side stream inside vfn_train_step
    rgb_gt, depth_gt, centroid, radius = bench.training_targets(model, uv, pose, K, dev, 64, 64)
    supervision.manual_seed(7)
    step = trainer.TrainStep(model, centroid, border_radius=radius, far=1.0)
    for _ in range(8):
        step(pose, uv, K, rgb_gt, depth_gt, epoch=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(pose, uv, K, rgb_gt, depth_gt, epoch=0)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # host time without the device queue pushing back: a synchronisation before every step (the queue is empty when the step is issued)
    iso = 0.0
    for _ in range(min(steps, 30)):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(pose, uv, K, rgb_gt, depth_gt, epoch=0)
        iso += time.perf_counter() - t1
    torch.cuda.synchronize()
    sel = step.last_colour_counts.tolist() if step.last_colour_counts is not None else None
    out[path] = {"took_one_call_path": step.one_call.why_not is None, "why_not": step.one_call.why_not,
                 "sparse_colour_branch": bool(model.sparse_colour_training) and step.one_call.why_not is None,
                 "samples_with_nonzero_weight": round(sel[0] / sel[1], 4) if sel else None,
                 "wall_ms_per_step": round(wall / steps * 1e3, 4), "host_issue_ms_per_step_back_to_back": round(host / steps * 1e3, 4),
                 "host_issue_ms_per_step_empty_queue": round(iso / min(steps, 30) * 1e3, 4)}
print(json.dumps(out))
