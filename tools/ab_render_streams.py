"""Interleaved A/B of the headline render (4096 rays x 64 + 64 samples, trained weights) with the batch issued as 1 .. 4 ranges of rays on as
many streams inside vfn_render_fwd (model.render_streams): the per-ray launches of one range beside the fused launches of another.
    python tools/ab_render_streams.py [rays] [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
built = bench.build_trained_scene(dev, rays, 64, 64, seed=0)
model, uv, pose, K = (built[:4] if built is not None else bench.build_scene(dev, rays, 64, 64, seed=0))
model.eval()


def run(streams, steps=40):
    model.render_streams = streams
    with torch.no_grad():
        for _ in range(5):
            model.render(pose, uv, K, 0, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.render(pose, uv, K, 0, False)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


with torch.no_grad():
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        model.render(pose, uv, K, 0, False)
for rep in range(reps):
    print("  ".join(f"streams={s}: {run(s):.4f} ms ({rays / run(s) / 1e3:.3f} M rays/s)" for s in (1, 2, 3, 4, 0)))
