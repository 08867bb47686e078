#!/usr/bin/env python3
"""Where a host-in / host-out dense-grid query spends its time (BASELINE.json configs[4]): python tools/grid_phase_times.py [res] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from vf_nerf_amd import grid, lib  # noqa: E402

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
model, _, _, _ = bench.build_scene(dev, 16, 64, 64, seed=0)
dec = model.fine_vector_field_network
samples = bench.reference_lattice(res)
n = samples.shape[0]
lat = grid.lattice_axes(samples)
runs = grid._rank_runs(n, 100000, 0, 1)
for w in (1, 2, 4, 8, 16, 32):
    t = time.perf_counter()
    ok = grid.lattice_rows_match(samples, lat[0], lat[1:], runs, workers=w)
    print(f"host verification, {w:2d} threads: {1e3 * (time.perf_counter() - t):7.1f} ms ({ok})")
t = time.perf_counter()
out = torch.empty((n, 3), pin_memory=True)
print(f"pinned [n,3] allocation (first): {1e3 * (time.perf_counter() - t):7.1f} ms")
del out
t = time.perf_counter()
out = torch.empty((n, 3), pin_memory=True)
print(f"pinned [n,3] allocation (cached): {1e3 * (time.perf_counter() - t):7.1f} ms")
d = torch.empty(1 << 22, 3, device=dev)
torch.cuda.synchronize()
t = time.perf_counter()
for lo in range(0, n, 1 << 22):
    out[lo:lo + (1 << 22)].copy_(d[: min(1 << 22, n - lo)], non_blocking=True)
torch.cuda.synchronize()
el = time.perf_counter() - t
print(f"download of [n,3] alone: {1e3 * el:7.1f} ms = {n * 12 / el / 1e9:.1f} GB/s")
del out
for fast in (True, False):
    grid.LATTICE_FAST_PATH = fast
    keep = None
    for c in range(calls):
        torch.cuda.synchronize()
        t = time.perf_counter()
        got = grid.get_set_predictions(dec, samples, 100000, dev)
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        print(f"{'lattice' if fast else 'upload '} call {c}: {1e3 * el:7.1f} ms = {n / el / 1e6:6.1f} M points/s ({grid.last_path})")
        keep = got
dsamples = samples[: 1 << 24].to(dev)
for c in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    grid.get_set_predictions(dec, dsamples, 100000, dev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print(f"resident 2^24 points call {c}: {1e3 * el:7.1f} ms = {(1 << 24) / el / 1e6:6.1f} M points/s")
