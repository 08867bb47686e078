#!/usr/bin/env python3
"""The weight-gradient kernel's launches of a training step, each ALONE on the device (csrc/vfn_dwf.hip): the batched 256 x 256 products of
the default training form, the encoding tile's 256 x 64 and the head's 32 x 256, at the step's point count.

    python tools/bench_dwf_shapes.py [points]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vf_nerf_amd import lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 629120
dev = torch.device("cuda", 0)
lib.load()
g = torch.Generator().manual_seed(0)
groups = lib.weight_grad_groups(m)
n_grp = lib.frag_groups(m)
x16 = lib.rows_to_frag(torch.relu(torch.randn(m, 256, generator=g)).to(dev), torch.float16)
dy = lib.rows_to_frag_f16s((torch.randn(m, 256, generator=g) * 1e-3).to(dev))
aux = torch.randn(m, lib.AUX_K, generator=g).to(dev)
dz = torch.randn(m, 4, generator=g).to(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


part0, db0 = torch.empty(groups, 256, 256, device=dev), torch.empty(groups, 256, device=dev)
part1 = torch.empty(groups, 256, 64, device=dev)
part2, db2 = torch.empty(groups, 32, 256, device=dev), torch.empty(groups, 32, device=dev)
rows = [("256 x 256  dY f16s x X f16 (one product, %d slabs)" % groups, lambda: lib.weight_grad_frag(0, dy, lib.DYF_FRAGF16S, x16, lib.XF_FRAG16, m, groups, part0, db0),
         m * 1024 + groups * 256 * 256 * 4),
        ("256 x 64   dY f16s x encoding tile", lambda: lib.weight_grad_frag(1, dy, lib.DYF_FRAGF16S, aux, lib.XF_AUX40, m, groups, part1, db0), m * (512 + 160)),
        ("32 x 256   head dz x X f16", lambda: lib.weight_grad_frag(2, dz, lib.DYF_DZ4, x16, lib.XF_FRAG16, m, groups, part2, db2), m * (512 + 16))]
for name, fn, nbytes in rows:
    us = timed(fn)
    print(f"{name:58s} {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s of operand bytes   ({m} points)")
