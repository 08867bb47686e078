"""The training-mode forward layer and its weight-gradient product with and without the activation fold, alone on the device:
    python tools/bench_fold.py [M]
row pass + product (what rounds 3-5 ran) against vfn_linear_rows_fold; vfn_weight_grad_partials_bf16_ld against ..._fold."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vf_nerf_amd import batchstat, lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
dev = torch.device("cuda:0")
k = n = 256
z_prev = torch.randn(m, k, device=dev)
coef = torch.stack([torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.ones(n, device=dev)]).contiguous()
w, b = torch.randn(n, k, device=dev) * 0.1, torch.randn(n, device=dev)
x, z = torch.empty(m, k, device=dev), torch.empty(m, n, device=dev)
part = torch.empty(lib.linear_rows_stat_parts(m), 2, n, device=dev)
dy = torch.randn(m, n, device=dev) * 1e-3
G = batchstat._groups(m)
dw, db = torch.empty(G, 256, 256, device=dev), torch.empty(G, 256, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = [("row pass (vfn_bstat_relu_rows)", lambda: lib.bstat_relu_rows(z_prev, coef, m, n, 1.0, x)),
        ("product on the stored activation (+ stats)", lambda: lib.linear_rows(x, w, b, m, n, k, z, stats_part=part, arith=lib.GEMM_SPLIT_F16, planes=lib.wplanes(n, k, dev))),
        ("folded product (+ stats)", lambda: lib.linear_rows_fold(z_prev, coef, n, 1.0, w, b, m, n, k, z, stats_part=part, arith=lib.GEMM_SPLIT_F16, planes=lib.wplanes(n, k, dev))),
        ("weight gradients on the stored activation", lambda: lib.weight_grad_partials_bf16_cols(dy, x, m, G, dw, db)),
        ("weight gradients, folded", lambda: lib.weight_grad_partials_bf16_fold(dy, z_prev, coef, n, 1.0, m, G, dw, db))]
for name, fn in rows:
    print(f"{name:48s} {timed(fn):7.3f} ms   (M = {m})")
