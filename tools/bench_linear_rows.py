"""Micro-benchmark of vfn_linear_rows (csrc/vfn_bstat.hip) against the fp32 matrix-pipe peak (157.3 TFLOP/s nominal):
    python tools/bench_linear_rows.py [M]
Prints TFLOP/s for the shapes of the training-mode path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vf_nerf_amd import lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
dev = "cuda:0"
for arith_name, arith, presplit in (("exact fp32", lib.GEMM_EXACT, False), ("split f16", lib.GEMM_SPLIT_F16, False),
                                    ("split f16, W's planes split once per call (vfn_linear_rows_ws)", lib.GEMM_SPLIT_F16, True),
                                    ("split bf16", lib.GEMM_SPLIT_BF16, False),
                                    ("split bf16, W's planes split once per call", lib.GEMM_SPLIT_BF16, True),
                                    ("bf16 in three parts (six products)", lib.GEMM_BF16X6, False)):
  print(f"--- {arith_name}")
  for name, k, n, trans, stats in (("fwd 256->256 + stats", 256, 256, False, True), ("fwd 256->256", 256, 256, False, False),
                                 ("dX 256<-256", 256, 256, True, False), ("fwd 40->256 + stats", 39, 256, False, True),
                                 ("fwd 289->256 + stats", 289, 256, False, True), ("fwd 256->259 tanh", 256, 259, False, False)):
      kp = (k + 7) & ~7
      a = torch.randn(m, kp, device=dev)
      a[:, k:] = 0
      w = torch.randn(k, n, device=dev) if trans else torch.randn(n, k, device=dev)
      b = None if trans else torch.randn(n, device=dev)
      c = torch.empty(m, (n + 7) & ~7, device=dev)
      part = torch.empty(lib.linear_rows_stat_parts(m), 2, n, device=dev) if stats else None
      act = lib.ACT_TANH if n == 259 else lib.ACT_NONE

      def run():
          lib.linear_rows(a, w, b, m, n, k, c, act=act, transpose_w=trans, stats_part=part, arith=arith,
                          planes=lib.wplanes(n, k, dev) if presplit else None)
      for _ in range(3):
          run()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      reps = 10
      e0.record()
      for _ in range(reps):
          run()
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / reps
      flops = 2.0 * m * k * n
      gb = 4.0 * m * (kp + ((n + 7) & ~7)) / 1e9                  # A read + C written, fp32
      print(f"{name:28s} M={m}: {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  ({flops / ms / 1e9 / 157.3 * 100:5.1f}% of 157.3)  {gb / ms:5.2f} TB/s of A + C")
