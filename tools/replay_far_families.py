"""The 8 000-step task (tests/golden/trained_far_run.npz), many device streams per family, one number per run: the mean over the run's sixteen
500-step windows of (the run's window mean / the mean of the reference's runs in that window) for the total loss — and the same number for
each REFERENCE run against the mean of the others (the yardstick: what one run of the reference differs from its siblings by).
    python tools/replay_far_families.py [streams] [exact_fp32_streams] > profiles/r06/replay_far_families.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import replay_reference_run as rr  # noqa: E402

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 12
exact = int(sys.argv[2]) if len(sys.argv) > 2 else 6
raw, recipe = rr.load_task("far")
ref = rr.windows(raw["runs.loss"], 500)                     # [R, 16]
R = ref.shape[0]
print(f"task: {recipe['epochs'] * recipe['steps_per_epoch']} steps x {recipe['n_rays']} rays; reference runs: {R}")
loo = [float((ref[i] / np.delete(ref, i, axis=0).mean(0)).mean()) for i in range(R)]
print(f"reference runs, each against the mean of the others: {np.round(loo, 4).tolist()}  (spread: std {np.std(loo, ddof=1):.4f});  final PSNR {np.round(raw['runs.psnr_before_after'][:, 1], 2).tolist()}")
mean_ref = ref.mean(0)
only = os.environ.get("VFN_FAMILIES")
for name, fam, n in (("default kernels, default 16-bit storages (step session)", "default", streams),
                     ("default kernels and storages, launch by launch (no session, dense colours)", "default_lbl", streams),
                     ("default kernels and storages, step session with the dense colour branch", "default_dense", streams),
                     ("f16x3 forward kernels, exact-fp32 backward kernels (launch by launch)", "fp32_backward", streams),
                     ("default kernels, fp32 storages (step session)", "fp32_storages", streams),
                     ("exact-fp32 kernels, fp32 storages (launch by launch)", "fp32", exact)):
    if n <= 0 or (only and fam not in only.split(",")):
        continue
    runs = [rr.replay(s, fam, task="far") for s in range(n)]
    per_run = np.array([float((rr.windows(r["loss"][None], 500)[0] / mean_ref).mean()) for r in runs])
    late = np.array([float((rr.windows(r["loss"][None], 500)[0][-4:] / mean_ref[-4:]).mean()) for r in runs])
    ps = np.array([r["psnr_before_after"][1] for r in runs])
    print(f"{name}: {n} runs, {1e3 * sum(r['seconds'] for r in runs) / sum(r['steps'] for r in runs):.3f} ms/step")
    print(f"    ratio over the whole run: mean {per_run.mean():.4f} +- {per_run.std(ddof=1) / np.sqrt(n):.4f} (standard error; single runs: std {per_run.std(ddof=1):.4f}, min {per_run.min():.4f}, max {per_run.max():.4f})")
    print(f"    ratio over the last 2 000 steps: mean {late.mean():.4f} +- {late.std(ddof=1) / np.sqrt(n):.4f};  final PSNR mean {ps.mean():.2f} +- {ps.std(ddof=1) / np.sqrt(n):.2f} dB (min {ps.min():.2f}, max {ps.max():.2f})", flush=True)
