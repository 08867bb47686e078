"""Which of the two 16-bit storages of the default training path costs the 3-4 % of loss at equal step count that
profiles/r06/replay_reference_run_far.md shows?  The 8 000-step task, default f16x3 kernels through the step session, four streams per
storage pair:  python tools/replay_storage_split.py > profiles/r06/replay_storage_split.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import replay_reference_run as rr  # noqa: E402

raw, recipe = rr.load_task("far")
ref = rr.reference_curves(raw)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print(f"ratio of the family's 500-step window means of the total loss to the reference's (three runs), {streams} device streams per family; last column: final PSNR (reference 17.73)")
for act, grad in (("f16", "f16"), ("f16", "fp32"), ("fp32", "f16"), ("fp32", "fp32")):
    runs = [rr.replay(s, f"storages:{act},{grad}", task="far") for s in range(streams)]
    cmp = rr.compare(ref, runs, window=500)
    r = np.array(cmp["quantities"]["loss"]["ratio_of_means"])
    print(f"activations {act:4s} gradients {grad:4s}: {1e3 * sum(x['seconds'] for x in runs) / sum(x['steps'] for x in runs):.3f} ms/step   mean ratio {r.mean():.4f}   last four windows "
          f"{np.round(r[-4:], 3).tolist()}   PSNR {cmp['psnr_vs_teacher_db']['replay_after_mean']:.2f}   issued as {runs[0]['issued_as']}", flush=True)
