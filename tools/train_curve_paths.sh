#!/bin/bash
# Round 5: the convergence comparison of the step's issue paths and arithmetics on the "recover" task (tools/train_curve.py: a trained
# teacher, the student 0.5 % off it, 1 500 optimizer steps at the shipped learning rate; 8 random streams per mode) -> one JSON line per mode
# in gpurun_out/train_curve_paths.jsonl.  Run on the GPU box from the repo root.
export VFN_CURVE_ONLY_DEFAULT=1 VFN_CURVE_STREAMS=${STREAMS:-8} VFN_CURVE_TASK=recover VFN_CURVE_PERTURB=0.005
out=gpurun_out/train_curve_paths.jsonl
mkdir -p gpurun_out; : > $out
python tools/train_curve.py 1500 2>/dev/null | tail -1 >> $out                                   # one C call per step (trainer.TrainStep)
VFN_ISSUE=call_sequence python tools/train_curve.py 1500 2>/dev/null | tail -1 >> $out           # the reference trainer's call sequence (step session)
VFN_ONE_CALL=0 python tools/train_curve.py 1500 2>/dev/null | tail -1 >> $out                    # launch by launch from Python, dense colour branch
VFN_CURVE_PRECISION=fp32 VFN_ONE_CALL=0 python tools/train_curve.py 1500 2>/dev/null | tail -1 >> $out   # exact-fp32 kernels, fp32 storages
python - <<'PY'
import json
for line in open("gpurun_out/train_curve_paths.jsonl"):
    d = json.loads(line)
    p, l = d["final_psnr_db"], d["final_loss"]
    mean = lambda v: sum(v) / len(v)
    sd = lambda v: (sum((x - mean(v)) ** 2 for x in v) / max(len(v) - 1, 1)) ** 0.5
    print(f'{d["kernels"]:34s} issue={d["issue"]:14s} one_call={d["one_call"]} psnr {mean(p):.3f} +- {sd(p):.3f} (from {d["psnr_before_db"][0]})  loss {mean(l):.4f} +- {sd(l):.4f}  {d["issued_as"][0]}')
PY
