#!/bin/bash
# The bench lines and host profiles of profiles/rNN/ that tools/profile_round.sh (the rocprofv3 passes) does not write:
#     bash tools/round_measurements.sh [out_dir]      (on the GPU box, from the repository root)
set -u
OUT=${1:-gpurun_out/measure}
mkdir -p "$OUT"
last() { tail -1; }
python bench.py 2> "$OUT/bench_default.err" | last > "$OUT/bench_default.json"
python bench.py --workload train --steps 30 2>/dev/null | last > "$OUT/bench_train.json"
python bench.py --workload train --rays 1024 --steps 60 2>/dev/null | last > "$OUT/bench_train_1024.json"
python bench.py --workload train --batch-statistics --train-weights random --steps 10 --warmup 3 2>/dev/null | last > "$OUT/bench_train_batch_statistics.json"
python tools/host_profile.py 1024 60 2>/dev/null | last > "$OUT/host_profile_1024.json"
python tools/host_profile.py 1024 60 4 2>/dev/null | last > "$OUT/host_profile_1024_4cores.json"
python tools/host_profile.py 4096 60 2>/dev/null | last > "$OUT/host_profile_4096.json"
VFN_SPARSE_COLOURS=0 python tools/host_profile.py 4096 40 2>/dev/null | last > "$OUT/host_profile_4096_dense.json"
python bench.py --workload view 2>/dev/null | last > "$OUT/bench_view.json"
python bench.py --workload view --as-evaluator --no-parity 2>/dev/null | last > "$OUT/bench_view_as_evaluator.json"
# round 5: the reference evaluator's UNCHANGED loop (dropin.install(patch_evaluator=False)) at its own 512-ray chunks, the 8 192-ray step,
# the weight-gradient launches alone, the step's idle gaps by call sequence
python bench.py --workload view --as-evaluator --unchanged-evaluator-loop --rays 512 --no-parity --steps 3 2>/dev/null | last > "$OUT/bench_view_as_evaluator_unchanged_loop.json"
python bench.py --workload view --as-evaluator --rays 512 --no-parity --steps 3 2>/dev/null | last > "$OUT/bench_view_as_evaluator_512.json"
python tools/host_profile.py 8192 10 2>/dev/null | last > "$OUT/host_profile_8192.json"
python bench.py --workload grid --grid-res 512 --steps 4 --warmup 6 2>/dev/null | last > "$OUT/bench_grid512.json"
python tools/grid_phase_times.py 512 4 2>/dev/null | grep -v Warn > "$OUT/grid_phase_times.txt"
python tools/bench_dwf_shapes.py 2>/dev/null | tail -3 > "$OUT/dwf_shapes.txt"
bash tools/micro_load_width.sh 2>/dev/null > "$OUT/load_width.txt" || true
python tools/bench_forward_modes.py 2>/dev/null | grep "ms " > "$OUT/forward_modes.txt"
bash tools/ab_sup_window.sh > "$OUT/ab_supervision_forward_placement.txt" 2>/dev/null
ls -la "$OUT"
