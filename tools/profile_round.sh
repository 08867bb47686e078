#!/bin/bash
# The rocprofv3 passes behind profiles/rNN/ (run on the GPU box from the repository root):  bash tools/profile_round.sh [out_dir]
# Kernel trace and every counter set in its OWN pass (MI355X_MICROARCH.md, HBM / rocprofv3 section); the program after `--` is python3
# itself (no wrapper: the profiler's preloaded library has initialised the GPU before the program starts).
set -u
R=$(pwd)
OUT=${1:-gpurun_out/export}
mkdir -p "$R/$OUT" "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-train --no-two-product-leg --no-other-scene-leg --no-shipped-rows --no-live-traffic"
# the SAME command un-profiled and under the kernel trace, back to back on this box: the line's event-timed launch
# duration must follow from the profiler's table, or profiled_vs_unprofiled.md says why not (clock and cycle counts of both runs)
$B --steps 20 > "$R/$OUT/bench_unprofiled_same_box.json" 2> "$R/gpurun_out/bench_unprofiled.err"
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_kt" -o kt -- $B --steps 20 > "$R/gpurun_out/prof_kt.log" 2>&1
grep '^{"metric"' "$R/gpurun_out/prof_kt.log" | tail -1 > "$R/$OUT/bench_profiled_same_box.json"
(cd "$R" && python3 tools/profiled_vs_unprofiled.py "$OUT/bench_unprofiled_same_box.json" gpurun_out/prof_kt.log "$(find gpurun_out/prof_kt -name 'kt_results.db' | head -1)" > "$OUT/profiled_vs_unprofiled.md" 2> "$OUT/profiled_vs_unprofiled.err")
rocprofv3 --pmc FETCH_SIZE -d "$R/gpurun_out/prof_fetch" -o pf -- $B --steps 5 > "$R/gpurun_out/prof_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$R/gpurun_out/prof_write" -o pw -- $B --steps 5 > "$R/gpurun_out/prof_write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$R/gpurun_out/pmc_mfma" -o pm -- $B --steps 5 > "$R/gpurun_out/pmc_mfma.log" 2>&1
cd "$R" && python3 tools/export_profiles.py gpurun_out "$OUT" f16x3 > "$OUT/export_f16x3.log" 2>&1
# the training step and the dense-grid stages: kernel tables
cd /tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_train" -o kt -- python3 $R/bench.py --workload train --steps 15 > "$R/gpurun_out/prof_train.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_train1024" -o kt -- python3 $R/bench.py --workload train --rays 1024 --steps 40 > "$R/gpurun_out/prof_train1024.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_grid" -o kt -- python3 $R/bench.py --workload grid-stages > "$R/gpurun_out/prof_grid.log" 2>&1
# round 5: the step issued as the reference trainer's own call sequence, and the GPU's idle gaps under either issue path
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_dropin" -o kt -- python3 $R/tools/drive_step.py 1024 100 drop_in > "$R/gpurun_out/prof_dropin.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_onecall" -o kt -- python3 $R/tools/drive_step.py 1024 100 one_call > "$R/gpurun_out/prof_onecall.log" 2>&1
# ... and ONE 4096-ray step on a time axis (start, end, queue of every kernel: what runs beside what)
rocprofv3 --kernel-trace -d "$R/gpurun_out/prof_tl4096" -o kt -- python3 $R/tools/drive_step.py 4096 30 one_call > "$R/gpurun_out/prof_tl4096.log" 2>&1
cd "$R"
db=$(find gpurun_out/prof_tl4096 -name "*_results.db" | head -1)
{ tail -1 gpurun_out/prof_tl4096.log; python3 tools/trace_timeline.py "$db" vfn_train_prep 0; } > "$OUT/step4096_onecall_timeline.txt"
rm -rf gpurun_out/prof_tl4096
for m in dropin onecall; do
  db=$(find gpurun_out/prof_$m -name "*_results.db" | head -1)
  { tail -1 gpurun_out/prof_$m.log; python3 tools/trace_gaps.py "$db" 600 12; python3 tools/topk.py "$db" 25; } > "$OUT/step1024_${m}_trace.txt"
done
rm -rf gpurun_out/prof_dropin gpurun_out/prof_onecall
python3 tools/topk.py gpurun_out/prof_train/kt_results.db 30 > "$OUT/train_kernel_stats.txt"
python3 tools/topk.py gpurun_out/prof_train1024/kt_results.db 30 > "$OUT/train1024_kernel_stats.txt"
python3 tools/topk.py gpurun_out/prof_grid/kt_results.db 30 > "$OUT/grid_stage_stats.txt"
grep '^{"metric"' gpurun_out/prof_train.log | tail -1 > "$OUT/bench_train_profiled.json"     # (the bench line of the profiled run, among the profiler's own log lines)
ls -la "$OUT"
# the raw rocprofv3 outputs are scratch (and would push gpurun_out/ past what is merged back)
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_mfma gpurun_out/prof_train gpurun_out/prof_train1024 gpurun_out/prof_grid
