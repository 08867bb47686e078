#!/bin/bash
# rocprofv3 passes over the training step (kernel trace, FETCH_SIZE, WRITE_SIZE, MFMA-busy counters; each in its OWN pass) ->
# gpurun_out/export_train/{bench_kernel_stats_train.csv, traffic_train.json, pmc_mfma_train.json}:   bash tools/profile_train.sh
set -u
# VFN_TRAIN_ARGS / VFN_TRAIN_TAG: extra bench.py arguments and the tag of the exported files (e.g. "--train-products 1" / train_p1)
TAG=${VFN_TRAIN_TAG:-train}
R=$(pwd); OUT=gpurun_out/export_$TAG; mkdir -p $R/$OUT $R/gpurun_out/pt
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload train ${VFN_TRAIN_ARGS:-}"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pt/prof_kt -o kt -- $B --steps 15 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pt/prof_fetch -o pf -- $B --steps 6 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pt/prof_write -o pw -- $B --steps 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pt/pmc_mfma -o pm -- $B --steps 6 > /dev/null 2>&1
cd $R && python3 tools/export_profiles.py gpurun_out/pt $OUT $TAG > $OUT/export_$TAG.log 2>&1
rm -rf gpurun_out/pt
ls $OUT
