#!/bin/bash
# VERDICT r05 next 4, the measured attempt at the 2.57 GB the chain writes per launch: what would the weight-gradient phase gain if the
# ReLU-masked half of every dY slot never crossed HBM?  An UPPER BOUND without writing the compaction: a build of csrc/vfn_dwf.hip that does
# not request every second dY piece (-DVFN_DWF_PROBE_HALF_DY: 25 % of the operand bytes gone, no decode work, wrong results) against the
# product build, on one box: the kernel alone (tools/bench_dwf_shapes.py) and the whole training step (bench.py --workload train).
#     (build container)  bash tools/build_unit_variant.sh half_dy vfn_dwf "-DVFN_DWF_PROBE_HALF_DY"
#     (GPU box)          bash tools/probe_dwf_half_dy.sh > gpurun_out/r06/probe_dwf_half_dy.txt
R=$(cd "$(dirname "$0")/.." && pwd)
for v in product half_dy product half_dy; do
  lib=""; [ $v = half_dy ] && lib="$R/tools/micro/libvfn_half_dy.so"
  echo "== $v"
  VFN_LIB=$lib python3 $R/tools/bench_dwf_shapes.py 2>/dev/null | head -1
  VFN_LIB=$lib python3 $R/bench.py --workload train --steps 30 --warmup 8 --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   training step', d['ms_per_step'], 'ms  (final loss', d['final_loss'], ')')"
done
