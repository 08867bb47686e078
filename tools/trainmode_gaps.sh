#!/bin/bash
# The batch-statistics training step (the non-default regime) on a time axis: how much of the step is the GPU idle (host-bound issue)?
#   bash tools/trainmode_gaps.sh > gpurun_out/r06/trainmode_gaps.txt
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/tm
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload train --batch-statistics --steps 10 --warmup 4 --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('un-profiled: training-mode step', d['ms_per_step'], 'ms')"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tm -o kt -- python3 $R/bench.py --workload train --batch-statistics --steps 10 --warmup 4 --no-parity > $R/gpurun_out/tm.log 2>&1
grep '^{"metric"' $R/gpurun_out/tm.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('under the kernel trace:', d['ms_per_step'], 'ms')"
cd $R
db=$(find gpurun_out/tm -name 'kt_results.db' | head -1)
python3 tools/trace_gaps.py "$db" 3000 14
python3 tools/topk.py "$db" 28
rm -rf gpurun_out/tm gpurun_out/tm.log
