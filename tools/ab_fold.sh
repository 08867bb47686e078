#!/bin/bash
# Same-box A/B of the training-mode step with / without the activation fold (batchstat.FOLD_ACTIVATIONS):  bash tools/ab_fold.sh
R=$(cd "$(dirname "$0")/.." && pwd)
for i in 1 2 3; do
  for f in 1 0; do
    VFN_FOLD_ACTIVATIONS=$f python3 $R/bench.py --workload train --batch-statistics --train-weights random --steps 10 --warmup 4 --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fold $f: training-mode step', d['ms_per_step'], 'ms')"
  done
done
