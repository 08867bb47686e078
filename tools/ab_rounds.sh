#!/bin/bash
# Same-box A/B of the three-products-everywhere render step: the round-3 tree (extracted to ab_r03/ by `git archive 55ff540 | tar -x -C ab_r03`
# and built there) against this tree, interleaved on one GPU box.  Usage (on the GPU box): bash tools/ab_rounds.sh [reps] > gpurun_out/r04/ab.txt
R=$(pwd)
REPS=${1:-3}
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline",{}); print(sys.argv[1], d["value"], d["ms_per_step"], r.get("kernel_ms", r.get("launch_ms")), r.get("frac"), r.get("effective_clock_ghz"))'
for i in $(seq $REPS); do
  (cd $R/ab_r03 && python bench.py --colour-products 3 --no-cpu-baseline --no-train --no-fp32-equivalent --steps 20 --warmup 5 2>/dev/null | python -c "$pick" r03_three_products_random_weights)
  (cd $R && python bench.py --weights random --no-live-traffic --no-cpu-baseline --no-train --no-two-product-leg --no-other-scene-leg --steps 20 --warmup 5 2>/dev/null | python -c "$pick" r04_three_products_random_weights)
  (cd $R && python bench.py --no-live-traffic --no-cpu-baseline --no-train --no-two-product-leg --no-other-scene-leg --steps 20 --warmup 5 2>/dev/null | python -c "$pick" r04_three_products_trained_weights)
done
