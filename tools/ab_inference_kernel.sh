B="python bench.py --no-cpu-baseline --no-train --no-two-product-leg --no-other-scene-leg --no-live-traffic --no-shipped-rows --steps 40 --warmup 10"
for rep in 1 2 3; do for v in base new; do for w in trained random; do
if [ $v = base ]; then export VFN_LIB=$PWD/tools/micro/libvfn_basemlp16.so; else unset VFN_LIB; fi
$B --weights $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v','$w',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['effective_clock_ghz'])"
done; done; done
