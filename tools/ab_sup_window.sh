for rep in 1 2; do for st in 2 3; do for n in 4096 1024; do
VFN_TRAIN_STREAMS=$st python tools/host_profile.py $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams',$st,'rays',$n,{k:v['wall_ms_per_step'] for k,v in d.items() if isinstance(v,dict) and 'wall_ms_per_step' in v})"
done; done; done
