# A/B of where the supervision batch's forward runs inside vfn_train_step (VFN_TRAIN_STREAMS: 2 = behind the fine pass's forward, 3 = beside the
# proposal pass as in round 4): wall ms per step of the four issue paths, two repetitions, one box
for rep in 1 2; do for st in 2 3; do for n in 4096 1024; do
VFN_TRAIN_STREAMS=$st python tools/host_profile.py $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('streams',$st,'rays',$n,{k:v['wall_ms_per_step'] for k,v in d.items() if isinstance(v,dict) and 'wall_ms_per_step' in v})"
done; done; done
