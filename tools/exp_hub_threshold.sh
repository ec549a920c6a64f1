cd $GRAFT_REPO_ROOT
run() { SS_HUB_THRESHOLD=$1 python bench.py --graph powerlaw --alpha $2 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('thr=$1 alpha=$2', round(d['ms_per_step'],4), {n:round(v['mean_launch_ms']*1e3,1) for n,v in k.items() if isinstance(v,dict)})"; }
for a in 0.5 0.9; do for t in 48 64 96 143 256; do run $t $a; done; done
