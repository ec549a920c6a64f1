#!/bin/bash
# experiment: collab-size graphs with rank^-0.5 / rank^-0.9 endpoints, per-family HIP-event times (bench.py kernel table);
# SS_HUB_OVERLAP=0: the hub passes in order on the launch stream
cd $GRAFT_REPO_ROOT
run() { python bench.py --graph $3 --alpha $1 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$2 $3 alpha=$1', round(d['ms_per_step'],4), {n:round(v['mean_launch_ms']*1e3,1) for n,v in k.items() if isinstance(v,dict)})"; }
for a in 0.5 0.9; do
  run $a overlap powerlaw
  SS_HUB_OVERLAP=0 run $a inorder powerlaw
done
run 0.5 overlap uniform
