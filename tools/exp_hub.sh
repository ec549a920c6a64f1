#!/bin/bash
# experiment: collab-size graphs with rank^-0.5 / rank^-0.9 endpoints (and the uniform graph), ms per step and per-family
# HIP-event times (bench.py kernel table): hub units hosted by the row launches (default) against launches of their own
# (SS_HUB_LAUNCHES=1)
cd $GRAFT_REPO_ROOT
run() { python bench.py --graph $3 --alpha $1 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$2 $3 alpha=$1', round(d['ms_per_step'],4), {n:round(v['mean_launch_ms']*1e3,1) for n,v in k.items() if isinstance(v,dict)})"; }
for a in 0.5 0.9; do
  run $a hosted powerlaw
  SS_HUB_LAUNCHES=1 run $a launches powerlaw
done
run 0.5 hosted uniform
SS_HUB_LAUNCHES=1 run 0.5 launches uniform
