#!/bin/bash
# experiment: grid of the query kernel at the bench's batch (65 536 pairs; 768 workgroups are resident at three wavefronts per SIMD)
cd $GRAFT_REPO_ROOT
run() { SS_PAIR_PER_GROUP=$1 SS_PAIR_GRID=$2 python bench.py --no-secondary --no-cpu-baseline --sustain-seconds 0 --settle-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('per_group=$1 grid=$2', round(d['ms_per_step'],4), 'pair_features us', round(k['pair_features']['mean_launch_ms']*1e3,2))"; }
run 2 2048; run 2 1536; run 2 1024; run 2 768; run 1 4096; run 1 3072; run 1 2304; run 1 1536; run 3 1536; run 2 2048
