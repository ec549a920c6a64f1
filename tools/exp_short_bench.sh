#!/bin/bash
# the driver's command line is `bench.py --gpus 1 --steps 20 --warmup 5`: a 9 ms timed region on a process that has just started.
# How far is that from the steady state, and does the settle phase close the gap?
cd $GRAFT_REPO_ROOT
run() { python bench.py --gpus 1 --steps $1 --warmup $2 --settle-seconds $3 --no-secondary --no-cpu-baseline --sustain-seconds $4 --no-kernel-table 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps=$1 warmup=$2 settle=$3', round(d['ms_per_step'],4), 'sustained', round(d['sustained']['ms_per_step'],4) if d.get('sustained') else None, 'dom launch ms', round(d['roofline']['mean_launch_ms'],4))"; }
for i in 1 2 3; do run 20 5 0 2; done
for i in 1 2 3; do run 20 5 0.5 2; done
run 200 20 0 2
run 200 20 0.5 2
