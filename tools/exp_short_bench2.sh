#!/bin/bash
# the driver's command line (`--steps 20 --warmup 5`) with the settle phase: timed region against the sustained rate, with and
# without the HIP events around the dominant kernel (SS_BENCH_EXP_NO_EVENTS=1: experiment only -- the line then has no roofline)
cd $GRAFT_REPO_ROOT
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --sustain-seconds 1 --no-kernel-table 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['ms_per_step'],4), 'sustained', round(d['sustained']['ms_per_step'],4), 'dom launch ms', round(d['roofline'].get('mean_launch_ms') or 0,4), 'frac', round(d['roofline'].get('frac') or 0,4))"; }
for i in 1 2 3; do run events; done
for i in 1 2; do SS_BENCH_EXP_NO_EVENTS=1 run noevents; done
python bench.py --api buddy --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('buddy', round(d['ms_per_step'],4), d['roofline']['kernel'][:40], round(d['roofline']['mean_launch_ms'],4), round(d['roofline']['frac'],3))"
python bench.py --api elph --batch 2048 --no-secondary --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('elph', round(d['ms_per_step'],4), d['roofline']['kernel'][:40], round(d['roofline']['mean_launch_ms'],4), round(d['roofline']['frac'],3))"
