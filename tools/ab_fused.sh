#!/bin/bash
# (GPU box) fused-stage variants: step time + the fused kernel's mean launch time on the bench graph, then citation2 / ppa size
run() { python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --sustain-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json, os
d = json.loads(sys.stdin.read())
print('   ', 'wg/cu', os.environ.get('SS_FUSED_WG_PER_CU', 'default'), '$*', 'step', round(d['ms_per_step'], 4), 'fused', round(d['kernels']['fused_first_hop_hll_hop']['mean_launch_ms'] * 1e3, 1), 'us')"; }
for w in ${WGS:-12 15 20 25 30 40}; do SS_FUSED_WG_PER_CU=$w run; done
for w in ${WGS_BIG:-15 20 30}; do SS_FUSED_WG_PER_CU=$w run --config citation2 --steps 5 --warmup 2; SS_FUSED_WG_PER_CU=$w run --config ppa --steps 5 --warmup 2; done
