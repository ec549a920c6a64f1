#!/bin/bash
# (GPU box) MinHash table hop variants: the kernel's mean launch time on the three shapes
run() { python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --sustain-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']['minhash_hop']
print('   ', '$*', 'step', round(d['ms_per_step'], 4), 'minhash hop', round(k['mean_launch_ms'] * 1e3, 1), 'us', round(k['frac_of_hbm_peak'], 3))"; }
run
run --config ppa --steps 5 --warmup 2
run --config citation2 --steps 5 --warmup 2
run --config citation2 --steps 5 --warmup 2
