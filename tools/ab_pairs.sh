#!/bin/bash
# (GPU box) query-kernel variants: the bench step's pair kernel span, the BUDDY precompute at collab size (h = 2) and a citation2-size (h = 3) link set
run() { python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --sustain-seconds 0 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d.get('kernels', {}).get('pair_features', {})
print('   ', '$*', 'step', round(d['ms_per_step'], 4), 'pair kernel', round(k.get('mean_launch_ms', 0) * 1e3, 1), 'us x', k.get('launches_per_step'))"; }
run
run
run --api buddy --steps 10 --warmup 3
run --api buddy --config citation2 --buddy-links 8000000 --steps 3 --warmup 1
run --api buddy --config citation2 --buddy-links 8000000 --buddy-negs 1000 --steps 3 --warmup 1
