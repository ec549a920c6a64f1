cd $GRAFT_REPO_ROOT
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --settle-seconds $1 --no-secondary --no-cpu-baseline --sustain-seconds 1 --no-kernel-table 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('settle=$1', round(d['ms_per_step'],4), 'sustained', round(d['sustained']['ms_per_step'],4), 'settle_steps', d['settle_steps'])"; }
run 2.0; run 2.0; run 0.5; run 0.5
