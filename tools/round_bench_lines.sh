#!/bin/bash
# usage (GPU box, via gpurun): bash tools/round_bench_lines.sh <prefix>   e.g. round2_v5
# the unprofiled bench lines of a round: gpurun_out/<prefix>_bench_<name>_full.json (copy them into profiles/ afterwards)
P=$1
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python bench.py "$@" 2>/dev/null | tail -1 > $O/${P}_bench_${name}_full.json; python -c "import json; d=json.load(open('$O/${P}_bench_${name}_full.json')); print('$name', round(d['ms_per_step'],4), d['value'])"; }
python bench.py 2>/dev/null | tail -1 > $O/${P}_bench_full.json; python -c "import json; d=json.load(open('$O/${P}_bench_full.json')); print('default', round(d['ms_per_step'],4), d['value'], d['roofline']['frac'], d['step_roofline']['frac'], d['cpu_baseline']['value'])"
run elph --no-cpu-baseline --api elph
run elph2048 --no-cpu-baseline --api elph --batch 2048
run buddy --no-cpu-baseline --api buddy
run cora --no-cpu-baseline --config cora
run ppa --no-cpu-baseline --config ppa --steps 10 --warmup 2
run citation2 --no-cpu-baseline --config citation2 --steps 5 --warmup 2
run powerlaw --no-cpu-baseline --graph powerlaw --alpha 0.5
run powerlaw09 --no-cpu-baseline --graph powerlaw --alpha 0.9
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --scaling strong 2>/dev/null | tail -1 > $O/${P}_bench_strong1_full.json; python -c "import json; d=json.load(open('$O/${P}_bench_strong1_full.json')); print('strong1', round(d['ms_per_step'],4))"
