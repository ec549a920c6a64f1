#!/bin/bash
# usage (one-GPU box): bash tools/two_ranks_one_gpu.sh [bench args]
# bench.py's N = 2 control flow (weak line + the `strong` figures: sharded links, replicated / row-sharded / peer-write build, exchange probe; the line carries `same_work_speedup`) with
# both ranks on device 0 and gloo instead of RCCL -- the numbers mean nothing (two processes share one GPU, exchanges go through the
# host); what it checks is that every collective of the N > 1 path is entered by every rank with matching shapes.
cd $GRAFT_REPO_ROOT
SS_BENCH_SINGLE_DEVICE=1 SS_BENCH_BACKEND=gloo python \
  bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --sustain-seconds 0 "$@"
