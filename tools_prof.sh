#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools_prof.sh <tag> [bench args]
# rocprofv3 kernel trace of bench.py -> gpurun_out/prof_<tag>/ (csv), summarised by tools_prof_summary.py
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
ls $R/gpurun_out/prof_$TAG
