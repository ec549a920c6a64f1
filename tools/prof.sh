#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof.sh <tag> [bench args]
# 1) rocprofv3 --kernel-trace --stats of bench.py  2) two PMC passes (FETCH_SIZE, WRITE_SIZE) in their own runs.
# (--settle-seconds 0: the untimed settle phase of a plain bench run would only add thousands of identical launches to the trace)
# Outputs land in gpurun_out/prof_<tag>/ ; tools/summarise_prof.py turns them into profiles/<tag>_*.{csv,json}
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --no-cpu-baseline --no-kernel-table --no-secondary --sustain-seconds 0 --settle-seconds 0 "$@" > $R/gpurun_out/prof_${TAG}_bench.log 2>&1
if [ -z "$NO_PMC" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -o ${TAG}_fetch -- python $R/bench.py --no-cpu-baseline --no-kernel-table --no-secondary --sustain-seconds 0 --settle-seconds 0 "$@" --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -o ${TAG}_write -- python $R/bench.py --no-cpu-baseline --no-kernel-table --no-secondary --sustain-seconds 0 --settle-seconds 0 "$@" --steps 3 --warmup 1 > /dev/null 2>&1
fi
ls $R/gpurun_out/prof_$TAG
