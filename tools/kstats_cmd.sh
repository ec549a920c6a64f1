#!/bin/bash
# usage (GPU box): bash tools/kstats_cmd.sh <tag> <command...> -> rocprofv3 kernel-trace stats of any command, per-kernel average us
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- "$@" > $R/gpurun_out/prof_${TAG}_cmd.log 2>&1
python - <<PY
import csv
t = "$TAG"
for r in csv.DictReader(open(f"$R/gpurun_out/prof_{t}/{t}_kernel_stats.csv")):
    if float(r["AverageNs"]) > 2000 and "ss::" in r["Name"]:
        print("  ", r["Name"][:72].ljust(72), r["Calls"].rjust(5), str(round(float(r["AverageNs"]) / 1e3, 1)).rjust(9))
PY
