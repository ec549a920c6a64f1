#!/bin/bash
# usage (GPU box): bash tools/kstats.sh <tag> [bench args] -> prints per-kernel average us (kernel-trace only, no PMC)
TAG=$1; shift
NO_PMC=1 bash $GRAFT_REPO_ROOT/tools/prof.sh $TAG "$@" > /dev/null 2>&1
python - <<PY
import csv, re
t = "$TAG"
print(t, re.search(r'"ms_per_step": ([0-9.]+)', open(f"$GRAFT_REPO_ROOT/gpurun_out/prof_{t}_bench.log").read()).group(1), "ms/step under rocprof")
for r in csv.DictReader(open(f"$GRAFT_REPO_ROOT/gpurun_out/prof_{t}/{t}_kernel_stats.csv")):
    if float(r["AverageNs"]) > 3000 and int(r["Calls"]) >= 20:
        print("  ", r["Name"][:64].ljust(64), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
