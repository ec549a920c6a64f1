#!/bin/bash
# usage (GPU box): bash tools/pmc_cmd.sh <tag> "<metric> <metric> ..." <command...> -> one rocprofv3 --pmc pass per metric over any
# command, per-kernel averages printed (characterisation only, not the roofline traffic)
TAG=$1; shift
METRICS=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for M in $METRICS; do
  rocprofv3 --pmc $M --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_$TAG -o ${TAG}_$M -- "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob, os, collections
out = collections.defaultdict(dict)
for path in sorted(glob.glob("$R/gpurun_out/pmcc_$TAG/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    name = None
    for r in csv.DictReader(open(path)):
        name = r["Counter_Name"]
        acc[r["Kernel_Name"].replace("void ", "").split("(")[0][:40]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k][name] = sum(v) / len(v)
metrics = sorted({m for v in out.values() for m in v})
lines = ["kernel".ljust(42) + " ".join(m[:16].rjust(17) for m in metrics)]
for k, v in sorted(out.items(), key=lambda kv: -len(kv[1])):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    lines.append(k.ljust(42) + " ".join((f"{v[m]:.2f}" if m in v else "-").rjust(17) for m in metrics))
open("$R/gpurun_out/pmc_cmd_$TAG.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
