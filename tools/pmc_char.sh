#!/bin/bash
# usage (GPU box): bash tools/pmc_char.sh <tag> -> one rocprofv3 --pmc pass per derived metric over a short bench run,
# per-kernel averages printed and written to gpurun_out/pmc_char_<tag>.txt (characterisation only, not the roofline traffic)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for M in VALUBusy VALUUtilization MemUnitBusy MemUnitStalled LdsBankConflict L2CacheHit MeanOccupancyPerCU SALUBusy; do
  rocprofv3 --pmc $M --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_$TAG -o ${TAG}_$M -- python $R/bench.py --no-cpu-baseline --no-secondary --no-kernel-table --sustain-seconds 0 --steps 3 --warmup 1 "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob, os, collections
out = collections.defaultdict(dict)
for path in sorted(glob.glob("$R/gpurun_out/pmcc_$TAG/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    name = None
    for r in csv.DictReader(open(path)):
        name = r["Counter_Name"]
        acc[r["Kernel_Name"].split("(")[0][:48]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k][name] = sum(v) / len(v)
metrics = sorted({m for v in out.values() for m in v})
lines = ["kernel".ljust(50) + " ".join(m[:14].rjust(15) for m in metrics)]
for k, v in sorted(out.items(), key=lambda kv: -len(kv[1])):
    if k.startswith("void at::") or k.startswith("__amd"):
        continue
    lines.append(k.ljust(50) + " ".join((f"{v[m]:.2f}" if m in v else "-").rjust(15) for m in metrics))
open("$R/gpurun_out/pmc_char_$TAG.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
