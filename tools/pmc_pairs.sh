#!/bin/bash
# usage (GPU box): bash tools/pmc_pairs.sh -- VALUBusy / MemUnitBusy / MemUnitStalled / MeanOccupancyPerCU of the query kernel on
# HBM-resident tables (citation2 size at h = 3, ppa size at h = 2; 4 M pairs per launch), one rocprofv3 --pmc pass per metric.
# round 2: h = 3: VALUBusy 42 %, 11.2 waves per CU, MemUnitStalled 0.45 % at 5.45 TB/s (HBM streams 6.3): within 14 % of what the
# memory can deliver for these gathers; h = 2: VALUBusy 52 %, 14.7 waves per CU at 6.84 TB/s.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for M in VALUBusy MemUnitBusy MemUnitStalled MeanOccupancyPerCU; do
  rocprofv3 --pmc $M --kernel-trace --output-format csv -d $R/gpurun_out/pmcp -o p_$M -- python $R/tools/probe_pairs.py --nodes 2927963 --hops 3 --batches 4194304 > /dev/null 2>&1
  rocprofv3 --pmc $M --kernel-trace --output-format csv -d $R/gpurun_out/pmcp2 -o p_$M -- python $R/tools/probe_pairs.py --nodes 576289 --hops 2 --batches 4194304 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for d in ("pmcp","pmcp2"):
    for path in sorted(glob.glob("$R/gpurun_out/%s/*_counter_collection.csv" % d)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "pair_features" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d, k, round(sum(v)/len(v),2))
PY
