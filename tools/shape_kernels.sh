cd /tmp && export TMPDIR=/tmp
for sh in 64,8,2 128,6,2 256,8,2 128,10,2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shape_$sh -o s -- python $GRAFT_REPO_ROOT/tools/probe_shapes.py --only $sh > /dev/null 2>&1
echo "== $sh"
python - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/shape_$sh/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r["Name"][:100], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
