#!/bin/bash
# What each part of the fused stage's wavefront costs, by leaving it out ("or the itemised table").
#   bash tools/ablate_fused.sh build     (build container) libraries with -DSS_FUSED_ABLATE=<mask> -> tools/ablate_lib/libss_ab<mask>.so
#   bash tools/ablate_fused.sh run       (GPU box)         the fused launch's mean HIP-event span in the bench step, per library
# Masks: ss_walks.hpp.  The ablated builds compute WRONG tables by construction; only their timing is used.
cd "$(dirname "$0")/.."
MASKS="0 1 2 4 8 12 16 32 60"
if [ "$1" = build ]; then
  mkdir -p tools/ablate_lib
  cd subgraph-sketching_amd/csrc
  OTHERS=$(ls build/*.o | grep -v ss_fused_hop.o)
  for m in $MASKS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I ../../include -DSS_FUSED_ABLATE=$m -c ss_fused_hop.hip -o /tmp/fused_ab$m.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate_lib/libss_ab$m.so $OTHERS /tmp/fused_ab$m.o || exit 1
  done
  ls -la ../../tools/ablate_lib
  exit 0
fi
for cfg in collab citation2; do
for m in $MASKS; do
  SS_LIB=$PWD/tools/ablate_lib/libss_ab$m.so python bench.py --no-cpu-baseline --no-secondary --sustain-seconds 0 --settle-seconds 0.5 --config $cfg --steps $([ $cfg = collab ] && echo 50 || echo 5) --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']['fused_first_hop_hll_hop']
print('$cfg mask $m: fused launch %.1f us, step %.4f ms' % (k['mean_launch_ms'] * 1e3, d['ms_per_step']))"
done
done
