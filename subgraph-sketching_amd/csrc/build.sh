#!/bin/bash
# Build libsubgraph_sketch.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# library travels with the repo snapshot to the GPU box.
set -e
cd "$(dirname "$0")"
OUT=../libsubgraph_sketch.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I ../../include $SS_EXTRA_FLAGS"
UNITS="ss_init ss_digest ss_csr ss_propagate ss_first_hop ss_fused_hop ss_count ss_pairs ss_heuristics ss_spmm ss_api ss_debug"
OBJS=""
PIDS=""
mkdir -p build
for f in $UNITS; do
  stale=0
  [ -f build/$f.o ] || stale=1
  for dep in $f.hip ss_common.hpp ss_walks.hpp ss_hub.hpp ../../include/subgraph_sketch.h ../../include/subgraph_sketch_debug.h build.sh; do
    [ -e $dep ] && [ $dep -nt build/$f.o ] && stale=1
  done
  if [ $stale = 1 ]; then
    rm -f build/$f.o  # a failed compile must not leave an older object for the link step to pick up
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS build/$f.o"
done
for pid in $PIDS; do  # a bare `wait` returns 0 whatever the jobs did: collect every job's own status
  wait $pid || { echo "build.sh: a compile job failed" >&2; exit 1; }
done
for f in $UNITS; do [ -f build/$f.o ] || { echo "build.sh: build/$f.o missing" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(realpath $OUT)"
