#!/bin/bash
# Build libsubgraph_sketch.so for gfx950 (cross-compiles without a GPU).  In-tree output so the
# library travels with the repo snapshot to the GPU box.
set -e
cd "$(dirname "$0")"
OUT=../libsubgraph_sketch.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I ../../include"
OBJS=""
for f in ss_init ss_csr ss_propagate ss_first_hop ss_count ss_pairs ss_heuristics ss_spmm ss_api; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ ss_common.hpp -nt build/$f.o ] || [ ss_walks.hpp -nt build/$f.o ] || [ ../../include/subgraph_sketch.h -nt build/$f.o ]; then
    mkdir -p build
    hipcc $FLAGS -c $f.hip -o build/$f.o &
  fi
  OBJS="$OBJS build/$f.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
echo "built $(realpath $OUT)"
