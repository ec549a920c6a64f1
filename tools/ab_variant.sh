#!/bin/bash
# usage (GPU box): bash tools/ab_variant.sh <unit[,unit...]> "<flags A>" "<flags B>" ... -- <command>
# Rebuilds the named translation unit(s) of the library with each -D flag set in turn and runs the command after each build, then
# restores the default build: an A/B/A on one box, same clocks (kernel experiments behind #ifdef switches).
UNIT=$1; shift
VARIANTS=()
while [ "$1" != "--" ]; do VARIANTS+=("$1"); shift; done
shift
for v in "${VARIANTS[@]}"; do
  for u in ${UNIT//,/ }; do rm -f $GRAFT_REPO_ROOT/subgraph-sketching_amd/csrc/build/$u.o; done
  SS_EXTRA_FLAGS="$v" bash $GRAFT_REPO_ROOT/subgraph-sketching_amd/csrc/build.sh > /dev/null || exit 1
  echo "=== $UNIT [$v]"
  "$@"
done
for u in ${UNIT//,/ }; do rm -f $GRAFT_REPO_ROOT/subgraph-sketching_amd/csrc/build/$u.o; done
bash $GRAFT_REPO_ROOT/subgraph-sketching_amd/csrc/build.sh > /dev/null
