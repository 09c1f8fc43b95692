#!/bin/bash
# time tools/octree_bench.py with every tuning variant under build/variants (GEOM=1 builds)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for f in build/variants/*.so; do
  echo "== $(basename $f .so) $@"
  HYP_LIB=$REPO/$f timeout 300 python tools/octree_bench.py 1e7 "$@" 2>&1 | grep "^final"
done
