#!/bin/bash
# per-kernel durations of tools/octree_bench.py for every tuning variant under build/variants (GEOM=1 builds)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for f in $REPO/build/variants/*.so; do
  n=$(basename $f .so)
  echo "== $n $@"
  cd /tmp; rm -rf /tmp/vt_$n
  HYP_LIB=$f timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/vt_$n -o t -- python $REPO/tools/octree_bench.py 1e7 "$@" 2>&1 | grep "^final"
  python - <<PY
import sqlite3, glob
for db in glob.glob("/tmp/vt_$n/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average from top_kernels limit 5"):
        print("   %-50s %4d calls  total %8.1f ms  avg %8.2f ms" % (r[0].split("(")[0][:50], r[1], r[2]/1e3, r[3]/1e3))
PY
done
