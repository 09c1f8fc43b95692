#!/bin/bash
# serial kernel trace of tools/octree_bench.py (BASELINE configs[3]): per-kernel durations of the imaging paths
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${TAG:-r02_octree}; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/tools/octree_bench.py ${1:-1e7} > $OUT/trace.log 2>&1
grep -v "^[WE]2026" $OUT/trace.log
python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/trace/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
        print("%-60s %6d %10.1f ms %9.1f us %6.2f" % (r[0].split("(")[0][:60], r[1], r[2]/1e3, r[3], r[4]))
PY
rm -rf $OUT/trace
