#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_tiled
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/tools/tiled_prof.py "$@" > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/trace/*.db"):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-60s calls %6d total %.1f ms avg %.1f us  %.1f%%" % (r[0][:60], r[1], r[2]/1e6, r[3]/1e3, r[4]))
PY
