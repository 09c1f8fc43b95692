#!/bin/bash
# rocprofv3 kernel trace of one command; prints the per-kernel table.  usage: r03_trace.sh tag <command...>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
tail -3 $OUT/trace.log
python - <<PY | tee $OUT/kernels.txt
import sqlite3, glob
for db in glob.glob("$OUT/trace/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-70s %6d %12.1f ms %10.1f us %6.2f" % (r[0].split("(")[0][:70], r[1], r[2]/1e3, r[3], r[4]))
PY
rm -rf $OUT/trace
