#!/bin/bash
# time the tuning variants under build/variants (1e8 packets) and trace each one serialised
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${TAG:-r02v}; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
python tools/variants.py run 1e8 "$@" 2>&1 | grep '^{' | tee $OUT/variants.log
python tools/variants.py run 1e8 tile_pools=1 "$@" 2>&1 | grep '^{' | tee -a $OUT/variants.log
if [ -n "${TRACE:-}" ]; then
cd /tmp
for f in $REPO/build/variants/*.so; do
  n=$(basename $f .so)
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$n -o t -- python $REPO/tools/variants.py one $f 1e8 tile_pools=1 "$@" > $OUT/trace_$n.log 2>&1
  echo "-- $n"
  python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/trace_$n/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 7"):
        print("%-50s %6d %10.1f ms %9.1f us %6.2f" % (r[0].split("(")[0][:50], r[1], r[2]/1e3, r[3], r[4]))
PY
  rm -rf $OUT/trace_$n
done
fi
