#!/bin/bash
# generic round-2 GPU job: tiled parity tests, bench with option sets, optional serial trace
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02b}; shift
OUT=$REPO/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
for TESTS in "${TESTS:-}" "${TESTS2:-}"; do if [ -n "$TESTS" ]; then
  echo "== tests: $TESTS"; timeout 1500 python -m pytest $TESTS -x -q -m gpu --timeout=300 --timeout_method=thread >> $OUT/tests.log 2>&1; echo rc $?; tail -5 $OUT/tests.log
fi; done
i=0
for opts in "$@"; do
  i=$((i+1))
  echo "== bench $opts"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $opts > $OUT/bench_$i.log 2>&1
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_$i.log") if l.startswith("{")][-1])
    print("   ms_per_step %.1f  kernel %.1f  value %.4g" % (d["ms_per_step"], d["lucy_kernel_ms"], d["value"]))
except Exception as e:
    print("   failed", e); print(open("$OUT/bench_$i.log").read()[-1500:])
PY
done
if [ -n "${TRACE:-}" ]; then
  cd /tmp
  echo "== serial trace $TRACE"; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline $TRACE > $OUT/trace.log 2>&1
  python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/trace/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-60s %6d %12.1f ms %10.1f us %6.2f" % (r[0].split("(")[0][:60], r[1], r[2]/1e3, r[3], r[4]))
PY
  rm -rf $OUT/trace
fi
