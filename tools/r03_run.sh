#!/bin/bash
# generic round-3 GPU job: tests, Voronoi / octree / Cartesian timings of the main library and of build/variants/*.so
#   TESTS="tests/a.py tests/b.py::c"  VOR="1e7 [opts]"  VORV="1e7 one [opts]" (variants, single species)  CAR="opts;opts"  OCT=1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03a}
OUT=$REPO/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
if [ -n "${TESTS:-}" ]; then
  echo "== tests: $TESTS"; timeout ${TEST_TIMEOUT:-1500} python -m pytest $TESTS -x -q -m gpu --timeout=600 --timeout_method=thread > $OUT/tests.log 2>&1; echo rc $?; tail -${TAIL:-15} $OUT/tests.log
fi
if [ -n "${VOR:-}" ]; then
  IFS=';' read -ra SETS <<< "$VOR"
  for o in "${SETS[@]}"; do echo "== voronoi $o"; timeout 600 python tools/voronoi_big_bench.py $o 2>&1 | tail -2 | tee -a $OUT/vor.log; done
fi
if [ -n "${VORV:-}" ]; then
  for f in build/variants/*.so; do
    IFS=';' read -ra SETS <<< "$VORV"
    for o in "${SETS[@]}"; do echo "== variant $(basename $f) $o"; HYP_LIB=$f timeout 600 python tools/voronoi_big_bench.py $o 2>&1 | tail -1 | tee -a $OUT/vorv.log; done
  done
fi
if [ -n "${OCT:-}" ]; then echo "== octree"; timeout 600 python tools/octree_bench.py $OCT 2>&1 | tail -12 | tee -a $OUT/oct.log; fi
if [ -n "${CAR:-}" ]; then
  IFS=';' read -ra SETS <<< "$CAR"
  i=0
  for o in "${SETS[@]}"; do
    i=$((i+1)); echo "== bench $o"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $o > $OUT/bench_$i.log 2>&1
    python - <<PY
import json
try:
    d = json.loads([l for l in open("$OUT/bench_$i.log") if l.startswith("{")][-1])
    print("   ms_per_step %.1f  kernel %.1f  value %.4g" % (d["ms_per_step"], d["lucy_kernel_ms"], d["value"]))
except Exception as e:
    print("   failed", e); print(open("$OUT/bench_$i.log").read()[-1500:])
PY
  done
fi
if [ -n "${CMD:-}" ]; then echo "== $CMD"; eval "$CMD" 2>&1 | tail -${TAIL:-30}; fi
