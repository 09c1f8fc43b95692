#!/bin/bash
# round-2 baseline: RCCL path at world size 1, serialised (one pool) kernel trace, normal bench
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02a; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
echo "== torchrun ws1"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --photons 2e6 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/torchrun_ws1.log 2>&1; echo rc $?; tail -3 $OUT/torchrun_ws1.log
echo "== bench"; timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log
cd /tmp
echo "== serial trace"; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o t -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --option tile_pools=1 > $OUT/trace_serial.log 2>&1; tail -1 $OUT/trace_serial.log
python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/trace_serial/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-60s %6d %14d %12.0f %6.2f" % (r[0].split("(")[0][:60], r[1], r[2], r[3], r[4]))
PY
rm -rf $OUT/trace_serial
