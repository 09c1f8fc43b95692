#!/bin/bash
# round-2 rocprofv3 capture of the benchmark on the GPU box (gpurun):
#   r02_serial : kernel trace, one slot pool (kernels do not overlap: per-kernel averages divide cleanly)
#   r02        : kernel trace + PMC passes of the default schedule (three pools on three streams)
# Raw rocpd databases are condensed on the box by tools/summarize_tiled.py (they exceed what gpurun copies back);
# copy gpurun_out/r02*_summary.md / _pmc.json into profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run_trace() {   # tag, bench args
  local TAG=$1; shift
  local OUT=$REPO/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
  echo "== kernel trace $TAG"; rocprofv3 --kernel-trace --stats -d $OUT/trace -o lucy -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
}
run_pmc() {     # tag, bench args
  local TAG=$1; shift
  local OUT=$REPO/gpurun_out/prof_$TAG
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
             "GRBM_GUI_ACTIVE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-60)
    echo "== pmc $set"
    rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_$name -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_$name.log 2>&1 || echo "pmc $set failed"
  done
}
run_trace r02_serial --option tile_pools=1
POOLS_NOTE="one pool: kernels run one after the other" SUMMARY_DIR=$REPO/gpurun_out python $REPO/tools/summarize_tiled.py r02_serial > /dev/null && rm -rf $REPO/gpurun_out/prof_r02_serial/trace
run_trace r02
run_pmc r02
SUMMARY_DIR=$REPO/gpurun_out python $REPO/tools/summarize_tiled.py r02 > /dev/null && rm -rf $REPO/gpurun_out/prof_r02/trace $REPO/gpurun_out/prof_r02/pmc_*/
ls $REPO/gpurun_out | head -30
