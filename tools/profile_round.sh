#!/bin/bash
# rocprofv3 capture on the GPU box (ROUND=r05 by default): for each workload one kernel-trace pass and one --pmc pass per counter set
# (counters only alongside --kernel-trace), condensed on the box by tools/summarize_profile_round.py into
# gpurun_out/${ROUND}prof/<workload>_summary.md and ${ROUND}_pmc.json (copy both into profiles/).
#   usage: [ROUND=r05] [PMC=0] [PACKETS=1e8] bash tools/profile_round.sh [workload ...]      workloads: car car1 oct_lucy oct_img vor vor1 amr sph
#   (<name>1 = the same with one slot pool: kernels do not overlap, per-kernel durations divide cleanly; PMC=0: trace pass only)
set -u
export ROUND=${ROUND:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${ROUND}prof; mkdir -p $OUT
# the box starts without gpurun_out/: continue from the committed summary so that a partial re-run keeps the other workloads
[ -f $OUT/${ROUND}_pmc.json ] || cp $REPO/profiles/${ROUND}_pmc.json $OUT/${ROUND}_pmc.json 2>/dev/null
export TMPDIR=/tmp
cd /tmp
WL=${@:-car car1 oct_lucy oct_img vor amr}
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
      "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_THREAD_CYCLES_VALU"
      "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
for w in $WL; do
  case $w in
    car)  CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras" ;;
    car1) CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --option tile_pools=1" ;;
    vor1|oct_lucy1|oct_img1|amr1|sph1) CMD="python $REPO/tools/workload.py ${w%1} ${PACKETS:-1e8} tile_pools=1" ;;
    *)    CMD="python $REPO/tools/workload.py $w ${PACKETS:-1e8} ${WOPTS:-}" ;;
  esac
  D=$OUT/raw_$w; rm -rf $D; mkdir -p $D
  echo "== $w: kernel trace"
  timeout 900 rocprofv3 --kernel-trace --stats -d $D/trace -o t -- $CMD > $D/trace.log 2>&1
  grep -h "PROFILE_TOTALS\|^{" $D/trace.log | tail -1 | cut -c1-300
  if [ "${PMC:-1}" != "0" ] && [ "${w%1}" == "$w" ]; then
    i=0
    for set in "${SETS[@]}"; do
      i=$((i+1))
      timeout 900 rocprofv3 --pmc $set --kernel-trace -d $D/pmc_$i -o pmc -- $CMD > $D/pmc_$i.log 2>&1 || echo "pmc pass '$set' failed"
    done
  fi
  python $REPO/tools/summarize_profile_round.py $w $D $OUT && rm -rf $D/trace $D/pmc_*/
done
ls -la $OUT
