#!/bin/bash
# rocprofv3 capture of the benchmark (run on the GPU box through gpurun).
# usage: tools/profile.sh <tag> [bench args...]
# Writes raw output under gpurun_out/prof_<tag>/ ; summaries are extracted by
# tools/summarize_profile.py into profiles/.
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 2 --warmup 1 --photons 2e7 --no-cpu-baseline $*"
echo "== kernel trace" ; rocprofv3 --kernel-trace --stats -d $OUT/trace -o lucy -- $BENCH > $OUT/trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-60)
  echo "== pmc $set"
  rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1 || echo "pmc $set failed"
done
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
ls -R $OUT | head -60
# condense on the box (the raw rocpd databases of a 1e8-packet tiled run exceed what gpurun copies back)
if [ -n "${SUMMARIZE:-}" ]; then
  SUMMARY_DIR=$REPO/gpurun_out python $REPO/tools/$SUMMARIZE $TAG > /dev/null && rm -rf $OUT/trace $OUT/pmc_*
fi
