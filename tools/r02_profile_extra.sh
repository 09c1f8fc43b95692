#!/bin/bash
# rocprofv3 of the configs[3] (octree + imaging) and configs[4] (real Voronoi tessellation) runs: kernel trace + SQ / TCC counters
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r02_extra; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for job in "octree:tools/octree_bench.py 1e7" "voronoi:tools/voronoi_big_bench.py 1e7"; do
  name=${job%%:*}; cmd=${job#*:}
  rocprofv3 --kernel-trace --stats -d $OUT/${name}_trace -o t -- python $REPO/$cmd > $OUT/${name}_trace.log 2>&1
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
    n2=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --kernel-trace -d $OUT/${name}_pmc_$n2 -o pmc -- python $REPO/$cmd > $OUT/${name}_pmc_$n2.log 2>&1 || echo "pmc $set failed"
  done
done
python $REPO/tools/summarize_extra.py $OUT $REPO/gpurun_out/r02_extra_summary.md
rm -rf $OUT/*_trace $OUT/*_pmc_*/
