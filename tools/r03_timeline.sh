#!/bin/bash
# rocprofv3 kernel trace of one command, reduced to a TIMELINE summary: how long the GPU had no kernel running, how much
# of the time 1 / 2 / 3+ kernels overlapped, and each kernel's share of the busy time.  usage: r03_timeline.sh tag <command...>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o t -- "$@" > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | cut -c1-300
python $REPO/tools/timeline.py $OUT/trace | tee $OUT/timeline.txt
rm -rf $OUT/trace
