#!/bin/bash
# kernel trace of the brick-tiled Lucy iteration: per-kernel totals and the per-generation timeline
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_tiled
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/tools/tiled_prof.py "$@" > $OUT/trace.log 2>&1
grep tiled $OUT/trace.log
python - <<PY
import sqlite3, glob, collections
for db in glob.glob("$OUT/trace/*.db"):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,start,end,vgpr_count,lds_size,workgroup_x,grid_x from kernels order by start"))
    seg, cur = [], []
    for r in rows:
        cur.append(r)
        if r[0].startswith("finish_kernel"): seg.append(cur); cur = []
    sg = seg[-1]
    d = collections.OrderedDict()
    for n, s, e, vg, lds, wg, grid in sg:
        k = n.split("(")[0][:34]
        x = d.setdefault(k, [0, 0.0, vg, lds, wg]); x[0] += 1; x[1] += (e - s) / 1e6
    span = (sg[-1][2] - sg[0][1]) / 1e6
    busy = sum(v[1] for v in d.values())
    print("last iteration: span %.1f ms, kernels busy %.1f ms (gaps %.1f ms)" % (span, busy, span - busy))
    for k, v in d.items():
        print("  %-36s calls %5d total %8.2f ms  %5.1f%%  vgpr %d lds %d wg %d" % (k, v[0], v[1], 100 * v[1] / span, v[2], v[3], v[4]))
    w = [(e - s) / 1e3 for n, s, e, *_ in sg if n.startswith("void tile_walk")]
    p = [(e - s) / 1e3 for n, s, e, *_ in sg if n.startswith("void tile_prep")]
    print("generations", len(w))
    print("walk us   ", [int(x) for x in w[:40]], "...", [int(x) for x in w[-8:]])
    print("prepare us", [int(x) for x in p[:40]], "...", [int(x) for x in p[-8:]])
PY
