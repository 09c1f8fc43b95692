"""Condense the rocprofv3 output of a brick-tiled bench run (gpurun_out/prof_<tag>/, rocpd sqlite)
into profiles/<tag>_summary.md: per-kernel time per Lucy iteration and PMC counters summed over
all dispatches of each kernel, also expressed per cell crossing.
usage: python tools/summarize_tiled.py <tag>"""
import collections, glob, json, os, sqlite3, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
L = ["# rocprofv3 summary `%s` (brick-tiled Lucy iteration)" % tag, ""]
bench = None
log = os.path.join(src, "trace.log")
if os.path.exists(log):
    js = [l for l in open(log) if l.startswith("{")]
    if js:
        bench = json.loads(js[-1])
        L += ["bench line of the traced run:", "", "```", js[-1].strip(), "```", ""]
n_iter = bench["steps"] + bench["warmup"] if bench else 1
crossings_iter = bench["config"]["crossings_per_packet"] * bench["config"]["packets_per_iteration"] if bench else float("nan")


def short(n):
    return n.split("(")[0].replace("void ", "")[:40]


for db in glob.glob(os.path.join(src, "trace", "*.db")):
    c = sqlite3.connect(db)
    L += ["## kernel stats (`rocprofv3 --kernel-trace --stats`, view `top_kernels`; durations in microseconds)", "",
          "| kernel | calls | total us | average us | % |", "|---|---|---|---|---|"]
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        L.append("| `%s` | %d | %d | %.0f | %.2f |" % (short(r[0]), r[1], r[2], r[3], r[4]))
    rows = list(c.execute("select name,start,end,vgpr_count,lds_size,workgroup_x from kernels order by start"))
    seg, cur = [], []
    for r in rows:
        cur.append(r)
        if r[0].startswith("finish_kernel"):
            seg.append(cur); cur = []
    L += ["", "per Lucy iteration (segments end at `finish_kernel`); `span` = first kernel start to last kernel end:", ""]
    for i, sg in enumerate(seg):
        d = collections.OrderedDict()
        for n, s, e, vg, lds, wg in sg:
            x = d.setdefault(short(n), [0, 0.0, vg, lds, wg]); x[0] += 1; x[1] += (e - s) / 1e6
        span = (sg[-1][2] - sg[0][1]) / 1e6
        if not any(k.startswith("tile_walk") for k in d): continue
        L.append("- iteration %d: span %.1f ms, sum of kernel durations %.1f ms (%s)" %
                 (i, span, sum(v[1] for v in d.values()), os.environ.get("POOLS_NOTE", "pools overlap on 3 streams")))
        for k, v in d.items():
            if v[1] > 0.05:
                L.append("  - `%s`: %d launches, %.1f ms, wg %d, LDS %d B" % (k, v[0], v[1], v[4], v[3]))
    L.append("")
L += ["## PMC counters (one `rocprofv3 --pmc` pass per counter group; kernels serialised by the profiler)", "",
      "Sum over all dispatches of all %d iterations of the run, and per cell crossing (%.4g crossings per iteration)." % (n_iter, crossings_iter), "",
      "| counter | kernel | sum | per crossing |", "|---|---|---|---|"]
tot = collections.defaultdict(float)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d): continue
    for db in glob.glob(os.path.join(d, "*.db")):
        c = sqlite3.connect(db)
        try:
            q = list(c.execute("select counter_name, kernel_name, sum(value) from counters_collection group by counter_name, kernel_name"))
        except Exception as e:
            L.append("| (%s: %s) | | | |" % (os.path.basename(d), e)); continue
        for name, kn, v in sorted(q):
            k = short(kn)
            if not (k.startswith("tile_") or k.startswith("lucy_kernel")): continue
            L.append("| %s | `%s` | %.6g | %.4g |" % (name, k, v, v / (n_iter * crossings_iter)))
            tot[name] += v
L += ["", "totals over the tile_* kernels, per crossing:", ""]
for name, v in sorted(tot.items()):
    L.append("- %s: %.6g  (%.4g per crossing)" % (name, v, v / (n_iter * crossings_iter)))
pmc_json = {"tag": tag, "crossings_per_iteration": crossings_iter, "iterations": n_iter,
            "per_crossing": {k: v / (n_iter * crossings_iter) for k, v in tot.items()}}
json.dump(pmc_json, open(os.path.join(os.environ.get("SUMMARY_DIR", os.path.join(root, "profiles")), tag + "_pmc.json"), "w"), indent=1)
out = os.path.join(os.environ.get("SUMMARY_DIR", os.path.join(root, "profiles")), tag + "_summary.md")
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L))
