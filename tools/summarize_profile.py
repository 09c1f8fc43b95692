"""Condense rocprofv3 output (gpurun_out/prof_<tag>/, rocpd sqlite) into
profiles/<tag>_summary.md.  usage: python tools/summarize_profile.py <tag> [kernel-substring]"""
import glob, os, sqlite3, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kern = sys.argv[2] if len(sys.argv) > 2 else "lucy_kernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
out = os.path.join(root, "profiles"); os.makedirs(out, exist_ok=True)
L = ["# rocprofv3 summary `%s`" % tag, ""]
for db in glob.glob(os.path.join(src, "trace", "*.db")):
    c = sqlite3.connect(db)
    L += ["## kernel stats (`rocprofv3 --kernel-trace --stats`, view top_kernels; durations in ns)", "",
          "| kernel | calls | total ns | average ns | % |", "|---|---|---|---|---|"]
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        L.append("| `%s` | %d | %d | %.0f | %.2f |" % (r[0][:90], r[1], r[2], r[3], r[4]))
    L += ["", "per-dispatch durations of `%s` (ns) and resources:" % kern, ""]
    rows = list(c.execute("select duration,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels where name like ? order by start", ("%" + kern + "%",)))
    for r in rows:
        L.append("- %d ns, grid %d x wg %d, LDS %d B, scratch %d B, VGPR %d, AGPR %d, SGPR %d" % r)
    L.append("")
log = os.path.join(src, "trace.log")
if os.path.exists(log):
    js = [l for l in open(log) if l.startswith("{")]
    if js: L += ["bench line of the traced run:", "", "```", js[-1].strip(), "```", ""]
L += ["## PMC counters (one rocprofv3 --pmc pass per row group; mean per dispatch of `%s`)" % kern, "",
      "| counter | mean per dispatch | dispatches |", "|---|---|---|"]
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d): continue
    for db in glob.glob(os.path.join(d, "*.db")):
        c = sqlite3.connect(db)
        try:
            q = c.execute("select counter_name, dispatch_id, sum(value) from counters_collection where kernel_name like ? group by counter_name, dispatch_id", ("%" + kern + "%",))
        except Exception as e:
            L.append("| (%s: %s) | | |" % (os.path.basename(d), e)); continue
        acc = {}
        for name, disp, v in q: acc.setdefault(name, []).append(v)
        for name, vals in sorted(acc.items()):
            L.append("| %s | %.6g | %d |" % (name, sum(vals) / len(vals), len(vals)))
open(os.path.join(out, tag + "_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
