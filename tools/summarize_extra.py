"""Condense the rocprofv3 output of tools/r02_profile_extra.sh (kernel trace + PMC passes of the configs[3] / configs[4]
runs) into one markdown file.  usage: python tools/summarize_extra.py <raw dir> <out.md>"""
import glob, sqlite3, sys
raw, out_path = sys.argv[1], sys.argv[2]
L = ["# rocprofv3 summary: configs[3] (octree + peel-off imaging) and configs[4] (voro++ tessellation), round 2", ""]
for name in ("octree", "voronoi"):
    L += ["## %s" % name, "", "```"]
    L += [l.rstrip() for l in open("%s/%s_trace.log" % (raw, name)) if not l.startswith(("W2", "E2", "I2"))]
    L += ["```", ""]
    for db in glob.glob("%s/%s_trace/**/*.db" % (raw, name), recursive=True):
        c = sqlite3.connect(db)
        reg = {}
        for n, vg, lds in c.execute("select name, vgpr_count, lds_size from kernels"):
            reg[n] = (vg, lds)
        L += ["| kernel | calls | total us | average us | vgpr | lds |", "|---|---|---|---|---|---|"]
        for r in c.execute("select name,total_calls,total_duration,average from top_kernels limit 8"):
            vg, lds = reg.get(r[0], ("?", "?"))
            L.append("| `%s` | %d | %d | %.0f | %s | %s |" % (r[0].split("(")[0][:70], r[1], r[2], r[3], vg, lds))
    L += ["", "| counter | kernel | sum over dispatches |", "|---|---|---|"]
    for d in sorted(glob.glob("%s/%s_pmc_*" % (raw, name))):
        for db in glob.glob(d + "/**/*.db", recursive=True):
            c = sqlite3.connect(db)
            try:
                for cn, kn, v in c.execute("select counter_name, kernel_name, sum(value) from counters_collection group by counter_name, kernel_name"):
                    k = kn.split("(")[0]
                    if any(t in k for t in ("lucy_kernel", "final_kernel", "final_defer_kernel", "peel_kernel")):
                        L.append("| %s | `%s` | %.6g |" % (cn, k[:60], v))
            except Exception as e:
                L.append("| (%s) | | |" % e)
    L.append("")
open(out_path, "w").write("\n".join(L) + "\n")
print("\n".join(L))
