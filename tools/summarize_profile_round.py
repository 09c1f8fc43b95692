"""Condense one workload of tools/profile_round.sh (round label from env ROUND, default r05): kernel table of the trace pass, counter sums of the PMC passes per kernel,
and their reduction per cell crossing (crossings of the whole profiled process, from the workload's own PROFILE_TOTALS /
bench line).  Writes <out>/<workload>_summary.md and merges <out>/<round>_pmc.json.
usage: python tools/summarize_profile_round.py <workload> <raw dir> <out dir>"""
import glob, json, os, re, sqlite3, sys
ROUND = os.environ.get("ROUND", "r05")
w, raw, out = sys.argv[1:4]
KEEP = ("tile_", "vtile_", "otile_", "atile_", "lucy_kernel", "final_kernel", "final_defer_kernel", "ff_walk_kernel", "peel_kernel", "reduce_copies", "finish_kernel")


def short(n):
    return n.split("(")[0].replace("void ", "")[:60]


log = open(os.path.join(raw, "trace.log")).read()
tot = None
for l in log.splitlines():
    if l.startswith("PROFILE_TOTALS "):
        tot = json.loads(l[len("PROFILE_TOTALS "):])
    elif l.startswith("{") and '"metric"' in l:
        b = json.loads(l)
        n_it = b.get("iterations_in_process", b["steps"] + b["warmup"])
        tot = {"workload": w, "packets": b["config"]["packets_per_iteration"] * n_it,
               "crossings": b["config"]["crossings_per_packet"] * b["config"]["packets_per_iteration"] * n_it, "n_dust": 1,
               "timed_ms": [b["lucy_kernel_ms"]], "ms_per_step": b["ms_per_step"], "bench_line": b}
if tot is None:
    sys.exit("no totals in %s/trace.log:\n%s" % (raw, log[-2000:]))
X = float(tot["crossings"])
L = ["# rocprofv3 summary, round %s: `%s`" % (ROUND.lstrip("r0"), w), "", "Workload output:", "", "```"]
L += [l for l in log.splitlines() if l and not l.startswith(("W2", "E2", "I2", "/opt/amdgpu")) and "amdgpu.ids" not in l][-8:]
# SURVEY section 8d, the rule bench.py uses too: a Lucy iteration moves 24 B x n_dust per crossing; an imaging iteration deposits nothing on its
# crossings -- 8 B x n_dust per crossing + 16 B x n_stokes per binned peel-off event
imaging = tot.get("events", 0) > 0
alg_per_crossing = (8.0 * tot["n_dust"] + 16.0 * tot["n_stokes"] * tot["events"] / X) if imaging else 24.0 * tot["n_dust"]
L += ["```", "", "Crossings of the whole profiled process (warm-up included): %.6g; packets %.4g; algorithmic bytes %s." % (X, tot["packets"],
      ("8 B x %d species per crossing + 16 B x %d Stokes x %.4g binned events = %.2f B per crossing (imaging iteration, SURVEY 8d)" % (tot["n_dust"], tot["n_stokes"], tot["events"], alg_per_crossing))
      if imaging else "24 B x %d species per crossing" % tot["n_dust"]), ""]
res = {"crossings": X, "packets": tot["packets"], "n_dust": tot["n_dust"], "timed_ms": tot.get("timed_ms"), "kernels": {}, "counters": {}}
for db in glob.glob(os.path.join(raw, "trace", "**", "*.db"), recursive=True):
    c = sqlite3.connect(db)
    reg = {}
    for n, vg, lds, wg in c.execute("select name, vgpr_count, lds_size, workgroup_x from kernels"):
        reg[n] = (vg, lds, wg)
    L += ["## kernel trace (`rocprofv3 --kernel-trace --stats`; microseconds)", "", "| kernel | calls | total us | average us | % | vgpr | LDS B | wg |", "|---|---|---|---|---|---|---|---|"]
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 14"):
        vg, lds, wg = reg.get(r[0], ("?", "?", "?"))
        L.append("| `%s` | %d | %d | %.1f | %.2f | %s | %s | %s |" % (short(r[0]), r[1], r[2], r[3], r[4], vg, lds, wg))
        res["kernels"][short(r[0])] = {"calls": r[1], "total_us": r[2], "avg_us": r[3], "vgpr": vg, "lds": lds}
    L.append("")
for d in sorted(glob.glob(os.path.join(raw, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        try:
            q = list(c.execute("select counter_name, kernel_name, sum(value), count(*) from counters_collection group by counter_name, kernel_name"))
        except Exception as e:
            L.append("(%s: %s)" % (os.path.basename(d), e)); continue
        for cn, kn, v, n in q:
            k = short(kn)
            if any(t in k for t in KEEP):
                res["counters"].setdefault(cn, {})[k] = v
if res["counters"]:
    L += ["## PMC counters (one `--pmc` pass per counter set), summed over the dispatches of each kernel; per cell crossing in brackets", ""]
    kernels = sorted({k for d in res["counters"].values() for k in d}, key=lambda k: -res["kernels"].get(k, {}).get("total_us", 0))
    main = [k for k in kernels if res["kernels"].get(k, {}).get("total_us", 0) > 0.02 * sum(v["total_us"] for v in res["kernels"].values())]
    L += ["| counter | " + " | ".join("`%s`" % k[:34] for k in main) + " | all kernels |", "|---|" + "---|" * (len(main) + 1)]
    per = {}
    for cn in sorted(res["counters"]):
        d = res["counters"][cn]
        s = sum(d.values())
        per[cn] = s / X
        L.append("| %s | " % cn + " | ".join("%.4g (%.4g)" % (d.get(k, 0.0), d.get(k, 0.0) / X) for k in main) + " | %.4g (%.4g) |" % (s, s / X))
    res["per_crossing"] = per
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        f, wr = per["FETCH_SIZE"] * 1024.0, per["WRITE_SIZE"] * 1024.0
        alg = alg_per_crossing
        res["bytes_per_crossing"] = {"fetched": f, "fetched_x2": 2 * f, "written": wr, "algorithmic": alg, "traffic_over_algorithmic": (f + wr) / alg,
                                     "traffic_over_algorithmic_reads_x2": (2 * f + wr) / alg}
        L += ["", "L2<->fabric bytes per crossing: fetched %.1f B (%.1f B with the guide's x2 for wide streaming reads), written %.1f B; algorithmic %.0f B -> traffic / algorithmic = %.2f (%.2f)."
              % (f, 2 * f, wr, alg, (f + wr) / alg, (2 * f + wr) / alg)]
    if "TCC_EA0_ATOMIC_sum" in per:
        L.append("Memory-side atomics per crossing: %.4f." % per["TCC_EA0_ATOMIC_sum"])
    v = per.get("SQ_INSTS_VALU")
    if v:
        f64 = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT64"))
        res["valu"] = {"per_crossing": v, "f64_and_int64_per_crossing": f64, "share": f64 / v}
        L.append("VALU wave-instructions per crossing %.3f, of which FP64 (add / mul / fma / trans) and 64-bit integer %.3f (%.0f %%); SALU %.3f."
                 % (v, f64, 100 * f64 / v, per.get("SQ_INSTS_SALU", 0.0)))
open(os.path.join(out, w + "_summary.md"), "w").write("\n".join(L) + "\n")
path = os.path.join(out, ROUND + "_pmc.json")
allres = json.load(open(path)) if os.path.exists(path) else {}
res.pop("kernels_full", None)
allres[w] = res
json.dump(allres, open(path, "w"), indent=1)
print("\n".join(L[-12:]))
