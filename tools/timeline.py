"""Timeline reduction of a rocprofv3 --kernel-trace database (rocpd sqlite): idle time, overlap depth, per-kernel shares."""
import glob, sqlite3, sys
from collections import defaultdict

def rows(db):
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [n for n in names if n == "kernels"] + [n for n in names if "kernel_dispatch" in n]
    for n in cand:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % n)]
        if "start" in cols and "end" in cols:
            namecol = "name" if "name" in cols else None
            if namecol is None:
                continue
            extra = [k for k in ("queue_id", "stream_id") if k in cols]
            q = "select %s, start, end%s from %s" % (namecol, "".join(", " + e for e in extra), n)
            return list(c.execute(q)), extra
    print("no kernel table with start/end/name found; tables:", names)
    for n in cand:
        print(n, [r[1] for r in c.execute("pragma table_info(%s)" % n)])
    sys.exit(1)

for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    R, extra = rows(db)
    R = [r for r in R if r[2] > r[1]]
    if not R:
        continue
    # the timed region of the workload: from the first tile_walk / lucy kernel of the LAST third of the trace to the end
    t0 = min(r[1] for r in R); t1 = max(r[2] for r in R)
    lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
    R = [r for r in R if r[1] >= lo]
    t0 = min(r[1] for r in R); t1 = max(r[2] for r in R)
    ev = []
    for r in R:
        ev.append((r[1], 1)); ev.append((r[2], -1))
    ev.sort()
    depth = 0; last = t0; hist = defaultdict(float)
    for t, d in ev:
        hist[min(depth, 4)] += t - last; last = t; depth += d
    span = t1 - t0
    print("span %.1f ms, %d dispatches" % (span / 1e6, len(R)))
    for k in sorted(hist):
        print("  %s kernels running: %8.1f ms  %5.1f %%" % (("%d" % k) if k < 4 else "4+", hist[k] / 1e6, 100 * hist[k] / span))
    # gaps with nothing running, by length
    gaps = []; depth = 0; last = None
    for t, d in ev:
        if depth == 0 and last is not None and t > last: gaps.append(t - last)
        depth += d
        if depth == 0: last = t
    gaps.sort()
    if gaps:
        n = len(gaps)
        print("  idle gaps: %d, median %.1f us, p90 %.1f us, max %.1f us, > 20 us: %d totalling %.1f ms" % (
            n, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[-1] / 1e3, sum(1 for g in gaps if g > 20e3), sum(g for g in gaps if g > 20e3) / 1e6))
    per = defaultdict(lambda: [0, 0.0])
    for r in R:
        k = r[0].split("(")[0][:60]; per[k][0] += 1; per[k][1] += r[2] - r[1]
    for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:10]:
        print("  %-60s %6d calls %9.1f ms summed, %7.1f us avg" % (k, n, d / 1e6, d / n / 1e3))
    if extra:
        qs = defaultdict(float)
        for r in R: qs[tuple(r[3:])] += r[2] - r[1]
        print("  busy per", extra, {k: round(v / 1e6, 1) for k, v in qs.items()})
