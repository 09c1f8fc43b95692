#!/bin/bash
# Host-side stress for the N > 1 path on a 1-GPU box: 8 independent processes, each with its own three HIP
# streams, ~2600 launches and host polls per Lucy iteration, time-slice ONE GPU.  If the host side (launch rate, polls, the
# runtime's locks) were the limit at 8 ranks per node, 8 concurrent processes would take longer than 8 x one process's work.
#   usage: bash tools/r04_host_contention.sh [packets per process, default 1.25e7 = configs[2]'s 1e9 / 8 scaled by 1/10]
N=${1:-1.25e7}
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out/r04_contention; mkdir -p $OUT
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --photons $N 2>/dev/null | tail -1; }
echo "== one process, $N packets per iteration"
run > $OUT/single.json; python - <<PY
import json; b = json.load(open("$OUT/single.json")); p = b["per_rank_ms_per_step"]
print("ms_per_step %.1f  launch_ms %.1f  kernel_wait_ms %.2f  device_propagate_ms %.1f" % (b["ms_per_step"], p["launch_ms"]["max"], p["kernel_wait_ms"]["max"], p["device_propagate_ms"]["max"]))
PY
echo "== eight concurrent processes on the same GPU, $N packets per iteration each"
T0=$(date +%s.%N)
for i in 0 1 2 3 4 5 6 7; do ( run > $OUT/p$i.json ) & done; wait
T1=$(date +%s.%N)
python - <<PY
import json
rows = [json.load(open("$OUT/p%d.json" % i)) for i in range(8)]
ms = [b["ms_per_step"] for b in rows]; la = [b["per_rank_ms_per_step"]["launch_ms"]["max"] for b in rows]; dev = [b["per_rank_ms_per_step"]["device_propagate_ms"]["max"] for b in rows]
s = json.load(open("$OUT/single.json"))["ms_per_step"]
print("ms_per_step per process: min %.1f max %.1f mean %.1f  (8 x single = %.1f)" % (min(ms), max(ms), sum(ms) / 8, 8 * s))
print("launch_ms per process:   min %.1f max %.1f; device_propagate_ms min %.1f max %.1f" % (min(la), max(la), min(dev), max(dev)))
print("wall time of the eight processes (start-up included): %.1f s" % ($T1 - $T0))
PY
