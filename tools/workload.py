"""The single-GPU workloads of BASELINE.json that are profiled besides the bench command (tools/profile_round.sh):
   python tools/workload.py oct_lucy|oct_img|vor|amr|sph [packets] [opt=value ...]
Runs one warm-up (a tenth of the packets) and two timed iterations; prints the timings and one line
`PROFILE_TOTALS {json}` with the crossings / packets of ALL iterations of the process (what a rocprofv3 pass sees)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
if os.environ.get("HYP_LIB"):        # a tuning variant built by tools/variants.py (one geometry, one species)
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from hyperion_amd.benchmark import make_octree_problem

which = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
if which == "amr":
    from hyperion_amd.benchmark import make_amr_problem
    p = make_amr_problem(n=64, levels=3)
    n_dust = 1
elif which == "sph":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_polar import config0_problem
    p = config0_problem(n_r=400, n_t=200, tau=3.0)
    n_dust = 1
elif which == "vor":
    from cases import voronoi_big_problem
    p = voronoi_big_problem(n_photons=n)
    n_dust = 2
else:
    p = make_octree_problem(max_level=7)
    n_dust = 1
eng = hyperion_amd.Engine(p)
for a in sys.argv[3:]:
    eng.set_option(a.split("=")[0], int(a.split("=")[1]))
tot = {"workload": which, "packets": 0, "crossings": 0, "n_dust": n_dust, "timed_ms": [], "timed_crossings": 0, "timed_packets": 0,
       "events": 0, "n_stokes": 4 if which == "oct_img" else 0}      # events: binned peel-off events x views of the imaging iterations (16 B x n_stokes each)


def run(m, it, timed):
    if which == "oct_img":
        _, st = eng.final_iteration(m)
    else:
        _, st = eng.lucy_iteration(m, it, want_output=False)
    tot["packets"] += m; tot["crossings"] += st["crossings"]
    if which == "oct_img":
        tot["events"] += eng.get_option("last_defer_events") * sum(len(g.theta) for g in p.peeled)
    if timed:
        ms = eng.last_kernel_ms()[0]
        tot["timed_ms"].append(ms); tot["timed_crossings"] += st["crossings"]; tot["timed_packets"] += m
        print("%s n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, killed_geo %d"
              % (which, m, ms, m / ms * 1e3, st["crossings"] / m, st["crossings"] / ms * 1e3, st["killed_geo"]), flush=True)


if which == "oct_img":
    eng.lucy_iteration(n // 10, 1, want_output=False)       # temperatures for the imaging iteration (not counted: another kernel)
run(n // 10, 1, False)
run(n, 2, True)
run(n, 3, True)
tot["lucy_mode"] = eng.get_option("last_lucy_mode")
tot["defer_rounds"] = eng.get_option("last_defer_rounds")
print("PROFILE_TOTALS " + json.dumps(tot), flush=True)
