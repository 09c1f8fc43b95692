"""Spherical polar grid 400 x 200 (configs[0]'s shape): the tiled Lucy iteration, with the counters of a -DHYP_PTILE_VERIFY build
   [HYP_LIB=build/variants/x.so] python tools/sph_probe.py [packets] [option=value ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
if os.environ.get("HYP_LIB"):
    import hyperion_amd.engine as E
    E._lib = E.load_library(os.environ["HYP_LIB"])
from test_gpu_polar import config0_problem
args = sys.argv[1:]
n = int(float(args.pop(0))) if args and args[0][0].isdigit() else 20_000_000
p = config0_problem(n_r=400, n_t=200, tau=3.0)
e = hyperion_amd.Engine(p)
for a in args:
    e.set_option(a.split("=")[0], int(a.split("=")[1]))
e.lucy_iteration(n // 10, 1, want_output=False)
for it in (2, 3):
    _, st = e.lucy_iteration(n, it, want_output=False)
    ms = e.last_kernel_ms()[0]
    print(json.dumps({"lib": os.path.basename(os.environ.get("HYP_LIB", "default")), "ms": round(ms, 1), "packets_per_s": n / ms * 1e3, "crossings": st["crossings"],
                      "killed_geo": st["killed_geo"], "interactions": st["interactions"], "mode": e.get_option("last_lucy_mode"),
                      "fast": e.get_option("last_walk_fast_steps"), "slow": e.get_option("last_walk_slow_steps"), "mismatch": e.get_option("last_walk_mismatch"), "why": [e.get_option("last_walk_why%d" % k) for k in range(8)]}), flush=True)
