"""Throughput of the imaging (final) iteration on the Cartesian benchmark grid and on config 4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
from cases import imaging_problem
from hyperion_amd.benchmark import make_octree_problem
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
for name, p in (("car64 tau=1, 1 view 256^2", imaging_problem(64, n_x=256, n_y=256)), ("config 4 octree", make_octree_problem(max_level=7))):
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(n // 5, 1, want_output=False)
    eng.final_iteration(n // 10)
    res, st = eng.final_iteration(n)
    ms = eng.last_kernel_ms()[0]
    print("%s: final n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, %.2f interactions/packet"
          % (name, n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, st["interactions"] / n), flush=True)
    eng.close()
