"""Throughput of the Voronoi walk at BASELINE config 5's scale (synthetic lattice tessellation)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hyperion_amd
from hyperion_amd.benchmark import make_voronoi_lattice_problem
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
p = make_voronoi_lattice_problem(n_side)
print("sites", p.n_cells, "species", p.n_dust)
if "--check" in sys.argv:
    from oracle_lib import Oracle
    a, sa = hyperion_amd.Engine(p).lucy_iteration(100000, 1)
    b, sb = Oracle(p).lucy_iteration(100000, 1)
    print("parity: crossings", sa["crossings"], sb["crossings"], "max rel", float(np.max(np.abs(a - b) / b.max())))
eng = hyperion_amd.Engine(p)
eng.lucy_iteration(n // 10, 1, want_output=False)
for it in (2, 3):
    _, st = eng.lucy_iteration(n, it, want_output=False)
    ms = eng.last_kernel_ms()[0]
    print("lucy n=%d kernel %.1f ms -> %.3e packets/s, %.1f crossings/packet, %.3e crossings/s, killed %d"
          % (n, ms, n / ms * 1e3, st["crossings"] / n, st["crossings"] / ms * 1e3, st["killed_geo"]))
