"""Emission threshold of the deferred propagation kernel on a Cartesian grid (tools/imaging_compare.py's cases):
   python tools/car_img_thr.py [packets]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
from cases import imaging_problem
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
cases = [("car 64^3 tau=1, 1 view 256^2 Stokes", imaging_problem(64, n_x=256, n_y=256)),
         ("car 64^3 tau=5, 3 views", imaging_problem(64, tau=5.0, n_x=128, n_y=128, theta=[30., 60., 90.], phi=[0., 120., 240.])),
         ("car 128^3 tau=1, 1 view 256^2 Stokes", imaging_problem(128, n_x=256, n_y=256))]
for name, p in cases:
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(n // 5, 1, want_output=False)
    line = name + ":"
    for it in (16, 32):
        for et in (8, 16, 32, 48):
            eng.set_option("final_interact_threshold", it); eng.set_option("final_emit_threshold", et)
            eng.final_iteration(n // 10)
            eng.final_iteration(n)
            line += " i%d/e%d %.1f" % (it, et, eng.last_kernel_ms()[0])
    print(line + " ms", flush=True)
    eng.close()
