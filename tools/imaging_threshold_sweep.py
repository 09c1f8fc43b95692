import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
from cases import imaging_problem
n = 4_000_000
for name, p in (("tau1", imaging_problem(64, n_x=256, n_y=256)), ("tau5x3", imaging_problem(64, tau=5.0, n_x=128, n_y=128, theta=[30., 60., 90.], phi=[0., 120., 240.]))):
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(n // 5, 1, want_output=False)
    for e in (8, 16, 32, 48):
        line = "%s emit=%d:" % (name, e)
        for i in (8, 16, 24, 32):
            eng.set_option("final_emit_threshold", e); eng.set_option("final_interact_threshold", i)
            eng.final_iteration(n // 10)
            eng.final_iteration(n)
            line += " i%d %.1f" % (i, eng.last_kernel_ms()[0])
        print(line, flush=True)
    eng.close()
