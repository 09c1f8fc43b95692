"""Imaging iteration of a run with the modified random walk (a 32^3 Cartesian grid, every cell chi_inv_planck-thick by tau_cell, realistic
dust, one view): deferred schedule (GEN + MRWF kernels) against the general kernel (gen_defer = 0).
   python tools/mrw_img_probe.py [packets] [tau_cell]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hyperion_amd
from hyperion_amd.benchmark import PC, make_benchmark_problem
from hyperion_amd.problem import PeeledImages
from test_gpu_mrw import thicken
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
tau_cell = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
p = thicken(make_benchmark_problem(32), tau_cell, gamma=2.0, n_mrw_max=10000)
p.config.n_inter_max = 10000
p.peeled = [PeeledImages(theta=[45.0], phi=[45.0], n_wav=20, wav_min=0.1, wav_max=3000.0, n_x=64, n_y=64, x_min=-1.5 * PC, x_max=1.5 * PC,
                         y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=1, ap_min=2 * PC, ap_max=2 * PC)]
for defer in (1, 0):
    e = hyperion_amd.Engine(p)
    e.set_option("gen_defer", defer)
    e.lucy_iteration(n, 1, want_output=False)
    print('lucy done %.1f ms' % e.last_kernel_ms()[0], flush=True)
    e.final_iteration(n // 10)
    _, st = e.final_iteration(n)
    ms = e.last_kernel_ms()[0]
    print("gen_defer %d: imaging %.1f ms (%.3g packets/s, %.0f crossings and %.1f interactions per packet, rounds %d)"
          % (defer, ms, n / ms * 1e3, st["crossings"] / n, st["interactions"] / n, e.get_option("last_defer_rounds")), flush=True)
    e.close()
