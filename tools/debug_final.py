import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import hyperion_amd
from hyperion_amd.benchmark import make_octree_problem
from oracle_lib import Oracle, lib
p = make_octree_problem(max_level=5, n_pix=32)
eng = hyperion_amd.Engine(p); orc = Oracle(p)
for it in (1, 2):
    eng.lucy_iteration(40000, it); orc.lucy_iteration(40000, it)
lib().orc_set_final_first_id.argtypes = [C.c_uint64]
def both(first, n):
    eng.final_launch(first, n); eng.final_accumulators(); ra, sa = eng.final_finish()
    lib().orc_set_final_first_id(first); rb, sb = orc.final_iteration(n)
    return sa, sb
lo, n = 0, 40000
sa, sb = both(lo, n); print(sa["crossings"], sb["crossings"], sa["killed_geo"], sb["killed_geo"], sa["interactions"], sb["interactions"])
while n > 1:
    h = n // 2
    sa, sb = both(lo, h)
    if sa["crossings"] != sb["crossings"]: n = h
    else: lo, n = lo + h, n - h
sa, sb = both(lo, 1)
print("first differing packet", lo, {k: sa[k] for k in ("crossings","interactions","killed_geo","energy_current")}, {k: sb[k] for k in ("crossings","interactions","killed_geo","energy_current")})
# how many packets differ in the first 2000?
cnt = 0
for i in range(lo, lo + 300):
    sa, sb = both(i, 1)
    if sa["crossings"] != sb["crossings"]:
        cnt += 1
        if cnt < 6: print(i, sa["crossings"], sb["crossings"], sa["interactions"], sb["interactions"], sa["killed_geo"], sb["killed_geo"])
print("differing among 300:", cnt)
