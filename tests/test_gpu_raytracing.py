"""Raytracing mode on the GPU (src/main/iter_raytracing.f90 + scattered-only peel-off in the final
iteration): parity with the CPU oracle on identical Philox streams for the reference's own
raytracing=True regression models (Cartesian, octree, AMR), sharding of the two parts, and the
reference's golden SEDs statistically."""
import numpy as np
import pytest

import hyperion_amd
from cases import golden_problem
from hyperion_amd.images import finalize_peeled
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


def compare_cubes(ra, rb, rtol=1e-9):
    for ga, gb in zip(ra, rb):
        for name in gb:
            np.testing.assert_allclose(ga[name], gb[name], rtol=rtol, atol=1e-11 * np.nanmax(np.abs(gb[name])), err_msg=name)


@pytest.mark.parametrize("grid", ["car", "oct", "amr", "sph", "cyl"])
@pytest.mark.parametrize("evenly", [False, True])
def test_raytracing_parity_with_oracle(grid, evenly):
    prob, _ = golden_problem("%s_peeloff_ray.%s.npz" % (grid, evenly))
    assert prob.config.raytracing
    eng, orc = hyperion_amd.Engine(prob), Oracle(prob)
    for it in (1, 2):
        eng.lucy_iteration(5000, it); orc.lucy_iteration(5000, it)
    ra, sa = eng.final_iteration(20000)
    rb, sb = orc.final_iteration(20000)
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    compare_cubes(ra, rb)
    # only scattered packets were peeled: origin slots "source" and "dust" (direct) of the basic-tracking group are empty
    assert np.all(ra[1]["sed"][:, 0] == 0) and np.all(ra[1]["sed"][:, 1] == 0) and ra[1]["sed"][:, 2].max() > 0
    ra, sa = eng.raytracing_iteration(8000, 12000)
    rb, sb = orc.raytracing_iteration(8000, 12000)
    assert sa["killed_geo"] == sb["killed_geo"] == 0 and sa["crossings"] == sb["crossings"]
    compare_cubes(ra, rb)
    assert ra[1]["sed"][0, 0].max() > 0 and ra[1]["sed"][0, 1].max() > 0       # direct source and thermal flux arrived
    assert np.all(ra[1]["sed"][1:, 0] == 0)                                      # raytraced flux is unpolarised
    eng.close(); orc.close()


def test_raytracing_sharded_equals_whole():
    """Multi-GPU contract on one GPU: rank 0 keeps the final-iteration cubes, the other rank starts
    from zero, each runs its id range of both parts, the blocks are summed."""
    import torch
    prob, _ = golden_problem("car_peeloff_ray.False.npz")
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(5000, 1)
    eng.final_iteration(10000)
    whole, _ = eng.raytracing_iteration(6000, 9000)
    eng.final_iteration(10000)
    eng.raytracing_launch(0, 0, 2500, 6000); eng.raytracing_launch(1, 0, 4000, 9000)
    part0 = eng.raytracing_accumulators_tensor().clone()
    eng.raytracing_launch(0, 2500, 3500, 6000, zero_first=True); eng.raytracing_launch(1, 4000, 5000, 9000)
    acc = eng.raytracing_accumulators_tensor()
    acc += part0
    torch.cuda.synchronize()
    both, st = eng.raytracing_finish()
    compare_cubes(both, whole, rtol=1e-11)
    eng.close()


def test_golden_raytracing_statistical():
    """GPU vs the Fortran-produced golden (test_peeloff.grid_type=car.raytracing=True): Stokes I of
    the SEDs against an ensemble with the golden's photon numbers."""
    prob, z = golden_problem("car_peeloff_ray.False.npz")
    K = 24
    cubes = []
    for k in range(K):
        prob.config.seed = -(700 + k)
        eng = hyperion_amd.Engine(prob)
        for it in range(1, 6):
            eng.lucy_iteration(1000, it)
        eng.final_iteration(5000)
        res, st = eng.raytracing_iteration(2000, 3000)
        assert st["killed_geo"] == 0
        cubes.append([finalize_peeled(p, r) for p, r in zip(prob.peeled, res)])
        eng.close()
    for g in range(3):
        gold = z["golden/group%d/seds" % (g + 1)]
        c = np.array([s[g]["seds"] for s in cubes])
        mean, sig = c.mean(axis=0), c.std(axis=0, ddof=1) * np.sqrt(1 + 1.0 / K)
        I = mean[0][:, :, -1, :]
        sel = (sig[0][:, :, -1, :] > 0) & (I > 0.02 * I.max())
        zI = (gold[0][:, :, -1, :] - I)[sel] / sig[0][:, :, -1, :][sel]
        assert np.abs(zI).max() < 5.0 and (zI ** 2).mean() < 4.0, (g, zI)
