"""Modified random walk (src/grid/grid_mrw_3d.f90) on the GPU: the reference's known-answer
temperature table (hyperion/model/tests/test_mrw.py:10-63) through the C ABI, and parity with
the CPU oracle on identical Philox streams for Cartesian (persistent and brick-tiled schedule),
octree and AMR grids, several dust species, and the imaging iteration."""
import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity, golden_problem
from hyperion_amd.benchmark import make_benchmark_problem
from hyperion_amd.problem import PeeledImages
from oracle_lib import Oracle
from test_oracle_mrw import D_REF, T_REF, converge, realistic_dust, single_cell_problem, temperature

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


@pytest.mark.parametrize("i", range(18))
def test_single_temperature(i):
    p = single_cell_problem([D_REF[i]])
    eng = hyperion_amd.Engine(p)
    e, _ = converge(eng)
    eng.close()
    t = temperature(p.dust[0], e[0, 0, 0, 0])
    assert T_REF[i] / t < 1.1 and t / T_REF[i] < 1.1


@pytest.mark.parametrize("i", range(1, 18, 4))
def test_multi_temperature(i):
    p = single_cell_problem(D_REF[i] * np.array([0.1, 0.2, 0.3, 0.4]))
    eng = hyperion_amd.Engine(p)
    e, _ = converge(eng)
    eng.close()
    for d in range(4):
        t = temperature(p.dust[d], e[d, 0, 0, 0])
        assert T_REF[i] / t < 1.1 and t / T_REF[i] < 1.1


# A random-walk step moves the packet by (distance to the closest wall) x (random unit vector):
# a position error along the wall normal is multiplied by |n + dr|, on average by exp(0.19), so
# the trajectory is chaotic and the 1-ulp differences between the device's and glibc's libm
# grow exponentially along it (measured with tools/mrw_growth.py on one packet: 3e-16 after 50
# interactions, 5e-15 after 100, 1e-13 after 200, 5e-11 after 400, 2e-7 after 800; flat 3e-15
# without the MRW).  Identical-stream parity is therefore checked on trajectories cut after
# N_INTER_PARITY interactions (the cut itself, killed_int, is part of what is compared); the
# full-length behaviour is covered by the reference's known-answer table above.
N_INTER_PARITY = 100


def min_cell_width(prob):
    if prob.grid_type == "car":
        return min(float(np.diff(w).min()) for w in prob.walls)
    if prob.grid_type == "oct":
        depth, stack, deepest = 0, [], 0
        for flag in prob.refined:           # depth-first flags: a refined cell is followed by its 8 children
            while stack and stack[-1] == 0:
                stack.pop()
            if stack:
                stack[-1] -= 1
            deepest = max(deepest, len(stack))
            if flag:
                stack.append(8)
        return 2.0 * min(prob.oct_half) / 2.0 ** deepest
    if prob.grid_type == "amr":
        b, n = prob.amr_bounds, prob.amr_n
        return float(((b[:, 1::2] - b[:, 0::2]) / n).min())
    return float(prob.vor_volume.min()) ** (1.0 / 3.0)


def thicken(prob, tau_cell=1.0, n_species=None, gamma=0.2, n_mrw_max=1000):
    """Swap in the realistic dust (it carries the Planck means the MRW needs), scale the density
    so that the densest cell has chi_inv_planck optical half-width `tau_cell`, switch MRW on.
    (A low gamma makes the random walk frequent without the millions of optical-wavelength
    scatterings per packet a truly diffusive set-up costs the oracle.)"""
    nd = prob.density.shape[0] if n_species is None else n_species
    dust = realistic_dust()
    prob.dust = [realistic_dust() for _ in range(nd)]
    rho = prob.density[:1] / prob.density.max()
    size = min_cell_width(prob)
    prob.specific_energy = None
    chi = float(dust.mo_chi_inv_planck[0])
    rho = rho * tau_cell / (chi * 0.5 * size)
    prob.density = np.concatenate([rho * f for f in (np.arange(1, nd + 1) / (0.5 * nd * (nd + 1)))], axis=0)
    prob.config.mrw, prob.config.mrw_gamma, prob.config.n_inter_mrw_max = True, gamma, n_mrw_max
    prob.config.n_inter_max = N_INTER_PARITY
    return prob


def run_both(prob, n, iters=2, n_img=0, options=None, atol_rel=1e-10):
    eng = hyperion_amd.Engine(prob)
    for k, v in (options or {}).items():
        eng.set_option(k, v)
    orc = Oracle(prob)
    for it in range(1, iters + 1):
        a, sa = eng.lucy_iteration(n, it)
        b, sb = orc.lucy_iteration(n, it)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        assert_parity(a, b, atol_rel=atol_rel)
    if n_img:
        ra, sa = eng.final_iteration(n_img)
        rb, sb = orc.final_iteration(n_img)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        for ga, gb in zip(ra, rb):
            for name in gb:
                np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-10 * np.nanmax(np.abs(gb[name])), err_msg=name)
    mode = eng.get_option("last_lucy_mode")
    eng.close(); orc.close()
    return a, sa, mode


def mrw_changes_result(prob, n):
    """The set-up is optically thick enough for the random walk to kick in: fewer
    interactions than the same packets without it."""
    prob.config.n_inter_max = 1000000
    eng = hyperion_amd.Engine(prob)
    _, s1 = eng.lucy_iteration(n, 1)
    eng.close()
    prob.config.mrw = False
    eng = hyperion_amd.Engine(prob)
    _, s0 = eng.lucy_iteration(n, 1)
    eng.close()
    prob.config.mrw = True
    prob.config.n_inter_max = N_INTER_PARITY
    return s1["interactions"] < 0.9 * s0["interactions"]


@pytest.mark.parametrize("nd", [1, 3])
def test_cartesian_parity(nd):
    p = thicken(make_benchmark_problem(8, n_photons=2000, n_iter=2), n_species=nd)
    assert mrw_changes_result(p, 500)
    run_both(p, 2000)


def test_cartesian_tiled_schedule_parity():
    p = thicken(make_benchmark_problem(32, n_photons=5000, n_iter=2), 0.5)
    _, _, mode = run_both(p, 5000, options={"lucy_mode": 1})
    assert mode == 1


def test_cartesian_imaging_parity():
    p = thicken(make_benchmark_problem(8, n_photons=2000, n_iter=2), n_species=2)
    from hyperion_amd.benchmark import PC
    p.peeled = [PeeledImages(theta=[45.0, 120.0], phi=[45.0, 200.0], n_wav=5, wav_min=0.1, wav_max=3000.0, n_x=8, n_y=8,
                             x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=2, ap_min=0.2 * PC,
                             ap_max=2.0 * PC, compute_stokes=True, track_origin="basic")]
    run_both(p, 2000, n_img=2000)


def test_octree_parity():
    p, _ = golden_problem("oct_specific_energy.False.False.npz")
    thicken(p)
    assert mrw_changes_result(p, 500)
    run_both(p, 2000)


def test_amr_parity():
    p, _ = golden_problem("amr_specific_energy.False.False.npz")
    thicken(p)
    assert mrw_changes_result(p, 500)
    run_both(p, 2000)


def test_step_limit_kills_like_the_oracle():
    p = single_cell_problem([1e8], n_inter_mrw_max=3)
    _, st, _ = run_both(p, 500, iters=1)
    assert st["killed_int"] > 0


def test_errors():
    p = single_cell_problem([1.0])
    p.dust = [realistic_dust(mo_kappa_planck=1)]
    with pytest.raises(hyperion_amd.EngineError, match="kappa_planck"):
        hyperion_amd.Engine(p)
    v, _ = golden_problem("vor_lattice.npz")
    thicken(v)
    eng = hyperion_amd.Engine(v)
    with pytest.raises(hyperion_amd.EngineError, match="not implemented for Voronoi"):
        eng.lucy_iteration(100, 1)
    eng.close()


def test_imaging_with_mrw_on_the_deferred_schedule():
    """Round 4: the polychromatic imaging iteration of a run with the modified random walk on the deferred schedule
    (final_defer_kernel<.., GEN, MRWF>: every MRW step peeled off as isotropic emission, iter_final.f90:165-183; a packet set
    aside between rounds remembers the step it is at): = the general kernel = the oracle, one round and many."""
    from hyperion_amd.benchmark import PC
    p = thicken(make_benchmark_problem(8, n_photons=2000, n_iter=2), n_species=2)
    p.peeled = [PeeledImages(theta=[45.0, 120.0], phi=[45.0, 200.0], n_wav=5, wav_min=0.1, wav_max=3000.0, n_x=8, n_y=8,
                             x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=2, ap_min=0.2 * PC,
                             ap_max=2.0 * PC, compute_stokes=True, track_origin="basic")]
    for peel_events in (0, 4096):
        out = []
        for defer in (1, 0):
            eng = hyperion_amd.Engine(p)
            eng.set_option("gen_defer", defer)
            if peel_events and defer:
                eng.set_option("peel_events", peel_events)
            eng.lucy_iteration(2000, 1)
            res, st = eng.final_iteration(3000)
            rounds = eng.get_option("last_defer_rounds")
            assert (rounds >= (3 if peel_events else 1)) if defer else rounds == 0
            eng.close()
            out.append((res, st))
        (ra, sa), (rb, sb) = out
        for k in INT_KEYS:
            assert sa[k] == sb[k], (k, sa, sb)
        for ga, gb in zip(ra, rb):
            for name in gb:
                np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-10 * np.nanmax(np.abs(gb[name])), err_msg=name)
    orc = Oracle(p)
    orc.lucy_iteration(2000, 1)
    ro, so = orc.final_iteration(3000)
    orc.close()
    for k in INT_KEYS:
        assert sa[k] == so[k], (k, sa, so)
