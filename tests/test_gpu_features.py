"""GPU <-> oracle parity of the round-2 features through the C-ABI: n_photons, frequency-resolved specific energy, the
partial diffusion approximation, image filters and the convergence quantile (all on identical Philox streams)."""
import numpy as np
import pytest

import hyperion_amd
from cases import golden_problem, pda_block_problem
from hyperion_amd.benchmark import make_benchmark_problem
from hyperion_amd.problem import PeeledImages
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
KEYS = ("killed_geo", "killed_int", "crossings", "interactions")


def both(prob, n, iters=1, **opts):
    eng = hyperion_amd.Engine(prob)
    for k, v in opts.items():
        eng.set_option(k, v)
    orc = Oracle(prob)
    out = []
    for it in range(1, iters + 1):
        a, sa = eng.lucy_iteration(n, it)
        b, sb = orc.lucy_iteration(n, it)
        assert all(sa[k] == sb[k] for k in KEYS), (sa, sb)
        out.append((a, b))
    return eng, orc, out


def test_n_photons_parity():
    """The reference counts distinct packets per cell (it runs them one after the other and remembers the last one).  The
    device keeps, per lane, the set of cells its current packet has been counted in: the same integers as the oracle's, in
    every cell, whatever the order the packets run in."""
    prob = make_benchmark_problem(12, n_photons=20000, n_iter=1)
    prob.config.output_n_photons = "last"
    eng, orc, _ = both(prob, 20000)
    g, c = eng.n_photons(), orc.n_photons()
    assert g.shape == c.shape and g.dtype == np.int64 and c.max() > 1000
    np.testing.assert_array_equal(g, c)
    assert eng.get_option("n_photons_inexact") == 0
    eng.close(); orc.close()
    # an opaque block: its heart is starved, a few packets wander about in its skin for hundreds of interactions
    prob = pda_block_problem(pda=False)
    eng, orc, _ = both(prob, 30000)
    g, c = eng.n_photons(), orc.n_photons()
    eng.close(); orc.close()
    assert (c <= 32).sum() > 500
    np.testing.assert_array_equal(g, c)


@pytest.mark.parametrize("mrw", [False, True])
def test_specific_energy_spectrum_parity(mrw):
    if mrw:
        # optically thick cells with the realistic dust: the random walk deposits through the emissivity bin fractions
        # (deposit_specific_energy_spectrum, grid_physics_3d.f90:367-395)
        from test_gpu_mrw import thicken
        prob = thicken(make_benchmark_problem(8, n_photons=8000, n_iter=2), tau_cell=4.0)
        edges = np.logspace(np.log10(prob.dust[0].nu[0]), np.log10(prob.dust[0].nu[-1]), 9)
        edges[0] *= 0.999; edges[-1] *= 1.001
        n = 8000
    else:
        prob = make_benchmark_problem(10, n_photons=20000, n_iter=2, tau=2.0)
        edges = np.logspace(9.0, 17.0, 9)          # covers the whole dust table (3e9 .. 3e16 Hz)
        n = 20000
    prob.config.output_specific_energy_spectrum = "last"
    prob.config.spectrum_bin_edges = edges
    eng, orc, res = both(prob, n, iters=2)
    gs, e_out = eng.specific_energy_spectrum()
    cs = orc.specific_energy_spectrum()
    eng.close(); orc.close()
    np.testing.assert_array_equal(e_out, prob.config.spectrum_bin_edges)
    assert gs.shape == cs.shape == (8,) + prob.density.shape and (cs > 0).sum(axis=0).max() > 2
    np.testing.assert_allclose(gs, cs, rtol=1e-8, atol=1e-11 * cs.max())
    a, _ = res[-1]
    if not mrw:
        np.testing.assert_allclose(gs.sum(axis=0), a, rtol=1e-6)      # the reference's own identity


def test_spectrum_with_sublimation_and_several_species():
    prob, _ = golden_problem("car_specific_energy.False.True.npz")          # three species
    for d, mode in zip(prob.dust, ("fast", "slow", "cap")):
        d.sublimation_mode = mode
        d.sublimation_specific_energy = 3.0e4
    prob.config.output_specific_energy_spectrum = "all"
    prob.config.spectrum_bin_edges = np.logspace(10.0, 16.0, 5)
    eng, orc, res = both(prob, 20000, iters=2)
    gs, _ = eng.specific_energy_spectrum()
    cs = orc.specific_energy_spectrum()
    np.testing.assert_allclose(eng.density(), orc.density(), rtol=1e-9)
    eng.close(); orc.close()
    np.testing.assert_allclose(gs, cs, rtol=1e-8, atol=1e-12 * cs.max())


def test_pda_parity_exact_branch():
    """Fewer than 10 000 PDA cells: the reference eliminates, the device iterates the same system to 1e-12; the outer loop
    stops at 1e-5 on both sides."""
    prob = pda_block_problem()
    eng, orc, res = both(prob, 30000)
    a, b = res[0]
    n_dev, n_orc = eng.get_option("pda_last_cells"), orc.pda_last_cells()
    g, c = eng.n_photons(), orc.n_photons()
    eng.close(); orc.close()
    assert n_orc > 64 and n_dev == n_orc
    np.testing.assert_array_equal(g < 30, c < 30)
    pda = (c < 30)
    np.testing.assert_allclose(a[0][~pda], b[0][~pda], rtol=1e-9)
    np.testing.assert_allclose(a[0][pda], b[0][pda], rtol=1e-4)


def test_pda_exact_branch_with_an_empty_pocket_pivots_like_the_oracle():
    """A hollow of eight empty cells inside the opaque block: two empty PDA cells side by side have no Rosseland depth between
    them, solve_pda_indiv_exact clamps the sum at 1e-100 (grid_pda_3d.f90:222) and the row of such a cell carries coefficients
    of 1e100 next to rows of order one -- the system lineq_gausselim is given, solved here with partial pivoting on the
    device and by the oracle's own pivoting elimination."""
    prob = pda_block_problem()
    n = prob.density.shape[1]
    prob.density[0, n // 2 - 1:n // 2 + 1, n // 2 - 1:n // 2 + 1, n // 2 - 1:n // 2 + 1] = 0.0
    # (a packet that gets into the pocket bounces between its opaque walls for hundreds of interactions, and trajectories that long
    # are not the same on the device and on the host -- one ulp of libm per interaction, DESIGN.md section 2 --: cut them like the
    # MRW parity tests do)
    prob.config.n_inter_max = 60
    eng, orc, res = both(prob, 30000)
    a, b = res[0]
    n_dev, n_orc = eng.get_option("pda_last_cells"), orc.pda_last_cells()
    c = orc.n_photons()
    eng.close(); orc.close()
    assert n_orc > 64 and n_dev == n_orc
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    pda = (c < 30)
    filled = prob.density[0] > 0
    assert (pda & ~filled).sum() == 8          # the pocket is solved for, not sampled
    np.testing.assert_allclose(a[0][~pda], b[0][~pda], rtol=1e-9)
    np.testing.assert_allclose(a[0][pda & filled], b[0][pda & filled], rtol=1e-4)


def test_pda_iterative_branch_follows_the_reference_sweeps():
    """More than 10 000 PDA cells: Gauss-Seidel in cell order down to 1e-4 per sweep; the hyperplane-ordered sweeps of the
    device give each cell the operands of the sequential loop, so the two agree far below that tolerance."""
    prob = make_benchmark_problem(26, n_photons=20, n_iter=1, tau=30.0)      # 20 packets: every interior cell is starved
    prob.config.pda = True
    eng, orc, res = both(prob, 20)
    a, b = res[0]
    assert orc.pda_last_cells() == 24 ** 3 and eng.get_option("pda_last_cells") == 24 ** 3
    eng.close(); orc.close()
    np.testing.assert_allclose(a, b, rtol=1e-9)


def test_pda_on_a_cylindrical_grid_runs_the_reference_model():
    """The reference's own PDA model (test_pinte_specific_energy, tau = 1e4): device and oracle on the same streams."""
    prob, gold = golden_problem("pinte_specific_energy.tau=10000.npz")
    prob.config.n_inter_max = 100             # cut trajectories (MRW: chaotic in the last bit), PDA unaffected
    eng, orc, res = both(prob, 20000)
    a, b = res[0]
    assert orc.pda_last_cells() > 300
    g, c = eng.n_photons(), orc.n_photons()
    eng.close(); orc.close()
    np.testing.assert_array_equal(g < 30, c < 30)
    ok = (c >= 300)
    np.testing.assert_allclose(a[0][ok], b[0][ok], rtol=1e-8)
    pda = (c < 30) & (prob.density[0] > 0)
    lr = np.log10(a[0][pda] / b[0][pda])
    assert np.abs(np.median(lr)) < 1e-3 and (np.abs(lr) < 0.05).mean() > 0.9


def test_filters_parity():
    prob, _ = golden_problem("car_peeloff.False.npz")
    base = prob.peeled[0]
    nu = np.logspace(np.log10(base.nu_min), np.log10(base.nu_max), 12)
    bell = np.exp(-0.5 * ((np.log10(nu) - np.log10(nu[6])) / 0.3) ** 2)
    flt = PeeledImages(theta=base.theta, phi=base.phi, n_wav=2, compute_image=True, n_x=6, n_y=5, x_min=base.x_min, x_max=base.x_max,
                       y_min=base.y_min, y_max=base.y_max, compute_sed=True, n_ap=3, ap_min=base.ap_max / 10, ap_max=base.ap_max,
                       uncertainties=True, track_origin="basic",
                       filters=[(nu, bell, nu[6]), (nu[:5], np.array([0.0, 0.3, 1.0, 0.3, 0.0]), nu[2])])
    prob.peeled = [base, flt]
    eng = hyperion_amd.Engine(prob); orc = Oracle(prob)
    eng.lucy_iteration(5000, 1); orc.lucy_iteration(5000, 1)
    rg, sg = eng.final_iteration(20000)
    # filters only change the deposit: the problem runs on the deferred schedule (hyp_defer.h: the peel kernel spreads a packet
    # over the filters), on the inline plain kernel and on the general kernel alike
    assert eng.get_option("plain_imaging") == 1 and eng.get_option("last_defer_rounds") >= 1
    eng.set_option("defer_peel", 0)
    ri, si = eng.final_iteration(20000)
    assert eng.get_option("last_defer_rounds") == 0
    eng.set_option("plain_imaging", 0)
    rn, sn = eng.final_iteration(20000)
    rc, sc = orc.final_iteration(20000)
    eng.close(); orc.close()
    assert rc[1]["sed"].sum() > 0 and rc[1]["img"].sum() > 0
    for res, st in ((rg, sg), (ri, si), (rn, sn)):
        assert all(st[k] == sc[k] for k in KEYS)
        for x, y in zip(res, rc):
            for k in y:
                np.testing.assert_allclose(x[k], y[k], rtol=1e-9, atol=1e-14 * np.abs(y[k]).max())


def test_convergence_value_on_the_device():
    prob = make_benchmark_problem(8, n_photons=5000, n_iter=3)
    eng = hyperion_amd.Engine(prob); orc = Oracle(prob)
    prev = None
    for it in (1, 2, 3):
        eng.lucy_iteration(5000, it)
        b, _ = orc.lucy_iteration(5000, it)
        st, v = eng.convergence_value(99.0)
        if prev is None:
            assert st == 3
        else:
            so, vo = orc.convergence_value_against(prev, 99.0)
            assert st == 0 and so == 0
            assert abs(v - vo) <= 1e-9 * vo
        prev = b
    st, v = eng.convergence_value(50.0)          # nothing changed since the last call
    assert st == 1 and v == 0.0
    eng.close(); orc.close()


def test_no_sources_is_a_valid_setup_for_dust_only_raytracing():
    """setup_rt.f90:228-239: without sources the run is refused only if an iteration that needs them is asked for; the
    raytracing iteration then images the thermal emission alone -- the same cubes as the dust part of a run with the
    source (same packet ids, same specific energy)."""
    import copy
    from cases import imaging_problem
    prob = imaging_problem(tau=1.0)
    prob.config.raytracing = True
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(20000, 1, want_output=False)
    se = eng.specific_energy()
    a, sa = eng.raytracing_iteration(0, 4000)
    eng.close()
    assert float(np.nansum(a[0]["img"][0])) > 0
    p2 = copy.deepcopy(prob)
    p2.sources = []
    p2.specific_energy = se
    eng = hyperion_amd.Engine(p2)
    b, sb = eng.raytracing_iteration(9999, 4000)          # the source packets are dropped: n_raytracing_photons_sources = 0
    assert sa["crossings"] == sb["crossings"]
    for name in a[0]:
        np.testing.assert_allclose(b[0][name], a[0][name], rtol=1e-9, atol=1e-12 * np.nanmax(np.abs(a[0][name])), err_msg=name)
    with pytest.raises(hyperion_amd.EngineError, match=r"no sources set up - need sources for initial iteration\(s\)"):
        eng.lucy_iteration(100, 1)
    with pytest.raises(hyperion_amd.EngineError, match="no sources set up - need sources for last iteration"):
        eng.final_iteration(100)
    eng.close()
