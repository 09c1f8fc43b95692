"""Monochromatic final iteration (src/main/iter_final_mono.f90, src/grid/grid_monochromatic.f90) on
the GPU: parity with the CPU oracle on identical Philox streams for the reference's Pascucci
benchmark model (spherical polar grid, stellar sphere, mono + raytracing), for Cartesian / octree
models with several dust species and origin tracking, sharded launches, and the reference's own
monochromatic regression tests (hyperion/model/tests/test_mono.py) through the C ABI."""
import numpy as np
import pytest

import hyperion_amd
from cases import golden_problem, imaging_problem
from hyperion_amd.benchmark import LSUN, PC, load_test_dust
from hyperion_amd.problem import PeeledImages, Problem, RunConfig, Source
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")
C_CGS = 29979245800.0


def compare_cubes(ra, rb, rtol=1e-9):
    for ga, gb in zip(ra, rb):
        for name in gb:
            np.testing.assert_allclose(ga[name], gb[name], rtol=rtol, atol=1e-11 * np.nanmax(np.abs(gb[name])), err_msg=name)


def run_mono_both(prob, n_lucy, n_src, n_dust, iters=2, n_ray=0):
    eng, orc = hyperion_amd.Engine(prob), Oracle(prob)
    for it in range(1, iters + 1):
        eng.lucy_iteration(n_lucy, it); orc.lucy_iteration(n_lucy, it)
    ra, sa = eng.mono_iteration(n_src, n_dust)
    rb, sb = orc.mono_iteration(n_src, n_dust)
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    compare_cubes(ra, rb)
    if n_ray:
        ra, sa = eng.raytracing_iteration(n_ray, n_ray)
        rb, sb = orc.raytracing_iteration(n_ray, n_ray)
        assert sa["crossings"] == sb["crossings"]
        compare_cubes(ra, rb)
    eng.close(); orc.close()
    return ra, sa


@pytest.mark.parametrize("tau", ["0.1", "10"])
def test_pascucci_model_parity(tau):
    prob, _ = golden_problem("pascucci.tau=%s.npz" % tau)
    ra, st = run_mono_both(prob, 2000, 2000, 2000, iters=2, n_ray=2000)
    assert ra[0]["sed"][0].max() > 0


def mono_problem(base, wavelengths, n_dust=1):
    """Turns an imaging test problem into a monochromatic one."""
    base.config.monochromatic = True
    base.config.frequencies = C_CGS / (np.asarray(wavelengths) * 1e-4)
    for pl in base.peeled:
        pl.inu_min, pl.inu_max = 1, len(wavelengths)
        pl.n_wav = len(wavelengths)
    return base


def test_cartesian_two_species_detailed_origin():
    p = imaging_problem(n=10, tau=2.0, track_origin="detailed")
    p.dust = [p.dust[0], load_test_dust()]
    p.density = np.concatenate([0.6 * p.density, 0.4 * p.density], axis=0)
    p = mono_problem(p, [0.3, 1.0, 5.0, 30.0, 200.0])
    ra, st = run_mono_both(p, 20000, 20000, 20000)
    assert st["interactions"] > 0
    # source and dust slots both received flux
    assert ra[0]["sed"][0].sum() > 0


def test_subrange_of_frequencies_and_sharding():
    """A group that images frequencies 2..4 of 5 (inu_min / inu_max); the same iteration as two
    shards of every launch summed equals the whole."""
    p = imaging_problem(n=8, tau=1.0)
    p = mono_problem(p, [0.5, 1.0, 10.0, 100.0, 500.0])
    p.peeled[0].inu_min, p.peeled[0].inu_max, p.peeled[0].n_wav = 2, 4, 3
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(20000, 1)
    whole, sw = eng.mono_iteration(6000, 4000)
    first = True
    for which, n in ((0, 6000), (1, 4000)):
        for inu in range(5):
            eng.mono_launch(which, inu, 0, n // 2, n, zero_first=first)
            first = False
            eng.mono_launch(which, inu, n // 2, n - n // 2, n)
    both, sb = eng.mono_finish()
    eng.close()
    for k in INT_KEYS:
        assert sw[k] == sb[k]
    compare_cubes(both, whole, rtol=1e-12)
    assert whole[0]["sed"].shape[-1] == 3
    orc = Oracle(p)
    orc.lucy_iteration(20000, 1)
    ro, so = orc.mono_iteration(6000, 4000)
    orc.close()
    compare_cubes(whole, ro)


def _single_cell(densities, energies, wavelengths, n_view_sed=True):
    """hyperion/model/tests/test_mono.py: one Cartesian cell, no sources, given specific energies."""
    x = np.array([-1.0, 1.0])
    cfg = RunConfig()
    cfg.monochromatic = True
    cfg.frequencies = C_CGS / (np.asarray(wavelengths) * 1e-4)
    cfg.n_initial_iter = 0
    nd = len(densities)
    peel = [PeeledImages(theta=[45.0], phi=[45.0], n_x=4, n_y=4, x_min=-2.0, x_max=2.0, y_min=-2.0, y_max=2.0,
                         n_ap=1, ap_min=10.0, ap_max=10.0, track_origin="detailed", n_wav=len(wavelengths))]
    dens = np.array(densities, dtype=float).reshape(nd, 1, 1, 1)
    se = np.array(energies, dtype=float).reshape(nd, 1, 1, 1)
    # the C ABI needs at least one source; a vanishing one far below the dust emission stands in for "no sources"
    src = [Source(type="point", luminosity=1e-30, temperature=6000.0, position=(0.0, 0.0, 0.0))]
    return Problem(walls=[x, x, x], density=dens, dust=[load_test_dust() for _ in range(nd)], sources=src, config=cfg,
                   peeled=peel, specific_energy=se)


def test_reference_check_weighting():
    """test_mono.py:40-100 (regression test of the reference): with two dust populations of very
    different energies, the SED of the first must equal the SED of a model that has only that
    population (the model is optically thin) -- per-dust weighting of mean_prob."""
    wav = np.logspace(-1.0, 4.0, 10)
    p2 = _single_cell([1e-10, 1e-10], [1e8, 1e-4], wav)
    p1 = _single_cell([1e-10], [1e8], wav)
    out = []
    for p in (p2, p1):
        eng = hyperion_amd.Engine(p)
        res, st = eng.mono_iteration(0, 50000)
        eng.close()
        nsrc = len(p.sources)
        out.append(res[0]["sed"][0, nsrc + 0, 0, 0, :])         # detailed origin slot: dust_emit of dust 0
    a, b = out
    ok = b > 0
    assert ok.sum() >= 8
    assert np.all((a[ok] / b[ok] < 1.02) & (b[ok] / a[ok] < 1.02))


def test_reference_zero_probability_does_not_crash():
    """test_mono.py:10-37: a dust type whose emission probability vanishes at some wavelengths."""
    p = _single_cell([1.0, 0.5], [1e-3, 1e-3], [0.01, 0.1, 1.0, 10.0, 100.0, 1000.0])
    eng, orc = hyperion_amd.Engine(p), Oracle(p)
    ra, sa = eng.mono_iteration(0, 2000)
    rb, sb = orc.mono_iteration(0, 2000)
    eng.close(); orc.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k]
    compare_cubes(ra, rb)


def test_not_monochromatic_is_refused():
    p = imaging_problem(n=6)
    eng = hyperion_amd.Engine(p)
    with pytest.raises(hyperion_amd.EngineError, match="monochromatic mode was not requested"):
        eng.mono_iteration(10, 10)
    eng.close()


def test_lte_map_source_in_monochromatic_mode():
    """A luminosity-map source with the 'lte' spectrum in the monochromatic iteration: its packets carry the emission
    probability of the dust of their cell at the frequency (source_type.f90:455-459)."""
    from test_oracle_units import map_source_problem
    p, _ = map_source_problem(lte=True, n=5, tau=1.0)
    p = mono_problem(p, [1.0, 10.0, 100.0, 500.0])
    p.peeled = [PeeledImages(theta=[40.0], phi=[10.0], n_x=4, n_y=4, x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC,
                             n_ap=1, ap_min=2 * PC, ap_max=2 * PC, track_origin="basic", n_wav=4, inu_min=1, inu_max=4)]
    ra, st = run_mono_both(p, 20000, 10000, 10000)
    assert ra[0]["sed"][0, 0].max() > 0 and ra[0]["sed"][0, 1].max() > 0


# ---- round 4: the launches of a plain problem's monochromatic iteration on the deferred schedule (hyp_defer.h:
# final_defer_kernel<.., true, true> writes events, peel_kernel walks them sorted by cell into the launch's frequency plane) ----

def _mono_deferred_vs_inline(prob, n_lucy, n_src, n_dust, peel_events=0, oracle=True):
    """The same iteration on the deferred schedule (default), on the general kernel with inline peel-off (mono_defer = 0)
    and on the CPU oracle: integer tallies equal, cubes to 1e-9."""
    out = []
    for defer in (1, 0):
        eng = hyperion_amd.Engine(prob)
        eng.set_option("mono_defer", defer)
        if peel_events and defer:
            eng.set_option("peel_events", peel_events)
        eng.lucy_iteration(n_lucy, 1)
        res, st = eng.mono_iteration(n_src, n_dust)
        assert eng.get_option("last_mono_deferred") == defer
        rounds = eng.get_option("last_defer_rounds") if defer else 0
        eng.close()
        out.append((res, st, rounds))
    (ra, sa, rounds), (rb, sb, _) = out
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    compare_cubes(ra, rb)
    if oracle:
        orc = Oracle(prob)
        orc.lucy_iteration(n_lucy, 1)
        ro, so = orc.mono_iteration(n_src, n_dust)
        orc.close()
        for k in INT_KEYS:
            assert sa[k] == so[k], (k, sa, so)
        compare_cubes(ra, ro)
    return ra, sa, rounds


@pytest.mark.parametrize("forced_first", [True, False])
def test_mono_deferred_equals_inline_cartesian(forced_first):
    p = mono_problem(imaging_problem(n=10, tau=2.0, track_origin="basic"), [0.5, 2.0, 20.0, 300.0])
    p.config.forced_first_interaction = forced_first
    ra, st, _ = _mono_deferred_vs_inline(p, 20000, 20000, 20000)
    assert st["interactions"] > 20000 and np.nansum(ra[0]["img"]) > 0


def test_mono_deferred_many_rounds_small_event_buffer():
    """An event buffer of 4096 slots: packets are set aside between rounds with the energy they were emitted with
    (Packet::e_init), which the energy threshold of the forced scatterings needs."""
    p = mono_problem(imaging_problem(n=8, tau=4.0), [1.0, 30.0])
    p.config.monochromatic_energy_threshold = 1e-6
    _, st, rounds = _mono_deferred_vs_inline(p, 10000, 6000, 6000, peel_events=4096)
    assert rounds >= 3


def test_mono_deferred_two_species_and_raytracing_peels_scattered_light_only():
    p = imaging_problem(n=8, tau=2.0, track_origin="detailed")
    p.dust = [p.dust[0], load_test_dust()]
    p.density = np.concatenate([0.7 * p.density, 0.3 * p.density], axis=0)
    p = mono_problem(p, [0.4, 4.0, 40.0])
    p.config.raytracing = True
    _mono_deferred_vs_inline(p, 20000, 15000, 15000)


def test_mono_deferred_octree():
    from hyperion_amd.benchmark import make_octree_problem
    p = mono_problem(make_octree_problem(max_level=4, n_pix=16), [1.0, 10.0, 100.0])
    _mono_deferred_vs_inline(p, 20000, 10000, 10000)


def test_mono_with_a_stellar_sphere_on_the_deferred_schedule():
    """A star with a radius (the Pascucci / Pinte benchmark models): re-absorption, re-emission at the launch's frequency and the
    4 mu / limb-darkened peel-off on the deferred schedule (final_defer_kernel<.., true, true, true>) = the general kernel = the oracle."""
    p = mono_problem(imaging_problem(n=8, tau=2.0, theta=[40.0, 110.0], phi=[10.0, 220.0]), [1.0, 10.0, 100.0])
    p.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=0.2 * PC, limb_darkening=True)]
    _, st, _ = _mono_deferred_vs_inline(p, 10000, 10000, 10000)
    assert st["interactions"] > 0


def test_pascucci_model_runs_deferred():
    prob, _ = golden_problem("pascucci.tau=1.npz")
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(2000, 1)
    eng.mono_iteration(500, 500)
    assert eng.get_option("last_mono_deferred") == 1
    eng.close()


def test_mono_deferred_with_the_modified_random_walk_switched_on():
    """The modified random walk belongs to the polychromatic iterations (iter_final_mono.f90 has none): a run that has it on
    still takes the deferred schedule for its monochromatic launches (the Pinte benchmark set-up)."""
    from test_oracle_mrw import realistic_dust
    p = mono_problem(imaging_problem(n=8, tau=20.0, theta=[40.0], phi=[10.0]), [1.0, 100.0])
    p.dust = [realistic_dust()]
    p.density = p.density * (1.0 / float(p.dust[0].mo_chi_inv_planck[0]))
    p.config.mrw = True
    p.config.mrw_gamma = 2.0
    p.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=0.05 * PC)]
    p.config.monochromatic_energy_threshold = 1e-4
    _, st, _ = _mono_deferred_vs_inline(p, 5000, 4000, 4000)
    assert st["interactions"] > 0
