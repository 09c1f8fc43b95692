"""Imaging (final) iteration with peel-off: HIP path vs CPU oracle on identical
Philox streams, plus the RNG-free analytic answers the reference's own tests
state (hyperion/model/tests/test_image.py, test_sed.py)."""
import numpy as np
import pytest

import hyperion_amd
from cases import golden_problem, imaging_problem
from hyperion_amd.benchmark import LSUN, PC, make_benchmark_problem
from hyperion_amd.images import finalize_peeled
from hyperion_amd.problem import PeeledImages
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")


def run_both(prob, n_lucy, n_img, lucy_iters=1):
    eng = hyperion_amd.Engine(prob)
    orc = Oracle(prob)
    for it in range(1, lucy_iters + 1):
        eng.lucy_iteration(n_lucy, it, want_output=False)
        orc.lucy_iteration(n_lucy, it)
    ra, sa = eng.final_iteration(n_img)
    rb, sb = orc.final_iteration(n_img)
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
    for ga, gb in zip(ra, rb):
        assert set(ga) == set(gb)
        for name in gb:
            scale = np.nanmax(np.abs(gb[name]))
            np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-11 * scale, err_msg=name)
    eng.close()
    orc.close()
    return ra, sa


@pytest.mark.parametrize("evenly", [False, True])
def test_reference_peeloff_model(evenly):
    """test_bit_level.py:175-236: three groups, polarised kmh dust, origin tracking."""
    prob, _ = golden_problem("car_peeloff.%s.npz" % evenly)
    res, st = run_both(prob, 5000, 20000, lucy_iters=2)
    assert res[1]["sed"].shape == (4, 4, 1, 2, 4) and res[2]["img"].shape == (4, 12, 1, 6, 6, 4)
    assert np.abs(res[0]["sed"][1]).max() > 0 and np.abs(res[0]["sed"][2]).max() > 0     # Q, U populated


@pytest.mark.parametrize("kw", [
    {}, {"uncertainties": True}, {"ignore_optical_depth": True}, {"compute_sed": False}, {"compute_image": False},
    {"compute_stokes": False}, {"track_origin": "scatterings", "track_n_scat": 2}, {"track_origin": "detailed"},
    {"d_min": -0.3 * PC, "d_max": 0.4 * PC}, {"theta": [0.5, 90.0, 179.5, 33.0], "phi": [0.0, 270.0, 45.0, 359.0]},
    {"n_ap": 1, "ap_min": 0.5 * PC, "ap_max": 0.5 * PC},
])
def test_image_configurations(kw):
    run_both(imaging_problem(**kw), 20000, 30000)


@pytest.mark.parametrize("algo", ["none", "wr99", "baes16"])
def test_forced_first_interaction(algo):
    p = imaging_problem(tau=0.3)
    p.config.forced_first_interaction = algo != "none"
    if algo != "none":
        p.config.forced_first_interaction_algorithm = algo
    run_both(p, 10000, 30000)


def test_multi_dust_imaging():
    prob, _ = golden_problem("car_specific_energy.False.True.npz")
    prob.peeled = [PeeledImages(theta=[60.0, 120.0], phi=[10.0, 200.0], n_wav=4, wav_min=0.05, wav_max=500.0,
                                n_x=8, n_y=8, x_min=-PC, x_max=PC, y_min=-PC, y_max=PC,
                                n_ap=2, ap_min=0.3 * PC, ap_max=1.5 * PC, track_origin="detailed", uncertainties=True)]
    run_both(prob, 5000, 20000)


def test_direct_source_attenuation_is_exp_minus_tau():
    """test_image.py:876-915: a source seen through uniform, purely absorbing
    dust is dimmed by exp(-chi rho d).  Here: central source, tau = 1 to the face
    along the line of sight (theta=90, phi=0), kill_on_absorb so that only direct
    light reaches the image; ratio to the dust-free run = e^-1."""
    def flux(tau):
        p = make_benchmark_problem(10, tau=tau)
        p.dust[0].albedo[:] = 0.0
        p.config.kill_on_absorb = True
        p.config.forced_first_interaction = False
        p.peeled = [PeeledImages(theta=[90.0], phi=[0.0], n_wav=1, wav_min=0.01, wav_max=1e5, compute_image=False,
                                 n_ap=1, ap_min=3 * PC, ap_max=3 * PC, compute_stokes=False)]
        eng = hyperion_amd.Engine(p)
        res, st = eng.final_iteration(100000)
        return res[0]["sed"].sum()
    assert flux(1.0) / flux(1e-12) == pytest.approx(np.exp(-1.0), rel=1e-6)


def test_sed_uncertainty_is_one_over_sqrt_n():
    """test_sed.py:473-501: for equal-weight packets sigma/F = 1/sqrt(N)."""
    p = make_benchmark_problem(6, tau=1e-12)
    p.config.forced_first_interaction = False
    p.peeled = [PeeledImages(theta=[30.0], phi=[60.0], n_wav=1, wav_min=0.001, wav_max=1e6, compute_image=False,
                             n_ap=1, ap_min=3 * PC, ap_max=3 * PC, uncertainties=True, compute_stokes=False)]
    n = 40000
    eng = hyperion_amd.Engine(p)
    res, st = eng.final_iteration(n)
    out = finalize_peeled(p.peeled[0], res[0])
    assert out["seds_unc"].sum() / out["seds"].sum() == pytest.approx(1.0 / np.sqrt(n), rel=1e-3)
    # and the flux itself: L / (4 pi) per steradian, nu F_nu summed over the one bin
    from hyperion_amd.images import dnunorm
    assert out["seds"].sum() * dnunorm(p.peeled[0]) == pytest.approx(LSUN, rel=1e-9)


def test_total_is_source_plus_dust_components():
    """test_image.py:672-677 with basic origin tracking."""
    p = imaging_problem(track_origin="basic")
    q = imaging_problem()
    a, _ = run_both(p, 20000, 30000)
    b, _ = run_both(q, 20000, 30000)
    np.testing.assert_allclose(a[0]["img"].sum(axis=1), b[0]["img"][:, 0], rtol=1e-9, atol=1e-12 * b[0]["img"].max())


def test_full_size_image_512():
    """BASELINE config 4's detector (512 x 512, one view, Stokes) on the 128^3
    Cartesian grid, 5e6 imaging packets: image total equals SED of the largest
    aperture, flux conservation against the no-dust case bound."""
    p = make_benchmark_problem(128)
    p.peeled = [PeeledImages(theta=[45.0], phi=[45.0], n_wav=1, wav_min=0.01, wav_max=1e5,
                             n_x=512, n_y=512, x_min=-2 * PC, x_max=2 * PC, y_min=-2 * PC, y_max=2 * PC,
                             n_ap=1, ap_min=4 * PC, ap_max=4 * PC)]
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(2_000_000, 1, want_output=False)
    res, st = eng.final_iteration(5_000_000)
    img, sed = res[0]["img"], res[0]["sed"]
    assert img.shape == (4, 1, 1, 512, 512, 1)
    assert img[0].sum() == pytest.approx(sed[0].sum(), rel=1e-9)
    assert 0.3 * LSUN < sed[0].sum() < 1.0 * LSUN          # attenuated but not lost
    assert st["killed_geo"] == 0
    # the central pixel block holds the direct source light
    c = img[0, 0, 0, 254:258, 254:258, 0].sum()
    assert c > 0.1 * img[0].sum()


def test_binned_images_parity_and_scaling():
    """images_binned.f90 on the GPU: packets that leave the grid alive in the final iteration are binned by
    direction; parity with the oracle on identical streams (cubes of the binned group = group index n_peeled),
    together with a peeled group, origin tracking and uncertainties; a killed packet is not binned."""
    from test_oracle_units import binned_problem
    p = binned_problem()
    p.binned.uncertainties = True
    p.config.n_inter_max = 3                    # some packets are killed: they must not be binned
    eng, orc = hyperion_amd.Engine(p), Oracle(p)
    for it in (1, 2):
        eng.lucy_iteration(20000, it); orc.lucy_iteration(20000, it)
    ra, sa = eng.final_iteration(60000)
    rb, sb = orc.final_iteration(60000)
    eng.close(); orc.close()
    for k in ("crossings", "interactions", "killed_geo", "killed_int"):
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["killed_int"] > 0
    assert len(ra) == len(rb) == 2
    for ga, gb in zip(ra, rb):
        for name in gb:
            np.testing.assert_allclose(ga[name], gb[name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(gb[name])), err_msg=name)
    assert ra[1]["sed"].shape == (4, 4, 12, 1, 5) and ra[1]["sed2"].max() > 0


def test_binned_images_on_a_cartesian_grid_with_run():
    """run(): the binned cubes come back separately (RunResult.binned) and every direction bin is filled."""
    from hyperion_amd.problem import PeeledImages
    from hyperion_amd.run import run_problem
    p = imaging_problem(n=8, tau=0.5)
    p.config.forced_first_interaction = False
    p.config.n_initial_iter = 1
    p.config.n_initial_photons = 20000
    p.config.n_last_photons = 100000
    p.binned = PeeledImages(theta=[0.0], phi=[0.0], n_wav=3, wav_min=0.1, wav_max=1000.0, compute_image=False,
                            n_ap=1, ap_min=3 * PC, ap_max=3 * PC)
    p.n_binned_theta, p.n_binned_phi = 3, 4
    r = run_problem(p)
    assert len(r.peeled) == 1 and r.binned is not None
    s = r.binned["seds"]
    assert s.shape == (4, 1, 12, 1, 3)
    assert np.all(s[0].sum(axis=(0, 2, 3)) > 0)
    # total over the bins / (n_theta n_phi) = the 4 pi average the peeled image approximates
    tot = s[0, 0, :, 0, :].mean(axis=0).sum()
    peel = r.peeled[0]["seds"][0, 0, 0, -1, :].sum()
    assert tot == pytest.approx(peel, rel=0.1)


@pytest.mark.parametrize("raytracing", [False, True])
def test_inside_observer_parity(raytracing):
    """Peel-off towards an observer INSIDE the grid (images_peeled.f90:158-205, 236): direction to the observer's
    position, optical depth up to the observer (tmax = d), 1 / (4 pi d^2) dilution, longitude / latitude maps; in the
    imaging iteration and (raytracing on) in the raytracing iteration, against the oracle on identical streams."""
    from test_oracle_units import inside_observer_problem
    p, d = inside_observer_problem(tau=1.0)
    p.peeled[0].phi = np.array([250.0])
    p.peeled[0].theta = np.array([80.0])
    p.peeled[0].n_wav, p.peeled[0].wav_min, p.peeled[0].wav_max = 4, 0.1, 1000.0
    p.peeled[0].track_origin = "basic"
    p.config.raytracing = raytracing
    eng, orc = hyperion_amd.Engine(p), Oracle(p)
    for it in (1, 2):
        eng.lucy_iteration(20000, it); orc.lucy_iteration(20000, it)
    ra, sa = eng.final_iteration(40000)
    rb, sb = orc.final_iteration(40000)
    for k in ("crossings", "interactions", "killed_geo", "killed_int"):
        assert sa[k] == sb[k], (k, sa, sb)
    # point sources: the problem runs on the deferred schedule (the peel kernel walks towards the observer's position and
    # stops there); without it, on the general kernel -- same tallies, same cubes
    assert eng.get_option("plain_imaging") == 1 and eng.get_option("last_defer_rounds") >= 1
    for name in rb[0]:
        np.testing.assert_allclose(ra[0][name], rb[0][name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(rb[0][name])), err_msg=name)
    eng.set_option("defer_peel", 0)
    rg, sg = eng.final_iteration(40000)
    assert eng.get_option("last_defer_rounds") == 0
    for k in ("crossings", "interactions", "killed_geo", "killed_int"):
        assert sg[k] == sb[k], (k, sg, sb)
    for name in rb[0]:
        np.testing.assert_allclose(rg[0][name], rb[0][name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(rb[0][name])), err_msg=name)
    eng.set_option("defer_peel", 1)
    if raytracing:
        ra, sa = eng.final_iteration(40000)
        ra, sa = eng.raytracing_iteration(20000, 20000)
        rb, sb = orc.raytracing_iteration(20000, 20000)
        assert sa["crossings"] == sb["crossings"]
    eng.close(); orc.close()
    for name in rb[0]:
        np.testing.assert_allclose(ra[0][name], rb[0][name], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(rb[0][name])), err_msg=name)
    img = ra[0]["img"][0]
    assert img.shape == (4, 1, 5, 9, 4) and img.sum() > 0


def _images_equal(ra, rb, rtol=1e-9):
    for ga, gb in zip(ra, rb):
        assert set(ga) == set(gb)
        for name in gb:
            scale = np.nanmax(np.abs(gb[name]))
            np.testing.assert_allclose(ga[name], gb[name], rtol=rtol, atol=1e-11 * scale, err_msg=name)


@pytest.mark.parametrize("kw", [{}, {"uncertainties": True, "track_origin": "detailed", "theta": [10.0, 80.0, 150.0], "phi": [0.0, 120.0, 300.0]}])
@pytest.mark.parametrize("peel_events", [0, 4096])
def test_deferred_peeloff_equals_inline(kw, peel_events):
    """hyp_defer.h: the events written by the propagation kernel and walked by the peel kernel give the images of the
    inline peel-off (the same sums in another order), also when the event buffer is so small that the iteration takes
    many rounds, with packets set aside and id ranges returned between them."""
    prob = imaging_problem(tau=3.0, **kw)
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(20000, 1, want_output=False)
    assert eng.get_option("plain_imaging") == 1 and eng.get_option("defer_peel") == 1
    if peel_events:
        eng.set_option("peel_events", peel_events)
    ra, sa = eng.final_iteration(30000)
    rounds = eng.get_option("last_defer_rounds")
    assert rounds >= (3 if peel_events else 1)
    assert eng.get_option("last_defer_events") >= 30000
    # the peel kernel took the events ordered by cell (peel_sort, the default); as written they give the same images
    assert eng.get_option("peel_sort") == 1
    eng.set_option("peel_sort", 0)
    rc, sc = eng.final_iteration(30000)
    for k in INT_KEYS:
        assert sa[k] == sc[k], (k, sa, sc)
    _images_equal(ra, rc)
    eng.set_option("defer_peel", 0)
    rb, sb = eng.final_iteration(30000)
    assert eng.get_option("last_defer_rounds") == 0
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
    _images_equal(ra, rb)
    eng.close()


def _ff_problem(grid):
    if grid == "car":
        return imaging_problem(tau=2.0, uncertainties=True)
    if grid == "oct":
        from hyperion_amd.benchmark import make_octree_problem
        return make_octree_problem(max_level=5, n_pix=32)
    prob, _ = golden_problem("%s_peeloff.False.npz" % grid)       # test_bit_level.py:175-236 on that grid
    prob.config.forced_first_interaction = True
    return prob


@pytest.mark.parametrize("grid", ["car", "oct", "amr", "sph"])
@pytest.mark.parametrize("peel_events", [0, 4096])
def test_forced_first_prepass_equals_the_walk_inside_the_propagation_kernel(grid, peel_events):
    """hyp_defer.h: ff_walk_kernel emits every packet, makes its escape walk and samples the forced first interaction
    (iter_final.f90:191-209) ahead of the rounds and leaves an EmitRec per id; final_defer_kernel<.., false> has no emission
    code and starts its packets from the records.  Same packets, same draws: the tallies are equal and the images are the
    same sums -- also over many rounds (id ranges returned, packets set aside) and against the inline schedule."""
    prob = _ff_problem(grid)
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(20000, 1, want_output=False)
    assert eng.get_option("plain_imaging") == 1 and eng.get_option("ff_prepass") == 1
    if peel_events:
        eng.set_option("peel_events", peel_events)
    ra, sa = eng.final_iteration(30000)
    assert eng.get_option("last_ff_prepass") == 1
    assert eng.get_option("last_defer_rounds") >= (3 if peel_events else 1)
    eng.set_option("ff_prepass", 0)
    rb, sb = eng.final_iteration(30000)
    assert eng.get_option("last_ff_prepass") == 0
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["energy_current"] == sb["energy_current"]
    _images_equal(ra, rb)
    eng.set_option("defer_peel", 0)
    rc, sc = eng.final_iteration(30000)
    for k in INT_KEYS:
        assert sa[k] == sc[k], (k, sa, sc)
    _images_equal(ra, rc)
    eng.close()


def test_forced_first_prepass_is_skipped_without_forced_first_interaction():
    p = imaging_problem(tau=0.3)
    p.config.forced_first_interaction = False
    eng = hyperion_amd.Engine(p)
    eng.lucy_iteration(10000, 1, want_output=False)
    eng.final_iteration(10000)
    assert eng.get_option("last_defer_rounds") >= 1 and eng.get_option("last_ff_prepass") == 0
    eng.close()


def test_sharded_imaging_iteration_equals_the_whole_one():
    """hyp_final_launch(first_id, n_local) on two id ranges (what two ranks do), blocks summed like the all-reduce does,
    then hyp_final_finish: the cubes of the whole iteration -- also through the deferred schedule with an event buffer
    small enough to need several rounds per shard."""
    import torch
    prob = imaging_problem(tau=2.0, uncertainties=True)
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(20000, 1, want_output=False)
    eng.set_option("peel_events", 8192)
    whole, sw = eng.final_iteration(24000)
    assert eng.get_option("last_defer_rounds") > 2
    eng.final_launch(0, 10000)
    a = eng.final_accumulators_tensor().clone()
    eng.final_launch(10000, 14000)
    b = eng.final_accumulators_tensor()
    b += a                                              # the all-reduce of two ranks
    parts, sp = eng.final_finish()
    for k in INT_KEYS:
        assert sp[k] == sw[k], (k, sp, sw)
    _images_equal(parts, whole)
    eng.close()


# --- deferred against inline peel-off on the other grids and modes (the models the staged schedule of round 3 was tested on;
# that schedule was measured slower and is gone, hyp_defer.h is the one deferred schedule) --------------------------------------

def _deferred_vs_inline(prob, n_lucy, n_img, peel_events=0):
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(n_lucy, 1, want_output=False)
    assert eng.get_option("plain_imaging") == 1
    eng.set_option("defer_peel", 1)
    if peel_events:
        eng.set_option("peel_events", peel_events)
    ra, sa = eng.final_iteration(n_img)
    rounds = eng.get_option("last_defer_rounds")
    assert rounds >= 1 and eng.get_option("last_defer_events") >= (0 if prob.config.raytracing else n_img)
    eng.set_option("defer_peel", 0)
    rb, sb = eng.final_iteration(n_img)
    assert eng.get_option("last_defer_rounds") == 0
    eng.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
    _images_equal(ra, rb)
    return rounds, ra


@pytest.mark.parametrize("kw", [{}, {"uncertainties": True, "track_origin": "detailed", "theta": [10.0, 80.0, 150.0], "phi": [0.0, 120.0, 300.0]}])
def test_deferred_imaging_equals_inline_cartesian_thick(kw):
    """Optically thick Cartesian model with forced first interaction, one round and many (an event buffer of 4096 slots)"""
    _deferred_vs_inline(imaging_problem(tau=3.0, **kw), 20000, 30000)
    rounds, _ = _deferred_vs_inline(imaging_problem(tau=3.0, **kw), 20000, 30000, peel_events=4096)
    assert rounds >= 5


def test_deferred_imaging_without_forced_first_interaction_and_scattered_only():
    p = imaging_problem(tau=2.0)
    p.config.forced_first_interaction = False
    _deferred_vs_inline(p, 20000, 30000)
    p.config.raytracing = True          # the final iteration then peels scattered packets only
    _deferred_vs_inline(p, 20000, 30000)


def test_deferred_imaging_on_the_tree_and_polar_grids():
    """Octree (vertex source: packets killed by the propagation check), AMR, Voronoi and spherical polar grids."""
    from cases import golden_problem
    from hyperion_amd.benchmark import make_octree_problem
    _deferred_vs_inline(make_octree_problem(max_level=5, n_pix=32), 20000, 40000)
    for name in ("amr_peeloff.False.npz", "sph_peeloff.False.npz", "oct_peeloff.True.npz"):
        prob, _ = golden_problem(name)
        _deferred_vs_inline(prob, 5000, 20000)
    from hyperion_amd.problem import PeeledImages
    prob, _ = golden_problem("vor_lattice.npz")          # central point source: the plain imaging kernels apply
    prob.peeled = [PeeledImages(theta=[45.0, 100.0], phi=[45.0, 250.0], n_wav=3, wav_min=0.1, wav_max=1000.0, n_x=8, n_y=8, x_min=-1.5 * PC, x_max=1.5 * PC,
                                y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=2, ap_min=0.2 * PC, ap_max=2.0 * PC, compute_stokes=True)]
    _deferred_vs_inline(prob, 5000, 20000)


# --- the imaging iteration's propagation half on the tiled schedule (IMG kernels of hyp_tiled.h; defer_peel = 2 forces it) ----------

def _tiled_vs_inline(prob, n_lucy, n_img, oracle=False, **opts):
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(n_lucy, 1, want_output=False)
    assert eng.get_option("plain_imaging") == 1
    eng.set_option("defer_peel", 2)
    for k, v in opts.items():
        eng.set_option(k, v)
    ra, sa = eng.final_iteration(n_img)
    assert eng.get_option("last_tiled_imaging") == 1, "the tiled imaging schedule did not run"
    flushes, events = eng.get_option("last_defer_rounds"), eng.get_option("last_defer_events")
    assert flushes >= 1 and events >= (0 if prob.config.raytracing else n_img)
    eng.set_option("defer_peel", 0)
    rb, sb = eng.final_iteration(n_img)
    assert eng.get_option("last_defer_rounds") == 0 and eng.get_option("last_tiled_imaging") == 0
    eng.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
    _images_equal(ra, rb)
    return flushes, ra


@pytest.mark.parametrize("kw", [{}, {"uncertainties": True, "track_origin": "detailed", "theta": [10.0, 80.0, 150.0], "phi": [0.0, 120.0, 300.0]}])
def test_tiled_imaging_equals_inline_cartesian(kw):
    """Optically thick Cartesian model with forced first interaction: emission and the forced first interaction ahead (ff_walk_kernel),
    interactions and emissions leave events from the slot-pool kernels, walks from LDS bricks; one pool and three, small tasks,
    an event buffer that has to be emptied several times on the way"""
    _tiled_vs_inline(imaging_problem(tau=3.0, **kw), 20000, 30000, tile_slots=8192, tile_task=256, tile_pools=1)
    flushes, _ = _tiled_vs_inline(imaging_problem(tau=3.0, **kw), 20000, 90000, tile_slots=12288, tile_task=512, tile_pools=3, peel_events=65536)
    assert flushes >= 3


def test_tiled_imaging_without_forced_first_interaction_and_scattered_only():
    p = imaging_problem(tau=2.0)
    p.config.forced_first_interaction = False
    _tiled_vs_inline(p, 20000, 30000, tile_slots=8192, tile_task=256)
    p.config.raytracing = True          # the final iteration then peels scattered packets only
    _tiled_vs_inline(p, 20000, 30000, tile_slots=8192, tile_task=256)


def test_tiled_imaging_on_the_tree_voronoi_and_polar_grids():
    from cases import golden_problem
    from hyperion_amd.benchmark import make_octree_problem
    _tiled_vs_inline(make_octree_problem(max_level=5, n_pix=32), 20000, 40000, tile_slots=8192, tile_task=256, ot_cells=9)
    for name in ("amr_peeloff.False.npz", "sph_peeloff.False.npz", "oct_peeloff.True.npz"):
        prob, _ = golden_problem(name)
        _tiled_vs_inline(prob, 5000, 20000, tile_slots=4096, tile_task=256)
    from hyperion_amd.problem import PeeledImages
    prob, _ = golden_problem("vor_lattice.npz")
    prob.peeled = [PeeledImages(theta=[45.0, 100.0], phi=[45.0, 250.0], n_wav=3, wav_min=0.1, wav_max=1000.0, n_x=8, n_y=8, x_min=-1.5 * PC, x_max=1.5 * PC,
                                y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=2, ap_min=0.2 * PC, ap_max=2.0 * PC, compute_stokes=True)]
    _tiled_vs_inline(prob, 5000, 20000, tile_slots=4096, tile_task=256, vt_cells=12)


def test_tiled_imaging_falls_back_to_the_deferred_rounds_and_counts_once():
    """An event buffer that does not hold a few generations' worth of events: the tiled schedule declines BEFORE the forced-first
    pre-pass has run (the pre-pass counts its crossings and kills; the deferred rounds run it themselves) -- tallies and images of
    the inline kernel"""
    prob = imaging_problem(tau=3.0)
    eng = hyperion_amd.Engine(prob)
    eng.lucy_iteration(20000, 1, want_output=False)
    eng.set_option("defer_peel", 2)
    eng.set_option("peel_events", 4096)
    ra, sa = eng.final_iteration(30000)
    assert eng.get_option("last_tiled_imaging") == 0 and eng.get_option("last_defer_rounds") >= 5
    eng.set_option("defer_peel", 0)
    rb, sb = eng.final_iteration(30000)
    eng.close()
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    _images_equal(ra, rb)


# ---- round 4: the direct light of a point source towards a view walked once (hyp_defer.h: direct_column_kernel) ----

def _memo_vs_walked(prob, n_lucy, n_img):
    """The deferred schedule with the recorded walk of the direct light (default) and with every event walked
    (direct_memo = 0): identical integer tallies -- the recorded walk is only taken for pairs whose first propagation check
    falls behind its last crossing -- and the same images up to the association of the optical-depth sum."""
    out = []
    for memo in (1, 0):
        eng = hyperion_amd.Engine(prob)
        eng.set_option("direct_memo", memo)
        eng.set_option("defer_peel", 1)
        eng.lucy_iteration(n_lucy, 1, want_output=False)
        res, st = eng.final_iteration(n_img)
        assert eng.get_option("last_defer_rounds") >= 1 and eng.get_option("last_direct_memo") == memo
        eng.close()
        out.append((res, st))
    (ra, sa), (rb, sb) = out
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    _images_equal(ra, rb)
    return ra, sa


def test_direct_light_memo_cartesian_several_sources_and_views():
    from hyperion_amd.problem import Source
    p = imaging_problem(tau=2.0, theta=[20.0, 90.0, 160.0], phi=[10.0, 200.0, 330.0], track_origin="detailed")
    p.sources = [Source(type="point", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0)),
                 Source(type="point", luminosity=0.5 * LSUN, temperature=4000.0, position=(0.31 * PC, -0.2 * PC, 0.44 * PC))]
    _memo_vs_walked(p, 20000, 40000)


def test_direct_light_memo_with_frequent_propagation_checks():
    """A check every ~5 steps: most direct walks meet a check before their last crossing and are walked; the tallies
    stay those of the walked schedule."""
    p = imaging_problem(tau=1.0)
    p.config.propagation_check_frequency = 0.2
    _memo_vs_walked(p, 10000, 30000)


def test_direct_light_memo_octree_source_on_a_vertex():
    """BASELINE configs[3]'s situation: the source sits on a vertex of the tree (packets are killed in find_wall's
    negative-t branch there, SURVEY / DESIGN section 2); same kills, same crossings with and without the recorded walk."""
    from hyperion_amd.benchmark import make_octree_problem
    _, st = _memo_vs_walked(make_octree_problem(max_level=5, n_pix=32), 20000, 60000)
    assert st["crossings"] > 0


# ---- round 4: sources with a surface on the deferred schedule (final_defer_kernel<.., GEN>, peel_kernel<.., GEN>) ----

def _sphere_problem(base, radius, limb=False, extra_point=False):
    from hyperion_amd.problem import Source
    base.sources = [Source(type="sphere", luminosity=LSUN, temperature=6000.0, position=(0.0, 0.0, 0.0), radius=radius, limb_darkening=limb)]
    if extra_point:
        base.sources.append(Source(type="point", luminosity=0.3 * LSUN, temperature=4000.0, position=(0.4 * PC, 0.1 * PC, -0.3 * PC)))
    return base


def _gen_deferred_vs_general(prob, n_lucy, n_img, peel_events=0, oracle=True):
    """Spherical sources (emission from the surface, limb darkening, re-absorption and re-emission, walks blocked by a source):
    the imaging iteration on the deferred schedule (default), on the general kernel with inline peel-off (gen_defer = 0) and on
    the CPU oracle -- integer tallies equal, cubes to 1e-9."""
    out = []
    for defer in (1, 0):
        eng = hyperion_amd.Engine(prob)
        eng.set_option("gen_defer", defer)
        if peel_events and defer:
            eng.set_option("peel_events", peel_events)
        assert eng.get_option("plain_imaging") == 0 and eng.get_option("gen_defer") == defer
        eng.lucy_iteration(n_lucy, 1, want_output=False)
        res, st = eng.final_iteration(n_img)
        rounds = eng.get_option("last_defer_rounds")
        assert (rounds >= 1) == bool(defer)
        eng.close()
        out.append((res, st, rounds))
    (ra, sa, rounds), (rb, sb, _) = out
    for k in INT_KEYS:
        assert sa[k] == sb[k], (k, sa, sb)
    assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
    _images_equal(ra, rb)
    # ... and with the propagation half on the slot-pool schedule (round 6: the GEN instances of tile_emit / tile_interact <IMG> -- general
    # emitter, the escape walk of the forced first interaction in the emission kernel, re-emission events, walks that watch t_src), pools
    # far smaller than the packet count, with and without the end-game on the deferred rounds
    for end_game in (1, 0):
        eng = hyperion_amd.Engine(prob)
        for k, v in dict(defer_peel=2, tile_slots=6144, tile_task=256, tile_pools=2, img_end_game=end_game).items():
            eng.set_option(k, v)
        if peel_events:
            eng.set_option("peel_events", max(peel_events, 65536))
        eng.lucy_iteration(n_lucy, 1, want_output=False)
        rt, stt = eng.final_iteration(n_img)
        assert eng.get_option("last_tiled_imaging") == 1, "the tiled imaging schedule did not run"
        eng.close()
        for k in INT_KEYS:
            assert sa[k] == stt[k], ("tiled", end_game, k, sa, stt)
        assert sa["energy_current"] == pytest.approx(stt["energy_current"], rel=1e-12)
        _images_equal(ra, rt)
    if oracle:
        orc = Oracle(prob)
        orc.lucy_iteration(n_lucy, 1)
        ro, so = orc.final_iteration(n_img)
        orc.close()
        for k in INT_KEYS:
            assert sa[k] == so[k], (k, sa, so)
        _images_equal(ra, ro)
    return ra, sa, rounds


@pytest.mark.parametrize("limb", [False, True])
def test_sphere_source_deferred_equals_general_cartesian(limb):
    p = _sphere_problem(imaging_problem(n=10, tau=2.0, theta=[30.0, 100.0], phi=[20.0, 250.0], track_origin="basic"), 0.12 * PC, limb=limb)
    ra, st, _ = _gen_deferred_vs_general(p, 20000, 30000)
    assert st["interactions"] > 0 and np.nansum(ra[0]["sed"]) > 0


def test_sphere_and_point_sources_many_rounds():
    """A big star (many re-absorptions) next to a point source whose light it blocks for some views; an event buffer of
    4096 slots: packets are set aside before interactions AND before re-emissions."""
    p = _sphere_problem(imaging_problem(n=8, tau=3.0, theta=[60.0, 120.0], phi=[15.0, 195.0]), 0.25 * PC, extra_point=True)
    _, st, rounds = _gen_deferred_vs_general(p, 10000, 20000, peel_events=4096)
    assert rounds >= 3


def test_sphere_source_deferred_spherical_grid_and_octree():
    import sys, os
    from test_gpu_polar import config0_problem
    p = _sphere_problem(config0_problem(n_r=40, n_t=24, tau=2.0, log_r=True, peeled=True), 0.004 * PC)
    _gen_deferred_vs_general(p, 20000, 20000)
    from hyperion_amd.benchmark import make_octree_problem
    q = _sphere_problem(make_octree_problem(max_level=4, n_pix=16), 0.03 * PC)
    _gen_deferred_vs_general(q, 20000, 20000)


def test_external_box_and_point_on_a_voronoi_lattice_deferred():
    """BASELINE configs[4]'s sources (a point + an external box, inward normals in the emission events, no peel-off of the box's
    light) and two species on a Voronoi lattice: GEN kernels = general kernel = oracle."""
    from hyperion_amd.benchmark import make_voronoi_lattice_problem
    from hyperion_amd.problem import PeeledImages
    p = make_voronoi_lattice_problem(n=6, tau=1.5, n_photons=20000, n_iter=1)
    p.peeled = [PeeledImages(theta=[35.0, 120.0], phi=[25.0, 260.0], n_wav=3, wav_min=0.1, wav_max=1000.0, n_x=8, n_y=8,
                             x_min=-1.5 * PC, x_max=1.5 * PC, y_min=-1.5 * PC, y_max=1.5 * PC, n_ap=2, ap_min=0.3 * PC, ap_max=2.0 * PC, compute_stokes=True)]
    _gen_deferred_vs_general(p, 20000, 20000)


def test_spotted_star_deferred():
    from test_oracle_units import spotted_star_problem
    p = spotted_star_problem()
    eng = hyperion_amd.Engine(p)
    if eng.get_option("gen_defer") != 1:
        eng.close()
        pytest.skip("the spotted-star model has no peeled group the deferred schedule applies to")
    eng.close()
    _gen_deferred_vs_general(p, 5000, 20000)


# --- end-game of the imaging iteration on the tiled schedule (round 6) ------------------------------------------------

def _thick_scattering_problem(n=24, tau=6.0, albedo=0.9, n_pix=16):
    """An optically thick cube with a high albedo: tens of scatterings per packet and a long tail of packets with hundreds --
    after the last packet id is handed out, the longest histories keep the generations of the tiled schedule going."""
    p = imaging_problem(n=n, tau=tau, n_x=n_pix, n_y=n_pix)
    p.dust[0].albedo = np.full_like(p.dust[0].albedo, albedo)
    p.config.forced_first_interaction = True
    return p


def test_tiled_imaging_end_game_equals_oracle():
    """Small N, identical streams: the propagation half on the tiled schedule with small pools (many generations), its last packets
    handed to the deferred rounds (tile_to_susp_kernel: packets about to interact AND packets on their way) -- tallies equal to the
    oracle's, cubes to rounding; the same without the end-game and on the deferred schedule alone."""
    prob = _thick_scattering_problem()
    n = 200_000
    orc = Oracle(prob)
    orc.lucy_iteration(50_000, 1)
    want, sw = orc.final_iteration(n)
    orc.close()
    seen = {}
    for name, opts in (("end-game", dict(defer_peel=2, tile_slots=3 * 8192, tile_task=1024)),
                       ("generations", dict(defer_peel=2, tile_slots=3 * 8192, tile_task=1024, img_end_game=0)),
                       ("deferred", dict(defer_peel=3))):
        eng = hyperion_amd.Engine(prob)
        eng.lucy_iteration(50_000, 1)
        for k, v in opts.items():
            eng.set_option(k, v)
        got, sg = eng.final_iteration(n)
        seen[name] = (eng.get_option("last_tiled_imaging"), eng.get_option("last_end_game"), eng.get_option("last_generations"))
        eng.close()
        for k in INT_KEYS:
            assert sg[k] == sw[k], (name, k, sg, sw)
        for ga, gb in zip(got, want):
            for key in gb:
                np.testing.assert_allclose(ga[key], gb[key], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(gb[key])), err_msg=name + " " + key)
    assert seen["end-game"][0] == 1 and seen["end-game"][1] > 0, seen
    assert seen["generations"][0] == 1 and seen["generations"][1] == 0, seen
    assert seen["deferred"][0] == 0
    assert seen["end-game"][2] < seen["generations"][2], seen           # the tail of generations is gone


def test_tiled_imaging_end_game_at_scale():
    """4e6 imaging packets of the same model on a 64^3 grid.  Forced onto the tiled schedule, the end-game stops the generations when
    the last id is out and one packet per lane of the deferred grid is left: tallies equal to the deferred schedule's and to the
    generations-to-the-end run, cubes to rounding, a quarter of the generations, less time.  Left to itself (defer_peel = 1) the engine
    does NOT tile this model -- five cell crossings per flight: the deferred rounds are faster (0.122 against 0.162 s) -- and its
    choice is within 1.3 x the best of the three."""
    import time
    prob = _thick_scattering_problem(n=64)
    n = 4_000_000
    res = {}
    for name, opts in (("auto", {}), ("end-game", dict(defer_peel=2)), ("generations", dict(defer_peel=2, img_end_game=0)), ("deferred", dict(defer_peel=3))):
        eng = hyperion_amd.Engine(prob)
        eng.lucy_iteration(100_000, 1)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.final_iteration(n // 8)          # warm-up: buffers, tables
        t0 = time.perf_counter()
        got, st = eng.final_iteration(n)
        dt = time.perf_counter() - t0
        res[name] = (got, st, dt, eng.get_option("last_tiled_imaging"), eng.get_option("last_end_game"), eng.get_option("last_generations"))
        eng.close()
    print({k: (v[2], v[3], v[4], v[5]) for k, v in res.items()})
    assert res["end-game"][3] == 1 and res["end-game"][4] > 0 and res["generations"][3] == 1 and res["deferred"][3] == 0
    assert res["auto"][3] == 0          # short flights: the deferred rounds
    for name in ("generations", "deferred", "auto"):
        for k in INT_KEYS:
            assert res["end-game"][1][k] == res[name][1][k], (name, k)
        for ga, gb in zip(res["end-game"][0], res[name][0]):
            for key in gb:
                np.testing.assert_allclose(ga[key], gb[key], rtol=1e-9, atol=1e-11 * np.nanmax(np.abs(gb[key])), err_msg=name + " " + key)
    assert res["end-game"][5] < 400 and res["end-game"][5] < res["generations"][5]
    assert res["end-game"][2] < res["generations"][2] * 1.02
    assert res["auto"][2] < 1.3 * min(v[2] for v in res.values()), {k: v[2] for k, v in res.items()}
