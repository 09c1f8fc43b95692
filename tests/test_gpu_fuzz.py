"""Randomised cross-feature parity (round 6): the reference's own regression models on every grid type, with their
densities, sources, run configuration, image groups AND the engine's schedule options drawn at random per case, against
the CPU oracle on identical Philox streams.  The dedicated parity tests vary one feature at a time; a case here combines
e.g. a 3-species octree model, a two-interaction cap, the BAES16 forced first interaction, an `additional` specific energy,
a `scatterings` origin cube with uncertainties and the tiled schedule with 4096-slot pools.  Deterministic: a case is a
function of its number.  Integer tallies must be equal; sums to the tolerance of tests/cases.py::assert_parity."""
import copy

import numpy as np
import pytest

import hyperion_amd
from cases import assert_parity, golden_problem
from hyperion_amd.problem import PeeledImages
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu
INT_KEYS = ("crossings", "interactions", "killed_geo", "killed_int")
BASES = ["car_peeloff.False.npz", "oct_peeloff.True.npz", "amr_peeloff.False.npz", "sph_peeloff.True.npz", "cyl_peeloff.False.npz",
         "car_specific_energy.True.True.npz", "oct_specific_energy.True.True.npz", "sph_specific_energy.True.True.npz",
         "cyl_specific_energy.False.True.npz", "amr_specific_energy.True.True.npz", "vor_config5.npz"]


def random_case(case):
    rng = np.random.RandomState(1000 + case)
    prob, _ = golden_problem(BASES[case % len(BASES)])
    prob = copy.deepcopy(prob)
    cfg = prob.config
    cfg.seed = -int(rng.randint(1, 2 ** 30))
    # medium: overall optical depth over two decades, a tenth of the cells emptied (masked cells stay empty)
    rho = prob.density * 10.0 ** rng.uniform(-1.0, 0.8)
    rho = np.where(rng.uniform(size=rho.shape) < 0.1, 0.0, rho)
    prob.density = rho
    # sources: a random subset (at least one), luminosities over a decade, evenly or by luminosity
    keep = [s for s in prob.sources if rng.uniform() < 0.7] or [prob.sources[0]]
    for s in keep:
        s.luminosity = s.luminosity * 10.0 ** rng.uniform(-0.5, 0.5)
        if s.temperature is not None:
            s.temperature = float(s.temperature * rng.uniform(0.7, 1.5))
    prob.sources = keep
    cfg.sample_sources_evenly = bool(rng.uniform() < 0.5)
    # run configuration
    cfg.n_inter_max = int(rng.choice([2, 5, 1000000]))
    cfg.kill_on_absorb = bool(rng.uniform() < 0.1)
    cfg.kill_on_scatter = bool(rng.uniform() < 0.1)
    ff = rng.choice(["off", "wr99", "baes16"])
    cfg.forced_first_interaction = ff != "off"
    if ff != "off":
        cfg.forced_first_interaction_algorithm = str(ff)
        cfg.baes16_xi = float(rng.uniform(0.2, 0.8))
    cfg.propagation_check_frequency = float(rng.choice([1e-3, 0.1, 1.0]))
    if rng.uniform() < 0.4:
        prob.specific_energy = np.full(prob.density.shape, 10.0 ** rng.uniform(-3.0, -1.0))
        cfg.specific_energy_type = str(rng.choice(["initial", "additional"]))
    # image groups: one or two, random views, origin tracking, uncertainties, bins
    groups = []
    for _ in range(int(rng.randint(1, 3))):
        nv = int(rng.randint(1, 4))
        lim = float(max(abs(np.asarray(w)).max() for w in prob.walls)) if prob.grid_type in ("car",) else None
        base = prob.peeled[0] if prob.peeled else None
        xm = base.x_max if base is not None else (lim if lim else 1.0)
        groups.append(PeeledImages(theta=rng.uniform(0.0, 180.0, nv), phi=rng.uniform(0.0, 360.0, nv),
                                   n_wav=int(rng.randint(1, 6)), wav_min=0.05, wav_max=2000.0,
                                   n_x=int(rng.randint(1, 7)), n_y=int(rng.randint(1, 7)), x_min=-xm, x_max=xm, y_min=-xm, y_max=xm,
                                   n_ap=int(rng.randint(1, 4)), ap_min=0.1 * xm, ap_max=2.0 * xm,
                                   track_origin=str(rng.choice(["no", "basic", "detailed", "scatterings"])), track_n_scat=int(rng.randint(0, 4)),
                                   uncertainties=bool(rng.uniform() < 0.5), compute_stokes=True))
    prob.peeled = groups
    # the engine's schedule: persistent / tiled with pools far smaller than the packet count; inline / deferred / tiled imaging
    opts = {}
    if rng.uniform() < 0.6:
        opts.update(lucy_mode=1, tile_slots=int(rng.choice([4096, 12288, 49152])), tile_task=int(rng.choice([256, 1024])),
                    tile_drain=int(rng.choice([0, 300])), tile_pools=int(rng.randint(1, 4)))
        if prob.grid_type == "oct":
            opts["ot_cells"] = int(rng.choice([9, 200]))
        if prob.grid_type == "vor":
            opts["vt_cells"] = int(rng.choice([7, 40]))
    else:
        opts["lucy_mode"] = 0
    opts["defer_peel"] = int(rng.choice([0, 1, 2]))
    if rng.uniform() < 0.3:
        opts["peel_events"] = int(rng.choice([64 * 1024, 160 * 1024]))
    return prob, opts


@pytest.mark.parametrize("case", range(110))
def test_random_combination_matches_oracle(case):
    prob, opts = random_case(case)
    n_lucy, n_img = 12000, 8000
    eng, orc = hyperion_amd.Engine(prob), Oracle(prob)
    for k, v in opts.items():
        eng.set_option(k, v)
    for it in (1, 2):
        a, sa = eng.lucy_iteration(n_lucy, it)
        b, sb = orc.lucy_iteration(n_lucy, it)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (case, opts, it, k, sa, sb)
        assert sa["energy_current"] == pytest.approx(sb["energy_current"], rel=1e-12)
        assert_parity(a, b, atol_rel=1e-10)
    ga, sa = eng.final_iteration(n_img)
    gb, sb = orc.final_iteration(n_img)
    for k in INT_KEYS:
        assert sa[k] == sb[k], (case, opts, "final", k, sa, sb)
    for xa, xb in zip(ga, gb):
        for name in xb:
            np.testing.assert_allclose(xa[name], xb[name], rtol=1e-9, atol=1e-10 * np.nanmax(np.abs(xb[name])), err_msg="case %d %s %s" % (case, opts, name))
    eng.close(); orc.close()


@pytest.mark.parametrize("case", range(110, 176))
def test_random_combination_with_extended_sources_and_raytracing(case):
    """The same draw plus, per case: one source turned into a sphere with a radius (re-absorption and re-emission by the source,
    the general emitters and the GEN kernels of the deferred schedule), limb darkening, and the raytracing iteration on top of the
    imaging iteration (thermal emission of the dust and the sources' own light peeled off per frequency bin)."""
    prob, opts = random_case(case)
    rng = np.random.RandomState(5000 + case)
    half = {"car": 3.08568025e18, "oct": None, "amr": None, "sph_pol": None, "cyl_pol": None, "vor": None}[prob.grid_type]
    if prob.grid_type != "vor" and rng.uniform() < 0.6:       # (the Voronoi model's external source stays as it is)
        s = prob.sources[int(rng.randint(0, len(prob.sources)))]
        s.type = "sphere"
        s.radius = float(10.0 ** rng.uniform(15.0, 16.5))     # 3e-4 .. 1e-2 of the grids' half-width: inside one cell or across a few
        s.limb_darkening = bool(rng.uniform() < 0.5)
    ray = bool(rng.uniform() < 0.6)
    prob.config.raytracing = ray
    n_lucy, n_img = 10000, 6000
    eng, orc = hyperion_amd.Engine(prob), Oracle(prob)
    for k, v in opts.items():
        eng.set_option(k, v)
    for it in (1, 2):
        a, sa = eng.lucy_iteration(n_lucy, it)
        b, sb = orc.lucy_iteration(n_lucy, it)
        for k in INT_KEYS:
            assert sa[k] == sb[k], (case, opts, it, k, sa, sb)
        assert_parity(a, b, atol_rel=1e-10)
    ga, sa = eng.final_iteration(n_img)
    gb, sb = orc.final_iteration(n_img)
    for k in INT_KEYS:
        assert sa[k] == sb[k], (case, opts, "final", k, sa, sb)
    if ray:
        ga, sa = eng.raytracing_iteration(4000, 4000)
        gb, sb = orc.raytracing_iteration(4000, 4000)
        assert sa["crossings"] == sb["crossings"] and sa["killed_geo"] == sb["killed_geo"], (case, sa, sb)
    for xa, xb in zip(ga, gb):
        for name in xb:
            np.testing.assert_allclose(xa[name], xb[name], rtol=1e-9, atol=1e-10 * np.nanmax(np.abs(xb[name])), err_msg="case %d %s %s" % (case, opts, name))
    eng.close(); orc.close()
