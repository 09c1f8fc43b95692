"""The HIP path against the reference's own golden outputs (statistical, the
RNG streams differ): hyperion/model/tests/data/test_specific_energy.grid_type=car.*.rtout."""
import numpy as np
import pytest

import hyperion_amd
from cases import golden_problem

pytestmark = pytest.mark.gpu
K = 16


@pytest.mark.parametrize("name", ["False.False", "True.False", "False.True", "True.True"])
def test_all_five_iterations_match_reference_golden(name):
    prob, z = golden_problem("car_specific_energy.%s.npz" % name)
    gold = z["golden/specific_energy"]                     # (5, n_dust, 3, 5, 7)
    chains = []
    for s in range(K):
        prob.config.seed = -(7000 + s)
        eng = hyperion_amd.Engine(prob)
        chains.append([eng.lucy_iteration(10000, it)[0] for it in range(1, 6)])
        eng.close()
    chains = np.array(chains)                              # (K, 5, ...)
    prob.config.seed = -424242
    eng = hyperion_amd.Engine(prob)
    big = np.array([eng.lucy_iteration(1000000, it)[0] for it in range(1, 6)])
    eng.close()
    sigma = chains.std(axis=0, ddof=1)
    zs = (gold - big) / sigma
    assert np.abs(zs).max() < 6.5
    assert abs(zs.mean()) < 0.25
    assert 0.7 < (zs ** 2).mean() < 1.6
    w = prob.density * prob.volumes
    for it in range(5):
        assert (gold[it] * w).sum() == pytest.approx((big[it] * w).sum(), rel=0.04)
    assert (gold * w).sum() == pytest.approx((big * w).sum(), rel=0.02)


@pytest.mark.parametrize("tau", ["1000", "10000", "100000", "1000000"])
def test_pinte_benchmark_run_matches_reference_golden(tau):
    """The whole run() sequence on the GPU for the reference's Pinte benchmark model (cylindrical polar grid,
    stellar sphere, polarising dust, Lucy iterations with the convergence test and the modified random walk,
    monochromatic final iteration, raytracing): the golden test_pinte_seds.tau=*.rtout against K GPU realisations
    at its own packet numbers, exactly as tests/test_oracle_golden.py does with the oracle."""
    from hyperion_amd.run import run_problem
    prob, z = golden_problem("pinte_seds.tau=%s.npz" % tau)
    gold = z["golden/seds"]
    K = 32      # (12 until round 5: the statistic on the last iteration's absorbed luminosity below needs sigma to 13 %, not 21 %)
    # Round 6: the realisations run with `reproducible = 1` (one wave, sums in program order: tests/test_gpu_reproducible.py), eight
    # engines side by side on host threads -- a seed gives the same realisation, and this test the same numbers, on every run.  The
    # bounds that round 5 had loosened after run-to-run failures (well-sampled bins, |z| of the last iteration) are back where they were.
    import copy
    from golden_stats import ensemble
    w = prob.density * prob.volumes

    def one(k):
        p = copy.deepcopy(prob)
        p.config.seed = -(900 + k)
        r = run_problem(p, engine_options={"reproducible": 1})
        assert r.final_stats["killed_geo"] == 0
        return (r.peeled[0]["seds"], r.n_iterations, (r.iterations[-1].specific_energy * w).sum(), r.iterations[-1].specific_energy[0],
                [it.killed_int for it in r.iterations], sum(it.killed_geo for it in r.iterations))
    runs = ensemble(one, list(range(K)), 8)
    S, n_it, e_last, se_last, killed_int, killed_geo = (list(x) for x in zip(*runs))
    # the golden ran all 10 iterations without converging (99th percentile rule at 5000 packets); so does the GPU
    assert int(z["golden/iterations"]) == 10 and not bool(z["golden/converged"])
    assert min(n_it) >= 9
    S = np.array(S)
    assert S.shape[1:] == gold.shape
    I, sg = S.mean(axis=0)[0, 0, :, 0, :], S.std(axis=0, ddof=1)[0, 0, :, 0, :]
    g = gold[0, 0, :, 0, :]
    sel = (sg > 0) & (I > 1e-3 * I.max())
    zs = (g - I)[sel] / sg[sel]
    well = sg[sel] < 0.3 * I[sel]
    # (the thickest disc: only the face-on view is well sampled; the count is a sanity check on the selection, not a physical statement)
    assert well.sum() > (10 if tau == "1000000" else 20), well.sum()
    assert np.abs(zs[well]).max() < 6.0 and (zs[well] ** 2).mean() < 3.0 and abs(zs[well].mean()) < 1.0
    if (~well).any():
        assert zs[~well].min() > -6.0 and (zs[~well] > 6.0).mean() < 0.1
    # the specific energy of the last iteration, cell by cell: over the cells the realisations agree on (scatter below 30 %) the
    # golden is neither above nor below -- the median of log10(golden / mean) is zero to a few per cent (the oracle: -0.0001 at
    # tau = 1e6 over 329 cells).  This is the unbiased statistic; the TOTAL below is not: at tau = 1e6 five mid-plane cells hold two
    # thirds of it, their energies still grow from iteration to iteration (1.4e33 -> 1e36 over the ten) and differ by factors of
    # 0.2 - 1.3 between the golden and the mean, in the oracle's realisations exactly as in the device's
    se = np.array(se_last)
    m, sd = se.mean(axis=0), se.std(axis=0, ddof=1)
    gold_se = z["golden/specific_energy_last"][0]
    okc = (m > 0) & (gold_se > 0) & (sd < 0.3 * m) & (w[0] > 0)
    assert okc.sum() > 100
    assert abs(np.median(np.log10(gold_se[okc] / m[okc]))) < 0.05
    e_gold = (z["golden/specific_energy_last"] * w).sum()
    # The absorbed luminosity of the LAST iteration against the K realisations, in log space (the total is a product of
    # feedbacks through the temperatures of a few mid-plane cells and scatters log-normally: sigma = 0.02 dex at tau = 1e3,
    # 0.10 dex at tau = 1e6).  With the reproducible realisations of round 6 the statistic is a fixed number per model:
    # z = +2.26, +0.83, -0.42, -2.41 for tau = 1e3 ... 1e6 (the oracle's own 32 seeds gave +3.0, +0.7, -0.5, -3.0 in round 5: the
    # same golden against another sample of the same distribution).  Bound |z| < 4.  The two ends keep their signs:
    # the golden's last iteration is brighter than the ensemble at tau = 1e3 (one packet's random walk through six adjacent inner
    # mid-plane cells leaves 5-18 x their mean energy in the golden: +18 % on the total) and fainter at tau = 1e6 (the ten cells
    # that hold two thirds of the total sit at ranks 0.0-0.67 of the realisations' heavy-tailed distributions, sd / mean 0.4-1.4
    # per cell).  A line-by-line review of the random walk against grid_mrw_3d.f90:29-202, iter_lucy.f90:133-152 and
    # dust_type_4elem.f90:286-291,400-419 in round 6 (entry test alpha_inv_planck x distance > gamma with density > 0 only,
    # diff_coeff over ALL species, ct = -ln(y) / D (R0 / pi)^2, deposit E ct kappa_planck(E_cell), b_nu = j_nu / kappa_nu sampled at the
    # cell's bracket, opacities NOT refreshed after the walk, the 1000-step cap) found no difference; the killed-packet counters of
    # the same runs (below), which depend on exactly that path, agree with the reference's to a few per cent.
    le = np.log10(np.array(e_last))
    z_tot = (np.log10(e_gold) - le.mean()) / (le.std(ddof=1) * np.sqrt(1.0 + 1.0 / K))
    print("pinte tau=%s: z of the last iteration's absorbed luminosity %.2f, well-sampled bins %d" % (tau, z_tot, well.sum()))
    assert abs(z_tot) < 4.0, (e_gold, sorted(e_last), z_tot)
    # killed_photons_int of the ten Lucy iterations (tests/golden/killed_counts.json): the reference's own counters
    from test_oracle_golden import check_killed_counts, killed_counts
    gk = killed_counts("test_pinte_seds.tau=%s" % tau)
    assert [g[0] for g in gk["iterations"]] == [0] * 10 and all(k_geo == 0 for k_geo in killed_geo)
    n_common = min(len(k) for k in killed_int)          # (a realisation may meet the convergence rule after nine iterations)
    assert n_common >= 9
    check_killed_counts([g[1] for g in gk["iterations"]][:n_common], [k[:n_common] for k in killed_int], ("gpu", tau))


@pytest.mark.parametrize("tau", ["0.1", "1", "10", "100"])
def test_monochromatic_pascucci_benchmark_matches_reference_golden(tau):
    """The reference's Pascucci benchmark outputs (spherical polar grid, stellar sphere, monochromatic final iteration at 61
    wavelengths + raytracing) against realisations of the HIP engine: the statistics of tests/test_oracle_golden.py."""
    from test_oracle_golden import check_pascucci_golden
    check_pascucci_golden(hyperion_amd.Engine, tau)


@pytest.mark.parametrize("tau", ["1000", "10000", "100000", "1000000"])
def test_pinte_benchmark_images_match_reference_golden(tau):
    """The reference's Pinte benchmark IMAGES (cylindrical polar grid, MRW, monochromatic + raytracing, 51 x 51 Stokes images of
    two nearly edge-on views) against realisations of the HIP engine: annuli, peak pixel and totals as in tests/test_oracle_golden.py."""
    from test_oracle_golden import check_pinte_images_golden
    check_pinte_images_golden(hyperion_amd.Engine, tau)


class _EngineRunner:
    def __init__(self, prob):
        self.e = hyperion_amd.Engine(prob)

    def __call__(self, n, it):
        return self.e.lucy_iteration(n, it)[0]

    def close(self):
        self.e.close()


def test_pooled_bias_over_all_specific_energy_goldens():
    """The HIP engine against all 20 specific-energy goldens x 5 iterations of the reference (Cartesian, octree, AMR,
    spherical, cylindrical), pooled: the weighted mean ratio within 1 %, each grid type within 2.5 %."""
    from test_oracle_golden import pooled_specific_energy_bias
    mean, n, per = pooled_specific_energy_bias(_EngineRunner, n_packets=1000000)
    assert n == 100
    assert abs(mean - 1.0) < 0.01, (mean, per)
    for grid, r in per.items():
        assert abs(r - 1.0) < 0.025, (grid, r, per)


@pytest.mark.parametrize("ray", [False, True])
@pytest.mark.parametrize("grid", ["car", "oct", "amr", "sph", "cyl"])
@pytest.mark.parametrize("evenly", [False, True])
def test_peeloff_goldens_pixel_by_pixel_and_polarisation_amplitude(grid, evenly, ray):
    """The reference's twenty peel-off goldens against realisations of the HIP engine at their own packet numbers: image
    geometry pixel by pixel with the mirror-image rejection, and the amplitude of the polarised signal -- the same checks,
    with the same bounds, that tests/test_oracle_golden.py applies to the oracle."""
    from golden_stats import peeloff_golden_stats
    from test_oracle_golden import check_image_geometry, check_polarisation_amplitude
    S = peeloff_golden_stats(hyperion_amd.Engine, grid, evenly, ray)
    assert S["killed"] == 0
    check_image_geometry(S, ("gpu", grid, evenly, ray))
    check_polarisation_amplitude(S, ("gpu", grid, evenly, ray))


def test_pooled_polarisation_amplitude_and_flux_over_all_peeloff_goldens():
    """Polarisation amplitude (1 = the reference's), total / direct / scattered / thermal flux ratios pooled over the twenty
    goldens, HIP engine: bounds of test_oracle_golden.check_pooled_peeloff_statistics."""
    from golden_stats import pooled_peeloff_statistics
    from test_oracle_golden import check_pooled_peeloff_statistics
    check_pooled_peeloff_statistics(pooled_peeloff_statistics(hyperion_amd.Engine))


@pytest.mark.parametrize("tau", [1000, 1000000])
def test_pda_golden_pinte_specific_energy(tau):
    """The reference's PDA outputs (test_pinte_specific_energy.tau=*) against K realisations of the whole run on the GPU:
    same statistic as tests/test_oracle_features.py::test_pda_golden_pinte_specific_energy."""
    from hyperion_amd.run import run_problem
    prob, gold = golden_problem("pinte_specific_energy.tau=%d.npz" % tau)
    prob.config.output_n_photons = "last"
    K = 8
    logs, nph = [], []
    for seed in range(K):
        prob.config.seed = -5000 - seed
        r = run_problem(prob)
        assert r.n_iterations == 3
        logs.append(np.log10(r.iterations[-1].specific_energy[0, 0])); nph.append(r.iterations[-1].n_photons[0])
    logs, nph = np.array(logs), np.array(nph).mean(axis=0)
    g = np.log10(gold["golden/specific_energy_3"][0, 0])
    rho = prob.density[0, 0]
    mu, sd = logs.mean(axis=0), logs.std(axis=0, ddof=1)
    z = (g - mu) / np.sqrt(sd ** 2 * (1 + 1.0 / K) + 1e-6)
    interior = np.zeros(rho.shape, dtype=bool)
    interior[1:-1, 1:-1] = True
    pda_like = (nph < 30) & (rho > 0) & interior
    sampled = (nph >= 300) & (rho > 0)
    assert pda_like.sum() > 300 and sampled.sum() > 100
    zc = np.clip(z, -6.0, 6.0)
    assert abs(np.median(z[pda_like])) < 0.6 and (zc[pda_like] ** 2).mean() < 5.0 and (np.abs(z[pda_like]) > 6).mean() < 0.08
    assert abs(np.median(z[sampled])) < 0.6 and (zc[sampled] ** 2).mean() < 3.0
