"""Statistics of the reference's peel-off goldens (test_peeloff.grid_type=*.raytracing=*.rtout,
hyperion/model/tests/test_bit_level.py:175-236) against an ensemble of realisations computed by a
runner -- the CPU oracle (tests/test_oracle_golden.py) or the HIP engine (tests/test_gpu_golden.py),
which share the call surface lucy_iteration / final_iteration / raytracing_iteration / close.

The golden is ONE realisation of 5 x 1000 Lucy + 5000 imaging packets with the reference's own RNG;
the runner supplies the expectation and, from K realisations at the golden's packet numbers, the
noise of every statistic.  Three things are measured here that sums of cubes cannot see:

  * the AMPLITUDE of the linear polarisation (scatter_stokes, src/dust/dust_type_4elem.f90:603-690,
    P1..P4 interpolation :543-546): a matched filter of the golden's Q and U image pixels on the
    expected Q and U pattern, a = sum(w g m) / sum(w m^2).  a = 1 if the polarisation degree is
    right, 0.5 if it were halved, -1 if its sign were wrong;
  * the GEOMETRY of the images (image axes, src/images/images_peeled.f90:209-211; pixel index,
    src/images/image_type.f90:364-365): the goldens image five off-centre point sources, so pixel
    by pixel z-scores and the correlation with the expected image and with its mirror images;
  * the FLUX scale (peel-off weights, images_peeled.f90:218-254): golden / expected totals, to be
    pooled over all goldens by the caller.
"""
import numpy as np

from cases import golden_problem
from hyperion_amd.images import finalize_peeled

_CACHE = {}


def peeloff_run(make, prob, seed, n_lucy, n_img):
    """program main's sequence for these models with runner `make(prob)`; cubes as image_write leaves them."""
    prob.config.seed = seed
    o = make(prob)
    for it in range(1, prob.config.n_initial_iter + 1):
        o.lucy_iteration(n_lucy, it)
    res, st = o.final_iteration(n_img)
    if prob.config.raytracing:
        # main.f90:296-303: the raytracing iteration adds direct and thermal emission to the cubes
        scale = n_img / 5000.0
        res, st2 = o.raytracing_iteration(int(prob.config.n_ray_photons_sources * scale), int(prob.config.n_ray_photons_dust * scale))
        st["killed_geo"] += st2["killed_geo"]
        st["killed_int"] += st2["killed_int"]
    o.close()
    return [finalize_peeled(p, r) for p, r in zip(prob.peeled, res)], st


def _corr(a, b):
    a, b = a.ravel() - a.mean(), b.ravel() - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def _matched_amplitude(x, m, w):
    return float((w * x * m).sum() / (w * m * m).sum())


def _with_seed(prob, seed):
    """A shallow copy of the problem with its own run configuration: the realisations of an ensemble may run side by side."""
    import copy
    p = copy.copy(prob)
    p.config = copy.copy(prob.config)
    p.config.seed = seed
    return p


def ensemble(fn, seeds, workers=1):
    """[fn(seed) for seed in seeds], on a thread pool when the runner can run side by side (the oracle: one OpenMP thread per call)."""
    if workers <= 1:
        return [fn(s) for s in seeds]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(fn, seeds))


def peeloff_golden_stats(make, grid, evenly, ray, k=40, big=(100000, 600000), seed0=100, make_serial=None, workers=1):
    """Everything the tests assert about one peel-off golden, computed once per (runner, golden).

    Expectation: one large run when the cubes are linear in the packets' contributions (raytracing off), the mean of the
    K realisations when the raytraced thermal emission makes them depend non-linearly on temperatures that come from
    5 x 1000 Lucy packets (raytracing on); in that case every statistic of realisation i uses the mean of the others."""
    key = (getattr(make, "__name__", repr(make)), grid, bool(evenly), bool(ray), k, big)
    if key in _CACHE:
        return _CACHE[key]
    prob, z = golden_problem("%s_peeloff%s.%s.npz" % (grid, "_ray" if ray else "", evenly))
    runs = ensemble(lambda seed: peeloff_run(make_serial or make, _with_seed(prob, seed), seed, 1000, 5000),
                    [-(seed0 + i) for i in range(k)], workers if make_serial is not None else 1)
    killed = sum(r[1]["killed_geo"] + r[1]["killed_int"] for r in runs)
    samples = [r[0] for r in runs]
    if ray:
        bigres = None
    else:
        bigres, st = peeloff_run(make, prob, -5, big[0], big[1])
        killed += st["killed_geo"] + st["killed_int"]
    out = {"killed": killed, "groups": [], "n_groups": len(prob.peeled), "k": k, "problem": prob, "golden": z,
           "samples": samples, "big": bigres}
    amp_num = {"gold": 0.0, "samples": np.zeros(k)}
    amp_den = 0.0
    amp_den_s = np.zeros(k)
    for g in range(len(prob.peeled)):
        G = {}
        for name in ("seds", "images"):
            gold = z["golden/group%d/%s" % (g + 1, name)]
            cube = np.array([s[g][name] for s in samples])
            assert gold.shape == cube.shape[1:], (name, gold.shape, cube.shape)
            sd = cube.std(axis=0, ddof=1)
            if ray:
                mean = cube.mean(axis=0)
                sd = sd * np.sqrt(1.0 + 1.0 / k)
            else:
                mean = bigres[g][name]
            G[name] = {"gold": gold, "mean": mean, "sd": sd, "cube": cube}
        # --- image geometry: Stokes I summed over origins and wavelengths, per (view, y, x) pixel
        im = G["images"]
        gI, mI = im["gold"][0].sum(axis=(0, 4)), im["mean"][0].sum(axis=(0, 4))
        cI = im["cube"][:, 0].sum(axis=(1, 5))
        sI = cI.std(axis=0, ddof=1) * (np.sqrt(1.0 + 1.0 / k) if ray else 1.0)
        G["pixel_gold"], G["pixel_mean"], G["pixel_sd"] = gI, mI, sI
        G["corr"] = [_corr(gI[v], mI[v]) for v in range(gI.shape[0])]
        G["corr_mirror_x"] = [_corr(gI[v], mI[v][:, ::-1]) for v in range(gI.shape[0])]
        G["corr_mirror_y"] = [_corr(gI[v], mI[v][::-1, :]) for v in range(gI.shape[0])]
        G["corr_rot180"] = [_corr(gI[v], mI[v][::-1, ::-1]) for v in range(gI.shape[0])]
        # --- flux scale: all wavelengths of the largest aperture / of the image, all views and origins
        sed = G["seds"]
        G["sed_total"] = (float(sed["gold"][0][:, :, -1, :].sum()), float(sed["mean"][0][:, :, -1, :].sum()),
                          float(sed["cube"][:, 0][:, :, :, -1, :].sum(axis=(1, 2, 3)).std(ddof=1)))
        G["image_total"] = (float(gI.sum()), float(mI.sum()), float(cI.sum(axis=(1, 2, 3)).std(ddof=1)))
        if sed["gold"].shape[1] == 4:
            # 'basic' origin tracking (orig(), image_type.f90:117-134): source emission, dust emission, scattered source, scattered
            # dust.  Each class has its own peel-off weight (images_peeled.f90:218-254): emitted isotropically by a source, by
            # the dust, or redirected with the phase function.
            for name, sl in (("source", [0]), ("dust", [1]), ("scattered", [2, 3])):
                G["sed_" + name] = (float(sed["gold"][0][sl][:, :, -1, :].sum()), float(sed["mean"][0][sl][:, :, -1, :].sum()),
                                    float(sed["cube"][:, 0][:, sl][:, :, :, -1, :].sum(axis=(1, 2, 3)).std(ddof=1)))
        # --- polarisation amplitude: Q and U pixels (summed over origins: the groups with origin tracking bin the same
        # packets), matched filter on the expected pattern, weights 1 / variance of a realisation's pixel
        for ist in (1, 2):
            # summed over wavelengths as well: at long wavelengths a scattering is a rare event of large weight, the
            # variance of such a bin cannot be estimated from K realisations; the sum is dominated by the optical and
            # near-infrared bins, where every realisation holds hundreds of scatterings per pixel.  The floor on the
            # variance keeps a pixel that happens to scatter little in the K realisations from carrying the sum.
            gq, mq = im["gold"][ist].sum(axis=(0, 4)), im["mean"][ist].sum(axis=(0, 4))
            cq = im["cube"][:, ist].sum(axis=(1, 5))
            var = cq.var(axis=0, ddof=1)
            w = 1.0 / (var + 1e-2 * var.max())
            G["amp_terms_%d" % ist] = (float((w * gq * mq).sum()), float((w * mq * mq).sum()))
            if g == 0 or g == 1:        # group 3 (detailed tracking) bins the packets of group 2 again
                amp_num["gold"] += (w * gq * mq).sum()
                amp_den += (w * mq * mq).sum()
                for i in range(k):
                    mi = (cube_mean_without(cq, i) if ray else mq)
                    amp_num["samples"][i] += (w * cq[i] * mi).sum()
                    amp_den_s[i] += (w * mi * mi).sum()
        out["groups"].append(G)
    out["amp_gold"] = float(amp_num["gold"] / amp_den)
    out["amp_samples"] = amp_num["samples"] / amp_den_s
    out["amp_snr2"] = float(amp_den)          # sum (m / sigma)^2: the polarised signal of one realisation, in sigma^2
    _CACHE[key] = out
    return out


def cube_mean_without(c, i):
    return (c.sum(axis=0) - c[i]) / (c.shape[0] - 1)


def pooled(values, sigmas):
    """Inverse-variance weighted mean and its standard error."""
    v, s = np.asarray(values, dtype=float), np.asarray(sigmas, dtype=float)
    w = 1.0 / (s * s)
    return float((w * v).sum() / w.sum()), float(1.0 / np.sqrt(w.sum()))


ALL_PEELOFF_GOLDENS = [(g, e, r) for r in (False, True) for g in ("car", "oct", "amr", "sph", "cyl") for e in (False, True)]


def pooled_peeloff_statistics(make, goldens=ALL_PEELOFF_GOLDENS, **kw):
    """Over the given goldens: pooled polarisation amplitude (value, standard error from the realisations' own scatter of the
    same estimator), pooled golden / expected flux of SEDs and of images (value, standard error), and the per-golden numbers."""
    amp, amp_sd, fs, fs_sd, fi, fi_sd, per = [], [], [], [], [], [], {}
    cls = {"source": ([], []), "dust": ([], []), "scattered": ([], [])}
    for grid, evenly, ray in goldens:
        S = peeloff_golden_stats(make, grid, evenly, ray, **kw)
        a_sd = float(np.std(S["amp_samples"], ddof=1))
        amp.append(S["amp_gold"]); amp_sd.append(a_sd)
        G = S["groups"][0]      # the three groups bin the same packets: flux from the first (two views, no tracking)
        g_, m_, sd_ = G["sed_total"]
        fs.append(g_ / m_); fs_sd.append(sd_ / m_)
        g_, m_, sd_ = G["image_total"]
        fi.append(g_ / m_); fi_sd.append(sd_ / m_)
        for name in cls:          # by origin class, from the group with 'basic' tracking
            g_, m_, sd_ = S["groups"][1]["sed_" + name]
            if m_ > 0 and sd_ > 0:
                cls[name][0].append(g_ / m_); cls[name][1].append(sd_ / m_)
        per[(grid, evenly, ray)] = {"amp": S["amp_gold"], "amp_sd": a_sd, "amp_samples_mean": float(np.mean(S["amp_samples"])),
                                    "sed": fs[-1], "sed_sd": fs_sd[-1], "image": fi[-1], "image_sd": fi_sd[-1]}
    out = {"amp": pooled(amp, amp_sd), "sed": pooled(fs, fs_sd), "image": pooled(fi, fi_sd), "per": per}
    for name in cls:
        out["sed_" + name] = pooled(*cls[name]) + (len(cls[name][0]),)
    return out
