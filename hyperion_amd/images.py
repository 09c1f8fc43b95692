"""Write-time normalisation of the peeled SED / image cubes
(``image_write``, ``src/images/image_type.f90:608-788``): the engine returns the
raw flux sums already scaled by ``energy_total/energy_current``; here they are
converted to nu*F_nu, uncertainties become sqrt(sum x^2), and SED apertures are
accumulated outwards."""
from __future__ import annotations

import numpy as np


def dnunorm(peeled):
    r = peeled.nu_max / peeled.nu_min
    n = float(peeled.n_wav)
    return r ** (0.5 / n) - r ** (-0.5 / n)


def finalize_peeled(peeled, raw, frequencies=None):
    """raw: {'sed','sed2','img','img2'} in the .rtout layout
    (n_stokes, n_orig, n_view, n_ap, n_nu) / (n_stokes, n_orig, n_view, n_y, n_x, n_nu)."""
    out = {}
    if frequencies is not None:
        # use_exact_nu (image_type.f90:675-683, 733-741): F_nu at the group's frequencies -> nu F_nu
        nu = np.asarray(frequencies, dtype=float)[peeled.inu_min - 1:peeled.inu_max]
        if "sed" in raw:
            out["seds"] = np.cumsum(raw["sed"] * nu, axis=3)
            if peeled.uncertainties:
                unc = np.sqrt(raw["sed2"]) * nu
                out["seds_unc"] = np.sqrt(np.cumsum(unc * unc, axis=3))
        if "img" in raw:
            out["images"] = raw["img"] * nu
            if peeled.uncertainties:
                out["images_unc"] = np.sqrt(raw["img2"]) * nu
        return out
    # with filters the flux stays F_nu dnu: the filter already carries the normalisation (image_type.f90:644-651)
    norm = 1.0 if getattr(peeled, "filters", None) else dnunorm(peeled)
    if "sed" in raw:
        sed = raw["sed"] / norm
        out["seds"] = np.cumsum(sed, axis=3)
        if peeled.uncertainties:
            unc = np.sqrt(raw["sed2"]) / norm
            out["seds_unc"] = np.sqrt(np.cumsum(unc * unc, axis=3))
    if "img" in raw:
        out["images"] = raw["img"] / norm
        if peeled.uncertainties:
            out["images_unc"] = np.sqrt(raw["img2"]) / norm
    return out
