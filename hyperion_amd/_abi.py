"""ctypes mirror of ``include/hyperion_amd.h`` and the marshalling of a
:class:`hyperion_amd.problem.Problem` into it.  The structs are plain C (no
torch types).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .problem import Problem, SUBLIMATION_MODES, TRACK_ORIGIN

ABI_VERSION = 2      # HYP_ABI_VERSION of include/hyperion_amd.h these mirrors follow

MAX_DUST = 8
_dp = C.POINTER(C.c_double)


class DustDesc(C.Structure):
    _fields_ = [
        ("n_nu", C.c_int32), ("n_mu", C.c_int32), ("n_jnu", C.c_int32), ("n_enu", C.c_int32),
        ("n_e", C.c_int32), ("sublimation_mode", C.c_int32), ("version", C.c_int32), ("is_lte", C.c_int32),
        ("sublimation_specific_energy", C.c_double), ("minimum_specific_energy", C.c_double),
        ("nu", _dp), ("albedo", _dp), ("chi", _dp), ("mu", _dp),
        ("P1", _dp), ("P2", _dp), ("P3", _dp), ("P4", _dp),
        ("emiss_nu", _dp), ("emiss_jnu", _dp), ("emiss_var", _dp),
        ("mo_specific_energy", _dp), ("mo_chi_rosseland", _dp),
        ("mo_kappa_planck", _dp), ("mo_chi_inv_planck", _dp),
    ]


class SpotDesc(C.Structure):
    _fields_ = [
        ("longitude", C.c_double), ("latitude", C.c_double), ("radius", C.c_double), ("luminosity", C.c_double),
        ("temperature", C.c_double), ("spectrum_type", C.c_int32), ("n_spec", C.c_int32), ("spec_nu", _dp), ("spec_fnu", _dp),
    ]


class SourceDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("spectrum_type", C.c_int32), ("peeloff", C.c_int32), ("n_spec", C.c_int32),
        ("limb_darkening", C.c_int32), ("n_points", C.c_int32),
        ("luminosity", C.c_double), ("temperature", C.c_double),
        ("position", C.c_double * 3), ("radius", C.c_double), ("box", C.c_double * 6),
        ("spec_nu", _dp), ("spec_fnu", _dp),
        ("direction", C.c_double * 2), ("points", _dp), ("point_lum", _dp), ("map", _dp),
        ("n_spots", C.c_int32), ("reserved_spots", C.c_int32), ("spots", C.POINTER(SpotDesc)),
    ]


class GridDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32), ("n3", C.c_int32),
        ("w1", _dp), ("w2", _dp), ("w3", _dp),
        ("n_cells", C.c_int64), ("refined", C.POINTER(C.c_int32)),
        ("oct_center", C.c_double * 3), ("oct_half", C.c_double * 3),
        ("vor_sites", _dp), ("vor_volume", _dp),
        ("vor_idx", C.POINTER(C.c_int32)), ("vor_neighs", C.POINTER(C.c_int32)), ("vor_box", C.c_double * 6),
        ("n_amr_levels", C.c_int32), ("n_amr_grids", C.c_int32),
        ("amr_level", C.POINTER(C.c_int32)), ("amr_n", C.POINTER(C.c_int32)), ("amr_bounds", _dp),
        ("vor_bb", _dp),
    ]


class Config(C.Structure):
    _fields_ = [
        ("seed", C.c_int64), ("n_inter_max", C.c_int64), ("n_reabs_max", C.c_int64),
        ("kill_on_absorb", C.c_int32), ("kill_on_scatter", C.c_int32),
        ("sample_sources_evenly", C.c_int32), ("enforce_energy_range", C.c_int32),
        ("forced_first_interaction", C.c_int32), ("forced_first_interaction_algorithm", C.c_int32),
        ("specific_energy_type", C.c_int32), ("raytracing", C.c_int32),
        ("baes16_xi", C.c_double), ("propagation_check_frequency", C.c_double),
        ("n_inter_mrw_max", C.c_int64), ("mrw_gamma", C.c_double), ("mrw", C.c_int32), ("monochromatic", C.c_int32),
        ("monochromatic_energy_threshold", C.c_double), ("frequencies", _dp), ("n_frequencies", C.c_int32), ("reserved2", C.c_int32),
        ("pda", C.c_int32), ("count_photons", C.c_int32), ("n_spectrum_bins", C.c_int32), ("reserved3", C.c_int32),
        ("spectrum_bin_edges", _dp),
    ]


class PeeledDesc(C.Structure):
    _fields_ = [
        ("n_view", C.c_int32), ("inside_observer", C.c_int32), ("ignore_optical_depth", C.c_int32),
        ("compute_image", C.c_int32), ("compute_sed", C.c_int32), ("n_x", C.c_int32), ("n_y", C.c_int32),
        ("n_ap", C.c_int32), ("n_nu", C.c_int32), ("track_origin", C.c_int32), ("track_n_scat", C.c_int32),
        ("uncertainties", C.c_int32), ("compute_stokes", C.c_int32), ("reserved0", C.c_int32),
        ("x_min", C.c_double), ("x_max", C.c_double), ("y_min", C.c_double), ("y_max", C.c_double),
        ("ap_min", C.c_double), ("ap_max", C.c_double), ("nu_min", C.c_double), ("nu_max", C.c_double),
        ("d_min", C.c_double), ("d_max", C.c_double), ("peeloff_origin", C.c_double * 3),
        ("theta", _dp), ("phi", _dp), ("inu_min", C.c_int32), ("inu_max", C.c_int32),
        ("use_filters", C.c_int32), ("reserved_f", C.c_int32),
        ("filt_n", C.POINTER(C.c_int32)), ("filt_nu", _dp), ("filt_tr", _dp),
    ]


class ProblemDesc(C.Structure):
    _fields_ = [
        ("grid", GridDesc), ("config", Config),
        ("n_dust", C.c_int32), ("n_sources", C.c_int32), ("n_peeled", C.c_int32), ("reserved0", C.c_int32),
        ("dust", C.POINTER(DustDesc)), ("sources", C.POINTER(SourceDesc)), ("peeled", C.POINTER(PeeledDesc)),
        ("density", _dp), ("specific_energy", _dp),
        ("binned", C.POINTER(PeeledDesc)), ("n_binned_theta", C.c_int32), ("n_binned_phi", C.c_int32),
    ]


class IterStats(C.Structure):
    _fields_ = [
        ("energy_current", C.c_double), ("energy_abs_tot", C.c_double * MAX_DUST),
        ("killed_geo", C.c_uint64), ("killed_int", C.c_uint64),
        ("crossings", C.c_uint64), ("interactions", C.c_uint64), ("n_packets", C.c_uint64),
    ]

    def as_dict(self):
        return {"energy_current": self.energy_current, "energy_abs_tot": list(self.energy_abs_tot),
                "killed_geo": self.killed_geo, "killed_int": self.killed_int,
                "crossings": self.crossings, "interactions": self.interactions,
                "n_packets": self.n_packets}


SOURCE_TYPES = {"point": 1, "sphere": 2, "map": 4, "extern_sph": 5, "extern_box": 6, "plane_parallel": 7, "point_collection": 8}


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


class MarshalledProblem:
    """Owns the C structs and keeps every referenced numpy buffer alive."""

    def __init__(self, prob: Problem):
        self._keep = []
        keep = self._keep.append

        def arr(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep(a)
            return _ptr(a)

        d = ProblemDesc()
        if prob.grid_type in ("car", "sph_pol", "cyl_pol"):
            n1, n2, n3 = prob.shape
            d.grid.type = {"car": 1, "sph_pol": 5, "cyl_pol": 6}[prob.grid_type]
            d.grid.n1, d.grid.n2, d.grid.n3 = n1, n2, n3
            d.grid.w1, d.grid.w2, d.grid.w3 = (arr(w) for w in prob.walls)
            d.grid.n_cells = n1 * n2 * n3
        elif prob.grid_type == "oct":
            d.grid.type = 2
            ref = np.ascontiguousarray(prob.refined, dtype=np.int32)
            keep(ref)
            d.grid.n_cells = ref.size
            d.grid.refined = ref.ctypes.data_as(C.POINTER(C.c_int32))
            for k in range(3):
                d.grid.oct_center[k] = float(prob.oct_center[k])
                d.grid.oct_half[k] = float(prob.oct_half[k])
        elif prob.grid_type == "vor":
            d.grid.type = 3
            d.grid.n_cells = prob.vor_sites.shape[0]
            d.grid.vor_sites = arr(prob.vor_sites)
            d.grid.vor_volume = arr(prob.vor_volume)
            idx = np.ascontiguousarray(prob.vor_idx, dtype=np.int32)
            nei = np.ascontiguousarray(prob.vor_neighs, dtype=np.int32)
            keep(idx)
            keep(nei)
            d.grid.vor_idx = idx.ctypes.data_as(C.POINTER(C.c_int32))
            d.grid.vor_neighs = nei.ctypes.data_as(C.POINTER(C.c_int32))
            for k in range(6):
                d.grid.vor_box[k] = float(prob.vor_box[k])
            if prob.vor_bb is not None:
                d.grid.vor_bb = arr(np.asarray(prob.vor_bb, dtype=np.float64).reshape(-1, 6))
        elif prob.grid_type == "amr":
            d.grid.type = 4
            lev = np.ascontiguousarray(prob.amr_level, dtype=np.int32)
            nn = np.ascontiguousarray(prob.amr_n, dtype=np.int32)
            keep(lev)
            keep(nn)
            d.grid.n_cells = prob.n_cells
            d.grid.n_amr_grids = lev.size
            d.grid.n_amr_levels = int(lev.max())
            d.grid.amr_level = lev.ctypes.data_as(C.POINTER(C.c_int32))
            d.grid.amr_n = nn.ctypes.data_as(C.POINTER(C.c_int32))
            d.grid.amr_bounds = arr(prob.amr_bounds)
        else:
            raise ValueError("Unexpected coordinate type: %s" % prob.grid_type)

        c = prob.config
        d.config.seed = int(c.seed)
        d.config.n_inter_max = int(c.n_inter_max)
        d.config.n_reabs_max = int(c.n_reabs_max)
        d.config.kill_on_absorb = int(c.kill_on_absorb)
        d.config.kill_on_scatter = int(c.kill_on_scatter)
        d.config.sample_sources_evenly = int(c.sample_sources_evenly)
        d.config.enforce_energy_range = int(c.enforce_energy_range)
        d.config.forced_first_interaction = int(c.forced_first_interaction)
        algo = {"wr99": 1, "baes16": 2}.get(c.forced_first_interaction_algorithm)
        if algo is None:
            raise ValueError("Unknown forced first interaction algorithm: %s"
                             % c.forced_first_interaction_algorithm)
        d.config.forced_first_interaction_algorithm = algo
        if c.specific_energy_type not in ("initial", "additional"):
            raise ValueError("specific_energy_type should be 'additional' or 'initial'")
        d.config.specific_energy_type = 1 if c.specific_energy_type == "additional" else 0
        d.config.raytracing = int(bool(c.raytracing))
        d.config.mrw = int(bool(c.mrw))
        d.config.mrw_gamma = float(c.mrw_gamma)
        d.config.n_inter_mrw_max = int(c.n_inter_mrw_max)
        d.config.baes16_xi = float(c.baes16_xi)
        d.config.monochromatic = int(bool(c.monochromatic))
        d.config.monochromatic_energy_threshold = float(c.monochromatic_energy_threshold)
        if c.monochromatic:
            d.config.frequencies = arr(c.frequencies)
            d.config.n_frequencies = int(np.asarray(c.frequencies).size)
        d.config.propagation_check_frequency = float(c.propagation_check_frequency)
        d.config.pda = int(bool(c.pda))
        d.config.count_photons = int(bool(c.pda) or c.output_n_photons != "none")
        if c.output_specific_energy_spectrum != "none":
            if c.spectrum_bin_edges is None:
                raise ValueError("specific_energy_spectrum_bin_edges should be present in the input when "
                                 "output_specific_energy_spectrum is enabled")
            edges = np.asarray(c.spectrum_bin_edges, dtype=np.float64)
            if edges.size < 2 or np.any(edges[1:] <= edges[:-1]):
                raise ValueError("specific_energy_spectrum_bin_edges should be strictly increasing")
            d.config.n_spectrum_bins = edges.size - 1
            d.config.spectrum_bin_edges = arr(edges)

        nd = prob.n_dust
        if nd > MAX_DUST:
            raise ValueError("at most %d dust species are supported" % MAX_DUST)
        dusts = (DustDesc * max(nd, 1))()
        for i, du in enumerate(prob.dust):
            x = dusts[i]
            x.n_nu, x.n_mu = du.nu.size, du.mu.size
            x.n_jnu, x.n_enu = du.emiss_var.size, du.emiss_nu.size
            x.n_e = 0 if du.mo_specific_energy is None else du.mo_specific_energy.size
            x.sublimation_mode = SUBLIMATION_MODES[du.sublimation_mode]
            x.version = int(du.version)
            x.is_lte = int(du.is_lte)
            x.sublimation_specific_energy = float(du.sublimation_specific_energy)
            x.minimum_specific_energy = float(du.minimum_specific_energy)
            for k in ("nu", "albedo", "chi", "mu", "P1", "P2", "P3", "P4",
                      "emiss_nu", "emiss_jnu", "emiss_var"):
                setattr(x, k, arr(getattr(du, k)))
            x.mo_specific_energy = arr(du.mo_specific_energy) if du.mo_specific_energy is not None else None
            x.mo_chi_rosseland = arr(du.mo_chi_rosseland) if du.mo_chi_rosseland is not None else None
            x.mo_kappa_planck = arr(du.mo_kappa_planck) if du.mo_kappa_planck is not None else None
            x.mo_chi_inv_planck = arr(du.mo_chi_inv_planck) if du.mo_chi_inv_planck is not None else None
        keep(dusts)
        d.n_dust = nd
        d.dust = C.cast(dusts, C.POINTER(DustDesc))

        ns = len(prob.sources)
        srcs = (SourceDesc * max(ns, 1))()
        for i, s in enumerate(prob.sources):
            x = srcs[i]
            if s.type not in SOURCE_TYPES:
                raise ValueError("unknown type in source list: %s" % s.type)
            x.type = SOURCE_TYPES[s.type]
            x.peeloff = int(s.peeloff)
            x.limb_darkening = int(bool(s.limb_darkening))
            x.direction[0], x.direction[1] = float(s.direction[0]), float(s.direction[1])
            if s.type == "point_collection":
                pts = np.ascontiguousarray(s.points, dtype=np.float64).reshape(-1, 3)
                x.n_points = pts.shape[0]
                x.points = arr(pts)
                x.point_lum = arr(s.point_luminosity)
            if s.spots:
                if s.type != "sphere":
                    raise ValueError("only spherical sources can have spots")
                sp = (SpotDesc * len(s.spots))()
                for k, q in enumerate(s.spots):
                    sp[k].longitude, sp[k].latitude, sp[k].radius = float(q.longitude), float(q.latitude), float(q.radius)
                    sp[k].luminosity = float(q.luminosity)
                    if q.spectrum_nu is not None:
                        sp[k].spectrum_type, sp[k].n_spec = 1, int(np.size(q.spectrum_nu))
                        sp[k].spec_nu, sp[k].spec_fnu = arr(q.spectrum_nu), arr(q.spectrum_fnu)
                    elif q.temperature is not None:
                        sp[k].spectrum_type, sp[k].temperature = 2, float(q.temperature)
                    else:
                        raise ValueError("Spot cannot have LTE spectrum")
                keep(sp)
                x.n_spots = len(s.spots)
                x.spots = C.cast(sp, C.POINTER(SpotDesc))
            if s.type == "map":
                if s.map is None or np.size(s.map) != prob.n_cells:
                    raise ValueError("map source needs a luminosity map with one value per cell")
                x.map = arr(np.ascontiguousarray(s.map, dtype=np.float64).ravel())
            x.luminosity = float(s.luminosity)
            for k in range(3):
                x.position[k] = float(s.position[k])
            x.radius = float(s.radius)
            for k in range(6):
                x.box[k] = float(s.box[k])
            if s.spectrum_nu is not None:
                x.spectrum_type = 1
                x.n_spec = int(np.size(s.spectrum_nu))
                x.spec_nu = arr(s.spectrum_nu)
                x.spec_fnu = arr(s.spectrum_fnu)
            elif s.temperature is not None:
                x.spectrum_type = 2
                x.temperature = float(s.temperature)
            elif s.lte and s.type == "map":
                x.spectrum_type = 3
            else:
                raise ValueError("source needs a spectrum or a temperature")
        keep(srcs)
        d.n_sources = ns
        d.sources = C.cast(srcs, C.POINTER(SourceDesc))

        npl = len(prob.peeled)
        groups = list(prob.peeled) + ([prob.binned] if prob.binned is not None else [])
        pls = (PeeledDesc * max(len(groups), 1))()
        for i, p in enumerate(groups):
            x = pls[i]
            x.n_view = p.n_view
            x.inside_observer = int(p.inside_observer)
            x.ignore_optical_depth = int(p.ignore_optical_depth)
            x.compute_image = int(p.compute_image)
            x.compute_sed = int(p.compute_sed)
            x.n_x, x.n_y, x.n_ap, x.n_nu = int(p.n_x), int(p.n_y), int(p.n_ap), int(p.n_wav)
            x.track_origin = TRACK_ORIGIN[p.track_origin]
            x.track_n_scat = int(p.track_n_scat)
            x.uncertainties = int(p.uncertainties)
            x.compute_stokes = int(p.compute_stokes)
            x.x_min, x.x_max, x.y_min, x.y_max = p.x_min, p.x_max, p.y_min, p.y_max
            x.ap_min, x.ap_max = p.ap_min, p.ap_max
            x.nu_min, x.nu_max = p.nu_min, p.nu_max
            x.d_min, x.d_max = p.d_min, p.d_max
            for k in range(3):
                x.peeloff_origin[k] = float(p.peeloff_origin[k])
            x.theta = arr(p.theta)
            x.phi = arr(p.phi)
            x.inu_min, x.inu_max = int(p.inu_min), int(p.inu_max)
            if p.filters:
                x.use_filters = 1
                fn = np.array([np.size(f[0]) for f in p.filters], dtype=np.int32)
                keep(fn)
                x.filt_n = fn.ctypes.data_as(C.POINTER(C.c_int32))
                x.filt_nu = arr(np.concatenate([np.asarray(f[0], dtype=np.float64) for f in p.filters]))
                x.filt_tr = arr(np.concatenate([np.asarray(f[1], dtype=np.float64) for f in p.filters]))
                x.n_nu = len(p.filters)
        keep(pls)
        d.n_peeled = npl
        d.peeled = C.cast(pls, C.POINTER(PeeledDesc))
        if prob.binned is not None:
            d.binned = C.cast(C.byref(pls, npl * C.sizeof(PeeledDesc)), C.POINTER(PeeledDesc))
            d.n_binned_theta, d.n_binned_phi = int(prob.n_binned_theta), int(prob.n_binned_phi)

        d.density = arr(prob.density)
        d.specific_energy = arr(prob.specific_energy) if prob.specific_energy is not None else None
        self.desc = d
        self.problem = prob

    def peeled_shapes(self, g, n_orig):
        """(sed_shape, img_shape) of group g in the .rtout layout."""
        if g == len(self.problem.peeled):        # the binned group: n_theta x n_phi views
            p = self.problem.binned
            n_view = self.problem.n_binned_theta * self.problem.n_binned_phi
        else:
            p = self.problem.peeled[g]
            n_view = p.n_view
        ns = 4 if p.compute_stokes else 1
        sed = (ns, n_orig, n_view, p.n_ap, p.n_wav) if p.compute_sed else None
        img = (ns, n_orig, n_view, p.n_y, p.n_x, p.n_wav) if p.compute_image else None
        return sed, img
