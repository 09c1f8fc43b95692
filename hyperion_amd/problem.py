"""In-memory description of one radiative-transfer run: the flat-array form of
a Hyperion ``.rtin`` input file.

The reference passes a model from its Python front-end to its Fortran core as
an HDF5 file (``hyperion/model/model.py:1025-1080`` -> ``src/main/main.f90``);
this module is the array-level equivalent that crosses our C-ABI
(``include/hyperion_amd.h``).  Field names and meaning follow the ``.rtin``
contract (``docs/advanced/model_file.rst`` of the reference, readers
``src/main/setup_rt.f90:38-302``, ``src/dust/dust_type_4elem.f90:78-293``,
``src/sources/source_type.f90:102-322``,
``src/grid/grid_geometry_cartesian_3d.f90:77-134``).

A Problem can be stored as a plain ``.npz`` (no pickles) so that inputs made in
a container with h5py travel to a box without it.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

SUBLIMATION_MODES = {"no": 0, "fast": 1, "slow": 2, "cap": 3}
TRACK_ORIGIN = {"no": 0, "basic": 1, "yes": 1, "detailed": 2, "scatterings": 3}
C_CGS = 29979245800.0


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


@dataclass
class Dust:
    """One ``/Dust/dust_NNN`` group."""
    nu: np.ndarray
    albedo: np.ndarray
    chi: np.ndarray
    mu: np.ndarray
    P1: np.ndarray  # (n_nu, n_mu)
    P2: np.ndarray
    P3: np.ndarray
    P4: np.ndarray
    emiss_nu: np.ndarray
    emiss_jnu: np.ndarray  # (n_enu, n_jnu)
    emiss_var: np.ndarray  # (n_jnu,) specific energies
    mo_specific_energy: Optional[np.ndarray] = None
    mo_chi_rosseland: Optional[np.ndarray] = None
    mo_kappa_planck: Optional[np.ndarray] = None      # MRW: dust.f90:88-100
    mo_chi_inv_planck: Optional[np.ndarray] = None
    mo_temperature: Optional[np.ndarray] = None       # not used by the engine: specific energy <-> temperature
    version: int = 2
    is_lte: bool = True
    sublimation_mode: str = "no"
    sublimation_specific_energy: float = 0.0
    minimum_specific_energy: float = 0.0

    def __post_init__(self):
        for k in ("nu", "albedo", "chi", "mu", "P1", "P2", "P3", "P4",
                  "emiss_nu", "emiss_jnu", "emiss_var"):
            setattr(self, k, _f64(getattr(self, k)))
        for k in ("mo_specific_energy", "mo_chi_rosseland", "mo_kappa_planck", "mo_chi_inv_planck", "mo_temperature"):
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, _f64(v))
        n_nu, n_mu = self.nu.size, self.mu.size
        for k in ("P1", "P2", "P3", "P4"):
            if getattr(self, k).shape != (n_nu, n_mu):
                raise ValueError("%s should have shape (n_nu, n_mu)" % k)
        if self.emiss_jnu.shape != (self.emiss_nu.size, self.emiss_var.size):
            raise ValueError("emiss_jnu should have shape (n_enu, n_jnu)")
        if self.sublimation_mode not in SUBLIMATION_MODES:
            raise ValueError("Unknown dust sublimation mode: %s" % self.sublimation_mode)


@dataclass
class Spot:
    """A spot on a spherical source (``Spot N`` sub-group, ``src/sources/source_type.f90:150-188``): centre
    (longitude, latitude) and angular radius in degrees, luminosity, own spectrum or temperature."""
    longitude: float = 0.0
    latitude: float = 0.0
    radius: float = 0.0
    luminosity: float = 0.0
    temperature: Optional[float] = None
    spectrum_nu: Optional[np.ndarray] = None
    spectrum_fnu: Optional[np.ndarray] = None


@dataclass
class Source:
    """One ``/Sources/source_NNNNN`` group: 'point', 'sphere' (position, radius,
    limb_darkening; can re-absorb packets), 'extern_sph' (position +
    radius) or 'extern_box' (box = xmin,xmax,ymin,ymax,zmin,zmax)
    (``src/sources/source_type.f90:102-322``)."""
    type: str = "point"
    luminosity: float = 0.0
    position: tuple = (0.0, 0.0, 0.0)
    radius: float = 0.0
    box: tuple = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    temperature: Optional[float] = None
    spectrum_nu: Optional[np.ndarray] = None
    spectrum_fnu: Optional[np.ndarray] = None
    peeloff: bool = True
    limb_darkening: bool = False
    direction: tuple = (0.0, 0.0)                       # plane_parallel: theta, phi of the beam (deg)
    points: Optional[np.ndarray] = None                 # point_collection: (n, 3) positions
    point_luminosity: Optional[np.ndarray] = None       # point_collection: (n,) luminosities (luminosity = their sum)
    map: Optional[np.ndarray] = None                    # map: luminosity per cell, shape of one density species ('Luminosity map')
    spots: List[Spot] = field(default_factory=list)     # sphere: spots (the reference's source type 3)
    lte: bool = False                                   # map sources only: spectrum = emissivity of the dust in the emitting cell


@dataclass
class PeeledImages:
    """One ``/Output/Peeled/group_NNNNN`` group
    (``src/images/images_peeled.f90:272-380``, ``image_type.f90:153-335``)."""
    theta: np.ndarray
    phi: np.ndarray
    n_wav: int = 1
    wav_min: float = 1.0      # micron
    wav_max: float = 1000.0   # micron
    compute_image: bool = True
    n_x: int = 1
    n_y: int = 1
    x_min: float = -1.0
    x_max: float = 1.0
    y_min: float = -1.0
    y_max: float = 1.0
    compute_sed: bool = True
    n_ap: int = 1
    ap_min: float = 1.0
    ap_max: float = 1.0
    track_origin: str = "no"
    track_n_scat: int = 0
    uncertainties: bool = False
    compute_stokes: bool = True
    inside_observer: bool = False
    ignore_optical_depth: bool = False
    d_min: float = -np.inf
    d_max: float = np.inf
    peeloff_origin: tuple = (0.0, 0.0, 0.0)
    # monochromatic mode: 1-based range of RunConfig.frequencies imaged by this group (image_type.f90:243-258)
    inu_min: int = 0
    inu_max: int = 0
    # filter convolution (image_type.f90:173-181,285-291): list of (nu[], transmission[], nu0), replaces the n_wav bins
    filters: Optional[list] = None
    io_bytes: int = 8          # attr io_bytes: precision of the cubes in the .rtout (image_type.f90:325-333)

    def __post_init__(self):
        self.theta = _f64(np.atleast_1d(self.theta))
        self.phi = _f64(np.atleast_1d(self.phi))

    @property
    def n_view(self):
        return self.theta.size

    @property
    def nu_min(self):
        return C_CGS / (self.wav_max * 1.0e-4)

    @property
    def nu_max(self):
        return C_CGS / (self.wav_min * 1.0e-4)


@dataclass
class RunConfig:
    """Root attributes of the ``.rtin`` with the reference's defaults
    (``hyperion/conf/conf_files.py:48-73``, ``src/main/setup_rt.f90:38-302``)."""
    seed: int = -124902
    n_inter_max: int = 1000000
    n_reabs_max: int = 1000000
    kill_on_absorb: bool = False
    kill_on_scatter: bool = False
    sample_sources_evenly: bool = False
    enforce_energy_range: bool = True
    forced_first_interaction: bool = True
    forced_first_interaction_algorithm: str = "wr99"
    baes16_xi: float = 0.5
    propagation_check_frequency: float = 1.0e-3
    specific_energy_type: str = "initial"
    n_initial_iter: int = 5
    n_initial_photons: int = 0
    n_last_photons: int = 0
    n_ray_photons_sources: int = 0     # raytracing iteration (src/main/setup_rt.f90:169-245)
    n_ray_photons_dust: int = 0
    check_convergence: bool = False
    convergence_absolute: float = 0.0
    convergence_relative: float = 0.0
    convergence_percentile: float = 100.0
    output_specific_energy: str = "last"
    output_density: str = "none"
    output_density_diff: str = "none"                  # src/grid/grid_generic.f90:114-130
    output_n_photons: str = "none"                     # src/grid/grid_generic.f90:40-46
    output_specific_energy_spectrum: str = "none"      # src/main/setup_rt.f90:77-104, src/grid/grid_generic.f90:71-93
    spectrum_bin_edges: Optional[np.ndarray] = None    # table /specific_energy_spectrum_bin_edges, column nu (Hz)
    physics_io_bytes: int = 8                          # root attr: precision of the grid datasets of the .rtout
    copy_input: bool = False                           # root attr: copy the input into /Input instead of linking it (main.f90:133-150)
    mrw: bool = False
    mrw_gamma: float = 1.0             # src/main/setup_rt.f90:106-113
    n_inter_mrw_max: int = 1000
    pda: bool = False
    monochromatic: bool = False        # src/main/setup_rt.f90:49-57,220-222: use_exact_nu with the /frequencies table
    frequencies: Optional[np.ndarray] = None
    monochromatic_energy_threshold: float = 1.0e-10
    n_last_photons_sources: int = 0
    n_last_photons_dust: int = 0
    raytracing: bool = False


@dataclass
class Problem:
    """grid_type 'car': `walls` = [x(n1+1), y(n2+1), z(n3+1)], density
    (n_dust, n3, n2, n1).  grid_type 'sph_pol': walls = [r, theta, phi]; 'cyl_pol': walls =
    [w, z, phi] (src/grid/grid_geometry_spherical_3d.f90:90-203, grid_geometry_cylindrical_3d.f90:90-175).  grid_type 'oct': `refined` (depth-first flags of all
    cells), `oct_center`, `oct_half` (half-widths of the top cell), density
    (n_dust, n_cells) -- src/grid/grid_geometry_octree.f90:184-246."""
    walls: List[np.ndarray]
    density: np.ndarray
    dust: List[Dust]
    sources: List[Source]
    config: RunConfig = field(default_factory=RunConfig)
    peeled: List[PeeledImages] = field(default_factory=list)
    specific_energy: Optional[np.ndarray] = None
    # /Output/Binned/group_00001 (src/images/images_binned.f90): a PeeledImages carrying the image settings (its
    # viewing angles are ignored) and the number of direction bins
    binned: Optional[PeeledImages] = None
    n_binned_theta: int = 0
    n_binned_phi: int = 0
    grid_type: str = "car"
    geometry_id: str = ""
    refined: Optional[np.ndarray] = None
    oct_center: tuple = (0.0, 0.0, 0.0)
    oct_half: tuple = (1.0, 1.0, 1.0)
    # grid_type 'vor' (src/grid/grid_geometry_voronoi.f90:96-188): sites (n,3),
    # cell volumes, CSR neighbour lists (ids 0-based, -1..-6 = domain walls), box
    vor_sites: Optional[np.ndarray] = None
    vor_volume: Optional[np.ndarray] = None
    vor_idx: Optional[np.ndarray] = None
    vor_neighs: Optional[np.ndarray] = None
    vor_box: tuple = (0.0, 1.0, 0.0, 1.0, 0.0, 1.0)
    vor_bb: Optional[np.ndarray] = None     # (n, 6) bb_min, bb_max of the cells (table `cells`): random_position_cell
    # grid_type 'amr' (src/grid/grid_geometry_amr.f90:111-180): the grids of all levels, level by
    # level: amr_level (g,) 1-based, amr_n (g,3) = n1,n2,n3, amr_bounds (g,6) = xmin,xmax,...;
    # density (n_dust, n_cells) with the cells of grid after grid, x fastest
    amr_level: Optional[np.ndarray] = None
    amr_n: Optional[np.ndarray] = None
    amr_bounds: Optional[np.ndarray] = None

    def __post_init__(self):
        self.walls = [_f64(w) for w in self.walls]
        self.density = _f64(self.density)
        if self.grid_type in ("car", "sph_pol", "cyl_pol"):
            if self.density.ndim == 3:
                self.density = self.density[None]
            n1, n2, n3 = self.shape
            want = (len(self.dust), n3, n2, n1)
        elif self.grid_type == "oct":
            self.refined = np.ascontiguousarray(np.asarray(self.refined).astype(np.int32))
            if self.refined.ndim != 1 or (self.refined.size - 1) % 8 != 0:
                raise ValueError("refined should have shape 8 * n + 1")
            if self.density.ndim == 1:
                self.density = self.density[None]
            want = (len(self.dust), self.refined.size)
        elif self.grid_type == "vor":
            self.vor_sites = _f64(self.vor_sites).reshape(-1, 3)
            self.vor_volume = _f64(self.vor_volume)
            self.vor_idx = np.ascontiguousarray(self.vor_idx, dtype=np.int32)
            self.vor_neighs = np.ascontiguousarray(self.vor_neighs, dtype=np.int32)
            n = self.vor_sites.shape[0]
            if self.vor_idx.size != n + 1 or self.vor_volume.size != n or self.vor_idx[-1] != self.vor_neighs.size:
                raise ValueError("inconsistent Voronoi neighbour lists")
            if self.density.ndim == 1:
                self.density = self.density[None]
            want = (len(self.dust), n)
        elif self.grid_type == "amr":
            self.amr_level = np.ascontiguousarray(self.amr_level, dtype=np.int32).reshape(-1)
            self.amr_n = np.ascontiguousarray(self.amr_n, dtype=np.int32).reshape(-1, 3)
            self.amr_bounds = _f64(self.amr_bounds).reshape(-1, 6)
            if not (self.amr_level.size == self.amr_n.shape[0] == self.amr_bounds.shape[0]) or np.any(np.diff(self.amr_level) < 0):
                raise ValueError("amr grids must be listed level by level")
            if self.density.ndim == 1:
                self.density = self.density[None]
            want = (len(self.dust), int(np.prod(self.amr_n, axis=1).sum()))
        else:
            raise ValueError("Unexpected coordinate type: %s" % self.grid_type)
        if self.density.shape != want:
            raise ValueError("density array has wrong shape %r, expected %r" % (self.density.shape, want))
        if self.config.monochromatic:
            if self.config.frequencies is None or len(self.config.frequencies) < 1:
                raise ValueError("monochromatic mode needs RunConfig.frequencies")
            self.config.frequencies = _f64(self.config.frequencies)
            for pl in self.peeled:
                if pl.inu_min == 0 and pl.inu_max == 0:
                    pl.inu_min, pl.inu_max = 1, self.config.frequencies.size
                pl.n_wav = pl.inu_max - pl.inu_min + 1
        if self.specific_energy is not None:
            self.specific_energy = _f64(self.specific_energy)
            if self.specific_energy.shape != self.density.shape:
                raise ValueError("specific_energy array has wrong number of dust types")

    @property
    def shape(self):
        if self.grid_type == "oct":
            return (self.refined.size,)
        if self.grid_type == "vor":
            return (self.vor_sites.shape[0],)
        if self.grid_type == "amr":
            return (int(np.prod(self.amr_n, axis=1).sum()),)
        return tuple(w.size - 1 for w in self.walls)

    @property
    def n_cells(self):
        return int(np.prod(self.shape))

    @property
    def n_dust(self):
        return len(self.dust)

    def octree_cells(self):
        """(centres (n,3), half-widths (n,3), level (n,)) of every octree cell in
        file order (pre-order depth-first, children x-fastest)."""
        n = self.refined.size
        c = np.zeros((n, 3)); h = np.zeros((n, 3)); lev = np.zeros(n, dtype=np.int32)
        c[0] = self.oct_center; h[0] = self.oct_half
        stack = [[0, 0]] if self.refined[0] else []
        filled = 1
        while stack:
            par, k = stack[-1]
            if k == 8:
                stack.pop(); continue
            stack[-1][1] = k + 1
            i = filled; filled += 1
            s = np.array([1 if k & 1 else -1, 1 if k & 2 else -1, 1 if k & 4 else -1])
            c[i] = c[par] + s * h[par] / 2.0
            h[i] = h[par] / 2.0
            lev[i] = lev[par] + 1
            if self.refined[i]:
                stack.append([i, 0])
        if filled != n:
            raise ValueError("refined array is not self-consistent")
        return c, h, lev

    @property
    def volumes(self):
        if self.grid_type == "oct":
            _, h, _ = self.octree_cells()
            return 8.0 * h[:, 0] * h[:, 1] * h[:, 2]
        if self.grid_type == "vor":
            return np.maximum(self.vor_volume, 0.0)
        if self.grid_type == "amr":
            b, n = self.amr_bounds, self.amr_n
            v = ((b[:, 1] - b[:, 0]) / n[:, 0]) * ((b[:, 3] - b[:, 2]) / n[:, 1]) * ((b[:, 5] - b[:, 4]) / n[:, 2])
            return np.repeat(v, np.prod(n, axis=1))
        if self.grid_type == "sph_pol":   # grid_geometry_spherical_3d.f90:146-155: dr3 * dcost * dphi / 3
            r, t, ph = self.walls
            return (np.diff(ph)[:, None, None] * (np.cos(t[:-1]) - np.cos(t[1:]))[None, :, None]
                    * np.diff(r ** 3)[None, None, :] / 3.0)
        if self.grid_type == "cyl_pol":   # grid_geometry_cylindrical_3d.f90:140-147: dw2 * dz * dphi / 2
            w, z, ph = self.walls
            return np.diff(ph)[:, None, None] * np.diff(z)[None, :, None] * np.diff(w ** 2)[None, None, :] / 2.0
        dx, dy, dz = (np.diff(w) for w in self.walls)
        return dz[:, None, None] * dy[None, :, None] * dx[None, None, :]

    # ---------------------------------------------------------------- npz I/O
    def to_npz(self, path, dust_library=None):
        """dust_library: optional {filename: Dust}; a species whose tables equal a
        library entry is stored as a reference to that sibling .npz file."""
        arrays = {}
        meta = {"grid_type": self.grid_type, "geometry_id": self.geometry_id,
                "config": {k: v for k, v in self.config.__dict__.items() if k not in ("frequencies", "spectrum_bin_edges")}, "dust": [], "sources": [], "peeled": [],
                "oct_center": list(self.oct_center), "oct_half": list(self.oct_half), "vor_box": list(self.vor_box)}
        for i, w in enumerate(self.walls):
            arrays["walls_%d" % (i + 1)] = w
        if self.config.frequencies is not None:
            arrays["config/frequencies"] = np.asarray(self.config.frequencies, dtype=float)
        if self.config.spectrum_bin_edges is not None:
            arrays["config/spectrum_bin_edges"] = np.asarray(self.config.spectrum_bin_edges, dtype=float)
        if self.refined is not None:
            arrays["refined"] = self.refined
        for k in ("vor_sites", "vor_volume", "vor_idx", "vor_neighs", "vor_bb", "amr_level", "amr_n", "amr_bounds"):
            if getattr(self, k) is not None:
                arrays[k] = getattr(self, k)
        arrays["density"] = self.density
        if self.specific_energy is not None:
            arrays["specific_energy"] = self.specific_energy
        for i, d in enumerate(self.dust):
            m = {}
            ref = None
            for fname, lib in (dust_library or {}).items():
                if all(np.array_equal(getattr(d, k), getattr(lib, k)) for k in
                       ("nu", "chi", "albedo", "mu", "P1", "P2", "P3", "P4", "emiss_nu", "emiss_jnu", "emiss_var")):
                    ref = fname
            for k, v in d.__dict__.items():
                if isinstance(v, np.ndarray):
                    if ref is None:
                        arrays["dust%d/%s" % (i, k)] = v
                elif v is not None:
                    m[k] = v
            if ref is not None:
                m["__tables_from__"] = ref
            meta["dust"].append(m)
        for i, s in enumerate(self.sources):
            m = {}
            for k, v in s.__dict__.items():
                if k == "spots":
                    if v:
                        m["spots"] = [{kk: (vv.tolist() if isinstance(vv, np.ndarray) else vv) for kk, vv in q.__dict__.items()} for q in v]
                elif isinstance(v, np.ndarray):
                    arrays["source%d/%s" % (i, k)] = v
                elif v is not None:
                    m[k] = list(v) if isinstance(v, tuple) else v
            meta["sources"].append(m)
        for i, p in enumerate(self.peeled):
            m = {}
            for k, v in p.__dict__.items():
                if k == "filters":
                    if v:
                        m["n_filters"] = len(v)
                        for j, f in enumerate(v):
                            arrays["peeled%d/filter%d_nu" % (i, j)] = np.asarray(f[0], dtype=float)
                            arrays["peeled%d/filter%d_tr" % (i, j)] = np.asarray(f[1], dtype=float)
                        m["filter_nu0"] = [float(f[2]) for f in v]
                elif isinstance(v, np.ndarray):
                    arrays["peeled%d/%s" % (i, k)] = v
                else:
                    m[k] = list(v) if isinstance(v, tuple) else v
            meta["peeled"].append(m)
        if self.binned is not None:
            meta["binned"] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.binned.__dict__.items() if not isinstance(v, np.ndarray)}
            meta["n_binned"] = [int(self.n_binned_theta), int(self.n_binned_phi)]
        arrays["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(path, **arrays)

    @classmethod
    def from_npz(cls, path):
        z = np.load(path, allow_pickle=False)
        meta = json.loads(bytes(z["meta_json"]).decode())

        def sub(prefix):
            return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}

        dust = []
        for i, m in enumerate(meta["dust"]):
            kw = dict(m)
            ref = kw.pop("__tables_from__", None)
            if ref is not None:
                import os
                lib = np.load(os.path.join(os.path.dirname(os.path.abspath(path)), ref), allow_pickle=False)
                kw.update({k: lib[k] for k in lib.files if k not in ("version",)})
            kw.update(sub("dust%d/" % i))
            dust.append(Dust(**kw))
        sources = []
        for i, m in enumerate(meta["sources"]):
            kw = dict(m)
            kw.update(sub("source%d/" % i))
            if "position" in kw:
                kw["position"] = tuple(kw["position"])
            if "box" in kw:
                kw["box"] = tuple(kw["box"])
            if "direction" in kw:
                kw["direction"] = tuple(kw["direction"])
            if "spots" in kw:
                kw["spots"] = [Spot(**{kk: (np.asarray(vv, dtype=float) if isinstance(vv, list) else vv) for kk, vv in q.items()}) for q in kw["spots"]]
            sources.append(Source(**kw))
        peeled = []
        for i, m in enumerate(meta["peeled"]):
            kw = dict(m)
            kw.update(sub("peeled%d/" % i))
            nf, nu0 = kw.pop("n_filters", 0), kw.pop("filter_nu0", [])
            if nf:
                kw["filters"] = [(kw.pop("filter%d_nu" % j), kw.pop("filter%d_tr" % j), nu0[j]) for j in range(nf)]
            for k in ("d_min", "d_max"):
                if kw.get(k) is None:
                    kw.pop(k, None)
            if "peeloff_origin" in kw:
                kw["peeloff_origin"] = tuple(kw["peeloff_origin"])
            peeled.append(PeeledImages(**kw))
        binned, nb = None, (0, 0)
        if "binned" in meta:
            kw = dict(meta["binned"])
            for k in ("d_min", "d_max"):
                if kw.get(k) is None:
                    kw.pop(k, None)
            kw["peeloff_origin"] = tuple(kw.get("peeloff_origin", (0.0, 0.0, 0.0)))
            binned, nb = PeeledImages(theta=[0.0], phi=[0.0], **kw), meta["n_binned"]
        cfg = RunConfig(**meta["config"])
        if "config/frequencies" in z.files:
            cfg.frequencies = z["config/frequencies"]
        if "config/spectrum_bin_edges" in z.files:
            cfg.spectrum_bin_edges = z["config/spectrum_bin_edges"]
        walls = [z[k] for k in ("walls_1", "walls_2", "walls_3") if k in z.files]
        return cls(walls=walls, density=z["density"],
                   dust=dust, sources=sources, config=cfg, peeled=peeled,
                   specific_energy=z["specific_energy"] if "specific_energy" in z.files else None,
                   binned=binned, n_binned_theta=int(nb[0]), n_binned_phi=int(nb[1]),
                   grid_type=meta.get("grid_type", "car"), geometry_id=meta.get("geometry_id", ""),
                   refined=z["refined"] if "refined" in z.files else None,
                   oct_center=tuple(meta.get("oct_center", (0.0, 0.0, 0.0))),
                   oct_half=tuple(meta.get("oct_half", (1.0, 1.0, 1.0))),
                   vor_box=tuple(meta.get("vor_box", (0.0, 1.0, 0.0, 1.0, 0.0, 1.0))),
                   **{k: z[k] for k in ("vor_sites", "vor_volume", "vor_idx", "vor_neighs", "vor_bb", "amr_level", "amr_n", "amr_bounds") if k in z.files})
