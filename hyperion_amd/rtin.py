"""Reader for Hyperion ``.rtin`` HDF5 input files -> :class:`Problem`.

This is the file-contract side of the drop-in boundary
(``scripts/hyperion:39-92`` + ``src/main/setup_rt.f90`` of the reference).
It needs ``h5py``; where h5py is absent (the system Python of this image) use a
``.npz`` written by :meth:`Problem.to_npz` instead.
"""
from __future__ import annotations

import numpy as np

from .problem import Dust, PeeledImages, Problem, RunConfig, Source, Spot


def _s(v):
    if isinstance(v, (bytes, np.bytes_)):
        return v.decode()
    return str(v)


def _version_tuple(v):
    out = []
    for part in v.strip().split("."):
        digits = "".join(ch for ch in part if ch.isdigit())
        if digits == "" or not part[0].isdigit():
            break
        out.append(int(digits))
    return tuple(out)


def _b(v):
    s = _s(v).strip().lower()
    if s in ("yes", "y", "true"):
        return True
    if s in ("no", "n", "false"):
        return False
    raise ValueError("cannot interpret %r as a boolean" % (v,))


def read_dust_group(g, minimum_specific_energy=0.0):
    """``src/dust/dust_type_4elem.f90:78-293``"""
    a = g.attrs
    op = g["optical_properties"][...]
    em = g["emissivities"][...]
    mo = g["mean_opacities"][...]
    version = int(a["version"])
    kw = dict(
        nu=op["nu"], albedo=op["albedo"], chi=op["chi"],
        mu=g["scattering_angles"][...]["mu"],
        P1=op["P1"], P2=op["P2"], P3=op["P3"], P4=op["P4"],
        emiss_nu=em["nu"], emiss_jnu=em["jnu"],
        emiss_var=g["emissivity_variable"][...]["specific_energy"],
        mo_specific_energy=mo["specific_energy"],
        mo_chi_rosseland=mo["chi_rosseland"],
        mo_kappa_planck=mo["kappa_planck"] if "kappa_planck" in mo.dtype.names else None,
        # dust_type_4elem.f90:231-237: version-1 files carry the Rosseland mean in its place
        mo_chi_inv_planck=mo["chi_rosseland" if version == 1 else "chi_inv_planck"]
        if ("chi_rosseland" if version == 1 else "chi_inv_planck") in mo.dtype.names else None,
        mo_temperature=mo["temperature"] if "temperature" in mo.dtype.names else None,
        version=version, is_lte=_b(a["lte"]),
        sublimation_mode=_s(a["sublimation_mode"]).strip(),
        minimum_specific_energy=float(minimum_specific_energy),
    )
    if _s(a["emissvar"]).strip() != "E":
        raise ValueError("Only emissvar='E' supported at this time")
    if kw["sublimation_mode"] != "no":
        kw["sublimation_specific_energy"] = float(a["sublimation_specific_energy"])
    return Dust(**kw)


def read_rtin(path):
    import h5py  # deliberately local: the engine itself never needs HDF5

    with h5py.File(path, "r") as f:
        a = f.attrs
        cfg = RunConfig()
        # src/main/setup_rt.f90:38-45
        if "python_version" not in a or _version_tuple(_s(a["python_version"])) < (0, 8, 7):
            raise ValueError("cannot read files made with the Python module before version 0.8.7")
        cfg.seed = int(a["seed"]) if "seed" in a else -124902
        cfg.n_inter_max = int(a["n_inter_max"])
        cfg.n_reabs_max = int(a["n_reabs_max"])
        cfg.kill_on_absorb = _b(a["kill_on_absorb"])
        cfg.kill_on_scatter = _b(a["kill_on_scatter"]) if "kill_on_scatter" in a else False
        cfg.sample_sources_evenly = _b(a["sample_sources_evenly"]) if "sample_sources_evenly" in a else False
        cfg.enforce_energy_range = _b(a["enforce_energy_range"]) if "enforce_energy_range" in a else True
        if "forced_first_scattering" in a:
            cfg.forced_first_interaction = _b(a["forced_first_scattering"])
        else:
            cfg.forced_first_interaction = _b(a["forced_first_interaction"])
            cfg.forced_first_interaction_algorithm = _s(a["forced_first_interaction_algorithm"]).strip()
            if "forced_first_interaction_baes16_xi" in a:
                cfg.baes16_xi = float(a["forced_first_interaction_baes16_xi"])
        cfg.propagation_check_frequency = float(a["propagation_check_frequency"]) \
            if "propagation_check_frequency" in a else 1.0e-3
        cfg.specific_energy_type = _s(a["specific_energy_type"]).strip() if "specific_energy_type" in a else "initial"
        cfg.n_initial_iter = int(a["n_initial_iter"])
        cfg.n_initial_photons = int(a["n_initial_photons"]) if cfg.n_initial_iter > 0 else 0
        cfg.mrw = _b(a["mrw"])
        if cfg.mrw:
            cfg.mrw_gamma = float(a["mrw_gamma"])
            cfg.n_inter_mrw_max = int(a["n_inter_mrw_max"])
        cfg.pda = _b(a["pda"])
        cfg.monochromatic = _b(a["monochromatic"])
        cfg.raytracing = _b(a["raytracing"])
        if cfg.monochromatic:       # src/main/setup_rt.f90:49-57,220-222; hyperion/model/model.py:133-137
            cfg.frequencies = np.asarray(f["frequencies"][...]["nu"], dtype=float)
            cfg.monochromatic_energy_threshold = float(a["monochromatic_energy_threshold"]) \
                if "monochromatic_energy_threshold" in a else 1.0e-10
            cfg.n_last_photons_sources = int(a["n_last_photons_sources"]) if "n_last_photons_sources" in a else 0
            cfg.n_last_photons_dust = int(a["n_last_photons_dust"]) if "n_last_photons_dust" in a else 0
        cfg.n_last_photons = int(a["n_last_photons"]) if "n_last_photons" in a else 0
        if cfg.raytracing:
            cfg.n_ray_photons_sources = int(a["n_ray_photons_sources"]) if "n_ray_photons_sources" in a else 0
            cfg.n_ray_photons_dust = int(a["n_ray_photons_dust"]) if "n_ray_photons_dust" in a else 0
        if cfg.n_initial_iter > 0:
            cfg.check_convergence = _b(a["check_convergence"])
            if cfg.check_convergence:
                cfg.convergence_absolute = float(a["convergence_absolute"])
                cfg.convergence_relative = float(a["convergence_relative"])
                cfg.convergence_percentile = float(a["convergence_percentile"])
        # src/main/setup_rt.f90:77-104,247-283: every /Output switch is one of all / last / none
        out = f["Output"].attrs
        for key in ("output_density", "output_density_diff", "output_specific_energy", "output_n_photons"):
            v = _s(out[key]).strip()
            if v not in ("all", "last", "none"):
                raise ValueError("%s should be one of all/last/none" % key)
            setattr(cfg, key, v)
        if "output_specific_energy_spectrum" in out:
            cfg.output_specific_energy_spectrum = _s(out["output_specific_energy_spectrum"]).strip()
            if cfg.output_specific_energy_spectrum not in ("all", "last", "none"):
                raise ValueError("output_specific_energy_spectrum should be one of all/last/none")
        if cfg.output_specific_energy_spectrum != "none":
            if "specific_energy_spectrum_bin_edges" not in f:
                raise ValueError("specific_energy_spectrum_bin_edges should be present in the input when "
                                 "output_specific_energy_spectrum is enabled")
            cfg.spectrum_bin_edges = np.asarray(f["specific_energy_spectrum_bin_edges"][...]["nu"], dtype=float)
        # src/main/setup_rt.f90:207-215, src/main/main.f90:133-150
        cfg.physics_io_bytes = int(a["physics_io_bytes"]) if "physics_io_bytes" in a else 8
        if cfg.physics_io_bytes not in (4, 8):
            raise ValueError("unexpected value of physics_io_bytes (should be 4 or 8)")
        cfg.copy_input = _b(a["copy_input"]) if "copy_input" in a else False

        geo = f["Grid/Geometry"]
        grid_type = _s(geo.attrs["grid_type"]).strip()
        extra = {}
        if grid_type == "car":
            walls = [geo["walls_1"][...]["x"], geo["walls_2"][...]["y"], geo["walls_3"][...]["z"]]
        elif grid_type == "sph_pol":
            walls = [geo["walls_1"][...]["r"], geo["walls_2"][...]["t"], geo["walls_3"][...]["p"]]
        elif grid_type == "cyl_pol":
            walls = [geo["walls_1"][...]["w"], geo["walls_2"][...]["z"], geo["walls_3"][...]["p"]]
        elif grid_type == "oct":
            walls = []
            ga = geo.attrs
            extra = dict(refined=np.asarray(geo["cells"][...]["refined"]).astype(np.int32),
                         oct_center=(float(ga["x"]), float(ga["y"]), float(ga["z"])),
                         oct_half=(float(ga["dx"]), float(ga["dy"]), float(ga["dz"])))
        elif grid_type == "vor":
            walls = []
            ga = geo.attrs
            cells = geo["cells"][...]
            extra = dict(vor_sites=np.asarray(cells["coordinates"], dtype=float),
                         vor_volume=np.asarray(cells["volume"], dtype=float),
                         vor_bb=np.concatenate([np.asarray(cells["bb_min"], dtype=float), np.asarray(cells["bb_max"], dtype=float)], axis=1)
                         if "bb_min" in cells.dtype.names else None,
                         vor_idx=np.asarray(geo["sparse_idx"][...]).astype(np.int32),
                         vor_neighs=np.asarray(geo["sparse_neighs"][...]).astype(np.int32),
                         vor_box=tuple(float(ga[k]) for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")))
        elif grid_type == "amr":
            walls = []
            lev, nn, bounds, paths = [], [], [], []
            for il in range(1, int(geo.attrs["nlevels"]) + 1):
                gl = geo["level_%05d" % il]
                for ig in range(1, int(gl.attrs["ngrids"]) + 1):
                    ga = gl["grid_%05d" % ig].attrs
                    lev.append(il)
                    nn.append([int(ga["n1"]), int(ga["n2"]), int(ga["n3"])])
                    bounds.append([float(ga[k]) for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")])
                    paths.append("level_%05d/grid_%05d" % (il, ig))
            extra = dict(amr_level=np.array(lev, dtype=np.int32), amr_n=np.array(nn, dtype=np.int32), amr_bounds=np.array(bounds))
        else:
            raise NotImplementedError("grid type %r is not supported yet" % grid_type)
        q = f["Grid/Quantities"]
        if grid_type == "amr":
            # read_grid_4d for AMR (src/grid/grid_io_amr_template.f90): one (n_dust, n3, n2, n1) array per grid
            def gather(name):
                parts = [np.asarray(q[p][name][...], dtype=float) for p in paths]
                return np.concatenate([a.reshape(a.shape[0], -1) for a in parts], axis=1)
            density = gather("density")
            spec = gather("specific_energy") if "specific_energy" in q[paths[0]] else None
        else:
            density = q["density"][...]
            spec = q["specific_energy"][...] if "specific_energy" in q else None
            # read_grid_4d (src/grid/grid_io.f90:78-81): the dataset must belong to this geometry
            for name in ("density", "specific_energy"):
                if name in q and "geometry" in q[name].attrs and "geometry" in geo.attrs \
                        and _s(q[name].attrs["geometry"]).strip() != _s(geo.attrs["geometry"]).strip():
                    raise ValueError("geometry IDs do not match for %s" % name)
        n_dust = density.shape[0]
        mse = np.zeros(n_dust)
        if "minimum_specific_energy" in q.attrs:
            mse = np.atleast_1d(q.attrs["minimum_specific_energy"]).astype(float)

        dust = []
        names = sorted(f["Dust"].keys())
        if len(names) != n_dust:
            raise ValueError("density array has wrong number of dust types")
        for i, n in enumerate(names):
            dust.append(read_dust_group(f["Dust"][n], mse[i]))

        sources = []
        for n in sorted(f["Sources"].keys()):
            g = f["Sources"][n]
            sa = g.attrs
            t = _s(sa["type"]).strip()
            if t == "point_collection":
                lum = np.asarray(g["luminosity"][...], dtype=float)
                s = Source(type=t, luminosity=float(lum.sum()), peeloff=_b(sa["peeloff"]), point_luminosity=lum,
                           points=np.asarray(g["position"][...], dtype=float).reshape(-1, 3))
            else:
                s = Source(type=t, luminosity=float(sa["luminosity"]), peeloff=_b(sa["peeloff"]))
            if t == "point":
                s.position = (float(sa["x"]), float(sa["y"]), float(sa["z"]))
            elif t == "sphere":
                # spots are sub-groups of the source group (source_type.f90:150-188)
                for k in sorted(g.keys()):
                    if not isinstance(g[k], type(g)):
                        continue
                    qa = g[k].attrs
                    q = Spot(longitude=float(qa["longitude"]), latitude=float(qa["latitude"]), radius=float(qa["radius"]),
                             luminosity=float(qa["luminosity"]))
                    qs = _s(qa["spectrum"]).strip()
                    if qs == "temperature":
                        q.temperature = float(qa["temperature"])
                    elif qs == "spectrum":
                        tab = g[k]["spectrum"][...]
                        q.spectrum_nu, q.spectrum_fnu = np.asarray(tab["nu"], dtype=float), np.asarray(tab["fnu"], dtype=float)
                    else:
                        raise ValueError("Spot cannot have LTE spectrum")
                    s.spots.append(q)
                s.position = (float(sa["x"]), float(sa["y"]), float(sa["z"]))
                s.radius = float(sa["r"])
                s.limb_darkening = _b(sa["limb"])
            elif t == "extern_sph":
                s.position = (float(sa["x"]), float(sa["y"]), float(sa["z"]))
                s.radius = float(sa["r"])
            elif t == "plane_parallel":
                s.position = (float(sa["x"]), float(sa["y"]), float(sa["z"]))
                s.radius = float(sa["r"])
                s.direction = (float(sa["theta"]), float(sa["phi"]))
            elif t == "point_collection":
                pass
            elif t == "map":       # source_type.f90:190-199; one dataset per AMR grid otherwise (grid_io_amr.f90)
                if grid_type == "amr":
                    s.map = np.concatenate([np.asarray(g[p_]["Luminosity map"][...], dtype=float).ravel() for p_ in paths])
                else:
                    s.map = np.asarray(g["Luminosity map"][...], dtype=float)
            elif t == "extern_box":
                s.box = tuple(float(sa[k]) for k in ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax"))
            else:
                raise NotImplementedError("source type %r is not supported yet" % t)
            st = _s(sa["spectrum"]).strip()
            if st == "temperature":
                s.temperature = float(sa["temperature"])
            elif st == "spectrum":
                tab = g["spectrum"][...]
                s.spectrum_nu = np.asarray(tab["nu"], dtype=float)
                s.spectrum_fnu = np.asarray(tab["fnu"], dtype=float)
            elif st == "lte" and t == "map":
                s.lte = True
            else:
                raise ValueError("Point source cannot have LTE spectrum")
            sources.append(s)

        def image_group(g, binned=False):
            """image_setup (src/images/image_type.f90:153-335); a peeled group also carries its viewing angles and
            observer settings (images_peeled.f90:306-345), a binned one n_theta / n_phi (images_binned.f90:42-56)."""
            pa = g.attrs
            if binned:
                p = PeeledImages(theta=[0.0], phi=[0.0])
            else:
                ang = g["angles"][...]
                p = PeeledImages(theta=ang["theta"], phi=ang["phi"])
                p.inside_observer = _b(pa["inside_observer"])
                p.ignore_optical_depth = _b(pa["ignore_optical_depth"])
                p.d_min, p.d_max = float(pa["d_min"]), float(pa["d_max"])
                if p.inside_observer:
                    p.peeloff_origin = tuple(float(pa["observer_" + k]) for k in "xyz")
                else:
                    p.peeloff_origin = tuple(float(pa["peeloff_" + k]) for k in "xyz")
            use_filters = _b(pa["use_filters"]) if "use_filters" in pa else False
            if use_filters:             # image_type.f90:173-181,285-291
                if cfg.monochromatic:
                    raise ValueError("cannot use filters in monochromatic mode")
                if cfg.raytracing and not binned:      # images_peeled.f90:349-351
                    raise ValueError("filter convolution cannot be used with raytracing")
                p.filters = []
                for i in range(1, int(pa["n_filt"]) + 1):
                    fg = g["filter_%05d" % i]
                    tab = fg[...]
                    p.filters.append((np.asarray(tab["nu"], dtype=float), np.asarray(tab["tn"], dtype=float), float(fg.attrs["nu0"])))
                p.n_wav = len(p.filters)
            else:
                p.n_wav = int(pa["n_wav"])
            if p.n_wav < 1:
                raise ValueError("n_nu should be >= 1")
            p.io_bytes = int(pa["io_bytes"]) if "io_bytes" in pa else 8
            if p.io_bytes not in (4, 8):
                raise ValueError("unexpected value of io_bytes (should be 4 or 8)")
            if use_filters:
                pass
            elif cfg.monochromatic:       # image_type.f90:243-258
                p.inu_min, p.inu_max = int(pa["inu_min"]), int(pa["inu_max"])
            else:
                p.wav_min, p.wav_max = float(pa["wav_min"]), float(pa["wav_max"])
            p.compute_image = _b(pa["compute_image"])
            if p.compute_image:
                p.n_x, p.n_y = int(pa["n_x"]), int(pa["n_y"])
                p.x_min, p.x_max = float(pa["x_min"]), float(pa["x_max"])
                p.y_min, p.y_max = float(pa["y_min"]), float(pa["y_max"])
            p.compute_sed = _b(pa["compute_sed"])
            if p.compute_sed:
                p.n_ap = int(pa["n_ap"])
                p.ap_min, p.ap_max = float(pa["ap_min"]), float(pa["ap_max"])
            p.track_origin = _s(pa["track_origin"]).strip()
            p.track_n_scat = int(pa["track_n_scat"]) if "track_n_scat" in pa else 0
            p.uncertainties = _b(pa["uncertainties"])
            p.compute_stokes = _b(pa["compute_stokes"]) if "compute_stokes" in pa else True
            return p

        peeled = []
        if "Peeled" in f["Output"]:
            for n in sorted(f["Output/Peeled"].keys()):
                peeled.append(image_group(f["Output/Peeled"][n]))
        binned, nb = None, (0, 0)
        if "Binned" in f["Output"] and len(f["Output/Binned"].keys()) > 0:
            names = sorted(f["Output/Binned"].keys())
            if len(names) > 1:
                raise ValueError("can't have more than one binned image group")      # setup_rt.f90:324
            g = f["Output/Binned"][names[0]]
            binned, nb = image_group(g, binned=True), (int(g.attrs["n_theta"]), int(g.attrs["n_phi"]))

        return Problem(walls=walls, density=density, dust=dust, sources=sources, config=cfg,
                       peeled=peeled, specific_energy=spec, grid_type=grid_type,
                       binned=binned, n_binned_theta=nb[0], n_binned_phi=nb[1],
                       geometry_id=_s(geo.attrs["geometry"]), **extra)
