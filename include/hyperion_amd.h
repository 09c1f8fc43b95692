/*
 * hyperion_amd.h -- C-ABI of the MI355X photon-packet engine.
 *
 * The reference (hyperion-rt/hyperion) has no in-process FFI for this path:
 * its Python front-end writes an HDF5 .rtin and spawns the Fortran binary
 * (hyperion/model/model.py:1025-1080 -> scripts/hyperion:39-92 ->
 * src/main/main.f90).  This header is the array-based seam a maintainer would
 * bind instead of that subprocess (see INTEGRATION.md for the ctypes stub):
 * each entry point cites the reference routine it replaces.  Plain pointers
 * and sizes only; the caller owns every host buffer; descriptors are borrowed
 * for the duration of hyp_create() and copied to the device.
 *
 * Error convention (mirrors "non-zero exit + message in the log",
 * scripts/hyperion:98-104, hyperion/model/tests/test_fortran.py): every
 * function returns 0 on success and non-zero on failure; hyp_last_error()
 * then carries the reference's message text (e.g. "photon was not emitted
 * inside a cell", src/sources/source.f90:177).  No exceptions cross the ABI.
 * A handle is not thread-safe (the reference keeps the same state in
 * process-global Fortran module variables).
 */
#ifndef HYPERION_AMD_H
#define HYPERION_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HYP_MAX_DUST 8
#define HYP_ABI_VERSION 2

/* /Dust/dust_NNN of the .rtin -- src/dust/dust_type_4elem.f90:78-293 */
typedef struct hyp_dust_desc {
    int32_t n_nu;               /* optical_properties rows */
    int32_t n_mu;               /* scattering_angles rows */
    int32_t n_jnu;              /* emissivity_variable rows */
    int32_t n_enu;              /* emissivities rows */
    int32_t n_e;                /* mean_opacities rows (0 if absent) */
    int32_t sublimation_mode;   /* 0 no, 1 fast, 2 slow, 3 cap */
    int32_t version;
    int32_t is_lte;
    double  sublimation_specific_energy;
    double  minimum_specific_energy;  /* Grid/Quantities attr for this species */
    const double *nu;           /* [n_nu] Hz, increasing */
    const double *albedo;       /* [n_nu] */
    const double *chi;          /* [n_nu] cm^2/g */
    const double *mu;           /* [n_mu] */
    const double *P1;           /* [n_nu][n_mu] */
    const double *P2;
    const double *P3;
    const double *P4;
    const double *emiss_nu;     /* [n_enu] */
    const double *emiss_jnu;    /* [n_enu][n_jnu] */
    const double *emiss_var;    /* [n_jnu] specific energy */
    const double *mo_specific_energy; /* [n_e] or NULL */
    const double *mo_chi_rosseland;   /* [n_e] or NULL */
    const double *mo_kappa_planck;    /* [n_e] or NULL; needed with config.mrw (src/dust/dust.f90:88-93) */
    const double *mo_chi_inv_planck;  /* [n_e] or NULL; column chi_inv_planck (chi_rosseland in version-1 files, dust_type_4elem.f90:231-237) */
} hyp_dust_desc;

/* One spot of a spherical source: sub-group `Spot N` of the source group (src/sources/source_type.f90:150-188;
 * attrs longitude, latitude, radius [deg], luminosity, and its own spectrum / temperature). */
typedef struct hyp_spot_desc {
    double  longitude, latitude, radius;   /* degrees: angle3d_deg(lon, lat), cos(radius) */
    double  luminosity;
    double  temperature;
    int32_t spectrum_type;  /* 1 tabulated spectrum, 2 blackbody temperature */
    int32_t n_spec;
    const double *spec_nu;  /* [n_spec] */
    const double *spec_fnu; /* [n_spec] */
} hyp_spot_desc;

/* /Sources/source_NNNNN -- src/sources/source_type.f90:102-322 */
typedef struct hyp_source_desc {
    int32_t type;          /* 1 point, 2 sphere (position, radius, limb_darkening), 5 extern_sph (position, radius), 6 extern_box (box),
                              7 plane_parallel (position, radius, direction), 8 point_collection (points, point_lum) */
    int32_t spectrum_type; /* 1 tabulated spectrum, 2 blackbody temperature, 3 lte (map sources only) */
    int32_t peeloff;
    int32_t n_spec;
    int32_t limb_darkening; /* sphere: attr `limb` (source_type.f90:142) */
    int32_t n_points;       /* point_collection: number of points */
    double  luminosity;
    double  temperature;
    double  position[3];
    double  radius;
    double  box[6];
    const double *spec_nu;
    const double *spec_fnu;
    double  direction[2];     /* plane_parallel: attrs theta, phi (deg) of the beam (source_type.f90:239-256) */
    const double *points;     /* point_collection: [n_points][3] dataset `position` (source_type.f90:258-277) */
    const double *point_lum;  /* point_collection: [n_points] dataset `luminosity` */
    const double *map;        /* map (type 4): [n_cells] dataset `Luminosity map`, cell order of the density (source_type.f90:190-199,
                                 grid_load_pdf_map src/grid/grid_geometry_common_3d.f90:47-63); spectrum_type 3 = 'lte': the dust emissivity
                                 of the emitting cell (select_dust_specific_energy_rho + dust_sample_j_nu, source_type.f90:486-491) */
    int32_t n_spots;          /* sphere: number of spots (the source then is the reference's type 3, a spotted sphere) */
    int32_t reserved_spots;
    const hyp_spot_desc *spots; /* [n_spots] */
} hyp_source_desc;

/* /Grid/Geometry -- src/grid/grid_geometry_cartesian_3d.f90:77-134 (type 1),
 * src/grid/grid_geometry_octree.f90:184-246 (type 2),
 * src/grid/grid_geometry_voronoi.f90:96-188 (type 3),
 * src/grid/grid_geometry_amr.f90:111-508 (type 4),
 * src/grid/grid_geometry_spherical_3d.f90:90-203 (type 5), src/grid/grid_geometry_cylindrical_3d.f90:90-175 (type 6) */
typedef struct hyp_grid_desc {
    int32_t type;          /* 1 cartesian, 2 octree, 3 voronoi, 4 amr, 5 spherical polar (r, theta, phi), 6 cylindrical polar (w, z, phi) */
    int32_t n1, n2, n3;    /* cartesian / spherical / cylindrical: cells per axis */
    const double *w1;      /* cartesian: [n1+1] walls */
    const double *w2;
    const double *w3;
    int64_t n_cells;       /* octree: number of cells, refined ones included */
    const int32_t *refined;/* octree: [n_cells] depth-first refinement flags (table `cells`) */
    double oct_center[3];  /* octree: attrs x, y, z of the top cell */
    double oct_half[3];    /* octree: attrs dx, dy, dz (half-widths) */
    /* voronoi (type 3), src/grid/grid_geometry_voronoi.f90:96-188; n_cells sites */
    const double *vor_sites;    /* [n_cells][3] table cells.coordinates */
    const double *vor_volume;   /* [n_cells] table cells.volume (<= 0: masked) */
    const int32_t *vor_idx;     /* [n_cells+1] sparse_idx (CSR offsets) */
    const int32_t *vor_neighs;  /* sparse_neighs: 0-based ids; -1..-6 = xmin,xmax,ymin,ymax,zmin,zmax walls */
    double vor_box[6];          /* attrs xmin,xmax,ymin,ymax,zmin,zmax */
    /* amr (type 4): level_NNNNN/grid_NNNNN groups flattened level by level; n_cells = sum n1*n2*n3,
     * cells numbered grid after grid with x fastest (src/core/type_cell_id_amr.f90:57-93) */
    int32_t n_amr_levels, n_amr_grids;
    const int32_t *amr_level;   /* [n_amr_grids] 1-based level of each grid, non-decreasing */
    const int32_t *amr_n;       /* [n_amr_grids][3] attrs n1,n2,n3 */
    const double *amr_bounds;   /* [n_amr_grids][6] attrs xmin,xmax,ymin,ymax,zmin,zmax */
    /* voronoi: bounding boxes of the cells, [n_cells][6] = bb_min[3], bb_max[3] of table `cells`, or NULL.  Needed by
     * random_position_cell (src/grid/grid_geometry_voronoi.f90:285-310: rejection sampling in the box) -- map sources,
     * raytraced / monochromatic dust emission */
    const double *vor_bb;
} hyp_grid_desc;

/* root attributes -- src/main/setup_rt.f90:38-302 */
typedef struct hyp_config {
    int64_t seed;
    int64_t n_inter_max;
    int64_t n_reabs_max;
    int32_t kill_on_absorb;
    int32_t kill_on_scatter;
    int32_t sample_sources_evenly;
    int32_t enforce_energy_range;
    int32_t forced_first_interaction;
    int32_t forced_first_interaction_algorithm; /* 1 wr99, 2 baes16 */
    int32_t specific_energy_type;               /* 0 initial, 1 additional */
    int32_t raytracing;                         /* root attr `raytracing`: the final iteration peels only scattered packets
                                                   (do_final(..., peeloff_scattering_only), src/main/main.f90:274) and the
                                                   image groups cache their binned spectra for hyp_raytracing_* */
    double  baes16_xi;
    double  propagation_check_frequency;
    /* modified random walk (src/grid/grid_mrw_3d.f90, src/main/setup_rt.f90:106-113) */
    int64_t n_inter_mrw_max;
    double  mrw_gamma;
    int32_t mrw;
    int32_t monochromatic;          /* root attr `monochromatic` (use_exact_nu): the final iteration is do_final_mono
                                       (src/main/iter_final_mono.f90) at the frequencies below, src/main/setup_rt.f90:49-57 */
    double  monochromatic_energy_threshold;   /* root attr, default 1e-10 */
    const double *frequencies;      /* [n_frequencies] table /frequencies, column nu (setup_rt.f90:220-222) */
    int32_t n_frequencies;
    int32_t reserved2;
    /* partial diffusion approximation (root attr `pda`): solve_pda after update_energy_abs, src/grid/grid_pda_3d.f90:84-172;
     * Cartesian, spherical and cylindrical grids (the others are built with grid_pda_disabled.f90: nothing to do) */
    int32_t pda;
    /* keep n_photons(cell), the number of packets that entered each cell in a Lucy iteration
     * (src/grid/grid_propagate_3d.f90:88-93,171-176): allocated with pda or /Output output_n_photons != 'none'
     * (src/grid/grid_physics_3d.f90:307-318) */
    int32_t count_photons;
    /* frequency-resolved specific energy (/Output output_specific_energy_spectrum != 'none', src/main/setup_rt.f90:77-104):
     * n_spectrum_bins bins with edges spectrum_bin_edges[n_spectrum_bins + 1] (Hz, strictly increasing; table
     * /specific_energy_spectrum_bin_edges column nu); 0 = off */
    int32_t n_spectrum_bins;
    int32_t reserved3;
    const double *spectrum_bin_edges;
} hyp_config;

/* /Output/Peeled/group_NNNNN -- src/images/images_peeled.f90:272-380,
 * src/images/image_type.f90:153-335 */
typedef struct hyp_peeled_desc {
    int32_t n_view;
    int32_t inside_observer;
    int32_t ignore_optical_depth;
    int32_t compute_image;
    int32_t compute_sed;
    int32_t n_x, n_y;
    int32_t n_ap;
    int32_t n_nu;
    int32_t track_origin;    /* 0 no, 1 basic, 2 detailed, 3 scatterings */
    int32_t track_n_scat;
    int32_t uncertainties;
    int32_t compute_stokes;
    int32_t reserved0;
    double  x_min, x_max, y_min, y_max;
    double  ap_min, ap_max;
    double  nu_min, nu_max;
    double  d_min, d_max;
    double  peeloff_origin[3];
    const double *theta;     /* [n_view] deg */
    const double *phi;       /* [n_view] deg */
    int32_t inu_min, inu_max; /* monochromatic: attrs inu_min, inu_max (1-based range of config.frequencies, image_type.f90:243-258);
                                 n_nu = inu_max - inu_min + 1 */
    /* filter convolution (attrs use_filters, n_filt and groups filter_NNNNN with tables nu, tn and attr nu0,
     * src/images/image_type.f90:173-181,285-291): n_nu = n_filt planes, a packet is binned into every filter whose
     * transmission at its frequency is positive with that weight (image_bin :467-475); not with raytracing or
     * monochromatic mode */
    int32_t use_filters;
    int32_t reserved_f;
    const int32_t *filt_n;   /* [n_nu] points of each filter curve */
    const double *filt_nu;   /* concatenated, increasing within a filter */
    const double *filt_tr;   /* concatenated transmissions (column tn) */
} hyp_peeled_desc;

typedef struct hyp_problem {
    hyp_grid_desc grid;
    hyp_config    config;
    int32_t n_dust;
    int32_t n_sources;
    int32_t n_peeled;
    int32_t reserved0;
    const hyp_dust_desc   *dust;
    const hyp_source_desc *sources;
    const hyp_peeled_desc *peeled;
    const double *density;           /* [n_dust][n3][n2][n1] (cartesian) or [n_dust][n_cells] (octree, voronoi, amr), as in the .rtin */
    const double *specific_energy;   /* same shape, or NULL */
    /* /Output/Binned/group_00001 (src/images/images_binned.f90:42-56): packets leaving the grid in the final iteration
     * are binned by direction into n_binned_theta x n_binned_phi views (cos(theta) in [-1,1], phi in [0,2pi));
     * `binned` describes the image like a peeled group (its n_view, theta, phi, inu_* are ignored); NULL = none.
     * The cubes are returned as group index n_peeled. */
    const hyp_peeled_desc *binned;
    int32_t n_binned_theta, n_binned_phi;
} hyp_problem;

typedef struct hyp_iter_stats {
    double   energy_current;               /* src/sources/source.f90:163 */
    double   energy_abs_tot[HYP_MAX_DUST]; /* grid_physics_3d.f90:605-611 */
    uint64_t killed_geo;                   /* src/main/counters.f90:8-10 */
    uint64_t killed_int;
    uint64_t crossings;                    /* cell steps taken (roofline unit) */
    uint64_t interactions;
    uint64_t n_packets;
} hyp_iter_stats;

typedef struct hyp_engine *hyp_handle;

/* setup_initial (src/main/setup_rt.f90:27): build tables, copy to the GPU
 * selected by `device` (HIP ordinal).  On failure *out is NULL and the message
 * is available from hyp_last_error(NULL). */
int  hyp_create(const hyp_problem *problem, int device, hyp_handle *out);
void hyp_destroy(hyp_handle h);
const char *hyp_last_error(hyp_handle h);
int  hyp_abi_version(void);
/* Digest of everything a hyp_problem points to, in a canonical order (FNV-1a 64 over scalars and arrays with their
 * lengths; needs no GPU): out[0] grid + density + specific energy, out[1] dust, out[2] sources, out[3] run configuration +
 * image groups.  Two marshallers that read the same .rtin (hyperion_amd/_abi.py from the Python reader, hyp_run.cpp's own
 * HDF5 reader) must produce the same four words: tests/test_native_driver_cpu.py compares them, which is how the marshalling
 * that feeds both the engine and the test oracle is checked independently of itself.  Returns non-zero on a malformed
 * problem (negative sizes). */
int  hyp_problem_digest(const hyp_problem *problem, uint64_t out[4]);

/* do_lucy (src/main/iter_lucy.f90:66-237): one whole temperature iteration on
 * one GPU.  `iteration` is the 1-based Lucy iteration (part of the RNG key).
 * specific_energy_out: [n_dust][n_cells] host buffer or NULL. */
int  hyp_lucy_iteration(hyp_handle h, uint64_t n_packets, int iteration,
                        double *specific_energy_out, hyp_iter_stats *stats);

/* The same iteration split for multi-GPU sharding (replaces the MPI chunk
 * dispatcher + MPI_Reduce/Bcast of src/mpi/mpi_routines.f90:62-323):
 *   launch      : zero the accumulators and propagate packet ids
 *                 [first_id, first_id + n_local) asynchronously;
 *   accumulators: device pointer + length (in doubles) of the contiguous
 *                 accumulator block [specific_energy_sum | energy_current |
 *                 killed_geo | killed_int | crossings | interactions]; waits
 *                 for the propagation to finish.  The caller all-reduces this
 *                 block in place across ranks (one RCCL call);
 *   finish      : update_energy_abs + sublimate_dust on the (reduced) block. */
int  hyp_lucy_launch(hyp_handle h, uint64_t first_id, uint64_t n_local, int iteration);
int  hyp_lucy_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles);
int  hyp_lucy_finish(hyp_handle h, double *specific_energy_out, hyp_iter_stats *stats);

/* do_final + peeloff_photon (src/main/iter_final.f90:60-273,
 * src/images/images_peeled.f90:95-270).  Same split as above; the accumulator
 * block is [all sed/img cubes | energy_current | killed counters ...].  With the deferred peel-off schedule (option
 * "defer_peel", the default where it applies) hyp_final_launch runs its rounds of {propagate, peel} to the end before it
 * returns; the split stays valid, only the overlap with host work is gone. */
int  hyp_final_iteration(hyp_handle h, uint64_t n_packets, hyp_iter_stats *stats);
int  hyp_final_launch(hyp_handle h, uint64_t first_id, uint64_t n_local);
int  hyp_final_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles);
int  hyp_final_finish(hyp_handle h, hyp_iter_stats *stats);
/* image cubes after hyp_final_finish, .rtout layout, scaled by
 * energy_total/energy_current (image_type.f90:136-151) but not yet
 * normalised by d(nu) (done at write time, image_type.f90:652-688).
 * which: 0 sed, 1 sed^2, 2 img, 3 img^2.  out may be NULL to query n. */
int  hyp_peeled_get(hyp_handle h, int group, int which, double *out, uint64_t *n_doubles);

/* do_raytracing (src/main/iter_raytracing.f90:30-143), only with config.raytracing: adds the direct
 * light of the sources (n_sources packets) and the thermal emission of the dust (n_dust packets,
 * emit_from_grid src/grid/grid_physics_3d.f90:691-753) to the cubes of the last final iteration,
 * every packet carrying the whole binned spectrum of its emitter (polychromatic peeloff_photon,
 * src/images/images_peeled.f90:218-254).  Call after hyp_final_finish (or alone if n_last_photons = 0).
 * Split form for sharding: launch(which = 0 sources / 1 dust) runs ids [first_id, first_id + n_local)
 * of n_total; zero_first clears the cubes first (ranks other than 0, so that one all-reduce of the
 * block from hyp_raytracing_accumulators gives final + raytraced flux); finish reports the counters. */
int  hyp_raytracing_iteration(hyp_handle h, uint64_t n_sources, uint64_t n_dust, hyp_iter_stats *stats);
int  hyp_raytracing_launch(hyp_handle h, int which, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first);
int  hyp_raytracing_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles);
int  hyp_raytracing_finish(hyp_handle h, hyp_iter_stats *stats);
int  hyp_peeled_n_orig(hyp_handle h, int group);

/* do_final_mono (src/main/iter_final_mono.f90:58-343 with src/grid/grid_monochromatic.f90), only with
 * config.monochromatic: for every frequency n_sources packets emitted by the sources at that frequency (their
 * energy carries the emission probability, source_type.f90:440-468) and n_dust packets emitted by the dust from
 * the cell pdf of that frequency; packets always scatter and lose (1 - albedo) of their energy until it drops
 * below monochromatic_energy_threshold of the initial one; every emission / scattering is peeled off into the
 * frequency's own image plane.  hyp_mono_iteration zeroes the cubes and runs all frequencies; the cubes are NOT
 * rescaled afterwards (the energies are absolute).  Split form for sharding: launch(which = 0 sources / 1 dust,
 * inu 0-based) runs ids [first_id, first_id + n_local) of n_total; the block of hyp_final_accumulators'
 * layout is returned by hyp_mono_accumulators; finish reports the counters summed over the launches. */
int  hyp_mono_iteration(hyp_handle h, uint64_t n_sources, uint64_t n_dust, hyp_iter_stats *stats);
int  hyp_mono_launch(hyp_handle h, int which, int inu, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first);
int  hyp_mono_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles);
int  hyp_mono_finish(hyp_handle h, hyp_iter_stats *stats);

/* current state, reference layout [n_dust][n_cells] */
int  hyp_get_specific_energy(hyp_handle h, double *out);
int  hyp_get_density(hyp_handle h, double *out);
int  hyp_set_specific_energy(hyp_handle h, const double *in);

/* n_photons of the last Lucy iteration (src/grid/grid_physics_3d.f90:38, output_grid src/grid/grid_generic.f90:40-46):
 * [n_cells] as doubles, whole-job counts once the accumulator block has been all-reduced; needs config.count_photons
 * or config.pda.  Call after hyp_lucy_accumulators / hyp_lucy_finish. */
int  hyp_get_n_photons(hyp_handle h, double *out);
/* frequency-resolved specific energy (src/grid/grid_generic.f90:71-93): out [n_bins][n_dust][n_cells] (may be NULL),
 * bin_edges_out [n_bins + 1] (may be NULL); needs config.n_spectrum_bins > 0 */
int  hyp_get_specific_energy_spectrum(hyp_handle h, double *out, double *bin_edges_out);
/* specific_energy_converged (src/grid/grid_physics_3d.f90:637-689): the `percentile` quantile of max(a/b, b/a)
 * between the specific energy at the previous call and the current one, computed on the device.  status: 0 value
 * computed, 1 nothing changed (value 0), 2 "could not check for convergence" (only zero cells changed), 3 first call */
int  hyp_convergence_value(hyp_handle h, double percentile, double *value, int *status);

/* measurement hooks for bench.py: duration (ms, HIP events on the engine's
 * stream) of the last propagation kernel and of the last finish step.  After a raytracing or monochromatic
 * iteration: the sum over the propagation kernels of all its launches (sources + dust, every wavelength), finish 0. */
int  hyp_last_kernel_ms(hyp_handle h, float *propagate_ms, float *finish_ms);
/* Options (integers; nothing in the environment changes the engine's behaviour).  The defaults are the measured optima; the tests
 * use the switches to show that the schedules agree with one another.
 *
 *   set and get
 *     "lucy_mode"           -1 auto (default), 0 persistent kernel with global atomics, 1 the tiled schedule of the grid (bricks of a
 *                           Cartesian / polar / AMR grid, clusters of Voronoi cells or octree subtrees in LDS)
 *     "tile_slots" "tile_task" "tile_pools"   slot pool of the tiled schedule: slots (0: 3 << 23, Cartesian grids from 1024 bricks 3 << 24,
 *                           within a third of the free memory and the packet count; imaging iteration 3 << 22, trees 3 << 23),
 *                           packets per walk task (0: 8192, Voronoi grids 16384), pools = streams (3)
 *     "tile_time_walk"      1: HIP events around every walk launch, read back as "last_walk_us" / "last_walk_launches" (bench.py)
 *     "vt_cells" "ot_cells" "at_cells" "pt_lds_kb"   most cells per Voronoi / octree cluster / AMR brick, LDS of a polar brick in KB
 *                           (0 / default: what 156 KB of LDS hold; the shapes built: "vt_max_cells" "vt_max_lds" "ot_max_cells" "at_max_cells"); "tile_drain" (packets in flight below which the Lucy iteration ends
 *                           in one drain launch; -1: 1 000 000), "tile_poll" (generations between two looks at the device), "tile_park" (a wave
 *                           whose task's queue is empty sends its last <= 48 walking packets back to their slots):
 *                           the tests' handles for exercising every path of the schedule on small problems
 *     "interact_threshold" "emit_threshold" "accum_copies" "blocks_per_cu" "chunk"   launch shape of the persistent kernels
 *     "defer_peel"          imaging iteration: 0 inline peel-off, 1 deferred (hyp_defer.h; default), 2 force / 3 forbid the
 *                           propagation half on the tiled schedule (default: from 4e6 packets)
 *     "peel_events"         capacity of the event buffer (default: at most 128 Mi events, fewer if the memory is not there)
 *     "peel_sort" "ff_prepass" "direct_memo"   1 (default): events peeled in cell order; emission + forced first interaction ahead of
 *                           the rounds; a point source's direct light walked once per (source, view)
 *     "gen_defer" "mono_defer"   1 (default): problems with extended sources / monochromatic launches image on the deferred schedule
 *     "plain_imaging"       the inline imaging kernel specialised for point sources (can only be switched off)
 *     "oct_neighbours"      0: the octree walk climbs and descends like the reference instead of using the neighbour table
 *     "reproducible"        1: every iteration runs as ONE wave on the persistent kernels (no tiled / deferred schedule, one accumulator
 *                           copy): floating-point sums are made in that wave's program order, so a seed gives the same bits on every
 *                           run, as the reference's serial binary does (src/main/main.f90:157-161).  For tests: ~1000 times slower
 *   get only: what the last iteration did
 *     "last_lucy_mode" "last_generations" "last_tile_slots" "last_walk_us" "last_walk_launches" "last_defer_rounds" "last_defer_events"
 *     "last_ff_prepass" "last_direct_memo" "last_tiled_imaging" "last_mono_deferred" "last_vt_exact_steps"
 *     "vt_clusters" "vt_max_cells" "ot_clusters" "at_slabs"   shape of the tiled schedule that was built
 *     "pda_last_cells"      cells the partial diffusion approximation solved
 *     "n_photons_inexact"   a packet visited more cells than its visited set holds: the n_photons of the last iteration are an upper bound
 *   get only: sharded runs, the geometry of the blocks that hyp_*_accumulators hand out
 *     "lucy_block_doubles" "image_block_doubles"   their lengths
 *     "lucy_flag_index" "image_flag_index"         index of a spare slot of the block's scalar tail that is never written on the
 *                           device: a rank whose launch failed adds 1 there before the all-reduce (to a zero block of the same
 *                           length), and hyp_*_finish on every rank returns "another rank reported an engine error" when the summed
 *                           slot is not zero.  That is mp_join's role (src/mpi/mpi_routines.f90) without a second collective. */
int  hyp_set_option(hyp_handle h, const char *name, int64_t value);
int  hyp_get_option(hyp_handle h, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif
