/*
 * hyp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, FP64) of the Hyperion Monte Carlo photon-packet
 * path: src/main/iter_lucy.f90, iter_final.f90, iter_final_mono.f90, iter_raytracing.f90 and what they call
 * under src/{core,grid,dust,sources,images} of the reference (six grid geometries, all source types,
 * MRW, peeled and binned images, inside observers).  It is the
 * parity checker for the HIP product path and the `cpu_baseline` leg of
 * bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * may load this library; the product (hyperion_amd/) never does.
 *
 * Parity status: the reference Fortran cannot be built here (its `fortranlib`
 * submodule, github.com/astrofrog/fortranlib @ unknown SHA, is absent), so the
 * random-number stream and the fortranlib sampling/interpolation arithmetic are
 * restated from their published behaviour ("parity unpinned at source level").
 * The oracle IS pinned statistically against the reference's own golden
 * outputs under hyperion/model/tests/data/ (tests/test_oracle_golden.py, tests/test_oracle_mrw.py):
 * test_specific_energy.grid_type={car,oct,amr,sph,cyl}.* (20 files), test_peeloff.grid_type={car,oct,amr,sph,cyl}.
 * raytracing={False,True}.* (20), test_pascucci.tau=* (4: monochromatic + raytracing, spherical grid),
 * test_pinte_seds.tau=* (3) and test_pinte_images.tau=* (2: cylindrical grid, MRW, monochromatic, raytracing),
 * the 18-density temperature table of test_mrw.py and the models of test_mono.py / test_spot_source.py;
 * test_pinte_specific_energy.* (4: the same disc with the PDA).  Features without a reference output
 * (Voronoi walk, binned images, map sources, inside observers) are pinned by analytic results and by
 * equivalence with golden-pinned features (tests/test_oracle_units.py).
 *
 * The descriptor structs below have the same memory layout as the product's
 * include/hyperion_amd.h so one ctypes builder serves both; the two
 * implementations share no code.
 */
#ifndef HYP_ORACLE_H
#define HYP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_DUST 8

/* One dust species: raw tables exactly as stored in the .rtin /Dust/dust_NNN
 * group (reader: src/dust/dust_type_4elem.f90:78-293). */
typedef struct orc_dust_desc {
    int32_t n_nu;               /* optical_properties rows */
    int32_t n_mu;               /* scattering_angles rows */
    int32_t n_jnu;              /* emissivity_variable rows */
    int32_t n_enu;              /* emissivities rows (frequencies) */
    int32_t n_e;                /* mean_opacities rows (0 if not supplied) */
    int32_t sublimation_mode;   /* 0 no, 1 fast, 2 slow, 3 cap */
    int32_t version;            /* dust file version attr */
    int32_t is_lte;
    double  sublimation_specific_energy;
    double  minimum_specific_energy; /* Grid/Quantities attr, this species */
    const double *nu;           /* [n_nu] */
    const double *albedo;       /* [n_nu] */
    const double *chi;          /* [n_nu] */
    const double *mu;           /* [n_mu] */
    const double *P1;           /* [n_nu][n_mu] */
    const double *P2;
    const double *P3;
    const double *P4;
    const double *emiss_nu;     /* [n_enu] */
    const double *emiss_jnu;    /* [n_enu][n_jnu] */
    const double *emiss_var;    /* [n_jnu] specific energies */
    const double *mo_specific_energy; /* [n_e] or NULL */
    const double *mo_chi_rosseland;   /* [n_e] or NULL (sublimation mode 2) */
    const double *mo_kappa_planck;    /* [n_e] or NULL; needed with config.mrw (src/dust/dust.f90:88-93) */
    const double *mo_chi_inv_planck;  /* [n_e] or NULL; column chi_inv_planck (chi_rosseland in version-1 files, dust_type_4elem.f90:231-237) */
} orc_dust_desc;

/* One spot of a spherical source: sub-group `Spot N` of the source group (src/sources/source_type.f90:150-188;
 * attrs longitude, latitude, radius [deg], luminosity, and its own spectrum / temperature). */
typedef struct orc_spot_desc {
    double  longitude, latitude, radius;   /* degrees: angle3d_deg(lon, lat), cos(radius) */
    double  luminosity;
    double  temperature;
    int32_t spectrum_type;  /* 1 tabulated spectrum, 2 blackbody temperature */
    int32_t n_spec;
    const double *spec_nu;  /* [n_spec] */
    const double *spec_fnu; /* [n_spec] */
} orc_spot_desc;

/* One source (reader: src/sources/source_type.f90:102-322). */
typedef struct orc_source_desc {
    int32_t type;          /* 1 point, 2 sphere, 4 map, 5 extern_sph, 6 extern_box, 7 plane_parallel, 8 point_collection */
    int32_t spectrum_type; /* 1 tabulated spectrum, 2 blackbody temperature, 3 lte (map sources only) */
    int32_t peeloff;
    int32_t n_spec;
    int32_t limb_darkening; /* sphere: attr `limb` (source_type.f90:142) */
    int32_t n_points;       /* point_collection: number of points */
    double  luminosity;
    double  temperature;
    double  position[3];
    double  radius;
    double  box[6];        /* xmin,xmax,ymin,ymax,zmin,zmax */
    const double *spec_nu;  /* [n_spec] */
    const double *spec_fnu; /* [n_spec] */
    double  direction[2];     /* plane_parallel: attrs theta, phi (deg) of the beam (source_type.f90:239-256) */
    const double *points;     /* point_collection: [n_points][3] dataset `position` (source_type.f90:258-277) */
    const double *point_lum;  /* point_collection: [n_points] dataset `luminosity` */
    const double *map;        /* map (type 4): [n_cells] dataset `Luminosity map`, cell order of the density (source_type.f90:190-199,
                                 grid_load_pdf_map src/grid/grid_geometry_common_3d.f90:47-63); spectrum_type 3 = 'lte': the dust emissivity
                                 of the emitting cell (select_dust_specific_energy_rho + dust_sample_j_nu, source_type.f90:486-491) */
    int32_t n_spots;          /* sphere: number of spots (the source then is the reference's type 3, a spotted sphere) */
    int32_t reserved_spots;
    const orc_spot_desc *spots; /* [n_spots] */
} orc_source_desc;

/* Grid geometry (reader: src/grid/grid_geometry_cartesian_3d.f90:77-134). */
typedef struct orc_grid_desc {
    int32_t type;          /* 1 = cartesian, 2 = octree, 3 = voronoi, 4 = amr */
    int32_t n1, n2, n3;
    const double *w1;      /* [n1+1] */
    const double *w2;      /* [n2+1] */
    const double *w3;      /* [n3+1] */
    /* octree (src/grid/grid_geometry_octree.f90:184-246): depth-first
     * `refined` flags of ALL cells, centre and half-widths of the top cell */
    int64_t n_cells;
    const int32_t *refined;
    double oct_center[3];
    double oct_half[3];
    /* voronoi (type 3, src/grid/grid_geometry_voronoi.f90:96-188) */
    const double *vor_sites;     /* [n_cells][3] */
    const double *vor_volume;    /* [n_cells] */
    const int32_t *vor_idx;      /* [n_cells+1] CSR offsets (sparse_idx) */
    const int32_t *vor_neighs;   /* neighbour ids, -1..-6 = xmin,xmax,ymin,ymax,zmin,zmax walls */
    double vor_box[6];
    /* amr (type 4, src/grid/grid_geometry_amr.f90:111-180): the grids of all levels, level by
     * level; cells are numbered grid after grid, x fastest (type_cell_id_amr.f90:57-93) */
    int32_t n_amr_levels, n_amr_grids;
    const int32_t *amr_level;    /* [n_amr_grids] 1-based level of each grid, non-decreasing */
    const int32_t *amr_n;        /* [n_amr_grids][3] n1, n2, n3 */
    const double *amr_bounds;    /* [n_amr_grids][6] xmin, xmax, ymin, ymax, zmin, zmax */
    /* voronoi: bounding boxes of the cells, [n_cells][6] = bb_min[3], bb_max[3] of table `cells`, or NULL.  Needed by
     * random_position_cell (src/grid/grid_geometry_voronoi.f90:285-310: rejection sampling in the box) -- map sources,
     * raytraced / monochromatic dust emission */
    const double *vor_bb;
} orc_grid_desc;

/* Run configuration: the root attributes of the .rtin
 * (reader: src/main/setup_rt.f90:38-302). */
typedef struct orc_config {
    int64_t seed;                    /* negative int, default -124902 */
    int64_t n_inter_max;
    int64_t n_reabs_max;
    int32_t kill_on_absorb;
    int32_t kill_on_scatter;
    int32_t sample_sources_evenly;
    int32_t enforce_energy_range;
    int32_t forced_first_interaction;
    int32_t forced_first_interaction_algorithm; /* 1 wr99, 2 baes16 */
    int32_t specific_energy_type;    /* 0 initial, 1 additional */
    int32_t raytracing;              /* root attribute `raytracing`: peel only scattered packets in the final iteration */
    double  baes16_xi;
    double  propagation_check_frequency;
    /* modified random walk (src/grid/grid_mrw_3d.f90, src/main/setup_rt.f90:106-113) */
    int64_t n_inter_mrw_max;
    double  mrw_gamma;
    int32_t mrw;
    int32_t monochromatic;           /* root attribute `monochromatic` (use_exact_nu, src/main/setup_rt.f90:49-57) */
    double  monochromatic_energy_threshold;   /* default 1e-10 */
    const double *frequencies;       /* [n_frequencies] table /frequencies column nu (setup_rt.f90:220-222) */
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
} orc_config;

/* One peeled image group (reader: src/images/images_peeled.f90:272-380,
 * src/images/image_type.f90:153-335). */
typedef struct orc_peeled_desc {
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
    double  nu_min, nu_max;  /* Hz (converted from wav_max/wav_min) */
    double  d_min, d_max;
    double  peeloff_origin[3];
    const double *theta;     /* [n_view] degrees */
    const double *phi;       /* [n_view] degrees */
    int32_t inu_min, inu_max; /* monochromatic: 1-based range of config.frequencies this group images (image_type.f90:243-258); n_nu = inu_max - inu_min + 1 */
    /* filter convolution (attrs use_filters, n_filt and groups filter_NNNNN with tables nu, tn and attr nu0,
     * src/images/image_type.f90:173-181,285-291): n_nu = n_filt planes, a packet is binned into every filter whose
     * transmission at its frequency is positive with that weight (image_bin :467-475); not with raytracing or
     * monochromatic mode */
    int32_t use_filters;
    int32_t reserved_f;
    const int32_t *filt_n;   /* [n_nu] points of each filter curve */
    const double *filt_nu;   /* concatenated, increasing within a filter */
    const double *filt_tr;   /* concatenated transmissions (column tn) */
} orc_peeled_desc;

typedef struct orc_problem {
    orc_grid_desc grid;
    orc_config    config;
    int32_t n_dust;
    int32_t n_sources;
    int32_t n_peeled;
    int32_t reserved0;
    const orc_dust_desc   *dust;     /* [n_dust] */
    const orc_source_desc *sources;  /* [n_sources] */
    const orc_peeled_desc *peeled;   /* [n_peeled] */
    const double *density;           /* [n_dust][n_cells] (n_dust,nz,ny,nx) */
    const double *specific_energy;   /* [n_dust][n_cells] or NULL */
    /* /Output/Binned/group_00001 (src/images/images_binned.f90:42-56): packets leaving the grid in the final iteration
     * are binned by direction into n_binned_theta x n_binned_phi views (cos(theta) in [-1,1], phi in [0,2pi));
     * `binned` describes the image like a peeled group (its n_view, theta, phi, inu_* are ignored); NULL = none.
     * The cubes are returned as group index n_peeled. */
    const orc_peeled_desc *binned;
    int32_t n_binned_theta, n_binned_phi;
} orc_problem;

typedef struct orc_iter_stats {
    double   energy_current;   /* sum of emitted packet energies */
    double   energy_abs_tot[ORC_MAX_DUST];
    uint64_t killed_geo;
    uint64_t killed_int;
    uint64_t crossings;        /* cell-wall crossings + partial steps */
    uint64_t interactions;     /* absorb+scatter events */
    uint64_t n_packets;
} orc_iter_stats;

typedef struct orc_state orc_state;

int  orc_create(const orc_problem *prob, orc_state **out);
void orc_destroy(orc_state *st);
const char *orc_last_error(const orc_state *st);
const char *orc_global_error(void);

/* One Lucy iteration (src/main/iter_lucy.f90:66-237) over packet ids
 * [first_id, first_id+n_local) of an iteration with n_total packets.  `iter`
 * is the 1-based Lucy iteration number (part of the RNG key).  With
 * n_local==n_total this is the whole iteration incl. update_energy_abs;
 * otherwise call orc_lucy_accumulate() + orc_lucy_finish(). */
int orc_lucy_iteration(orc_state *st, uint64_t n_packets, int iter,
                       int n_threads, double *specific_energy_out,
                       orc_iter_stats *stats);
int orc_lucy_accumulate(orc_state *st, uint64_t first_id, uint64_t n_local,
                        int iter, int n_threads, orc_iter_stats *stats);
int orc_lucy_finish(orc_state *st, double *specific_energy_out,
                    orc_iter_stats *stats);
int orc_set_accumulators(orc_state *st, const double *block);
/* raw accumulators after accumulate: [n_dust][n_cells] */
const double *orc_specific_energy_sum(const orc_state *st);
const double *orc_specific_energy(const orc_state *st);
const double *orc_density(const orc_state *st);
/* n_photons of the last Lucy iteration ([n_cells], NULL unless config.count_photons / pda); frequency-resolved specific
 * energy [n_bins][n_dust][n_cells] (NULL unless config.n_spectrum_bins) and its raw sums; number of cells the last
 * solve_pda treated; the value tested by specific_energy_converged (grid_physics_3d.f90:637-689) against `prev` */
const int64_t *orc_n_photons(const orc_state *st);
const double *orc_specific_energy_spectrum(const orc_state *st);
const double *orc_specific_energy_sum_spectrum(const orc_state *st);
int orc_pda_last_cells(const orc_state *st);
int orc_convergence_value(const orc_state *st, const double *prev, double percentile, double *value);

/* Final (imaging) iteration with peel-off (src/main/iter_final.f90:60-273).
 * Image/SED cubes are returned scaled (image_scale) but not dnu-normalised;
 * layout per group: sed[n_stokes][n_orig][n_view][n_ap][n_nu],
 * img[n_stokes][n_orig][n_view][n_y][n_x][n_nu] (the .rtout layout). */
/* do_raytracing (src/main/iter_raytracing.f90:30-143): adds the direct source and the thermal dust
 * emission to the (already scaled) cubes of the last orc_final_iteration. */
int orc_raytracing_iteration(orc_state *st, uint64_t n_sources, uint64_t n_dust, int n_threads, orc_iter_stats *stats);
/* do_final_mono (src/main/iter_final_mono.f90:58-230): all frequencies, source packets then dust packets; zeroes the cubes */
int orc_mono_iteration(orc_state *st, uint64_t n_sources, uint64_t n_dust, int n_threads, orc_iter_stats *stats);
/* one id range of one part (which = 0 sources, 1 dust) at frequency index inu (0-based) */
int orc_mono_accumulate(orc_state *st, int which, int inu, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first,
                        int n_threads, orc_iter_stats *stats);
int orc_raytracing_accumulate(orc_state *st, int which, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first,
                              int n_threads, orc_iter_stats *stats);
/* writable views of the cubes (tests emulate the all-reduce of the image block) */
double *orc_peeled_sed_rw(orc_state *st, int g);
double *orc_peeled_img_rw(orc_state *st, int g);
/* the two halves of orc_final_iteration for sharded runs: unscaled sums of an id range into zeroed cubes; the scaling
 * energy_total / energy_current (image_type.f90:136-151) once cubes and emitted energy have been summed over the ranks */
int orc_final_accumulate(orc_state *st, uint64_t first_id, uint64_t n_local, int n_threads, orc_iter_stats *stats);
int orc_final_scale(orc_state *st, double energy_current);
int orc_final_iteration(orc_state *st, uint64_t n_packets, int n_threads,
                        orc_iter_stats *stats);
int orc_peeled_n_orig(const orc_state *st, int group);
const double *orc_peeled_sed(const orc_state *st, int group);
const double *orc_peeled_img(const orc_state *st, int group);
const double *orc_peeled_sed2(const orc_state *st, int group);
const double *orc_peeled_img2(const orc_state *st, int group);

/* Unit-level probes used by tests (known-answer checks). */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                         uint32_t out[4]);
int    orc_walk_ray(const orc_state *st, const double r0[3],
                    const double v[3], double *path_out);
double orc_probe_uniform(int64_t seed, int iter, uint64_t packet_id, int k);
void   orc_probe_scatter(const orc_state *st, int dust, double nu,
                         const double a_in[4], const double s_in[4],
                         int64_t seed, uint64_t packet_id,
                         double a_out[4], double s_out[4]);
double orc_probe_sample_jnu(const orc_state *st, int dust, int jid,
                            double frac, double xi);
double orc_probe_planck(double T, int64_t seed, uint64_t packet_id);
void   orc_probe_rotate(const double loc_in[4], const double co_in[4],
                        double fin_out[4], double loc_back[4]);
void   orc_probe_optconsts(const orc_state *st, int dust, double nu,
                           double out3[3]);

#ifdef __cplusplus
}
#endif
#endif
