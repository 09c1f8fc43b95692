// hyp_device.h -- device-side data layout and math helpers of the photon-packet
// engine (gfx950).  Product code: never includes anything from oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int HYP_MAXD = 8;
#define SPOT_STRIDE 11
constexpr double HYP_PI = 3.14159265358979323846;
constexpr double HYP_TWOPI = 6.28318530717958647692;
constexpr double HYP_H_CGS = 6.6260755e-27;
constexpr double HYP_K_CGS = 1.380658e-16;
constexpr double HYP_C_CGS = 29979245800.0;
constexpr double HYP_STEF_BOLTZ = 5.67051e-5;
constexpr double HYP_DBL_MAX = 1.7976931348623157e308;
constexpr double HYP_INF = __builtin_huge_val();
constexpr double HYP_DBL_MIN = 2.2250738585072014e-308;

// One dust species, tables resident in HBM (read-mostly, L2/MALL cached).
// Built on the host by build_dust_tables() following dust_type_4elem.f90:78-293.
struct DDust {
    int n_nu, n_mu, n_jnu, n_enu;
    int zero_p2, have_e_range, sublimation_mode, pad0;
    double nu_min, nu_max, mu_min, mu_max;
    double e_min, e_max;                     // mean-opacity specific-energy range
    double minimum_specific_energy, sublimation_specific_energy;
    const double *nu, *log10_nu;             // [n_nu]
    const double *chi, *albedo;              // [n_nu]
    const double *log10_chi, *log10_albedo;  // [n_nu] (NaN where <= 0)
    const double *mu;                        // [n_mu]
    const double *P1, *P2, *P3, *P4;         // [n_nu][n_mu] normalised
    const double *P1_cdf, *P2_cdf;           // [n_nu][n_mu]
    const double *emiss_x;                   // [n_enu]
    const double *emiss_cdf;                 // [n_jnu][n_enu]
    const double *emiss_bp1;                 // [n_jnu][n_enu] power-law index+1 per bin
    const double *emiss_coarse;              // [n_jnu][n_ecoarse] every HYP_COARSE-th entry of emiss_cdf
    int n_ecoarse, pad2;
    const double *jnu_var, *log10_jnu_var;   // [n_jnu]
    const double *mo_e, *mo_chi_ross;        // [n_e] (sublimation mode 2) or null
    int n_e, pad1;
    // modified random walk (grid_mrw_3d.f90): Planck mean opacities and the b_nu = j_nu / kappa_nu pdfs
    const double *mo_kappa_planck, *mo_chi_inv_planck;   // [n_e] or null
    const double *bnu_cdf, *bnu_bp1, *bnu_coarse;         // same layout as emiss_cdf / emiss_bp1 / emiss_coarse
    // monochromatic mode: log10 of the normalised emissivity pdf of every row at the run's frequencies
    // (interpolate_pdf in dust_sample_emit_probability, dust_type_4elem.f90:356-377); -inf where it is zero
    const double *mono_log10_prob;           // [n_jnu][n_frequencies] or null
};

struct DSource {
    double pos[3];
    double temperature;
    double lum_pdf, lum_cdf;
    int spectrum_type, n_spec;
    const double *spec_x, *spec_cdf, *spec_bp1;
    int type, peeloff;        // 1 point, 2 sphere, 5 extern_sph, 6 extern_box
    int limb_darkening;       // sphere only
    int n_points;             // point_collection
    double dir_cost, dir_sint, dir_cosp, dir_sinp;   // plane_parallel: beam direction angle3d_deg(theta, phi)
    const double *points, *point_cdf;   // point_collection: [n][3] positions, luminosity cdf
    // spotted sphere (the reference's source type 3): spot_tab = [cdf over spots..., sphere (n_spots + 1)] then per spot
    // SPOT_STRIDE doubles {nx, ny, nz, cos(radius), spectrum_type, temperature, n_spec, off_x, off_cdf, off_bp1, off_mono}
    // with table offsets relative to spot_blob
    int n_spots;
    int vor_cell1;            // Voronoi grid, point source: cell of the source position + 1 (0: not set)
    const double *spot_tab, *spot_blob;
    const double *map_cdf;              // map (type 4): [n_cells] cumulative of the luminosity map; spectrum_type 3 = 'lte'
    double radius, box[6], face_cdf[6];
};

struct DPeeled {
    int n_view, ignore_optical_depth, compute_image, compute_sed;
    int n_x, n_y, n_ap, n_nu;
    int track_origin, track_n_scat, uncertainties, n_stokes;
    int n_orig, serial;     // serial: option "reproducible" -- deposits into the cubes lane by lane, in lane order (wave_accumulate)
    double x_min, x_max, y_min, y_max, ap_min, ap_max;
    double log10_nu_min, log10_nu_max, log10_ap_min, log10_ap_max;
    double d_min, d_max, origin[3];
    const double *view;     // [n_view][4] cost,sint,cosp,sinp
    double *sed, *sed2, *img, *img2;
    // raytracing caches (images_peeled.f90:57-82): spectra binned on this group's frequency grid
    const double *src_spec;       // [n_sources][n_nu]
    const double *dust_log10_em;  // [n_dust][nj_stride][n_nu]
    const double *dust_chi;       // [n_dust][n_nu]
    int inside_observer, pad_obs; // peel-off towards the point `origin` inside the grid (images_peeled.f90:158-205): lon / lat maps
    int nj_stride, inu_min;       // inu_min: monochromatic, 1-based first frequency of this group (image_type.f90:243-258)
    // filter convolution (image_type.f90:173-181,285-291,467-475): n_nu transmission curves, filt_off[n_nu + 1] into filt_nu / filt_tr
    int use_filters, view_base;           // view_base: global number of this group's first view
    const double *filt_off;       // integer-valued (the tables live in the constant blob of doubles)
    const double *filt_nu, *filt_tr;
};

// Octree cell record (32 B): grid_geometry_octree.f90 / type_grid_octree.f90:14-22.
// Half-widths are root half-width * 2^-level.
// one entry of a Voronoi cell's wall list: the neighbour (or -1..-6: a face of the box) and the neighbour's site
// (`loc` is used by the cluster-tiled schedule, hyp_vtile.h: index of the neighbour inside the cluster whose copy of the record this
// is, -1 for a face of the box, <= -2 where the neighbour belongs to another cluster: -(slot in the cluster's adjacency list) - 2)
struct alignas(32) VorWall { double x, y, z; int nb, loc; };
// Cluster-tiled Voronoi schedule (hyp_vtile.h).  A cluster's tables are one contiguous blob that a walk workgroup copies
// into LDS with 16-byte loads; sections, each padded to 16 bytes, in this order:
//   sx, sy, sz   [n_site] double   sites of the cluster's own cells (index < n_own) and of the cells of OTHER clusters that
//                                  share a wall with one of them ("ghosts", index >= n_own): the exact wall test needs them
//   wrec         [n_wall] float4   FP32 filter of the wall search: (n.x, n.y, n.z, |n|^2 / 2) x scale (x scale^2) per wall of an own cell, in
//                                  the cell's CSR order; n = neighbour's site - own site (a face of the box: the site's mirror image in it)
//   wlink        [n_wall] uint32   neighbour's index in the site table (bits 0-15) | position of THIS cell in the neighbour's
//                                  CSR list (bits 16-23; 255: none) | 1 + face for a face of the box (bits 24-27)
//   hdr          [n_own]  uint32   first wall record of the cell (bits 0-19) | number of walls (bits 20-27) | VT_HDR_EXACT
//   lmax         [n_own]  float    the cell's longest |n| x scale, rounded up: the error bounds of the filter use it for every wall of the cell
//   members      [n_own]  int      cell ids of the own cells
//   gcell, gpacked, gadj [n_site - n_own] int   per ghost: cell id, its vt_cluster word, slot of its cluster in vt_adj (VT_MAX_ADJ: none)
struct VtInfo {
    int blob16;              // offset of the blob in units of 16 bytes
    int n_own, n_site, n_wall;
    int cell0;               // first entry of the cluster in vt_members
    int ghost0;              // first entry of the cluster in vt_ghost
    float scale;             // power of two that brings the cluster's wall normals to order one (FP32 filter only)
    float abs_eps;           // 2^-49 x largest |coordinate| of the cluster x scale: the reference's own rounding of m - r
};
#define VT_HDR_EXACT 0x10000000u    // a neighbour listed twice, more than 32 walls, a wall 2^30 times shorter or longer than the mean: this cell's search skips the filter
#define VT_LINK_LOC(l) ((int)((l) & 0xffffu))
#define VT_LINK_BACK(l) ((int)(((l) >> 16) & 0xffu))
#define VT_LINK_BOX(l) ((int)(((l) >> 24) & 0xfu))       // 0: a neighbouring cell; 1 + face otherwise
#define VT_NO_BACK 255
struct VtGhost { int cell, adj; };   // a ghost's cell id and the slot of its cluster in this cluster's adjacency list (VT_MAX_ADJ: none)
#define VT_MAX_ADJ 64           // adjacent clusters listed per cluster (packets handed to further ones are counted in global memory)

struct alignas(16) OctCell {
    double x, y, z;
    int parent;              // -1 for the root
    signed char subcell;     // position inside the parent, bit0 = x, bit1 = y, bit2 = z
    unsigned char level;
    unsigned char refined;
    unsigned char pad;
};

// A pointer that comes out of a DProblem in memory is "flat" to the compiler (it cannot know that it does not point into LDS or
// scratch; neither an address-space cast pair nor an assumption changes that with this hipcc), and a flat load is issued to the
// LDS and the vector-memory path both, counts on lgkmcnt and vmcnt, and is not merged with its neighbours: an OctCell arrives as
// dwordx4 + dwordx2 + dword + ubyte.  The tables of the walks live in global memory: these loaders say so with address-space-1
// pointers to builtin types (hyp_ldg: one scalar; oct_cell_ldg: the 32-byte record as two global_load_dwordx4).  Measured
// (profiles/r03_tiled_log.md): configs[3] imaging 427 -> 390 ms; the lookup tables of the interaction / emission kernels neutral;
// the Voronoi wall records of the persistent kernel and of place_in_cell's site search are left as they were (two dwordx4 per
// record instead of dwordx4 + dwordx2 + dword read 32 bytes for 28 and cost 2.5 % of a configs[4] iteration).
#if defined(__HIPCC__)
#define HYP_AS1 __attribute__((address_space(1)))
typedef float hyp_v4f __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ T hyp_ldg(const T *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const HYP_AS1 T *)p;
#else
    return *p;
#endif
}
// unsafeAtomicAdd on an address in global memory: global_atomic_add_f64 instead of the flat form
__device__ __forceinline__ void hyp_atomic_add_g(double *p, double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)__hip_atomic_fetch_add((HYP_AS1 double *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p += v;
#endif
}
__device__ __forceinline__ OctCell oct_cell_ldg(const OctCell *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const HYP_AS1 hyp_v4f *q = (const HYP_AS1 hyp_v4f *)p;
    const hyp_v4f u = q[0], w = q[1];
    OctCell o;
    o.x = __hiloint2double(__float_as_int(u.y), __float_as_int(u.x)); o.y = __hiloint2double(__float_as_int(u.w), __float_as_int(u.z));
    o.z = __hiloint2double(__float_as_int(w.y), __float_as_int(w.x)); o.parent = __float_as_int(w.z);
    const unsigned int m = (unsigned int)__float_as_int(w.w);
    o.subcell = (signed char)(m & 0xffu); o.level = (unsigned char)((m >> 8) & 0xffu); o.refined = (unsigned char)((m >> 16) & 0xffu); o.pad = 0;
    return o;
#else
    return *p;
#endif
}
#endif

// Slots of the scalar tail that follows the per-cell accumulators.
// TAIL_RANK_ERROR: never written on the device; a rank of a sharded run that failed adds 1 there before the all-reduce
// (hyperion_amd/distributed.py, hyp_run.cpp), so the sum tells every rank.
enum { TAIL_ENERGY = 0, TAIL_KILLED_GEO = 1, TAIL_KILLED_INT = 2, TAIL_CROSSINGS = 3,
       TAIL_INTERACTIONS = 4, TAIL_RANK_ERROR = 5, TAIL_SIZE = 8 };

enum { ERR_NONE = 0, ERR_NU_RANGE = 1, ERR_NOT_IN_CELL = 2, ERR_NEGATIVE_T = 3, ERR_RAY_GRID = 4, ERR_INTERNAL = 5 };

// One grid of an AMR level (type_grid_amr.f90:12-21).  Walls are linspace(lo, hi, n+1) as the
// reference builds them (grid_geometry_amr.f90:124-137), stored once per grid.
struct AmrGrid {
    double lo[3], hi[3];
    int n[3];
    int go_off;              // offset of the (n1+2)(n2+2)(n3+2) goto table: grid to continue in + 1, 0 = stay
    int w_off[3];            // offsets of the three wall arrays in amr_walls
    unsigned int start;      // unique id of the first cell
};

// brick-tiled AMR schedule (hyp_atile.h): cells [o, o + n) per axis of grid `grid`; go_off = its slice of the goto table
// in DProblem::at_go, (n0 + 2)(n1 + 2)(n2 + 2) 16-bit entries (ghost layer included)
struct AtSlab { int grid, o[3], n[3], go_off; };

struct DProblem {
    int n1, n2, n3, n_dust;
    int n_sources, n_peeled, sample_sources_evenly, kill_on_absorb;
    int kill_on_scatter, forced_first, forced_algo, pad0;
    long long n_inter_max;
    unsigned long long n_cells;
    double baes16_xi;
    double check_p, check_log1mp;         // propagation_check_frequency p, log(1-p)
    uint32_t seed_key, pad1;
    int grid_type, pad3;                  // 1 cartesian, 2 octree, 3 voronoi, 4 amr, 5 spherical polar, 6 cylindrical polar
    const double *w[3], *ew[3];           // walls and 3*spacing(wall)
    // spherical / cylindrical polar grids (grid_type 5 / 6): w = (r, theta, phi) or (w, z, phi) walls, wr2 = w1^2,
    // tan(theta), tan^2(theta), cos(theta), tan(phi) of the walls; index of the theta = pi/2 wall (-2: none); 2 if n3 == 1 else 3
    const double *wr2, *wtant, *wtant2, *wcost, *wtanp;
    int midplane, n_dim;
    const OctCell *oct_cells;             // [n_cells]
    const int *oct_children;              // [n_cells][8], -1 where not refined
    const int *oct_neigh;                 // [n_cells][6] (axis * 2 + up): see geo_advance; n_cells = outside the grid
    double oct_half[3], oct_box[6], oct_eps;
    const double *vor_sites;              // voronoi: [n_cells][3]
    const int *vor_idx, *vor_neigh;       // CSR neighbour lists (ids >= 0, walls -1..-6)
    const VorWall *vor_walls;             // per CSR entry: the neighbour and its site (geo_find_wall)
    const int *vor_seed;                  // [vor_g^3] start site of the nearest-site walk
    const double *vor_volume;             // [n_cells]
    const double *vor_bb;                 // [n_cells][6] bb_min, bb_max of the cells (random_position_cell) or null
    double vor_box[6];
    int vor_g, pad4;
    // cluster-tiled schedule (hyp_vtile.h): cells grouped into spatially compact clusters whose tables fit in LDS
    const int *vt_cluster;                // [n_cells] cluster << 16 | index of the cell in its cluster
    const VtInfo *vt_info;                // [n_clusters]
    const float4 *vt_blob;                // the clusters' tables (VtInfo)
    const int *vt_members;                // [n_cells] cell ids, cluster by cluster
    const VtGhost *vt_ghost;              // ghosts of every cluster (VtInfo::ghost0)
    const int *vt_adj;                    // [n_clusters][VT_MAX_ADJ] adjacent clusters (unused slots: the cluster itself)
    // cluster-tiled octree schedule (hyp_otile.h): clusters = runs of sibling subtrees (contiguous cell ids) whose records fit in LDS
    const int *ot_cluster;                // [n_cells] cluster of the cell (-1: a cell above the clusters, never a leaf)
    const int *ot_c0, *ot_nc;             // [n_clusters] first cell id and number of cells
    const int *ot_kid_off;                // [n_clusters + 1] first row of the cluster in ot_kid
    const OctCell *ot_rec;                // [n_cells] oct_cells with `parent` replaced by the cell's row in its cluster's ot_kid slice (refined cells)
    const short *ot_kid;                  // [rows][8] children of the refined cells of a cluster as indices inside the cluster
    const short *ot_nb;                   // [n_cells][6] oct_neigh as an index inside the cell's cluster; -1: outside the grid, -2: in another cluster
    // brick-tiled AMR schedule (hyp_atile.h): every grid is cut into bricks of at most at_b[0] x at_b[1] x at_b[2] cells
    const AtSlab *at_slabs;               // [n_bricks]
    const short *at_go;                   // goto-table slices of the bricks, ghost layer included
    const int *at_grid_c0, *at_grid_nb;   // [n_amr_grids] first brick of a grid; [n_amr_grids][2] bricks along x and y:
    int at_b[3], pad_at;                  //   brick of a cell = c0 + ((i3 / b2) nb1 + i2 / b1) nb0 + i1 / b0
    const AmrGrid *amr_grids;             // amr: [n_amr_grids], level by level
    const int *amr_go;                    // goto tables of all grids
    const double *amr_walls;              // wall arrays of all grids
    const int *amr_cell_grid;             // [n_cells] grid of each unique cell id
    double amr_eps;                       // half the smallest cell width (grid_geometry_amr.f90:350)
    int n_amr_grids, n_amr_level1;        // level-1 grids come first
    // raytracing (iter_raytracing.f90): valid cells, current specific energy, absorbed energy per species
    const unsigned int *mask_map;         // [n_masked]
    unsigned long long n_masked;
    const double *specific_energy;        // [n_cells][n_dust]
    const double *energy_abs_tot;         // [n_dust]
    double energy_total;
    int peel_scattered_only;              // final iteration peels only scattered packets (raytracing on)
    int binned, n_bin_theta, n_bin_phi;   // binned images (images_binned.f90): index of their group in `peeled` (-1: none), direction bins
    int pad7;
    int any_intersect;                    // a source can re-absorb packets (spheres): source.f90:216
    long long n_reabs_max;
    // modified random walk: per-iteration tables of mrw_prepare_kernel and the cumulative of Min et al. (2009) eq. 6
    int mrw, pad5;
    long long n_inter_mrw_max;
    double mrw_gamma;
    const double *mrw_alpha, *mrw_diff;   // [n_cells] alpha_inv_planck, diff_coeff
    const double *mrw_kp;                 // [n_cells][n_dust] kappa_planck(specific_energy)
    const double *mrw_x, *mrw_y;          // [100]
    // monochromatic final iteration (iter_final_mono.f90): the frequency table and, refreshed by the host before every
    // launch, the part being run (0 = not a monochromatic launch, 1 = source packets, 2 = dust packets), the frequency
    // and the cell emission pdfs of grid_monochromatic.f90
    int mono_which, mono_inu, n_frequencies, pad6;
    double mono_nu, mono_n_total, mono_threshold;
    const double *mono_src_prob;          // [n_sources][n_frequencies] emission probability of each source at each frequency
    const double *mono_cdf;               // [n_dust][n_cells] cumulative of prob x energy over cells, normalised
    double mono_mean_prob[HYP_MAXD];
    const double *density;                // [n_cells][n_dust]   (cell-major)
    double *sum;                          // [n_copies][n_cells][n_dust] accumulators
    unsigned long long copy_stride;       // doubles between accumulator copies
    int n_copies, pad2;
    double *tail;                         // [TAIL_SIZE]
    const int *jnu_id;                    // [n_cells][n_dust]
    const double *jnu_frac;               // [n_cells][n_dust]
    unsigned long long *counter;          // packet-id dispenser
    int *err;                             // [0] code
    double *err_data;                     // [0..2]
    const DSource *sources;
    const DPeeled *peeled;
    int n_views_total, pad9;              // peeled views numbered through all groups (DPeeled::view_base)
    // n_photons (grid_propagate_3d.f90:88-93,171-176): packets that entered each cell in this Lucy iteration [n_cells];
    // visit_tab = per-lane set of the cells the lane's current packet has been counted in, see count_photon.
    // count_photons = 0: arrays absent.
    unsigned int *n_photons;
    unsigned long long *visit_tab;        // [lanes of the launch][HYP_VISIT_SLOTS]: see count_photon
    int *nphot_inexact;                   // set when a packet overflowed its table
    int count_photons;
    // frequency-resolved specific energy (grid_propagate_3d.f90:59-71,155-158,214-222): accumulators [n_bins][n_cells][n_dust],
    // log10 of the bin edges [n_bins + 1], fraction of each emissivity row in each bin [n_dust][nj_max][n_bins] (MRW deposits)
    int n_bins, nj_max, pad8;
    const double *log_nu_edges;
    double *sum_spec;
    const double *jnu_bin_frac;
    DDust dust[HYP_MAXD];
};

struct LaunchParams {
    unsigned long long first_id, end_id;
    uint32_t iter_tag;
    int chunk;
    int interact_threshold, emit_threshold;
};

// Deferred peel-off (hyp_defer.h): control block and buffers of one {propagate, peel} round
constexpr int HYP_PEEL_CHUNK = 512;      // event slots a wave reserves at a time (one same-address atomic per chunk)
struct PeelCtl {
    unsigned long long reserved;        // event slots reserved in this round (failed reservations count: may exceed the capacity)
    unsigned long long pair_cursor;     // (event, view) pairs handed out by the peel kernel
    unsigned long long written;         // events written in this round
    unsigned int n_susp[2], n_ret[2];   // packets set aside / id ranges returned, by round parity
    unsigned long long n_sorted;        // events of this round in DeferBuf::order (sorted peel-off)
    unsigned long long ff_cursor;       // packet ids handed out by the forced-first-interaction pre-pass (ff_walk_kernel)
};

// The direct light of a point source towards one view is the same walk for every packet (same start, same direction): its
// column density per dust species, crossings and fate, computed once per (source, view) by direct_column_kernel (hyp_defer.h);
// the peel kernel then gives an emission's (event, view) pair tau = sum_d chi_d(nu) x col[d] instead of walking it again.
struct DirectCol {
    double col[4];                      // sum of rho_d x t over the cells of the walk
    unsigned int crossings;             // cells crossed (what every such walk adds to the crossing tally)
    int status;                         // 0: walk it (inside observer, not a point source, ...);
};                                      // 1: leaves the grid; 2: ends in a failed wall search / an invalid cell (one killed packet per event)
struct DeferBuf {
    void *events;                       // PeelEvent<NDT, GEOM>[cap] (hyp_defer.h)
    unsigned long long cap;             // a multiple of HYP_PEEL_CHUNK
    PeelCtl *ctl;
    void *susp[2];                      // SuspRec<NDT, GEOM>[one per lane of the propagation grid], by round parity
    unsigned long long *ret[2];         // (next, end) pairs, one per wave of the propagation grid
    int cur;                            // parity of this round
    // sorted peel-off (hyp_defer.h: peel_sort_*): the round's events ordered by the cell they happened in, so that the lanes
    // of a peel wave walk side by side; null = the events are taken in the order they were written
    unsigned int *order;                // [cap] event slots by ascending key
    unsigned int *keys;                 // [cap] key of every event slot (HYP_SORT_EMPTY: nothing written there)
    unsigned int *bins;                 // [2 * n_bins]: counts (then cursors) | offsets
    unsigned int n_bins;
    // emission and forced first interaction made ahead of the rounds (hyp_defer.h: ff_walk_kernel): EmitRec<NDT>[packet ids of the
    // launch]; null = the propagation kernel emits and walks to the edge itself (ST_FF lanes)
    void *ff;
    const DirectCol *direct;            // [n_sources x n_views_total] or null (option direct_memo = 0, memory)
};
constexpr unsigned int HYP_SORT_EMPTY = 0xffffffffu;
constexpr int HYP_SORT_MAX_BINS = 4096;
constexpr int HYP_SORT_PER_WG = 8192;      // event slots per workgroup of the sort kernels

// ---------------------------------------------------------------------------
// Philox4x32-10, one stream pair per packet (counter = packet id, block, stream)
// ---------------------------------------------------------------------------
struct Rng {
    uint32_t key0, key1;
    uint32_t id_lo, id_hi;
    uint32_t blk_a, blk_b;
    double buf_a;
    int have_a;
    int countdown;             // cell steps left until the next propagation check
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t o[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32 x 32 -> 64 bit product per multiplier (v_mad_u64_u32) instead of separate high and low halves
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ double u64_to_unit(uint32_t hi, uint32_t lo)
{
    unsigned long long u = (((unsigned long long)hi << 32) | lo) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void rng_init(Rng &g, uint32_t key0, uint32_t key1, unsigned long long id)
{
    g.key0 = key0; g.key1 = key1; g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
    g.blk_a = 0; g.blk_b = 0; g.have_a = 0; g.countdown = 0; g.buf_a = 0.0;
}

__device__ __forceinline__ double rng_uniform(Rng &g)
{
    if (g.have_a) { g.have_a = 0; return g.buf_a; }
    uint32_t o[4];
    philox4x32_10(g.id_lo, g.id_hi, g.blk_a, 0u, g.key0, g.key1, o);
    g.blk_a++;
    g.buf_a = u64_to_unit(o[2], o[3]); g.have_a = 1;
    return u64_to_unit(o[0], o[1]);
}

// The reference draws one uniform per cell step and checks the packet's cell
// with probability p (grid_propagate_3d.f90:108).  The same Bernoulli process is
// generated from its gap distribution: steps until the next check
// = floor(log(1-u)/log(1-p)), one stream-B draw per check instead of per step.
// stream 1: the packet's own propagation checks; stream 2: those of a peel-off walk (its own block range, see peel_rng)
__device__ __forceinline__ int rng_check_gap(Rng &g, double p, double log1mp, uint32_t stream = 1u)
{
    if (p >= 1.0) return 0;
    if (!(p > 0.0)) return 2147483647;
    uint32_t o[4];
    philox4x32_10(g.id_lo, g.id_hi, g.blk_b, stream, g.key0, g.key1, o);
    g.blk_b++;
    double gap = floor(log(1.0 - u64_to_unit(o[0], o[1])) / log1mp);
    return gap >= 2147483647.0 ? 2147483647 : (int)gap;
}

__device__ __forceinline__ double rng_exp(Rng &g) { return -log(1.0 - rng_uniform(g)); }

// ---------------------------------------------------------------------------
// table helpers (fortranlib lib_array / type_pdf semantics)
// ---------------------------------------------------------------------------

// Fortran spacing(x): distance to the next larger representable number
__device__ __forceinline__ double spacing_d(double x)
{
    x = fabs(x);
    if (x == 0.0) return 2.2250738585072014e-308;
    return __longlong_as_double(__double_as_longlong(x) + 1) - x;
}

// j with x[j] <= xv < x[j+1]; xv == x[n-1] -> n-2; -1 outside (ascending x)
__device__ __forceinline__ int locate(const double *__restrict__ x, int n, double xv)
{
    if (!(xv >= x[0]) || !(xv <= x[n - 1])) return -1;
    if (xv == x[n - 1]) return n - 2;
    int jl = 0, ju = n - 1;
    while (ju - jl > 1) {
        int jm = (ju + jl) >> 1;
        if (xv >= x[jm]) jl = jm; else ju = jm;
    }
    return jl;
}

// locate() on a table in global memory (dust, source and run tables: everything but the walls staged in LDS): global loads
__device__ __forceinline__ int locate_g(const double *__restrict__ x, int n, double xv)
{
    const double x0 = hyp_ldg(x), xn = hyp_ldg(x + n - 1);
    if (!(xv >= x0) || !(xv <= xn)) return -1;
    if (xv == xn) return n - 2;
    int jl = 0, ju = n - 1;
    while (ju - jl > 1) {
        int jm = (ju + jl) >> 1;
        if (xv >= hyp_ldg(x + jm)) jl = jm; else ju = jm;
    }
    return jl;
}

// log-log interpolation with precomputed log10 tables; linear where an
// ordinate is not positive.  j = bracketing bin, lxv = log10(xv).
__device__ __forceinline__ double interp_loglog_at(const double *__restrict__ x, const double *__restrict__ lx,
                                                   const double *__restrict__ y, const double *__restrict__ ly,
                                                   int j, double xv, double lxv)
{
    double y1 = hyp_ldg(y + j), y2 = hyp_ldg(y + j + 1);
    if (y1 > 0.0 && y2 > 0.0) {
        const double lx1 = hyp_ldg(lx + j), ly1 = hyp_ldg(ly + j);
        double f = (lxv - lx1) / (hyp_ldg(lx + j + 1) - lx1);
        return exp10(ly1 + f * (hyp_ldg(ly + j + 1) - ly1));
    }
    const double x1 = hyp_ldg(x + j);
    return y1 + (xv - x1) / (hyp_ldg(x + j + 1) - x1) * (y2 - y1);
}

// bilinear, a[iy][ix]
__device__ __forceinline__ double bilinear(const double *__restrict__ a, int nx, int i, int j, double fx, double fy)
{
    const double *r0 = a + (size_t)j * nx + i, *r1 = r0 + nx;
    double a00 = hyp_ldg(r0), a10 = hyp_ldg(r0 + 1), a01 = hyp_ldg(r1), a11 = hyp_ldg(r1 + 1);
    return a00 * (1 - fx) * (1 - fy) + a10 * fx * (1 - fy) + a01 * (1 - fx) * fy + a11 * fx * fy;
}

// invert a piecewise power-law CDF (type_pdf sample_pdf with log=.true.)
__device__ __forceinline__ double sample_log_pdf(const double *__restrict__ x, const double *__restrict__ cdf,
                                                 const double *__restrict__ bp1, int n, double xi)
{
    int j = locate_g(cdf, n, xi);
    if (j < 0) j = 0;
    double c1 = hyp_ldg(cdf + j), c2 = hyp_ldg(cdf + j + 1);
    double f = (c2 > c1) ? (xi - c1) / (c2 - c1) : 0.0;
    double x1 = hyp_ldg(x + j), x2 = hyp_ldg(x + j + 1), b = hyp_ldg(bp1 + j);
    if (b != b) return x1 + f * (x2 - x1);
    if (fabs(b) < 1e-10) return x1 * pow(x2 / x1, f);
    return x1 * pow(1.0 + f * (pow(x2 / x1, b) - 1.0), 1.0 / b);
}

// The same inversion for two CDF rows at once (the two emissivity tables that bracket a cell's
// specific energy, dust_type_4elem.f90:379-398), with a two-level search: `coarse` holds every
// HYP_COARSE-th entry of a row (small enough to stay in L1), so the bracket is found with
// log2(n / HYP_COARSE) short-latency steps plus ONE trip to the row itself, instead of log2(n)
// dependent trips.  Both searches return the last j with cdf[j] <= xi, like locate() (whose
// edge rules are reproduced), so the result is the same number.
constexpr int HYP_COARSE = 8;
__device__ __forceinline__ void sample_log_pdf_pair(const double *__restrict__ x, const double *__restrict__ cdf_a, const double *__restrict__ cdf_b,
                                                    const double *__restrict__ bp1_a, const double *__restrict__ bp1_b,
                                                    const double *__restrict__ co_a, const double *__restrict__ co_b, int n, int nc, double xi,
                                                    double &xa, double &xb)
{
    const bool in_a = (xi >= hyp_ldg(cdf_a)) && (xi <= hyp_ldg(cdf_a + n - 1));
    const bool in_b = (xi >= hyp_ldg(cdf_b)) && (xi <= hyp_ldg(cdf_b + n - 1));
    int lo_a = 0, hi_a = nc, lo_b = 0, hi_b = nc;
    while (hi_a - lo_a > 1 || hi_b - lo_b > 1) {
        const int ma = (lo_a + hi_a) >> 1, mb = (lo_b + hi_b) >> 1;
        const double va = hyp_ldg(co_a + ma), vb = hyp_ldg(co_b + mb);
        if (hi_a - lo_a > 1) { if (xi >= va) lo_a = ma; else hi_a = ma; }
        if (hi_b - lo_b > 1) { if (xi >= vb) lo_b = mb; else hi_b = mb; }
    }
    const int base_a = lo_a * HYP_COARSE, base_b = lo_b * HYP_COARSE;
    double wa[HYP_COARSE + 1], wb[HYP_COARSE + 1];
#pragma unroll
    for (int k = 0; k <= HYP_COARSE; k++) {      // the window and the entry after it, all loads independent
        wa[k] = hyp_ldg(cdf_a + min(base_a + k, n - 1));
        wb[k] = hyp_ldg(cdf_b + min(base_b + k, n - 1));
    }
    int ca = 0, cb = 0;
#pragma unroll
    for (int k = 1; k < HYP_COARSE; k++) {
        ca += (base_a + k < n && wa[k] <= xi) ? 1 : 0;
        cb += (base_b + k < n && wb[k] <= xi) ? 1 : 0;
    }
    int ja = base_a + ca, jb = base_b + cb;
    if (ja == n - 1) ja = n - 2;
    if (jb == n - 1) jb = n - 2;
    if (!in_a) ja = 0;           // locate() = -1, sample_log_pdf then uses bin 0
    if (!in_b) jb = 0;
    // c1 = cdf[j], c2 = cdf[j+1]: from the window where possible
    double c1a = hyp_ldg(cdf_a + ja), c2a = hyp_ldg(cdf_a + ja + 1), c1b = hyp_ldg(cdf_b + jb), c2b = hyp_ldg(cdf_b + jb + 1);
    (void)wa; (void)wb;
    const double x1a = hyp_ldg(x + ja), x2a = hyp_ldg(x + ja + 1), ba = hyp_ldg(bp1_a + ja);
    const double x1b = hyp_ldg(x + jb), x2b = hyp_ldg(x + jb + 1), bb = hyp_ldg(bp1_b + jb);
    const double fa = (c2a > c1a) ? (xi - c1a) / (c2a - c1a) : 0.0;
    const double fb = (c2b > c1b) ? (xi - c1b) / (c2b - c1b) : 0.0;
    if (ba != ba) xa = x1a + fa * (x2a - x1a);
    else if (fabs(ba) < 1e-10) xa = x1a * pow(x2a / x1a, fa);
    else xa = x1a * pow(1.0 + fa * (pow(x2a / x1a, ba) - 1.0), 1.0 / ba);
    if (bb != bb) xb = x1b + fb * (x2b - x1b);
    else if (fabs(bb) < 1e-10) xb = x1b * pow(x2b / x1b, fb);
    else xb = x1b * pow(1.0 + fb * (pow(x2b / x1b, bb) - 1.0), 1.0 / bb);
}

// ---------------------------------------------------------------------------
// angles (fortranlib type_angle3d semantics).  Orientation: local azimuth in
// (0,pi) turns the direction towards increasing phi; pinned by the sign of
// Stokes U in the reference's golden peel-off outputs.
// ---------------------------------------------------------------------------
struct Angle { double cost, sint, cosp, sinp; };

__device__ __forceinline__ void angle_to_vector(const Angle &a, double &vx, double &vy, double &vz)
{
    vx = a.sint * a.cosp; vy = a.sint * a.sinp; vz = a.cost;
}

__device__ __forceinline__ void random_sphere_angle(Rng &g, Angle &a)
{
    double mu = 2.0 * rng_uniform(g) - 1.0;
    double phi = HYP_TWOPI * rng_uniform(g);
    a.cost = mu; a.sint = sqrt(1.0 - mu * mu);
    double s, c;
    sincos(phi, &s, &c);
    a.cosp = c; a.sinp = s;
}

__device__ __forceinline__ double clamp1(double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); }

// new direction = old direction `co` deflected by the local angle `loc`
__device__ __forceinline__ void rotate_angle(const Angle &loc, const Angle &co, Angle &fin)
{
    double cos_a = co.cost, sin_a = co.sint, cos_b = loc.cost, sin_b = loc.sint;
    double cos_C = loc.cosp, sin_C = fabs(loc.sinp);
    double cos_c = clamp1(cos_a * cos_b + sin_a * sin_b * cos_C);
    double sin_c = sqrt(1.0 - cos_c * cos_c);
    double cos_B, sin_B;
    if (fabs(sin_a) < 1e-12 || sin_c < 1e-12) {
        if (fabs(sin_a) < 1e-12) { cos_B = (cos_a > 0 ? -cos_C : cos_C); sin_B = sin_C; }
        else { cos_B = 1.0; sin_B = 0.0; }
    } else {
        cos_B = clamp1((cos_b - cos_a * cos_c) / (sin_a * sin_c));
        sin_B = sqrt(1.0 - cos_B * cos_B);
    }
    fin.cost = cos_c; fin.sint = sin_c;
    if (loc.sinp < 0.0) {   // new phi = old phi - B
        fin.cosp = co.cosp * cos_B + co.sinp * sin_B;
        fin.sinp = co.sinp * cos_B - co.cosp * sin_B;
    } else {                // new phi = old phi + B
        fin.cosp = co.cosp * cos_B - co.sinp * sin_B;
        fin.sinp = co.sinp * cos_B + co.cosp * sin_B;
    }
}

// local angle that takes `co` to `fin` (inverse of rotate_angle)
__device__ __forceinline__ void difference_angle(const Angle &co, const Angle &fin, Angle &loc)
{
    double cos_a = co.cost, sin_a = co.sint, cos_c = fin.cost, sin_c = fin.sint;
    double cos_B = co.cosp * fin.cosp + co.sinp * fin.sinp;
    double sin_Bs = co.cosp * fin.sinp - co.sinp * fin.cosp;   // sin(new phi - old phi)
    double cos_b = clamp1(cos_a * cos_c + sin_a * sin_c * cos_B);
    double sin_b = sqrt(1.0 - cos_b * cos_b);
    loc.cost = cos_b; loc.sint = sin_b;
    if (sin_b < 1e-12 || sin_a < 1e-12) {
        if (sin_a < 1e-12 && sin_b >= 1e-12) { loc.cosp = (cos_a > 0 ? -cos_B : cos_B); loc.sinp = sin_Bs; }
        else { loc.cosp = 1.0; loc.sinp = 0.0; }
        return;
    }
    double cos_C = clamp1((cos_c - cos_a * cos_b) / (sin_a * sin_b));
    double sin_C = sqrt(1.0 - cos_C * cos_C);
    loc.cosp = cos_C;
    loc.sinp = (sin_Bs >= 0.0) ? sin_C : -sin_C;
}

// dust_type_4elem.f90:603-690
__device__ __forceinline__ void scatter_stokes(double s[4], const Angle &a_coord, const Angle &a_scat,
                                               const Angle &a_final, double P1, double P2, double P3, double P4)
{
    double cos_a = a_coord.cost, sin_a = a_coord.sint;
    double cos_b = a_scat.cost, sin_b = a_scat.sint;
    double cos_c = a_final.cost, sin_c = a_final.sint;
    double cos_big_b = a_coord.cosp * a_final.cosp + a_coord.sinp * a_final.sinp;
    double cos_big_c = a_scat.cosp, sin_big_c = fabs(a_scat.sinp);
    double cos_big_a, sin_big_a;
    if (sin_big_c < 10.0 * HYP_DBL_MIN && sin_c < 10.0 * HYP_DBL_MIN) {
        cos_big_a = -cos_big_b * cos_big_c;
        sin_big_a = sqrt(1.0 - cos_big_a * cos_big_a);
    } else {
        cos_big_a = (cos_a - cos_b * cos_c) / (sin_b * sin_c);
        sin_big_a = sin_big_c * sin_a / sin_c;
    }
    double cos_2_i2 = 1.0 - 2.0 * sin_big_a * sin_big_a;
    double sin_2_i2 = 2.0 * sin_big_a * cos_big_a;
    double cos_2_alpha = 1.0 - 2.0 * a_scat.sinp * a_scat.sinp;
    double sin_2_alpha = -2.0 * a_scat.sinp * a_scat.cosp;
    double cos_2_beta = cos_2_i2;
    double sin_2_beta = (a_scat.sinp < 0.0) ? sin_2_i2 : -sin_2_i2;
    double I = s[0], Q = s[1], U = s[2], V = s[3];
    double RLS1 = P1 * I + P2 * (cos_2_alpha * Q + sin_2_alpha * U);
    double RLS2 = P2 * I + P1 * (cos_2_alpha * Q + sin_2_alpha * U);
    double RLS3 = -P4 * V + P3 * (-sin_2_alpha * Q + cos_2_alpha * U);
    double RLS4 = P3 * V + P4 * (-sin_2_alpha * Q + cos_2_alpha * U);
    s[0] = RLS1;
    s[1] = cos_2_beta * RLS2 + sin_2_beta * RLS3;
    s[2] = -sin_2_beta * RLS2 + cos_2_beta * RLS3;
    s[3] = RLS4;
}
