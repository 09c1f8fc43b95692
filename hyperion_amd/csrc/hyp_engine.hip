// hyp_engine.hip -- host side of the C-ABI (include/hyperion_amd.h): table
// construction, device residency, kernel launches, iteration epilogue.
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include "../../include/hyperion_amd.h"
#include "hyp_kernels.h"
#include "hyp_tiled.h"
#include "hyp_epilogue.h"
#include "hyp_pick.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace {

std::string g_error;   // message of a failed hyp_create

#define HIP_TRY(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_));                  \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

// A host-side pool of doubles that becomes one device allocation; tables are
// addressed by offset until upload, then by pointer.
struct Blob {
    std::vector<double> h;
    size_t put(const double *a, size_t n) { size_t o = h.size(); h.insert(h.end(), a, a + n); return o; }
    size_t put(const std::vector<double> &v) { return put(v.data(), v.size()); }
};

double seg_loglog(double x1, double x2, double y1, double y2)
{
    if (!(y1 > 0.0 && y2 > 0.0)) return 0.0;
    double b = std::log10(y1 / y2) / std::log10(x1 / x2);
    if (std::fabs(b + 1.0) < 1e-10) return x1 * y1 * std::log(x2 / x1);
    return y1 * (x2 * std::pow(x2 / x1, b) - x1) / (b + 1.0);
}

// interpolate_pdf(pdf, xv, bounds_error=.false., fill_value=0) of a log pdf set from (x, y[stride]): the normalised pdf
// interpolated in log-log (linear where an ordinate is not positive), 0 outside the table
double interp_log_pdf(const double *x, const double *y, size_t stride, int n, double xv)
{
    if (!(xv >= x[0]) || !(xv <= x[n - 1])) return 0.0;
    double norm = 0.0;
    for (int i = 0; i + 1 < n; i++) norm += seg_loglog(x[i], x[i + 1], y[(size_t)i * stride], y[(size_t)(i + 1) * stride]);
    if (!(norm > 0.0)) return 0.0;
    int j;
    if (xv == x[n - 1]) j = n - 2;
    else { int jl = 0, ju = n - 1; while (ju - jl > 1) { int jm = (ju + jl) >> 1; if (xv >= x[jm]) jl = jm; else ju = jm; } j = jl; }
    const double y1 = y[(size_t)j * stride] / norm, y2 = y[(size_t)(j + 1) * stride] / norm;
    if (y1 > 0.0 && y2 > 0.0) {
        const double f = (std::log10(xv) - std::log10(x[j])) / (std::log10(x[j + 1]) - std::log10(x[j]));
        return std::pow(10.0, std::log10(y1) + f * (std::log10(y2) - std::log10(y1)));
    }
    return y1 + (xv - x[j]) / (x[j + 1] - x[j]) * (y2 - y1);
}

// normalized_B_nu: source_type.f90:1088-1096
double normalized_B_nu(double nu, double T)
{
    const double a = 2.0 * HYP_H_CGS / HYP_C_CGS / HYP_C_CGS / HYP_STEF_BOLTZ * HYP_PI, b = HYP_H_CGS / HYP_K_CGS;
    const double T4 = T * T * T * T;
    return a * nu * nu * nu / (std::exp(b * nu / T) - 1.0) / T4;
}

// type_pdf set_pdf(x, y, log=.true.): normalised pdf, cdf and per-bin power-law
// index (+1) used by the device-side inversion.  Returns false if the integral
// vanishes.
bool build_log_pdf(const double *x, const double *y, int n, size_t stride,
                   std::vector<double> &cdf, std::vector<double> &bp1)
{
    std::vector<double> pdf(n);
    for (int i = 0; i < n; i++) pdf[i] = y[(size_t)i * stride];
    double norm = 0.0;
    for (int i = 0; i + 1 < n; i++) norm += seg_loglog(x[i], x[i + 1], pdf[i], pdf[i + 1]);
    if (!(norm > 0.0)) return false;
    for (int i = 0; i < n; i++) pdf[i] /= norm;
    cdf.assign(n, 0.0); bp1.assign(n, std::nan(""));
    for (int i = 1; i < n; i++) cdf[i] = cdf[i - 1] + seg_loglog(x[i - 1], x[i], pdf[i - 1], pdf[i]);
    double last = cdf[n - 1];
    for (int i = 0; i < n; i++) cdf[i] /= last;
    for (int i = 0; i + 1 < n; i++)
        if (pdf[i] > 0.0 && pdf[i + 1] > 0.0)
            bp1[i] = std::log10(pdf[i + 1] / pdf[i]) / std::log10(x[i + 1] / x[i]) + 1.0;
    return true;
}

// integral_loglog(x, y[, xmin, xmax]) of fortranlib (reference equivalent: hyperion/util/integrate.py
// integrate_loglog_subset): piecewise power laws, end points interpolated in log-log, limits
// clipped to the table.  `stride` lets y be a column of a row-major table.
double interp_seg_loglog(double x1, double x2, double y1, double y2, double x)
{
    if (y1 > 0.0 && y2 > 0.0) return y1 * std::pow(x / x1, std::log10(y2 / y1) / std::log10(x2 / x1));
    return y1 + (x - x1) / (x2 - x1) * (y2 - y1);
}

// interp1d_loglog of fortranlib at one abscissa inside [x[0], x[n-1]] (NaN outside)
double interp1d_loglog_host(const double *x, const double *y, int n, double xv)
{
    if (!(xv >= x[0] && xv <= x[n - 1])) return std::nan("");
    int j = (int)(std::upper_bound(x, x + n, xv) - x) - 1;
    if (j > n - 2) j = n - 2;
    const double y1 = y[j], y2 = y[j + 1];
    if (y1 > 0.0 && y2 > 0.0) {
        const double f = (std::log10(xv) - std::log10(x[j])) / (std::log10(x[j + 1]) - std::log10(x[j]));
        return std::pow(10.0, std::log10(y1) + f * (std::log10(y2) - std::log10(y1)));
    }
    return y1 + (xv - x[j]) / (x[j + 1] - x[j]) * (y2 - y1);
}

double integral_loglog_range(const double *x, const double *y, size_t stride, int n, double xmin, double xmax)
{
    if (xmin < x[0]) xmin = x[0];
    if (xmax > x[n - 1]) xmax = x[n - 1];
    if (!(xmax > xmin)) return 0.0;
    double s = 0.0;
    for (int i = 0; i + 1 < n; i++) {
        const double a = x[i], b = x[i + 1], ya0 = y[(size_t)i * stride], yb0 = y[(size_t)(i + 1) * stride];
        if (b <= xmin || a >= xmax) continue;
        const double xa = a < xmin ? xmin : a, xb = b > xmax ? xmax : b;
        const double ya = xa == a ? ya0 : interp_seg_loglog(a, b, ya0, yb0, xa);
        const double yb = xb == b ? yb0 : interp_seg_loglog(a, b, ya0, yb0, xb);
        s += seg_loglog(xa, xb, ya, yb);
    }
    return s;
}

double integral_loglog_all(const double *x, const double *y, size_t stride, int n)
{
    double s = 0.0;
    for (int i = 0; i + 1 < n; i++) s += seg_loglog(x[i], x[i + 1], y[(size_t)i * stride], y[(size_t)(i + 1) * stride]);
    return s;
}

double integral_linlog(const double *x, const double *y, int n)
{
    double s = 0.0;
    for (int i = 0; i + 1 < n; i++) {
        double y1 = y[i], y2 = y[i + 1], dx = x[i + 1] - x[i];
        if (y1 == y2) s += y1 * dx;
        else if (y1 > 0.0 && y2 > 0.0) s += (y2 - y1) * dx / std::log(y2 / y1);
    }
    return s;
}

double spacing(double x)
{
    x = std::fabs(x);
    if (x == 0.0) return DBL_MIN;
    return std::nextafter(x, INFINITY) - x;
}

struct DustOffsets {
    size_t nu, log10_nu, chi, albedo, log10_chi, log10_albedo, mu, P1, P2, P3, P4, P1_cdf, P2_cdf;
    size_t emiss_x, emiss_cdf, emiss_bp1, emiss_coarse, jnu_var, log10_jnu_var, mo_e, mo_chi_ross;
    size_t mo_kappa_planck, mo_chi_inv_planck, bnu_cdf, bnu_bp1, bnu_coarse, mono_prob;
    bool have_mo_e, have_mo_chi, have_mrw, have_pda;
};

struct SourceOffsets { size_t x, cdf, bp1; bool have; size_t points, point_cdf; bool have_points; size_t map_cdf; bool have_map; size_t spot_tab; bool have_spots; };
struct PeeledOffsets { size_t view, src_spec, dust_em, dust_chi, filt_off, filt_nu, filt_tr; };

}  // namespace

struct hyp_engine {
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    int n_cu = 0;

    DProblem hp;               // host copy of the device problem descriptor
    DProblem *d_problem = nullptr;
    double *d_blob = nullptr;
    OctCell *d_oct_cells = nullptr;
    int *d_oct_children = nullptr, *d_oct_neigh = nullptr;
    int oct_neighbours = 1;         // option: 0 = geo_advance climbs and descends as the reference does (for comparison)
    double *d_vor_sites = nullptr, *d_vor_volume = nullptr, *d_vor_bb = nullptr;
    unsigned int *d_mask_map = nullptr;
    bool ray_pending = false;
    AmrGrid *d_amr_grids = nullptr; int *d_amr_go = nullptr, *d_amr_cell_grid = nullptr; double *d_amr_walls = nullptr;
    int *d_vor_idx = nullptr, *d_vor_neigh = nullptr, *d_vor_seed = nullptr;
    VorWall *d_vor_walls = nullptr;
    DSource *d_sources = nullptr;
    DPeeled *d_peeled = nullptr;
    double *d_density = nullptr, *d_specific_energy = nullptr, *d_additional = nullptr;
    double *d_accum = nullptr;          // [copy0 | tail | pad][copy1]...
    size_t accum_stride = 0;            // doubles per copy slot
    int accum_copies_alloc = 0;
    int *d_jnu_id = nullptr;
    double *d_jnu_frac = nullptr;
    double *d_energy_abs_tot = nullptr;
    double *d_mrw_alpha = nullptr, *d_mrw_diff = nullptr, *d_mrw_kp = nullptr;   // per-iteration MRW tables
    double *d_scratch = nullptr;        // [n_dust*n_cells] layout conversions
    unsigned long long *d_counter = nullptr;
    int *d_err = nullptr;
    double *d_err_data = nullptr;
    double *d_img_accum = nullptr;      // all peeled cubes + tail
    size_t img_accum_n = 0;
    std::vector<size_t> sed_off, img_off, sed_n, img_n;
    std::vector<DPeeled> h_peeled;

    size_t n_cells = 0, n_elem = 0;
    int n_dust = 0;
    hyp_config cfg{};
    double energy_total = 0.0;
    bool lucy_pending = false, final_pending = false;
    uint64_t pending_packets = 0;
    float last_propagate_ms = 0.f, last_finish_ms = 0.f, ray_ms = 0.f;
    hyp_iter_stats last_stats{};

    // brick-tiled Lucy iteration (hyp_tiled.h)
    void *d_hot = nullptr, *d_cold = nullptr;
    int *d_slot_brick = nullptr, *d_order = nullptr;
    unsigned int *d_counts = nullptr, *d_cursor = nullptr;
    TileTask *d_tasks = nullptr;
    int *d_ilist = nullptr, *d_dlist = nullptr, *d_extra = nullptr;     // split schedule: per-task work lists
    TileCount *d_tcount = nullptr;
    // option (off): live timing of the dominant kernel for bench.py's roofline -- HIP events around every tile_walk launch on its
    // own stream and a device synchronisation at the end of the iteration; bench.py switches it on for one extra step
    int tile_time_walk = 0;
    std::vector<hipEvent_t> walk_events;
    double last_walk_ms = 0.0;
    int last_walk_launches = 0;
    TileCtl *d_ctl = nullptr;
    TileCtl *h_ctl = nullptr;           // pinned host copy
    int tile_slots_alloc = 0, tile_nd_alloc = 0;
    int lucy_mode = -1, tile_slots = 0 /* 0: 3 << 21 slots (octree and AMR: 3 << 22, configs[3] 119 -> 113 ms) */, tile_task = 0 /* 0: 8192 packets per task, 4096 on Voronoi grids */, tile_pools = 3, tile_drain = -1 /* -1: 1 000 000 packets in flight (profiles/r04_tiled_log.md) */, tile_park = 16;
    int last_lucy_mode = 0;
    hipStream_t pool_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_pool = nullptr;   // lucy_mode: -1 auto, 0 persistent, 1 brick-tiled
    int last_generations = 0;
    int tile_poll = 8;              // option: generations between two looks at the finished counter (a host sync)
    // cluster-tiled Voronoi schedule (hyp_vtile.h): tables built by build_vor_clusters()
    int vt_cells = 0;               // option: target cells per cluster (0: as many as the LDS budget allows)
    int tile_presort = 1;           // option: 1 = the Cartesian walk passes the kind of a packet's next interaction on with its slot (one species)
    int pt_vsplit = 1;              // option: spherical grids, 1 = two sort entries per brick (not yet interacted / the others)
    int pt_lds_kb = 128;            // option: LDS of the densities and accumulators of one polar-grid brick in KB (hyp_ptile.h)
    int vt_lds_kb = 156;            // option: LDS budget of one walk workgroup in KB (156: one 1024-thread workgroup per CU; 78: room for two of 512 threads)
    int vt_clusters = 0, vt_max_cells = 0, vt_built_for = -1;
    size_t vt_max_lds = 0;          // LDS of the largest cluster: tables + densities + accumulators
    int *d_vt_cluster = nullptr, *d_vt_members = nullptr, *d_vt_adj = nullptr;
    VtInfo *d_vt_info = nullptr; float4 *d_vt_blob = nullptr; VtGhost *d_vt_ghost = nullptr;
    std::vector<double> h_vor_sites; std::vector<int> h_vor_idx, h_vor_neigh;     // host copies for the cluster builder
    // cluster-tiled octree schedule (hyp_otile.h): tables built by build_oct_clusters()
    int ot_cells = 0;               // option: most cells per cluster (0: as many as the LDS budget allows)
    int ot_lds_kb = 156;            // option: LDS budget of one walk workgroup in KB (156: one 1024-thread workgroup per CU)
    int ot_clusters = 0, ot_max_cells = 0, ot_max_kids = 0, ot_built_for = -1;
    int *d_ot_cluster = nullptr, *d_ot_c0 = nullptr, *d_ot_nc = nullptr, *d_ot_kid_off = nullptr;
    OctCell *d_ot_rec = nullptr; short *d_ot_kid = nullptr, *d_ot_nb = nullptr;
    std::vector<OctCell> h_oct_cells; std::vector<int> h_oct_children, h_oct_neigh;      // host copies for the cluster builder
    // slab-tiled AMR schedule (hyp_atile.h): tables built by build_amr_slabs()
    int at_cells = 0;               // option: most cells per slab (0: as many as the LDS budget allows)
    int at_lds_kb = 156;            // option: LDS budget of one walk workgroup in KB (156: one 1024-thread workgroup per CU)
    int at_slabs_n = 0, at_max_cells = 0, at_max_go = 0, at_max_walls = 0, at_built_for = -1;
    AtSlab *d_at_slabs = nullptr; short *d_at_go = nullptr; int *d_at_grid_c0 = nullptr, *d_at_grid_nz = nullptr;      // (d_at_grid_nz: bricks along x, y per grid)
    std::vector<AmrGrid> h_amr_grids; std::vector<int> h_amr_go;

    // options
    int interact_threshold = 24, emit_threshold = 16, accum_copies = 16, blocks_per_cu = 0, chunk = 0;
    // the imaging iteration batches harder: the lanes that have just emitted walk to the observer (and, forced first
    // interaction, to the edge) together, so an emission of 48 lanes keeps 3 x the lanes busy in those walks than one of 16
    // (configs[3]: inline 76 -> 52 ms, deferred 60 -> 52 ms; profiles/r02_tiled_log.md).  -1 = measured optimum: interactions 16
    // deferred / 32 inline; emissions 48 deferred (16 on a Cartesian grid: its walks are cheap) / 32 inline.
    int final_interact_threshold = -1, final_emit_threshold = -1;

    // monochromatic final iteration
    std::vector<double> frequencies;
    double *d_mono_cdf = nullptr;       // [n_dust][n_cells]
    double *d_mono_mean = nullptr;      // [HYP_MAXD]
    DirectCol *d_direct = nullptr; size_t direct_cap = 0;       // direct light of the point sources, per (source, view): hyp_defer.h
    int direct_memo = 1, last_direct_memo = 0;                   // option direct_memo
    bool mono_pending = false;
    int gen_defer_opt = 1;              // option gen_defer: 1 = problems with spherical sources image on the deferred schedule, 0 = the general kernel
    int mono_defer_opt = 1;             // option mono_defer: 1 = monochromatic launches of plain problems on the deferred schedule, 0 = the general kernel
    int last_mono_deferred = 0;
    hyp_iter_stats mono_stats;

    // n_photons / frequency-resolved specific energy / PDA / convergence (hyp_epilogue.h)
    bool tile_unbuildable = false;  // the grid is beyond the limits of its tiled Lucy schedule's tables: auto mode stays on the persistent kernel
    bool plain_imaging = false;     // final_kernel<.., PLAIN>: see hyp_kernels.h
    bool inside_observers = false;  // a peeled group has an inside observer: deferred schedule or the general kernel, not the inline plain one
    bool ext_sources = false;       // point and external (box / sphere) sources with tabulated or blackbody spectra only: tile_emit_kernel<.., 2>
    bool mono_gen_defer = false;    // ... in a monochromatic run (final_defer_kernel<.., true, true, true>)
    bool gen_defer = false;         // sources with a surface: the imaging iteration on the deferred schedule (final_defer_kernel<.., GEN>, peel_kernel<.., GEN>)
    bool mono_defer = false;        // a monochromatic run of a problem that is plain otherwise: its launches run on the deferred schedule (final_defer_kernel<.., true, true>)
    bool lean_imaging = false;      // final_kernel<.., false, LEAN>: any sources, but no MRW / monochromatic / binned images / inside observers
    bool simple_sources = false;    // every source is a point source with a tabulated / blackbody spectrum (tile_emit_kernel<.., SIMPLE>)
    // deferred peel-off (hyp_defer.h): event buffer, control block, packets / id ranges carried between rounds
    int defer_peel = 1;             // option: 1 = deferred peel-off where plain_imaging holds (hyp_defer.h; large launches: propagation on the tiled schedule),
                                    //   2 = always on the tiled schedule where there is one, 3 = never, 0 = inline peel-off
    int last_tiled_imaging = 0;
    long long peel_events = 128ll << 20;    // option: capacity of the event buffer, in events (the ceiling: 8 per packet are asked for, and half as many
                                            // again and again while the allocation fails; 16 Mi until round 3: 1e8 packets then took 15 rounds)
    int peel_sort = 1;              // option: 1 = the peel kernel takes a round's events ordered by cell (hyp_defer.h: sorted peel-off)
    unsigned int *d_peel_order = nullptr, *d_peel_keys = nullptr, *d_peel_bins = nullptr;
    size_t peel_sort_cap = 0;
    int ff_prepass = 1;             // option: 1 = emission and the forced first interaction are made ahead of the rounds (hyp_defer.h: ff_walk_kernel)
    int last_ff_prepass = 0;        // whether the last imaging iteration did so
    void *d_ff = nullptr;           // EmitRec<n_dust>: one record per packet id of the launch
    size_t ff_cap = 0;              // bytes
    bool peel_events_exact = false;         // set by the option: use exactly that many (tests force many rounds with it)
    void *d_peel_events = nullptr, *d_peel_susp[2] = {nullptr, nullptr};
    unsigned long long *d_peel_ret[2] = {nullptr, nullptr};
    PeelCtl *d_peel_ctl = nullptr, *h_peel_ctl = nullptr;
    unsigned long long *h_peel_counter = nullptr;
    size_t peel_cap = 0, peel_event_bytes = 0, peel_lanes = 0;
    int last_defer_rounds = 0;
    unsigned long long last_defer_events = 0;
    bool count_photons = false, pda = false;
    int n_bins = 0, nj_max = 1;
    unsigned int *d_nphot = nullptr;      // [n_cells]
    unsigned long long *d_visit = nullptr;      // per-lane visited sets of count_photon, [visit_lanes][HYP_VISIT_SLOTS]
    size_t visit_lanes = 0;
    int *d_nphot_inexact = nullptr;
    int nphot_inexact = 0;          // a packet overflowed its visited set in the last counting iteration
    size_t ext_nphot = 0, ext_spec = 0, block_doubles = 0;        // offsets (doubles) of the extensions in the accumulator block; its length
    double *d_log_edges = nullptr, *d_bin_frac = nullptr, *d_spec = nullptr;
    std::vector<double> spectrum_edges;
    unsigned char *d_pda_mask = nullptr;
    unsigned int *d_pda_cells = nullptr, *d_pda_hp = nullptr;    // hp: [count | offsets (+1) | cursor], n_hp + 1 entries each
    double *d_pda_emean = nullptr, *d_pda_coef = nullptr;
    size_t pda_coef_alloc = 0;
    unsigned int *d_pda_id = nullptr;                            // [n_cells] index of a cell in the PDA list (0xffffffff: not one)
    double *d_pda_a = nullptr, *d_pda_b = nullptr, *d_pda_f = nullptr;   // dense system of the Gauss pivot branch
    size_t pda_dense_alloc = 0;
    PdaCtl *d_pda_ctl = nullptr;
    int pda_last_cells = 0, pda_last_outer = 0, pda_last_sweeps = 0;
    double *d_prev_se = nullptr, *d_ratio = nullptr;
    ConvCtl *d_conv_ctl = nullptr;
    bool have_prev = false;

    int set_error(const std::string &m) { err = m; return 1; }
};

namespace {

int set_error(const std::string &m) { g_error = m; return 1; }

template <typename T>
void free_dev(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

size_t lds_bytes(const DProblem &P) { return P.grid_type != 1 ? 0 : sizeof(double) * 2 * ((size_t)P.n1 + P.n2 + P.n3 + 3); }

RayKernel pick_ray_kernel(int nd, int grid_type)
{
#ifdef HYP_VARIANT_GEOM   // tuning builds (tools/variants.py) link one geometry unit only
    return pick_ray_kernel_g<HYP_VARIANT_GEOM>(nd);
#endif
    switch (grid_type) {
    case 2: return pick_ray_kernel_g<GEOM_OCT>(nd);
    case 3: return pick_ray_kernel_g<GEOM_VOR>(nd);
    case 4: return pick_ray_kernel_g<GEOM_AMR>(nd);
    case 5: return pick_ray_kernel_g<GEOM_SPH>(nd);
    case 6: return pick_ray_kernel_g<GEOM_CYL>(nd);
    default: return pick_ray_kernel_g<GEOM_CAR>(nd);
    }
}

LucyKernel pick_lucy_kernel(int nd, int grid_type)
{
#ifdef HYP_VARIANT_GEOM   // tuning builds (tools/variants.py) link one geometry unit only
    return pick_lucy_kernel_g<HYP_VARIANT_GEOM>(nd);
#endif
    switch (grid_type) {
    case 2: return pick_lucy_kernel_g<GEOM_OCT>(nd);
    case 3: return pick_lucy_kernel_g<GEOM_VOR>(nd);
    case 4: return pick_lucy_kernel_g<GEOM_AMR>(nd);
    case 5: return pick_lucy_kernel_g<GEOM_SPH>(nd);
    case 6: return pick_lucy_kernel_g<GEOM_CYL>(nd);
    default: return pick_lucy_kernel_g<GEOM_CAR>(nd);
    }
}

DeferKernels pick_defer_kernels(int nd, int grid_type)
{
#ifdef HYP_VARIANT_GEOM
    return pick_defer_kernels_g<HYP_VARIANT_GEOM>(nd);
#endif
    switch (grid_type) {
    case 2: return pick_defer_kernels_g<GEOM_OCT>(nd);
    case 3: return pick_defer_kernels_g<GEOM_VOR>(nd);
    case 4: return pick_defer_kernels_g<GEOM_AMR>(nd);
    case 5: return pick_defer_kernels_g<GEOM_SPH>(nd);
    case 6: return pick_defer_kernels_g<GEOM_CYL>(nd);
    default: return pick_defer_kernels_g<GEOM_CAR>(nd);
    }
}

LucyKernel pick_final_kernel(int nd, int grid_type, int mode)
{
#define PICK_FINAL(G) (mode == 0 || nd > 4 ? pick_final_kernel_g<G>(nd) : pick_final_special_g<G>(nd, mode))
#ifdef HYP_VARIANT_GEOM
    return PICK_FINAL(HYP_VARIANT_GEOM);
#endif
    switch (grid_type) {
    case 2: return PICK_FINAL(GEOM_OCT);
    case 3: return PICK_FINAL(GEOM_VOR);
    case 4: return PICK_FINAL(GEOM_AMR);
    case 5: return PICK_FINAL(GEOM_SPH);
    case 6: return PICK_FINAL(GEOM_CYL);
    default: return PICK_FINAL(GEOM_CAR);
    }
#undef PICK_FINAL
}

}  // namespace


// ---- brick- / cluster-tiled Lucy iteration (Cartesian and Voronoi grids): host-driven generations ----
namespace {

TileKernels pick_tile_kernels(int nd, int grid_type)
{
#ifdef HYP_VARIANT_GEOM
    return pick_tile_kernels_g<HYP_VARIANT_GEOM>(nd);
#endif
    switch (grid_type) {
    case 1: return pick_tile_kernels_g<GEOM_CAR>(nd);
    case 2: return pick_tile_kernels_g<GEOM_OCT>(nd);
    case 4: return pick_tile_kernels_g<GEOM_AMR>(nd);
    case 3: return pick_tile_kernels_g<GEOM_VOR>(nd);
    case 5: return pick_tile_kernels_g<GEOM_SPH>(nd);
    case 6: return pick_tile_kernels_g<GEOM_CYL>(nd);
    default: { TileKernels k; memset(&k, 0, sizeof k); return k; }
    }
}

void tile_shape(int nd, int &x, int &y, int &z)
{
    switch (nd) {
    case 1: x = TileShape<1>::X; y = TileShape<1>::Y; z = TileShape<1>::Z; break;
    case 2: x = TileShape<2>::X; y = TileShape<2>::Y; z = TileShape<2>::Z; break;
    case 3: x = TileShape<3>::X; y = TileShape<3>::Y; z = TileShape<3>::Z; break;
    default: x = TileShape<4>::X; y = TileShape<4>::Y; z = TileShape<4>::Z; break;
    }
}

int tile_bricks(const DProblem &P, int nd)
{
    int x, y, z;
    tile_shape(nd, x, y, z);
    return ((P.n1 + x - 1) / x) * ((P.n2 + y - 1) / y) * ((P.n3 + z - 1) / z);
}

// Bricks of a polar grid (hyp_ptile.h): boxes of (r, theta, phi) / (w, z, phi) indices whose densities and accumulators fit `cells`
// cells of LDS.  Packets move mostly along r, so the brick is long in the first index: at most 8 cells in phi, 32 in theta / z,
// and what is left of the budget in r; theta / z shrink before r falls below 16 cells.
void polar_tile_shape(const DProblem &P, int nd, int lds_kb, int &x, int &y, int &z)
{
    const long long cells = std::max<long long>(64, (long long)lds_kb * 1024 / (16ll * nd));
    z = (int)std::min<long long>(P.n3, 8);
    y = (int)std::min<long long>(P.n2, 32);
    while ((long long)y * z * 16 > cells && y > 1) y = (y + 1) / 2;
    while ((long long)y * z * 16 > cells && z > 1) z = (z + 1) / 2;
    x = (int)std::max<long long>(1, std::min<long long>(P.n1, cells / ((long long)y * z)));
}

// Number of bricks of a polar grid on the tiled schedule, or -1 when the grid has no such schedule: more bricks than the sort's
// tables hold (HYP_TILE_MAX_BRICKS), or a brick beyond the LDS of a CU (pt_lds_kb is an option; 160 KB per CU on gfx950)
long long polar_tile_bricks(const DProblem &P, int nd, int lds_kb)
{
    int bx, by, bz;
    polar_tile_shape(P, nd, lds_kb, bx, by, bz);
    const long long nb = (long long)((P.n1 + bx - 1) / bx) * ((P.n2 + by - 1) / by) * ((P.n3 + bz - 1) / bz);
    const size_t lds = sizeof(double) * 2 * (size_t)bx * by * bz * nd;
    return (nb <= HYP_TILE_MAX_BRICKS && lds + 4096 <= 160u * 1024u) ? nb : -1;      // (4 KB: the kernel's static LDS -- counters, brick histogram)
}

// LDS of one AMR brick (hyp_atile.h): n cells, g goto entries (16 bits), w walls
size_t amr_slab_lds(size_t n, size_t g, size_t w, int nd) { return sizeof(double) * (2 * n * nd + w) + sizeof(short) * g + 16; }

// LDS of one octree cluster (hyp_otile.h): n cells of which k are refined
size_t oct_cluster_lds(size_t n, size_t k, int nd) { return (sizeof(OctCell) + sizeof(double) * 2 * nd + sizeof(short) * 6) * n + sizeof(short) * 8 * k + 16; }

// LDS of one walk workgroup
size_t tile_walk_lds(hyp_handle h, const TileKernels &K, const TileGeom &T)
{
    if (h->hp.grid_type == 3)       // cluster: its tables (VtInfo) + densities + accumulators
        return h->vt_max_lds;
    if (h->hp.grid_type == 5 || h->hp.grid_type == 6)      // polar brick: densities + accumulators
        return sizeof(double) * 2 * (size_t)T.bx * T.by * T.bz * K.nd;
    if (h->hp.grid_type == 4)       // slab: densities + accumulators + walls + goto slice
        return amr_slab_lds((size_t)T.bx, (size_t)T.by, (size_t)T.bz, K.nd);
    if (h->hp.grid_type == 2)       // cluster: cell records + densities + accumulators + children of the refined cells + neighbour table
        return oct_cluster_lds((size_t)T.bx, (size_t)T.by, K.nd);
    return lds_bytes(h->hp) + sizeof(double) * 2 * (size_t)K.bx * K.by * K.bz * K.nd;      // walls + densities + accumulators of the brick
}

// `img`: the imaging iteration on the tiled schedule -- the event buffer the IMG kernels append to; `flush` empties it (sort +
// peel_kernel) and is called with every pool's stream idle, when the buffer could overflow before the next look and at the end
int run_tiled_generations(hyp_handle h, const TileKernels &K, const TileGeom &T0, uint64_t n_local, int n_pools, const DeferBuf *img = nullptr,
                          const std::function<int()> &flush = std::function<int()>())
{
    DeferBuf no_events;
    std::memset(&no_events, 0, sizeof no_events);
    const size_t lds_w = lds_bytes(h->hp);
    const size_t lds_int = lds_w;
    const TileWalkK walk_k = K.walk;
    const size_t lds_walk = tile_walk_lds(h, K, T0);
    const int grid_s = (T0.n_slots + 256 * HYP_SORT_PER_THREAD - 1) / (256 * HYP_SORT_PER_THREAD);
    const int grid_w = T0.n_slots / T0.task_size + T0.n_bricks + 1;
    // tile_interact: one workgroup per HYP_INTERACT_CHUNK entries of the pool's list (+ one for the extra list); tile_emit:
    // one per 256 free slots.  Workgroups beyond the lists' lengths (known on the device only) leave at once.
    const int grid_i = (T0.n_slots + HYP_INTERACT_CHUNK - 1) / HYP_INTERACT_CHUNK + 1;
    const int grid_e = (T0.n_slots + 255) / 256;
    if (hipFuncSetAttribute((const void *)walk_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_walk) != hipSuccess)
        return h->set_error("cannot reserve LDS for the tiled walk kernel");
    const size_t hot_sz = K.hot_bytes, cold_sz = K.cold_bytes;
    const int ri = h->hp.any_intersect ? 1 : 0, mi = h->hp.mrw ? 1 : 0;
    // Each pool of slots runs its own prepare -> sort -> walk sequence on its own stream, so the
    // latency-bound prepare of one pool overlaps the walk of the other.  The pools share only the
    // packet-id dispenser, the finished counter and the (atomic) accumulators.
    const size_t tasks_cap = (size_t)T0.n_slots / 256 + HYP_TILE_MAX_BRICKS + 2;
    int gen = 0, next_check = h->tile_poll;
    // Every packet in a slot makes one interaction per generation and is killed at n_inter_max of them (iter_lucy.f90:186-190,
    // iter_final.f90:255-259), and a slot takes a new id when its packet has ended: the generations are bounded by the interactions
    // of the packets that pass through one slot.  A sanity bound, not a schedule: a run that needs 1e6 generations is slow here
    // (launch-bound generations for a handful of packets; the Lucy iteration drains them in one kernel, the imaging iteration
    // has no such kernel) but it ends with the reference's result, not with an error.
    const long long per_slot = (long long)(n_local / ((uint64_t)T0.n_slots * (uint64_t)n_pools)) + 2;
    const long long max_gen_ll = std::max<long long>(200000, ((long long)h->cfg.n_inter_max + 2) * per_slot + 16);
    const int max_gen = (int)std::min<long long>(max_gen_ll, 2000000000ll);
    size_t n_timed = 0;
    // imaging: every generation can add at most one event per slot (plus the padding of the interaction chunks)
    const unsigned long long ev_per_gen = (unsigned long long)n_pools * ((unsigned long long)T0.n_slots + 64ull * (unsigned long long)grid_i);
    if (img) next_check = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((unsigned long long)h->tile_poll, img->cap / ev_per_gen));
    for (;; gen++) {
        for (int pool = 0; pool < n_pools; pool++) {
            TileGeom T = T0; T.pool = pool;
            hipStream_t st = pool == 0 ? h->stream : h->pool_stream[pool];
            void *hot = (char *)h->d_hot + hot_sz * (size_t)pool * T.n_slots;
            void *cold = (char *)h->d_cold + cold_sz * (size_t)pool * T.n_slots;
            int *slot_brick = h->d_slot_brick + (size_t)pool * T.n_slots;
            int *order = h->d_order + (size_t)pool * T.n_slots;
            TileTask *tasks = h->d_tasks + pool * tasks_cap;
            // counts and cursors by generation parity (tile_sort_kernel); `counts` = what this generation's sort reads
            const size_t par_off = (size_t)HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS;
            const int gp = gen & 1, gn = (gen + 1) & 1;
            unsigned *counts = h->d_counts + gp * par_off + pool * HYP_TILE_MAX_BRICKS, *cursor = h->d_cursor + gp * par_off + pool * HYP_TILE_MAX_BRICKS;
            unsigned *counts_next = h->d_counts + gn * par_off + pool * HYP_TILE_MAX_BRICKS, *cursor_next = h->d_cursor + gn * par_off + pool * HYP_TILE_MAX_BRICKS;
            int *ilist = h->d_ilist + (size_t)pool * 2 * T.n_slots, *dlist = h->d_dlist + (size_t)pool * 2 * T.n_slots;      // [staging | pool-wide list]
            int *extra = h->d_extra + (size_t)pool * 3 * HYP_TILE_EXTRA;
            T.gen = gen;
            TileCount *tcount = h->d_tcount + pool * tasks_cap;
            // walk (previous generation) left per-task lists: interactions, then emission into the freed slots
            if (gen == 0) tile_init_kernel<<<(T.n_slots + 255) / 256, 256, 0, st>>>(T, h->d_ctl, tasks, tcount, dlist);
            else
                (img ? K.interact_img : K.interact[ri][mi])<<<grid_i, 256, lds_int, st>>>(h->d_problem, T, h->d_ctl, hot, cold, slot_brick, tasks, ilist, dlist,
                                                                                     tcount, counts, extra, img ? *img : no_events);
            (img ? K.emit_img : h->simple_sources && K.emit_simple ? K.emit_simple : h->ext_sources && K.emit_ext ? K.emit_ext : K.emit)<<<grid_e, 256, lds_w, st>>>(h->d_problem, T, h->d_ctl, hot, cold, slot_brick, tasks, dlist, tcount, counts, extra, img ? *img : no_events);
            tile_sort_kernel<<<grid_s, 256, sizeof(unsigned) * (2 * (size_t)T.n_bricks + 512), st>>>(T, slot_brick, counts, counts_next, cursor, cursor_next, order, tasks, h->d_ctl);
            const bool timed = h->tile_time_walk && n_timed + 2 <= 16384;
            if (timed) {
                while (h->walk_events.size() < n_timed + 2) {
                    hipEvent_t e = nullptr;
                    if (hipEventCreate(&e) != hipSuccess) return h->set_error("cannot create a timing event");
                    h->walk_events.push_back(e);
                }
                (void)hipEventRecord(h->walk_events[n_timed], st);
            }
            walk_k<<<grid_w, K.walk_threads, lds_walk, st>>>(h->d_problem, T, h->d_ctl, hot, cold, order, tasks, slot_brick, ilist, dlist, tcount, counts_next);
            if (timed) { (void)hipEventRecord(h->walk_events[n_timed + 1], st); n_timed += 2; }
        }
        if (gen + 1 >= next_check || gen > max_gen) {
            next_check = gen + 1 + h->tile_poll;
            hipError_t e = hipMemcpyAsync(h->h_ctl, h->d_ctl, sizeof(TileCtl), hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) return h->set_error(std::string("tiled generation failed: ") + hipGetErrorString(e));
            if (img) {
                // how many more generations are sure to fit the event buffer decides when to look again; with fewer than two (or at
                // the end) the buffer is emptied: sort + peel, every pool's stream idle
                for (int pool = 1; pool < n_pools; pool++)
                    if (hipStreamSynchronize(h->pool_stream[pool]) != hipSuccess) return h->set_error("tiled imaging generation failed");
                unsigned long long reserved = 0;
                (void)hipMemcpy(&reserved, &img->ctl->reserved, sizeof reserved, hipMemcpyDeviceToHost);
                if (reserved > img->cap) return h->set_error("tiled imaging: the event buffer overflowed");
                unsigned long long room = (img->cap - reserved) / ev_per_gen;
                if (h->h_ctl->n_finished >= n_local || room < 2) {
                    if (flush()) return 1;
                    room = img->cap / ev_per_gen;
                }
                next_check = gen + 1 + (int)std::max<unsigned long long>(1, std::min<unsigned long long>((unsigned long long)h->tile_poll, room));
            }
            if (h->h_ctl->n_finished >= n_local) break;
            if (gen > max_gen) return h->set_error(img ? "tiled imaging iteration did not terminate" : "tiled Lucy iteration did not terminate");
            // few packets left and no ids to hand out: finish them in one launch
            const uint64_t in_flight = n_local - h->h_ctl->n_finished;
            const uint64_t drain_at = h->tile_drain >= 0 ? (uint64_t)h->tile_drain : 1000000ull;      // (flat between 4e5 and 1.5e6 since the drain takes its packets from one list, profiles/r04_tiled_log.md)
            if (!img && h->h_ctl->next_id >= h->h_ctl->end_id && in_flight <= drain_at) {      // (the drain kernel deposits: Lucy only)
                for (int pool = 1; pool < n_pools; pool++) {
                    (void)hipEventRecord(h->ev_pool, h->pool_stream[pool]);
                    (void)hipStreamWaitEvent(h->stream, h->ev_pool, 0);
                }
                TileGeom T = T0; T.n_slots = T0.n_slots * n_pools;
                // the slots that still hold a packet as one list (in the sort's order[] array: nobody sorts any more), then the drain
                (void)hipMemsetAsync(&h->d_ctl->n_live, 0, 2 * sizeof(unsigned int), h->stream);
                tile_live_kernel<<<(T.n_slots + HYP_PREP_CHUNK - 1) / HYP_PREP_CHUNK, 256, 0, h->stream>>>(T, h->d_slot_brick, h->d_order, h->d_ctl);
                T.drain_list = h->d_order;
                const int grid_d = (int)std::min<uint64_t>((in_flight + 255) / 256 + 1, (uint64_t)h->n_cu * 8);
                K.drain[ri][mi]<<<grid_d, 256, lds_w, h->stream>>>(h->d_problem, T, h->d_ctl, h->d_hot, h->d_cold, h->d_slot_brick);
                e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) return h->set_error(std::string("tiled drain failed: ") + hipGetErrorString(e));
                break;
            }
            int err = 0;
            (void)hipMemcpy(&err, h->d_err, sizeof(int), hipMemcpyDeviceToHost);
            if (err) break;
        }
    }
    h->last_generations = gen + 1;
    h->last_walk_ms = 0.0; h->last_walk_launches = 0;
    if (n_timed) {
        (void)hipDeviceSynchronize();
        for (size_t i = 0; i + 1 < n_timed; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->walk_events[i], h->walk_events[i + 1]) == hipSuccess) { h->last_walk_ms += ms; h->last_walk_launches++; }
        }
    }
#ifdef HYP_TILE_STATS
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h->h_ctl, h->d_ctl, sizeof(TileCtl), hipMemcpyDeviceToHost);
    {
        const unsigned long long *d = h->h_ctl->dbg;
        fprintf(stderr, "tile stats: generations %d, outer loops %llu, wave-steps %llu, lane-steps %llu (lane utilisation %.3f), waves %llu, "
                        "tasks %llu (mean %.0f packets), steps per outer loop %.2f\n", gen + 1, d[0], d[1], d[2], (double)d[2] / (64.0 * d[1]),
                d[3], d[4], (double)d[5] / d[4], (double)d[1] / d[0]);
        fprintf(stderr, "tile stats: service phases %llu (%.2f per outer loop), wave clocks in the service phase %.3f of the loop's (%.0f clocks per service phase, %.0f per outer loop)\n",
                d[7], (double)d[7] / d[0], (double)d[6] / d[8], (double)d[6] / d[7], (double)d[8] / d[0]);
        if (d[10]) fprintf(stderr, "tile stats: service phase = check + write-back %.3f (%.1f lanes), claim %.3f (%.1f lanes) of its clocks\n", (double)d[10] / d[6], (double)d[12] / d[7],
                           (double)d[11] / d[6], (double)d[13] / d[7]);
        if (d[15]) fprintf(stderr, "tile stats: propagation check / general wall search ran in %.3f of the service phases and took %.3f of their clocks\n", (double)d[14] / d[7], (double)d[15] / d[6]);
        fprintf(stderr, "tile stats: wave clocks waiting at the end of the task for the workgroup's last wave %.3f of the loop's\n", (double)d[9] / d[8]);
    }
#endif
    return 0;
}

// One iteration on the slot-pool schedule: the Lucy iteration (img == nullptr; `iter_tag` = the iteration number), or the imaging
// iteration's propagation half with its events appended to *img (run_tiled_imaging below)
int launch_tiled(hyp_handle h, uint64_t first_id, uint64_t n_local, uint32_t iter_tag, const DeferBuf *img = nullptr,
                 const std::function<int()> &flush = std::function<int()>())
{
    const DProblem &P = h->hp;
    const int nd = h->n_dust;
    const TileKernels K = pick_tile_kernels(nd, P.grid_type);
    if (!K.walk) return h->set_error("no tiled schedule for this grid geometry");
    TileGeom T;
    memset(&T, 0, sizeof T);
    if (P.grid_type == 3) {
        T.bx = h->vt_max_cells; T.by = 1; T.bz = 1;
        T.nbx = T.n_bricks = h->vt_clusters; T.nby = T.nbz = 1;
    } else if (P.grid_type == 4) {
        T.bx = h->at_max_cells; T.by = h->at_max_go; T.bz = h->at_max_walls;
        T.nbx = T.n_bricks = h->at_slabs_n; T.nby = T.nbz = 1;
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    } else if (P.grid_type == 2) {
        T.bx = h->ot_max_cells; T.by = h->ot_max_kids; T.bz = 1;
        T.nbx = T.n_bricks = h->ot_clusters; T.nby = T.nbz = 1;
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    } else if (P.grid_type == 5 || P.grid_type == 6) {
        polar_tile_shape(P, nd, h->pt_lds_kb, T.bx, T.by, T.bz);
        T.nbx = (P.n1 + T.bx - 1) / T.bx; T.nby = (P.n2 + T.by - 1) / T.by; T.nbz = (P.n3 + T.bz - 1) / T.bz;
        T.n_bricks = T.nbx * T.nby * T.nbz;
        // spherical grids: packets that have not interacted yet (radial for a central source: no cone wall is ever in reach, hyp_polar.h:
        // sph_cone_out_of_reach) sorted apart from the others, so that their waves skip the cone quadratics
        if (P.grid_type == 5 && h->pt_vsplit && 2 * T.n_bricks <= HYP_TILE_MAX_BRICKS) { T.vsplit = 2; T.n_bricks *= 2; }
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    } else {
        tile_shape(nd, T.bx, T.by, T.bz);
        T.nbx = (P.n1 + T.bx - 1) / T.bx; T.nby = (P.n2 + T.by - 1) / T.by; T.nbz = (P.n3 + T.bz - 1) / T.bz;
        T.n_bricks = T.nbx * T.nby * T.nbz;
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    }
    // the sort's tables (d_counts / d_cursor, tile_sort_kernel's LDS) hold HYP_TILE_MAX_BRICKS entries per pool
    if (T.n_bricks < 1 || T.n_bricks > HYP_TILE_MAX_BRICKS) return h->set_error("grid has too many bricks for the tiled schedule");
    int n_pools = std::max(1, std::min(h->tile_pools, HYP_TILE_MAX_POOLS));
    const long long want_slots = h->tile_slots > 0 ? h->tile_slots : ((P.grid_type == 2 || P.grid_type == 4) ? 3ll << 22 : 3ll << 21);
    long long slots = std::min<long long>(want_slots, (long long)n_local);
    if (slots < 65536) n_pools = 1;
    slots = (((slots + n_pools - 1) / n_pools + 255) / 256) * 256;       // per pool
    T.n_slots = (int)slots;
    const size_t all_slots = (size_t)slots * n_pools;
    T.task_size = h->tile_task <= 0 ? 8192 : h->tile_task < 256 ? 256 : h->tile_task;
    T.iter_tag = iter_tag; T.pool = 0; T.park = h->tile_park;
    T.imaging = img ? 1 : 0;
    const size_t hot_sz = K.hot_bytes, cold_sz = K.cold_bytes;
    if (all_slots > (size_t)h->tile_slots_alloc || nd != h->tile_nd_alloc) {
        free_dev(h->d_hot); free_dev(h->d_cold); free_dev(h->d_slot_brick); free_dev(h->d_order); free_dev(h->d_tasks);
        free_dev(h->d_ilist); free_dev(h->d_dlist); free_dev(h->d_tcount); free_dev(h->d_extra);
        const size_t n_tasks_max = HYP_TILE_MAX_POOLS * ((size_t)all_slots / 256 + HYP_TILE_MAX_BRICKS + 2);
        if (hipMalloc(&h->d_hot, hot_sz * all_slots) != hipSuccess || hipMalloc(&h->d_cold, cold_sz * all_slots) != hipSuccess ||
            hipMalloc(&h->d_slot_brick, sizeof(int) * all_slots) != hipSuccess || hipMalloc(&h->d_order, sizeof(int) * all_slots) != hipSuccess ||
            hipMalloc(&h->d_tasks, sizeof(TileTask) * n_tasks_max) != hipSuccess ||
            hipMalloc(&h->d_ilist, sizeof(int) * 2 * all_slots) != hipSuccess || hipMalloc(&h->d_dlist, sizeof(int) * 2 * all_slots) != hipSuccess ||
            hipMalloc(&h->d_tcount, sizeof(TileCount) * n_tasks_max) != hipSuccess ||
            hipMalloc(&h->d_extra, sizeof(int) * 3 * HYP_TILE_EXTRA * HYP_TILE_MAX_POOLS) != hipSuccess)
            return h->set_error("cannot allocate the packet pool of the tiled Lucy iteration");
        h->tile_slots_alloc = all_slots; h->tile_nd_alloc = nd;
    }
    if (!h->d_counts) {
        const size_t nb = sizeof(unsigned) * HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS;
        if (hipMalloc(&h->d_counts, 2 * nb) != hipSuccess || hipMalloc(&h->d_cursor, 2 * nb) != hipSuccess ||
            hipMalloc(&h->d_ctl, sizeof(TileCtl)) != hipSuccess || hipHostMalloc(&h->h_ctl, sizeof(TileCtl)) != hipSuccess)
            return h->set_error("cannot allocate the control blocks of the tiled Lucy iteration");
        if (hipEventCreateWithFlags(&h->ev_pool, hipEventDisableTiming) != hipSuccess)
            return h->set_error("cannot create the pool event of the tiled Lucy iteration");
    }
    for (int pool = 1; pool < n_pools; pool++)
        if (!h->pool_stream[pool] && hipStreamCreateWithFlags(&h->pool_stream[pool], hipStreamNonBlocking) != hipSuccess)
            return h->set_error("cannot create a stream for the tiled Lucy iteration");
    {
        static bool sort_attr = false;       // (n_bricks near HYP_TILE_MAX_BRICKS: more than the default 64 KB of dynamic LDS)
        if (!sort_attr) { (void)hipFuncSetAttribute((const void *)tile_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned) * (2 * HYP_TILE_MAX_BRICKS + 512))); sort_attr = true; }
    }
    TileCtl c0; memset(&c0, 0, sizeof(c0));
    c0.next_id = first_id; c0.end_id = first_id + n_local; c0.first_id = first_id;
    if (!img) (void)hipEventRecord(h->ev0, h->stream);        // (the imaging iteration's clock starts before its pre-pass)
    (void)hipMemsetAsync(h->d_hot, 0, hot_sz * all_slots, h->stream);          // state 0 = TS_DEAD
    (void)hipMemsetD32Async((hipDeviceptr_t)h->d_slot_brick, TILE_NEEDS_PREPARE, all_slots, h->stream);   // every slot is free
    (void)hipMemsetAsync(h->d_counts, 0, 2 * sizeof(unsigned) * HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS, h->stream);
    (void)hipMemsetAsync(h->d_cursor, 0, 2 * sizeof(unsigned) * HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS, h->stream);
    (void)hipMemcpyAsync(h->d_ctl, &c0, sizeof(c0), hipMemcpyHostToDevice, h->stream);
    (void)hipStreamSynchronize(h->stream);      // c0 lives on this stack frame; the other pools start after the resets
    const int rc = run_tiled_generations(h, K, T, n_local, n_pools, img, flush);
    for (int pool = 1; pool < n_pools; pool++) {      // join the other pools into the engine's stream
        (void)hipEventRecord(h->ev_pool, h->pool_stream[pool]);
        (void)hipStreamWaitEvent(h->stream, h->ev_pool, 0);
    }
    (void)hipEventRecord(h->ev1, h->stream);
    return rc;
}

}  // namespace

extern "C" {

int hyp_abi_version(void) { return HYP_ABI_VERSION; }

namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } }
    void i64(int64_t v) { bytes(&v, sizeof v); }
    void f64(double v) { if (v == 0.0) v = 0.0; bytes(&v, sizeof v); }      // -0.0 and 0.0 alike
    void arr(const double *p, size_t n) { i64(p ? (int64_t)n : -1); if (p) for (size_t i = 0; i < n; i++) f64(p[i]); }
    void arr32(const int32_t *p, size_t n) { i64(p ? (int64_t)n : -1); if (p) bytes(p, n * sizeof(int32_t)); }
};
void digest_peeled(Fnv &f, const hyp_peeled_desc &d, bool binned)
{
    for (int64_t v : {(int64_t)(binned ? 0 : d.n_view), (int64_t)d.inside_observer, (int64_t)d.ignore_optical_depth, (int64_t)d.compute_image, (int64_t)d.compute_sed,
                      (int64_t)d.n_x, (int64_t)d.n_y, (int64_t)d.n_ap, (int64_t)d.n_nu, (int64_t)d.track_origin, (int64_t)d.track_n_scat, (int64_t)d.uncertainties,
                      (int64_t)d.compute_stokes, (int64_t)d.use_filters}) f.i64(v);
    if (d.compute_image) for (double v : {d.x_min, d.x_max, d.y_min, d.y_max}) f.f64(v);
    if (d.compute_sed) for (double v : {d.ap_min, d.ap_max}) f.f64(v);
    for (double v : {d.nu_min, d.nu_max, d.d_min, d.d_max}) f.f64(v);
    if (d.inside_observer) for (double v : d.peeloff_origin) f.f64(v);
    if (!binned) { f.arr(d.theta, (size_t)d.n_view); f.arr(d.phi, (size_t)d.n_view); }
    if (d.use_filters) {
        f.arr32(d.filt_n, (size_t)d.n_nu);
        size_t tot = 0;
        for (int i = 0; i < d.n_nu && d.filt_n; i++) tot += (size_t)d.filt_n[i];
        f.arr(d.filt_nu, tot); f.arr(d.filt_tr, tot);
    }
}
}  // namespace

int hyp_problem_digest(const hyp_problem *pr, uint64_t out[4])
{
    if (!pr || !out) return 1;
    const hyp_grid_desc &g = pr->grid;
    if (pr->n_dust < 0 || pr->n_sources < 0 || pr->n_peeled < 0 || g.n_cells < 0) return 1;
    size_t nc = (size_t)g.n_cells;
    Fnv a;
    a.i64(g.type);
    if (g.type == 1 || g.type == 5 || g.type == 6) {
        nc = (size_t)g.n1 * g.n2 * g.n3;
        a.i64(g.n1); a.i64(g.n2); a.i64(g.n3);
        a.arr(g.w1, (size_t)g.n1 + 1); a.arr(g.w2, (size_t)g.n2 + 1); a.arr(g.w3, (size_t)g.n3 + 1);
    } else if (g.type == 2) {
        a.arr32(g.refined, nc);
        for (double v : g.oct_center) a.f64(v);
        for (double v : g.oct_half) a.f64(v);
    } else if (g.type == 3) {
        a.arr(g.vor_sites, 3 * nc); a.arr(g.vor_volume, nc); a.arr32(g.vor_idx, nc + 1);
        a.arr32(g.vor_neighs, g.vor_idx ? (size_t)g.vor_idx[nc] : 0);
        for (double v : g.vor_box) a.f64(v);
        a.arr(g.vor_bb, 6 * nc);
    } else if (g.type == 4) {
        a.i64(g.n_amr_levels); a.i64(g.n_amr_grids);
        a.arr32(g.amr_level, (size_t)g.n_amr_grids); a.arr32(g.amr_n, 3 * (size_t)g.n_amr_grids); a.arr(g.amr_bounds, 6 * (size_t)g.n_amr_grids);
        nc = 0;
        for (int k = 0; k < g.n_amr_grids && g.amr_n; k++) nc += (size_t)g.amr_n[3 * k] * g.amr_n[3 * k + 1] * g.amr_n[3 * k + 2];
    } else return 1;
    a.arr(pr->density, nc * (size_t)pr->n_dust);
    a.arr(pr->specific_energy, nc * (size_t)pr->n_dust);
    out[0] = a.h;
    Fnv d;
    d.i64(pr->n_dust);
    for (int i = 0; i < pr->n_dust; i++) {
        const hyp_dust_desc &D = pr->dust[i];
        for (int64_t v : {(int64_t)D.n_nu, (int64_t)D.n_mu, (int64_t)D.n_jnu, (int64_t)D.n_enu, (int64_t)D.n_e, (int64_t)D.sublimation_mode, (int64_t)D.version, (int64_t)D.is_lte}) d.i64(v);
        d.f64(D.sublimation_mode ? D.sublimation_specific_energy : 0.0); d.f64(D.minimum_specific_energy);
        d.arr(D.nu, (size_t)D.n_nu); d.arr(D.albedo, (size_t)D.n_nu); d.arr(D.chi, (size_t)D.n_nu); d.arr(D.mu, (size_t)D.n_mu);
        const size_t np = (size_t)D.n_nu * D.n_mu;
        d.arr(D.P1, np); d.arr(D.P2, np); d.arr(D.P3, np); d.arr(D.P4, np);
        d.arr(D.emiss_nu, (size_t)D.n_enu); d.arr(D.emiss_jnu, (size_t)D.n_enu * D.n_jnu); d.arr(D.emiss_var, (size_t)D.n_jnu);
        d.arr(D.mo_specific_energy, (size_t)D.n_e); d.arr(D.mo_chi_rosseland, (size_t)D.n_e); d.arr(D.mo_kappa_planck, (size_t)D.n_e);
        d.arr(D.mo_chi_inv_planck, (size_t)D.n_e);
    }
    out[1] = d.h;
    Fnv s;
    s.i64(pr->n_sources);
    for (int i = 0; i < pr->n_sources; i++) {
        const hyp_source_desc &S = pr->sources[i];
        for (int64_t v : {(int64_t)S.type, (int64_t)S.spectrum_type, (int64_t)S.peeloff}) s.i64(v);
        s.f64(S.luminosity);
        if (S.spectrum_type == 1) { s.arr(S.spec_nu, (size_t)S.n_spec); s.arr(S.spec_fnu, (size_t)S.n_spec); }
        else if (S.spectrum_type == 2) s.f64(S.temperature);
        if (S.type == 1 || S.type == 2 || S.type == 5 || S.type == 7) for (double v : S.position) s.f64(v);
        if (S.type == 2 || S.type == 5 || S.type == 7) s.f64(S.radius);
        if (S.type == 2) s.i64(S.limb_darkening);
        if (S.type == 6) for (double v : S.box) s.f64(v);
        if (S.type == 7) for (double v : S.direction) s.f64(v);
        if (S.type == 8) { s.arr(S.points, 3 * (size_t)S.n_points); s.arr(S.point_lum, (size_t)S.n_points); }
        if (S.type == 4) s.arr(S.map, nc);
        s.i64(S.type == 2 ? S.n_spots : 0);
        for (int k = 0; k < S.n_spots && S.type == 2 && S.spots; k++) {
            const hyp_spot_desc &T = S.spots[k];
            for (double v : {T.longitude, T.latitude, T.radius, T.luminosity}) s.f64(v);
            s.i64(T.spectrum_type);
            if (T.spectrum_type == 1) { s.arr(T.spec_nu, (size_t)T.n_spec); s.arr(T.spec_fnu, (size_t)T.n_spec); } else s.f64(T.temperature);
        }
    }
    out[2] = s.h;
    Fnv c;
    const hyp_config &K = pr->config;
    for (int64_t v : {K.seed, K.n_inter_max, K.n_reabs_max, (int64_t)K.kill_on_absorb, (int64_t)K.kill_on_scatter, (int64_t)K.sample_sources_evenly,
                      (int64_t)K.enforce_energy_range, (int64_t)K.forced_first_interaction, (int64_t)(K.forced_first_interaction ? K.forced_first_interaction_algorithm : 0),
                      (int64_t)K.specific_energy_type, (int64_t)K.raytracing, (int64_t)K.mrw, (int64_t)K.monochromatic, (int64_t)K.pda, (int64_t)K.count_photons,
                      (int64_t)K.n_spectrum_bins}) c.i64(v);
    if (K.forced_first_interaction && K.forced_first_interaction_algorithm == 2) c.f64(K.baes16_xi);
    c.f64(K.propagation_check_frequency);
    if (K.mrw) { c.i64(K.n_inter_mrw_max); c.f64(K.mrw_gamma); }
    if (K.monochromatic) { c.f64(K.monochromatic_energy_threshold); c.arr(K.frequencies, (size_t)K.n_frequencies); }
    if (K.n_spectrum_bins) c.arr(K.spectrum_bin_edges, (size_t)K.n_spectrum_bins + 1);
    c.i64(pr->n_peeled);
    for (int i = 0; i < pr->n_peeled; i++) {
        digest_peeled(c, pr->peeled[i], false);
        if (K.monochromatic) { c.i64(pr->peeled[i].inu_min); c.i64(pr->peeled[i].inu_max); }
    }
    c.i64(pr->binned ? 1 : 0);
    if (pr->binned) { digest_peeled(c, *pr->binned, true); c.i64(pr->n_binned_theta); c.i64(pr->n_binned_phi); }
    out[3] = c.h;
    return 0;
}

const char *hyp_last_error(hyp_handle h) { return h ? h->err.c_str() : g_error.c_str(); }

void hyp_destroy(hyp_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    free_dev(h->d_problem); free_dev(h->d_blob); free_dev(h->d_sources); free_dev(h->d_peeled);
    free_dev(h->d_oct_cells); free_dev(h->d_oct_children); free_dev(h->d_oct_neigh);
    free_dev(h->d_at_slabs); free_dev(h->d_at_go); free_dev(h->d_at_grid_c0); free_dev(h->d_at_grid_nz);
    free_dev(h->d_ot_cluster); free_dev(h->d_ot_c0); free_dev(h->d_ot_nc); free_dev(h->d_ot_kid_off); free_dev(h->d_ot_rec); free_dev(h->d_ot_kid); free_dev(h->d_ot_nb);
    free_dev(h->d_mono_cdf); free_dev(h->d_mono_mean); free_dev(h->d_direct);
    free_dev(h->d_vor_bb);
    free_dev(h->d_vor_sites); free_dev(h->d_vor_volume); free_dev(h->d_vor_idx); free_dev(h->d_vor_neigh); free_dev(h->d_vor_seed); free_dev(h->d_vor_walls);
    free_dev(h->d_mask_map);
    free_dev(h->d_vt_cluster); free_dev(h->d_vt_info); free_dev(h->d_vt_blob); free_dev(h->d_vt_members); free_dev(h->d_vt_adj); free_dev(h->d_vt_ghost);
    free_dev(h->d_amr_grids); free_dev(h->d_amr_go); free_dev(h->d_amr_walls); free_dev(h->d_amr_cell_grid);
    free_dev(h->d_density); free_dev(h->d_specific_energy); free_dev(h->d_additional);
    free_dev(h->d_accum); free_dev(h->d_jnu_id); free_dev(h->d_jnu_frac); free_dev(h->d_energy_abs_tot);
    free_dev(h->d_mrw_alpha); free_dev(h->d_mrw_diff); free_dev(h->d_mrw_kp);
    free_dev(h->d_scratch); free_dev(h->d_counter); free_dev(h->d_err); free_dev(h->d_err_data);
    free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
    free_dev(h->d_peel_order); free_dev(h->d_peel_keys); free_dev(h->d_peel_bins); free_dev(h->d_ff);
    free_dev(h->d_peel_ctl);
    if (h->h_peel_ctl) (void)hipHostFree(h->h_peel_ctl);
    if (h->h_peel_counter) (void)hipHostFree(h->h_peel_counter);
    free_dev(h->d_img_accum);
    free_dev(h->d_hot); free_dev(h->d_cold); free_dev(h->d_slot_brick); free_dev(h->d_order);
    free_dev(h->d_counts); free_dev(h->d_cursor); free_dev(h->d_tasks); free_dev(h->d_ctl);
    free_dev(h->d_ilist); free_dev(h->d_dlist); free_dev(h->d_tcount); free_dev(h->d_extra);
    for (hipEvent_t e : h->walk_events) (void)hipEventDestroy(e);
    free_dev(h->d_nphot); free_dev(h->d_visit); free_dev(h->d_nphot_inexact); free_dev(h->d_log_edges); free_dev(h->d_bin_frac); free_dev(h->d_spec);
    free_dev(h->d_pda_mask); free_dev(h->d_pda_cells); free_dev(h->d_pda_hp); free_dev(h->d_pda_emean); free_dev(h->d_pda_coef);
    free_dev(h->d_pda_id); free_dev(h->d_pda_a); free_dev(h->d_pda_b); free_dev(h->d_pda_f);
    free_dev(h->d_pda_ctl); free_dev(h->d_prev_se); free_dev(h->d_ratio); free_dev(h->d_conv_ctl);
    if (h->h_ctl) (void)hipHostFree(h->h_ctl);
    for (int i = 1; i < 4; i++) if (h->pool_stream[i]) (void)hipStreamDestroy(h->pool_stream[i]);
    if (h->ev_pool) (void)hipEventDestroy(h->ev_pool);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->ev3) (void)hipEventDestroy(h->ev3);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int run_finish_kernel(hyp_handle h, int mode, double scale, double *d_out_ref);
static int solve_pda(hyp_handle h);
static int sync_problem(hyp_handle h);
static int check_device_error(hyp_handle h);
static int mrw_prepare(hyp_handle h);
static int build_vor_clusters(hyp_handle h);
static int build_oct_clusters(hyp_handle h);
static int build_amr_slabs(hyp_handle h);

int hyp_create(const hyp_problem *pr, int device, hyp_handle *out)
{
    g_error.clear();
    if (out) *out = nullptr;
    if (!pr || !out) return set_error("null argument");
    if (pr->grid.type < 1 || pr->grid.type > 6) return set_error("Unexpected coordinate type (grid types: 1 cartesian, 2 octree, 3 voronoi, 4 amr, 5 spherical polar, 6 cylindrical polar)");
    if (pr->n_dust < 1 || pr->n_dust > HYP_MAX_DUST) return set_error("n_dust must be between 1 and 8");
    // no sources is a valid set-up for dust-only raytracing / monochromatic runs (setup_rt.f90:228-239): the iterations that
    // need sources refuse to start instead (hyp_lucy_launch, hyp_final_launch)
    if (pr->n_sources < 0 || (pr->n_sources > 0 && !pr->sources)) return set_error("invalid source list");
    if (pr->config.monochromatic && (pr->config.n_frequencies < 1 || !pr->config.frequencies)) return set_error("monochromatic mode needs a frequency table");
    const bool is_oct = pr->grid.type == 2, is_vor = pr->grid.type == 3, is_amr = pr->grid.type == 4;
    const bool is_sph = pr->grid.type == 5, is_cyl = pr->grid.type == 6, is_polar = is_sph || is_cyl;
    const bool is_xyz = pr->grid.type == 1;                 // Cartesian proper (walls staged in LDS, brick-tiled schedule)
    const bool is_car = is_xyz || is_polar;                 // three wall arrays, cells (i1, i2, i3)
    std::vector<AmrGrid> amr_grids;
    std::vector<int> amr_go, amr_cell_grid;
    std::vector<double> amr_walls;
    double amr_eps = 0.0;
    int amr_level1 = 0;
    int64_t amr_cells = 0;
    const int n[3] = {is_car ? pr->grid.n1 : 0, is_car ? pr->grid.n2 : 0, is_car ? pr->grid.n3 : 0};
    std::vector<int> vor_seed;
    int vor_g = 1;
    const double *win[3] = {pr->grid.w1, pr->grid.w2, pr->grid.w3};
    std::vector<OctCell> oct_cells;
    std::vector<int> oct_children, oct_neigh;
    if (is_vor) {
        // setup_grid_geometry: grid_geometry_voronoi.f90:96-188
        const int64_t nc = pr->grid.n_cells;
        if (nc < 1 || nc > 2000000000ll || !pr->grid.vor_sites || !pr->grid.vor_idx || !pr->grid.vor_neighs || !pr->grid.vor_volume)
            return set_error("voronoi grid needs sites, volumes and neighbour lists");
        const int32_t *idx = pr->grid.vor_idx, *nei = pr->grid.vor_neighs;
        for (int64_t i = 0; i < nc; i++) if (idx[i + 1] < idx[i]) return set_error("sparse_idx should be non-decreasing");
        for (int64_t k = 0; k < idx[nc]; k++) if (nei[k] < -6 || nei[k] >= nc) return set_error("neighbour index out of range");
        for (int a = 0; a < 3; a++) if (!(pr->grid.vor_box[2 * a + 1] > pr->grid.vor_box[2 * a])) return set_error("voronoi domain is empty");
        // seed grid of the nearest-site walk: nearest site of every seed-cell centre
        const double *S = pr->grid.vor_sites, *B = pr->grid.vor_box;
        auto d2 = [&](int i, const double r[3]) {
            double dx = S[3 * (size_t)i] - r[0], dy = S[3 * (size_t)i + 1] - r[1], dz = S[3 * (size_t)i + 2] - r[2];
            return dx * dx + dy * dy + dz * dz;
        };
        auto nearest_from = [&](const double r[3], int seed) {
            int cur = seed; double dcur = d2(cur, r);
            for (;;) {
                int best = cur; double dbest = dcur;
                for (int k = idx[cur]; k < idx[cur + 1]; k++) {
                    int nb = nei[k];
                    if (nb < 0) continue;
                    double d = d2(nb, r);
                    if (d < dbest) { dbest = d; best = nb; }
                }
                if (best == cur) return cur;
                cur = best; dcur = dbest;
            }
        };
        // about two seed cells per site: the walk from the seed to the nearest site is 0-1 hops for most positions (emission
        // from extended sources places every packet this way); the result does not depend on the seed
        vor_g = (int)std::ceil(std::cbrt((double)nc * 2.0));
        if (vor_g < 1) vor_g = 1;
        if (vor_g > 256) vor_g = 256;
        vor_seed.resize((size_t)vor_g * vor_g * vor_g);
        int last = 0;
        for (int k = 0; k < vor_g; k++) for (int j = 0; j < vor_g; j++) for (int i = 0; i < vor_g; i++) {
            double c[3] = {B[0] + (i + 0.5) / vor_g * (B[1] - B[0]), B[2] + (j + 0.5) / vor_g * (B[3] - B[2]),
                           B[4] + (k + 0.5) / vor_g * (B[5] - B[4])};
            last = nearest_from(c, last);
            vor_seed[((size_t)k * vor_g + j) * vor_g + i] = last;
        }
    } else if (is_polar) {
        // setup_grid_geometry: grid_geometry_spherical_3d.f90:90-203, grid_geometry_cylindrical_3d.f90:90-175
        const double pi = 3.14159265358979323846;
        for (int a = 0; a < 3; a++) if (n[a] < 1 || !win[a]) return set_error("grid walls missing");
        for (int i = 0; i <= n[0]; i++) if (win[0][i] < 0.0) return set_error(is_sph ? "r walls should be positive" : "w walls should be positive");
        for (int i = 0; i <= n[1] && is_sph; i++) if (win[1][i] < 0.0 || win[1][i] > pi) return set_error("theta walls should be between 0 and pi");
        for (int i = 0; i <= n[2]; i++) if (win[2][i] < 0.0 || win[2][i] > 2.0 * pi) return set_error("phi walls should be between 0 and 2*pi");
        static const char *names_s[3] = {"dr", "dt", "dphi"}, *names_c[3] = {"dw", "dz", "dphi"};
        for (int a = 0; a < 3; a++) for (int i = 0; i < n[a]; i++)
            if (win[a][i + 1] - win[a][i] == 0.0)
                return set_error(std::string("all ") + (is_sph ? names_s[a] : names_c[a]) + " values should be greater than zero");
        for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
            const double a0 = win[0][i], b0 = win[0][i + 1], dphi = win[2][k + 1] - win[2][k];
            const double vol = is_sph ? (b0 * b0 * b0 - a0 * a0 * a0) * (std::cos(win[1][j]) - std::cos(win[1][j + 1])) * dphi / 3.0
                                      : (b0 * b0 - a0 * a0) * (win[1][j + 1] - win[1][j]) * dphi / 2.0;
            if (vol == 0.0) return set_error("all volumes should be greater than zero");
        }
    } else if (is_car) {
        for (int a = 0; a < 3; a++) {
            if (n[a] < 1 || !win[a]) return set_error("grid walls missing");
            for (int i = 0; i < n[a]; i++)
                if (!(win[a][i + 1] - win[a][i] > 0.0))
                    return set_error(std::string("all d") + "xyz"[a] + " values should be greater than zero");
        }
        if (is_xyz && (size_t)n[0] + n[1] + n[2] + 3 > 9000) return set_error("grid has too many walls for LDS staging");
    } else if (is_amr) {
        // read_grid/read_level + setup_grid_geometry: grid_geometry_amr.f90:111-508
        const int ng = pr->grid.n_amr_grids, nl = pr->grid.n_amr_levels;
        if (ng < 1 || nl < 1 || !pr->grid.amr_level || !pr->grid.amr_n || !pr->grid.amr_bounds) return set_error("amr grid needs levels and grids");
        amr_grids.resize(ng);
        std::vector<int> level(ng);
        std::vector<std::array<double, 3>> width(ng);
        double min_width = DBL_MAX;
        for (int k = 0; k < ng; k++) {
            AmrGrid &g = amr_grids[k];
            level[k] = pr->grid.amr_level[k];
            if (level[k] < 1 || level[k] > nl || (k > 0 && level[k] < level[k - 1])) return set_error("amr grids must be listed level by level");
            if (level[k] == 1) amr_level1 = k + 1;
            for (int a = 0; a < 3; a++) {
                g.n[a] = pr->grid.amr_n[3 * k + a];
                g.lo[a] = pr->grid.amr_bounds[6 * k + 2 * a]; g.hi[a] = pr->grid.amr_bounds[6 * k + 2 * a + 1];
                if (g.n[a] < 1 || !(g.hi[a] > g.lo[a])) return set_error("all volumes should be greater than zero");
                g.w_off[a] = (int)amr_walls.size();
                // fortranlib linspace: x(i) = (xmax - xmin) * (i - 1) / (n - 1) + xmin
                for (int i = 0; i <= g.n[a]; i++) amr_walls.push_back((g.hi[a] - g.lo[a]) * (double)i / (double)g.n[a] + g.lo[a]);
                width[k][a] = (g.hi[a] - g.lo[a]) / (double)g.n[a];
                if (width[k][a] < min_width) min_width = width[k][a];
            }
            if (amr_cells > 2000000000ll) return set_error("amr grid has too many cells");
            g.start = (unsigned)amr_cells; amr_cells += (int64_t)g.n[0] * g.n[1] * g.n[2];
            g.go_off = (int)amr_go.size();
            amr_go.resize(amr_go.size() + (size_t)(g.n[0] + 2) * (g.n[1] + 2) * (g.n[2] + 2), 0);
        }
        if (amr_cells > 2000000000ll) return set_error("amr grid has too many cells");
        amr_eps = min_width / 2.0;
        auto aligned = [](double x1, double x2, double dx) {
            double r = std::fmod(std::fabs(x1 - x2), dx);
            if (r > 0.5 * dx) r = dx - r;
            return std::fabs(r / dx) < 1.e-8;
        };
        auto first_of_level = [&](int l) { for (int q = 0; q < ng; q++) if (level[q] == l) return q; return -1; };
        char msg[256];
        for (int k = 0; k < ng; k++) {
            const int ref = first_of_level(level[k]), igrid = k - ref + 1;
            for (int a = 0; a < 3; a++) {
                if (std::fabs(width[k][a] - width[ref][a]) > 1.e-10 * width[k][a]) {
                    std::snprintf(msg, sizeof msg, "Grids 1 and %d in level %d have differing cell widths in the %c direction", igrid, level[k], "xyz"[a]);
                    return set_error(msg);
                }
                if (!aligned(amr_grids[k].lo[a], amr_grids[ref].lo[a], width[ref][a])) {
                    std::snprintf(msg, sizeof msg, "Grids 1 and %d in level %d have edges that are not separated by an integer number of cells in the %c direction", igrid, level[k], "xyz"[a]);
                    return set_error(msg);
                }
            }
            if (level[k] > 1) {
                const int pref = first_of_level(level[k] - 1);
                if (pref < 0) return set_error("amr level without grids");
                for (int a = 0; a < 3; a++) {
                    const double rf = width[pref][a] / width[ref][a];
                    if (std::fabs(rf - std::nearbyint(rf)) > 1.e-10) {
                        std::snprintf(msg, sizeof msg, "Refinement factor in the %c direction between level %d and level %d is not an integer (%.3f)", "xyz"[a], level[k] - 1, level[k], rf);
                        return set_error(msg);
                    }
                    if (!aligned(amr_grids[k].lo[a], amr_grids[pref].lo[a], width[pref][a])) {
                        std::snprintf(msg, sizeof msg, "Grid %d in level %d is not aligned with cells in level %d in the %c direction", igrid, level[k], level[k] - 1, "xyz"[a]);
                        return set_error(msg);
                    }
                }
            }
        }
        auto in_grid = [&](int k, const double r[3]) {
            const AmrGrid &g = amr_grids[k];
            for (int a = 0; a < 3; a++) { if (r[a] < g.lo[a]) return false; if (r[a] > g.hi[a]) return false; }
            return true;
        };
        auto go_at = [&](int k, int i1, int i2, int i3) -> int & {
            const AmrGrid &g = amr_grids[k];
            return amr_go[g.go_off + ((size_t)i3 * (g.n[1] + 2) + i2) * (g.n[0] + 2) + i1];
        };
        auto wall = [&](int k, int a, int i) { return amr_walls[amr_grids[k].w_off[a] + i]; };
        // cells overlapped by a grid of the next level (:357-382)
        for (int l1 = nl - 1; l1 >= 1; l1--)
            for (int k1 = 0; k1 < ng; k1++) {
                if (level[k1] != l1) continue;
                const AmrGrid &g1 = amr_grids[k1];
                for (int k2 = 0; k2 < ng; k2++) {
                    if (level[k2] != l1 + 1) continue;
                    const AmrGrid &g2 = amr_grids[k2];
                    bool hit = true;
                    for (int a = 0; a < 3; a++) if (g1.hi[a] < g2.lo[a] || g1.lo[a] > g2.hi[a]) hit = false;
                    if (!hit) continue;
                    for (int i1 = 1; i1 <= g1.n[0]; i1++) for (int i2 = 1; i2 <= g1.n[1]; i2++) for (int i3 = 1; i3 <= g1.n[2]; i3++) {
                        const double r[3] = {0.5 * (wall(k1, 0, i1 - 1) + wall(k1, 0, i1)), 0.5 * (wall(k1, 1, i2 - 1) + wall(k1, 1, i2)),
                                             0.5 * (wall(k1, 2, i3 - 1) + wall(k1, 2, i3))};
                        if (in_grid(k2, r)) go_at(k1, i1, i2, i3) = k2 + 1;
                    }
                }
            }
        // one step outside each grid: the grid of the same or a coarser level found there (:384-486)
        for (int k1 = 0; k1 < ng; k1++) {
            const AmrGrid &g1 = amr_grids[k1];
            for (int l2 = level[k1]; l2 >= 1; l2--)
                for (int k2 = 0; k2 < ng; k2++) {
                    if (level[k2] != l2 || k2 == k1) continue;
                    const AmrGrid &g2 = amr_grids[k2];
                    bool close = true;
                    for (int a = 0; a < 3; a++)
                        if (g1.hi[a] < g2.lo[a] - width[k2][a] * 0.5 || g1.lo[a] > g2.hi[a] + width[k2][a] * 0.5) close = false;
                    if (!close) continue;
                    for (int a = 0; a < 3; a++) {
                        const int b = (a + 1) % 3, c = (a + 2) % 3;
                        for (int side = 0; side < 2; side++) {
                            int idx[3]; double r[3];
                            idx[a] = side ? g1.n[a] + 1 : 0;
                            r[a] = side ? g1.hi[a] + width[k1][a] * 0.5 : g1.lo[a] - width[k1][a] * 0.5;
                            for (int ib = 1; ib <= g1.n[b]; ib++) for (int ic = 1; ic <= g1.n[c]; ic++) {
                                idx[b] = ib; idx[c] = ic;
                                r[b] = 0.5 * (wall(k1, b, ib - 1) + wall(k1, b, ib)); r[c] = 0.5 * (wall(k1, c, ic - 1) + wall(k1, c, ic));
                                int &q = go_at(k1, idx[0], idx[1], idx[2]);
                                if (in_grid(k2, r) && q == 0) q = k2 + 1;
                            }
                        }
                    }
                }
        }
        amr_cell_grid.resize((size_t)amr_cells);
        for (int k = 0; k < ng; k++) {
            const AmrGrid &g = amr_grids[k];
            const size_t nc = (size_t)g.n[0] * g.n[1] * g.n[2];
            for (size_t c = 0; c < nc; c++) amr_cell_grid[g.start + c] = k;
        }
    } else {
        // setup_grid_geometry + octree_setup_indiv: grid_geometry_octree.f90:147-246
        const int64_t nc = pr->grid.n_cells;
        if (nc < 1 || nc > 2000000000ll || !pr->grid.refined) return set_error("octree needs a refined list");
        oct_cells.resize((size_t)nc);
        oct_children.assign((size_t)nc * 8, -1);
        for (int a = 0; a < 3; a++) if (!(pr->grid.oct_half[a] > 0.0)) return set_error("all volumes should be greater than zero");
        std::vector<double> hx((size_t)nc);   // x half-widths only to detect underflow of the level encoding
        OctCell &root = oct_cells[0];
        root.x = pr->grid.oct_center[0]; root.y = pr->grid.oct_center[1]; root.z = pr->grid.oct_center[2];
        root.parent = -1; root.subcell = -1; root.level = 0; root.refined = pr->grid.refined[0] == 1; root.pad = 0;
        std::vector<std::pair<int, int>> stack;
        if (root.refined) stack.push_back({0, 0});
        int64_t filled = 1;
        while (!stack.empty()) {
            int par = stack.back().first, k = stack.back().second;
            if (k == 8) { stack.pop_back(); continue; }
            stack.back().second = k + 1;
            if (filled >= nc) return set_error("refined array is not self-consistent");
            int c = (int)filled++;
            oct_children[(size_t)par * 8 + k] = c;
            const OctCell &pc = oct_cells[par];
            const int lev = pc.level;
            OctCell &cc = oct_cells[c];
            const double hpx = std::ldexp(pr->grid.oct_half[0], -lev), hpy = std::ldexp(pr->grid.oct_half[1], -lev),
                         hpz = std::ldexp(pr->grid.oct_half[2], -lev);
            cc.x = pc.x + ((k & 1) ? 1 : -1) * hpx / 2.0;
            cc.y = pc.y + ((k & 2) ? 1 : -1) * hpy / 2.0;
            cc.z = pc.z + ((k & 4) ? 1 : -1) * hpz / 2.0;
            cc.parent = par; cc.subcell = (signed char)k; cc.pad = 0;
            if (lev + 1 > 200) return set_error("octree too deep");
            cc.level = (unsigned char)(lev + 1);
            cc.refined = pr->grid.refined[c] == 1;
            if (cc.refined) stack.push_back({c, 0});
        }
        if (filled != nc) return set_error("refined array is not self-consistent");
        // neighbour across each face, no finer than the cell itself: what next_cell_int (:328-347) finds when its descent is
        // stopped at the cell's own level (geo_advance goes on from there)
        oct_neigh.assign((size_t)nc * 6, (int)nc);
        std::vector<int> subs(256);
        for (int64_t id = 1; id < nc; id++)
            for (int axis = 0; axis < 3; axis++)
                for (int up = 0; up < 2; up++) {
                    int cur = (int)id, depth = 0, n = (int)nc;
                    while (cur != 0) {
                        const int sub = oct_cells[cur].subcell, par = oct_cells[cur].parent;
                        if (((sub >> axis) & 1) != up) {
                            int S = oct_children[(size_t)par * 8 + (up ? (sub | (1 << axis)) : (sub & ~(1 << axis)))];
                            while (oct_cells[S].refined && depth > 0) {
                                const int sc = subs[--depth];
                                S = oct_children[(size_t)S * 8 + ((sc & ~(1 << axis)) | ((up ? 0 : 1) << axis))];
                            }
                            n = S;
                            break;
                        }
                        subs[depth++] = sub; cur = par;
                    }
                    oct_neigh[(size_t)id * 6 + 2 * axis + up] = n;
                }
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error("no HIP device available: the photon-packet engine requires an AMD GPU (gfx950)");
    if (device < 0 || device >= ndev) return set_error("invalid device ordinal");
    HIP_TRY(hipSetDevice(device));

    hyp_engine *h = new hyp_engine();
    h->device = device;
    h->cfg = pr->config;
    h->n_dust = pr->n_dust;
    h->n_cells = is_car ? (size_t)n[0] * n[1] * n[2] : is_amr ? (size_t)amr_cells : (size_t)pr->grid.n_cells;
    h->n_elem = h->n_cells * h->n_dust;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
    if (h->n_cu <= 0) h->n_cu = 256;

#define FAIL(msg) do { g_error = (msg); hyp_destroy(h); return 1; } while (0)
#define HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_error = std::string(#call) + ": " + hipGetErrorString(e_); hyp_destroy(h); return 1; } } while (0)

    Blob B;
    DProblem &P = h->hp;
    std::memset(&P, 0, sizeof(P));
    P.n1 = n[0]; P.n2 = n[1]; P.n3 = n[2]; P.n_dust = pr->n_dust;
    P.grid_type = pr->grid.type;
    if (is_oct) {
        double m = 0.0;
        for (int a = 0; a < 3; a++) {
            P.oct_half[a] = pr->grid.oct_half[a];
            P.oct_box[2 * a] = pr->grid.oct_center[a] - pr->grid.oct_half[a];
            P.oct_box[2 * a + 1] = pr->grid.oct_center[a] + pr->grid.oct_half[a];
            if (pr->grid.oct_half[a] > m) m = pr->grid.oct_half[a];
        }
        P.oct_eps = spacing(m) * 3.0;   // grid_geometry_octree.f90:243
    }
    P.n_sources = pr->n_sources; P.n_peeled = pr->n_peeled;
    P.sample_sources_evenly = pr->config.sample_sources_evenly;
    P.kill_on_absorb = pr->config.kill_on_absorb; P.kill_on_scatter = pr->config.kill_on_scatter;
    P.forced_first = pr->config.forced_first_interaction; P.forced_algo = pr->config.forced_first_interaction_algorithm;
    P.n_inter_max = pr->config.n_inter_max; P.n_reabs_max = pr->config.n_reabs_max; P.n_cells = h->n_cells; P.baes16_xi = pr->config.baes16_xi;
    {
        P.check_p = pr->config.propagation_check_frequency;
        P.check_log1mp = (P.check_p > 0.0 && P.check_p < 1.0) ? std::log1p(-P.check_p) : -1.0;
        int64_t sd = pr->config.seed;
        uint64_t s = (uint64_t)(sd < 0 ? -sd : sd);
        P.seed_key = (uint32_t)s ^ (uint32_t)(s >> 32);
    }

    // walls + 3*spacing(w): grid_geometry_cartesian_3d.f90:97-132
    // (Cartesian walls beyond 2^300: the reference's cell volumes dx dy dz overflow there; the wall search relies on path lengths
    // (w - r) / v staying finite, find_wall_ahead in hyp_kernels.h)
    for (int a = 0; a < 3 && is_car && !is_polar; a++)
        for (int i = 0; i <= n[a]; i++)
            if (!(std::fabs(win[a][i]) < 0x1p300)) return set_error("grid walls beyond 2^300 are not supported (cell volumes overflow)");
    size_t w_off[3] = {0, 0, 0}, ew_off[3] = {0, 0, 0};
    for (int a = 0; a < 3 && is_car; a++) {
        w_off[a] = B.put(win[a], n[a] + 1);
        std::vector<double> ew(n[a] + 1);
        for (int i = 0; i <= n[a]; i++) ew[i] = 3.0 * spacing(win[a][i]);
        // angles: ew = 3 * spacing(1) (spherical_3d.f90:199-201, cylindrical_3d.f90:171-173)
        if (is_polar && (a == 2 || (a == 1 && is_sph))) for (int i = 0; i <= n[a]; i++) ew[i] = 3.0 * spacing(1.0);
        ew_off[a] = B.put(ew);
    }
    size_t polar_off[5] = {0, 0, 0, 0, 0};
    int midplane = -2;
    if (is_polar) {
        std::vector<double> wr2(n[0] + 1), wtanp(n[2] + 1);
        for (int i = 0; i <= n[0]; i++) wr2[i] = win[0][i] * win[0][i];
        for (int i = 0; i <= n[2]; i++) wtanp[i] = std::tan(win[2][i]);
        polar_off[0] = B.put(wr2); polar_off[4] = B.put(wtanp);
        if (is_sph) {
            std::vector<double> wtant(n[1] + 1), wtant2(n[1] + 1), wcost(n[1] + 1);
            double m = DBL_MAX; int im = 0;
            for (int i = 0; i <= n[1]; i++) {
                wtant[i] = std::tan(win[1][i]); wtant2[i] = wtant[i] * wtant[i]; wcost[i] = std::cos(win[1][i]);
                const double d = std::fabs(win[1][i] - 3.14159265358979323846 / 2.0);
                if (d < m) { m = d; im = i; }
            }
            if (m < 1.e-6) midplane = im;       // :175: minloc(abs(w2 - pi/2)) if any is within 1e-6
            polar_off[1] = B.put(wtant); polar_off[2] = B.put(wtant2); polar_off[3] = B.put(wcost);
        }
    }

    // dust tables: dust_type_4elem.f90:78-293
    std::vector<DustOffsets> doff(pr->n_dust);
    for (int d = 0; d < pr->n_dust; d++) {
        const hyp_dust_desc &in = pr->dust[d];
        DDust &D = P.dust[d];
        DustOffsets &O = doff[d];
        const int nn = in.n_nu, nm = in.n_mu;
        if (nn < 2 || nm < 2 || in.n_jnu < 2 || in.n_enu < 2) FAIL("dust tables too short");
        D.n_nu = nn; D.n_mu = nm; D.n_jnu = in.n_jnu; D.n_enu = in.n_enu; D.n_e = in.n_e;
        D.sublimation_mode = in.sublimation_mode;
        D.sublimation_specific_energy = in.sublimation_specific_energy;
        D.minimum_specific_energy = in.minimum_specific_energy;
        D.nu_min = in.nu[0]; D.nu_max = in.nu[nn - 1];
        D.mu_min = in.mu[0]; D.mu_max = in.mu[nm - 1];
        for (int i = 0; i + 1 < nn; i++) if (!(in.nu[i + 1] > in.nu[i])) FAIL("dust frequencies should be monotonically increasing");
        std::vector<double> lnu(nn), lchi(nn), lalb(nn);
        for (int i = 0; i < nn; i++) {
            lnu[i] = std::log10(in.nu[i]);
            lchi[i] = in.chi[i] > 0.0 ? std::log10(in.chi[i]) : std::nan("");
            lalb[i] = in.albedo[i] > 0.0 ? std::log10(in.albedo[i]) : std::nan("");
        }
        O.nu = B.put(in.nu, nn); O.log10_nu = B.put(lnu);
        O.chi = B.put(in.chi, nn); O.albedo = B.put(in.albedo, nn);
        O.log10_chi = B.put(lchi); O.log10_albedo = B.put(lalb);
        O.mu = B.put(in.mu, nm);
        const size_t np = (size_t)nn * nm;
        std::vector<double> P1(in.P1, in.P1 + np), P2(in.P2, in.P2 + np), P3(in.P3, in.P3 + np), P4(in.P4, in.P4 + np);
        D.zero_p2 = 1;
        for (size_t i = 0; i < np; i++) if (P2[i] != 0.0) { D.zero_p2 = 0; break; }
        const double dmu = D.mu_max - D.mu_min;
        std::vector<double> C1(np, 0.0), C2(np, 0.0);
        for (int j = 0; j < nn; j++) {
            double *p1 = &P1[(size_t)j * nm], *p2 = &P2[(size_t)j * nm], *p3 = &P3[(size_t)j * nm], *p4 = &P4[(size_t)j * nm];
            double norm = integral_linlog(in.mu, p1, nm);
            if (norm == 0.0) FAIL("P1 matrix normalization is zero");
            for (int i = 0; i < nm; i++) {
                p1[i] = p1[i] / norm * dmu; p2[i] = p2[i] / norm * dmu;
                p3[i] = p3[i] / norm * dmu; p4[i] = p4[i] / norm * dmu;
            }
            double *c1 = &C1[(size_t)j * nm], *c2 = &C2[(size_t)j * nm];
            for (int i = 1; i < nm; i++) {
                double dx = in.mu[i] - in.mu[i - 1];
                c1[i] = c1[i - 1] + 0.5 * (p1[i] + p1[i - 1]) * dx;
                c2[i] = c2[i - 1] + 0.5 * (p2[i] + p2[i - 1]) * dx;
            }
            bool z1 = true, z2 = true;
            for (int i = 0; i < nm; i++) { if (c1[i] != 0.0) z1 = false; if (c2[i] != 0.0) z2 = false; }
            if (!z1) { double l = c1[nm - 1]; for (int i = 0; i < nm; i++) c1[i] /= l; }
            if (!z2) { double l = c2[nm - 1]; for (int i = 0; i < nm; i++) c2[i] /= l; }
        }
        O.P1 = B.put(P1); O.P2 = B.put(P2); O.P3 = B.put(P3); O.P4 = B.put(P4);
        O.P1_cdf = B.put(C1); O.P2_cdf = B.put(C2);
        // emissivities
        O.emiss_x = B.put(in.emiss_nu, in.n_enu);
        std::vector<double> cdf_all, bp1_all, cdf, bp1;
        for (int i = 0; i < in.n_jnu; i++) {
            if (!build_log_pdf(in.emiss_nu, in.emiss_jnu + i, in.n_enu, in.n_jnu, cdf, bp1)) FAIL("emissivity has zero integral");
            cdf_all.insert(cdf_all.end(), cdf.begin(), cdf.end());
            bp1_all.insert(bp1_all.end(), bp1.begin(), bp1.end());
        }
        O.emiss_cdf = B.put(cdf_all); O.emiss_bp1 = B.put(bp1_all);
        {   // every HYP_COARSE-th CDF entry of each row, for the two-level search of sample_log_pdf_pair
            const int nc = (in.n_enu + HYP_COARSE - 1) / HYP_COARSE;
            std::vector<double> coarse((size_t)in.n_jnu * nc);
            for (int i = 0; i < in.n_jnu; i++)
                for (int m = 0; m < nc; m++) coarse[(size_t)i * nc + m] = cdf_all[(size_t)i * in.n_enu + (size_t)m * HYP_COARSE];
            O.emiss_coarse = B.put(coarse);
        }
        std::vector<double> ljv(in.n_jnu);
        for (int i = 0; i < in.n_jnu; i++) ljv[i] = std::log10(in.emiss_var[i]);
        O.jnu_var = B.put(in.emiss_var, in.n_jnu); O.log10_jnu_var = B.put(ljv);
        O.have_mo_e = in.n_e > 0 && in.mo_specific_energy;
        O.have_mo_chi = O.have_mo_e && in.mo_chi_rosseland;
        if (O.have_mo_e) {
            for (int i = 1; i < in.n_e; i++)
                if (in.mo_specific_energy[i] < in.mo_specific_energy[i - 1]) FAIL("energy per unit mass is not monotonically increasing");
            O.mo_e = B.put(in.mo_specific_energy, in.n_e);
            D.e_min = in.mo_specific_energy[0]; D.e_max = in.mo_specific_energy[in.n_e - 1]; D.have_e_range = 1;
        }
        if (O.have_mo_chi) O.mo_chi_ross = B.put(in.mo_chi_rosseland, in.n_e);
        if (in.sublimation_mode == 2 && !O.have_mo_chi) FAIL("slow sublimation needs the Rosseland mean opacity table");
        O.have_pda = false;
        if (pr->config.pda) {       // setup_rt.f90:289-300; grid_pda_3d.f90 reads kappa_planck and chi_rosseland
            if (in.version == 1)
                FAIL("version 1 dust files can no longer be used when PDA is computed due to a bug - to fix this, re-generate the dust file using the latest version of Hyperion");
            if (!(O.have_mo_chi && in.mo_kappa_planck)) FAIL("PDA needs the kappa_planck and chi_rosseland mean opacities of every dust type");
            if (!pr->config.mrw) { O.mo_kappa_planck = B.put(in.mo_kappa_planck, in.n_e); O.have_pda = true; }
        }
        // modified random walk: Planck means + b_nu = j_nu / kappa_nu pdfs (dust_type_4elem.f90:289-291)
        O.have_mrw = false;
        if (pr->config.mrw) {
            if (!(O.have_mo_e && in.mo_kappa_planck && in.mo_chi_inv_planck))
                FAIL("MRW needs the kappa_planck and chi_inv_planck mean opacities of every dust type");
            O.mo_kappa_planck = B.put(in.mo_kappa_planck, in.n_e);
            O.mo_chi_inv_planck = B.put(in.mo_chi_inv_planck, in.n_e);
            std::vector<double> kap(nn), y(in.n_enu), bcdf_all, bbp1_all;
            for (int k = 0; k < nn; k++) kap[k] = in.chi[k] * (1.0 - in.albedo[k]);
            for (int i = 0; i < in.n_jnu; i++) {
                for (int k = 0; k < in.n_enu; k++)
                    y[k] = in.emiss_jnu[(size_t)k * in.n_jnu + i] / interp1d_loglog_host(in.nu, kap.data(), nn, in.emiss_nu[k]);
                if (!build_log_pdf(in.emiss_nu, y.data(), in.n_enu, 1, cdf, bp1)) FAIL("emissivity / kappa_nu has zero integral");
                bcdf_all.insert(bcdf_all.end(), cdf.begin(), cdf.end());
                bbp1_all.insert(bbp1_all.end(), bp1.begin(), bp1.end());
            }
            O.bnu_cdf = B.put(bcdf_all); O.bnu_bp1 = B.put(bbp1_all);
            const int nc = (in.n_enu + HYP_COARSE - 1) / HYP_COARSE;
            std::vector<double> coarse((size_t)in.n_jnu * nc);
            for (int i = 0; i < in.n_jnu; i++)
                for (int m = 0; m < nc; m++) coarse[(size_t)i * nc + m] = bcdf_all[(size_t)i * in.n_enu + (size_t)m * HYP_COARSE];
            O.bnu_coarse = B.put(coarse);
            O.have_mrw = true;
        }
    }
    // cumulative of Min et al. (2009) eq. 6 on 100 points: grid_mrw_3d.f90:157-195
    size_t mrw_x_off = 0, mrw_y_off = 0;
    if (pr->config.mrw) {
        std::vector<double> mx(100), my(100);
        for (int i = 0; i < 100; i++) {
            mx[i] = (double)i / 99.0;
            double y = 0.0;
            if (i == 99) y = 0.5;
            else for (long long j = 1;; j++) {
                const double term = std::pow(mx[i], (double)(j * j));
                if (term == 0.0) break;
                if (j % 2 == 0) y -= term; else y += term;
            }
            my[i] = y * 2.0;
        }
        mrw_x_off = B.put(mx); mrw_y_off = B.put(my);
    }

    // sources: source.f90:47-84, source_type.f90:102-322
    std::vector<DSource> hs(pr->n_sources);
    std::vector<SourceOffsets> soff(pr->n_sources);
    h->energy_total = 0.0;
    // luminosity of a point collection = sum of its members (source_type.f90:271)
    std::vector<double> src_lum(pr->n_sources);
    for (int i = 0; i < pr->n_sources; i++) {
        const hyp_source_desc &s = pr->sources[i];
        src_lum[i] = s.luminosity;
        if (s.type == 8 && s.point_lum && s.n_points > 0) { src_lum[i] = 0.0; for (int k = 0; k < s.n_points; k++) src_lum[i] += s.point_lum[k]; }
        h->energy_total += src_lum[i];
        soff[i].have_points = false; soff[i].have_map = false; soff[i].have_spots = false;
    }
    {
        double c = 0.0;
        for (int i = 0; i < pr->n_sources; i++) {
            const hyp_source_desc &s = pr->sources[i];
            DSource &S = hs[i];
            std::memset(&S, 0, sizeof(S));
            if (s.type != 1 && s.type != 2 && s.type != 4 && s.type != 5 && s.type != 6 && s.type != 7 && s.type != 8) FAIL("unknown type in source list: " + std::to_string(s.type));
            S.type = s.type; S.peeloff = s.peeloff; S.radius = s.radius; S.limb_darkening = s.limb_darkening;
            if (is_vor && s.type == 1) {
                // every packet of a point source starts in the same cell: find_cell (grid_geometry_voronoi.f90:196-229) once, here
                const double *Sx = pr->grid.vor_sites, *Bx = pr->grid.vor_box;
                const double r[3] = {s.position[0], s.position[1], s.position[2]};
                if (!(r[0] < Bx[0] || r[0] > Bx[1] || r[1] < Bx[2] || r[1] > Bx[3] || r[2] < Bx[4] || r[2] > Bx[5])) {
                    int id[3];
                    for (int a = 0; a < 3; a++) {
                        const double f = (r[a] - Bx[2 * a]) / (Bx[2 * a + 1] - Bx[2 * a]);
                        const int q = (int)(f * vor_g);
                        id[a] = q < 0 ? 0 : (q >= vor_g ? vor_g - 1 : q);
                    }
                    auto d2 = [&](int c) { const double dx = Sx[3 * (size_t)c] - r[0], dy = Sx[3 * (size_t)c + 1] - r[1], dz = Sx[3 * (size_t)c + 2] - r[2]; return dx * dx + dy * dy + dz * dz; };
                    int cur = vor_seed[((size_t)id[2] * vor_g + id[1]) * vor_g + id[0]];
                    double dcur = d2(cur);
                    for (;;) {
                        int best = cur; double dbest = dcur;
                        for (int k = pr->grid.vor_idx[cur]; k < pr->grid.vor_idx[cur + 1]; k++) {
                            const int nb = pr->grid.vor_neighs[k];
                            if (nb < 0) continue;
                            const double d = d2(nb);
                            if (d < dbest) { dbest = d; best = nb; }
                        }
                        if (best == cur) break;
                        cur = best; dcur = dbest;
                    }
                    S.vor_cell1 = cur + 1;
                }
            }
            if (s.type == 2) P.any_intersect = 1;      // s%intersect = .true.: source_type.f90:148
            if (s.type == 7) {      // plane_parallel: source_type.f90:239-256
                const double th = s.direction[0] * HYP_PI / 180.0, ph = s.direction[1] * HYP_PI / 180.0;
                S.dir_cost = std::cos(th); S.dir_sint = std::sin(th); S.dir_cosp = std::cos(ph); S.dir_sinp = std::sin(ph);
                if (s.peeloff) FAIL("plane parallel sources cannot be peeled off (source_emit_peeloff has no case for them)");
            }
            if (s.type == 8) {      // point_collection: source_type.f90:258-277
                if (s.n_points < 1 || !s.points || !s.point_lum) FAIL("point source collection needs positions and luminosities");
                std::vector<double> cdf(s.n_points);
                double tot = 0.0, c = 0.0;
                for (int k = 0; k < s.n_points; k++) tot += s.point_lum[k];
                for (int k = 0; k < s.n_points; k++) { c += s.point_lum[k] / tot; cdf[k] = c; }
                for (int k = 0; k < s.n_points; k++) cdf[k] /= c;
                S.n_points = s.n_points;
                soff[i].points = B.put(s.points, 3 * (size_t)s.n_points); soff[i].point_cdf = B.put(cdf);
                soff[i].have_points = true;
            }
            if (s.n_spots > 0) {    // spotted sphere: source_type.f90:150-188
                if (s.type != 2 || !s.spots) FAIL("only spherical sources can have spots");
                const int ns = s.n_spots;
                std::vector<double> tab((size_t)(ns + 1) + (size_t)ns * SPOT_STRIDE, 0.0);
                double tot = s.luminosity, cc = 0.0;
                for (int k = 0; k < ns; k++) tot += s.spots[k].luminosity;
                for (int k = 0; k <= ns; k++) { cc += (k < ns ? s.spots[k].luminosity : s.luminosity) / tot; tab[k] = cc; }
                for (int k = 0; k <= ns; k++) tab[k] /= cc;
                for (int k = 0; k < ns; k++) {
                    const hyp_spot_desc &q = s.spots[k];
                    double *t = tab.data() + (ns + 1) + (size_t)k * SPOT_STRIDE;
                    // angle3d_deg(lon, lat) as the reference passes them (theta = lon, phi = lat), then angle3d_to_vector3d
                    const double th = q.longitude * HYP_PI / 180.0, ph = q.latitude * HYP_PI / 180.0;
                    t[0] = std::sin(th) * std::cos(ph); t[1] = std::sin(th) * std::sin(ph); t[2] = std::cos(th);
                    t[3] = std::cos(q.radius * HYP_PI / 180.0);
                    t[4] = q.spectrum_type; t[5] = q.temperature; t[6] = q.n_spec;
                    if (q.spectrum_type == 1) {
                        std::vector<double> cdf, bp1;
                        if (!build_log_pdf(q.spec_nu, q.spec_fnu, q.n_spec, 1, cdf, bp1)) FAIL("source spectrum has zero integral");
                        t[7] = (double)B.put(q.spec_nu, q.n_spec); t[8] = (double)B.put(cdf); t[9] = (double)B.put(bp1);
                    } else if (q.spectrum_type != 2) FAIL("Spot cannot have LTE spectrum");
                    if (pr->config.monochromatic) {
                        std::vector<double> mp(pr->config.n_frequencies);
                        for (int f = 0; f < pr->config.n_frequencies; f++)
                            mp[f] = q.spectrum_type == 1 ? interp_log_pdf(q.spec_nu, q.spec_fnu, 1, q.n_spec, pr->config.frequencies[f])
                                                         : normalized_B_nu(pr->config.frequencies[f], q.temperature);
                        t[10] = (double)B.put(mp);
                    }
                }
                S.n_spots = ns;
                soff[i].spot_tab = B.put(tab); soff[i].have_spots = true;
            }
            if (s.type == 4) {      // map: source_type.f90:190-199, set_pdf(luminosity_map, map) over all cells
                if (!s.map) FAIL("map source needs a luminosity map");
                const size_t nc = h->n_cells;
                std::vector<double> cdf(nc);
                double tot = 0.0, cc = 0.0;
                for (size_t k = 0; k < nc; k++) tot += s.map[k];
                if (!(tot > 0.0)) FAIL("luminosity map is zero everywhere");
                for (size_t k = 0; k < nc; k++) { cc += s.map[k] / tot; cdf[k] = cc; }
                for (size_t k = 0; k < nc; k++) cdf[k] /= cc;
                soff[i].map_cdf = B.put(cdf); soff[i].have_map = true;
            }
            for (int k = 0; k < 6; k++) S.box[k] = s.box[k];
            if (s.type == 6) {   // face pdf ~ face areas: source_type.f90:233-237
                double dx = s.box[1] - s.box[0], dy = s.box[3] - s.box[2], dz = s.box[5] - s.box[4];
                double a[6] = {dy * dz, dy * dz, dz * dx, dz * dx, dx * dy, dx * dy}, cc = 0.0, tot = 0.0;
                for (int k = 0; k < 6; k++) tot += a[k];
                for (int k = 0; k < 6; k++) { cc += a[k] / tot; S.face_cdf[k] = cc; }
                for (int k = 0; k < 6; k++) S.face_cdf[k] /= cc;
            }
            S.pos[0] = s.position[0]; S.pos[1] = s.position[1]; S.pos[2] = s.position[2];
            S.temperature = s.temperature; S.spectrum_type = s.spectrum_type; S.n_spec = s.n_spec;
            S.lum_pdf = src_lum[i] / h->energy_total;
            c += S.lum_pdf; S.lum_cdf = c;
            soff[i].have = false;
            if (s.spectrum_type == 1) {
                for (int k = 0; k + 1 < s.n_spec; k++)
                    if (s.spec_nu[k + 1] < s.spec_nu[k]) FAIL("spectrum frequency should be monotonically increasing");
                std::vector<double> cdf, bp1;
                if (!build_log_pdf(s.spec_nu, s.spec_fnu, s.n_spec, 1, cdf, bp1)) FAIL("source spectrum has zero integral");
                soff[i].x = B.put(s.spec_nu, s.n_spec); soff[i].cdf = B.put(cdf); soff[i].bp1 = B.put(bp1);
                soff[i].have = true;
            } else if (s.spectrum_type == 3 && s.type == 4) {
                // 'lte': the emissivity of the dust in the emitting cell
            } else if (s.spectrum_type != 2)
                FAIL(std::string(s.type == 5 ? "External spherical source" : s.type == 6 ? "External box source" : s.type == 2 ? "Spherical source" : s.type == 7 ? "Plane parallel" : s.type == 8 ? "Point source collection" : "Point source") + " cannot have LTE spectrum");
        }
        for (int i = 0; i < pr->n_sources; i++) hs[i].lum_cdf /= c;
    }

    // peeled image groups: images_peeled.f90:272-380, image_type.f90:153-335
    // the binned image group (images_binned.f90:42-56), if any, is one more image group after the peeled ones:
    // n_theta x n_phi "views", never peeled into (P.n_peeled stays the number of peeled groups)
    const int n_groups = pr->n_peeled + (pr->binned ? 1 : 0);
    std::vector<hyp_peeled_desc> pdesc(pr->peeled, pr->peeled + pr->n_peeled);
    std::vector<double> binned_angles;
    if (pr->binned) {
        if (pr->config.monochromatic) FAIL("can't use binned images in exact wavelength mode");                       // setup_rt.f90:328
        if (pr->config.forced_first_interaction) FAIL("can't use binned images with forced first interaction");      // :329
        if (pr->n_binned_theta < 1 || pr->n_binned_phi < 1) FAIL("n_theta and n_phi should be positive");
        hyp_peeled_desc bd = *pr->binned;
        bd.n_view = pr->n_binned_theta * pr->n_binned_phi; bd.inside_observer = 0;
        binned_angles.assign((size_t)bd.n_view, 0.0);
        bd.theta = binned_angles.data(); bd.phi = binned_angles.data();
        pdesc.push_back(bd);
    }
    h->h_peeled.resize(n_groups);
    std::vector<PeeledOffsets> poff(n_groups);
    std::vector<int> ray_groups;
    h->sed_off.assign(n_groups, 0); h->img_off.assign(n_groups, 0);
    h->sed_n.assign(n_groups, 0); h->img_n.assign(n_groups, 0);
    size_t img_total = 0;
    int views_total = 0;
    for (int g = 0; g < n_groups; g++) {
        const hyp_peeled_desc &in = pdesc[g];
        DPeeled &G = h->h_peeled[g];
        std::memset(&G, 0, sizeof(G));
        G.view_base = views_total;
        if (g < pr->n_peeled && in.n_view > 0) views_total += in.n_view;
        if (in.inside_observer) {       // images_peeled.f90:312-315, 356-363
            if (in.compute_image && in.x_min < in.x_max) FAIL("longitudes should increase towards the left for inside observers");
            if (in.compute_sed) FAIL("computing SEDs for inside observers is not supported");
        }
        if (in.n_view < 1) FAIL("n_view should be a positive integer");
        G.n_view = in.n_view; G.ignore_optical_depth = in.ignore_optical_depth;
        G.compute_image = in.compute_image; G.compute_sed = in.compute_sed;
        G.n_x = in.n_x; G.n_y = in.n_y; G.n_ap = in.n_ap; G.n_nu = in.n_nu;
        if (pr->config.monochromatic) {     // image_type.f90:243-258
            if (in.inu_min < 1 || in.inu_min > pr->config.n_frequencies) FAIL("inu_min value is out of range");
            if (in.inu_max < 1 || in.inu_max > pr->config.n_frequencies) FAIL("inu_max value is out of range");
            G.n_nu = in.inu_max - in.inu_min + 1; G.inu_min = in.inu_min;
            if (G.n_nu != in.n_nu) FAIL("n_nu of a monochromatic image group should be inu_max - inu_min + 1");
        }
        G.track_origin = in.track_origin; G.track_n_scat = in.track_n_scat; G.uncertainties = in.uncertainties;
        G.n_stokes = in.compute_stokes ? 4 : 1;
        switch (in.track_origin) {
        case 0: G.n_orig = 1; break;
        case 1: G.n_orig = 4; break;
        case 2: G.n_orig = 2 * (pr->n_sources + pr->n_dust); break;
        case 3: G.n_orig = 4 + 2 * in.track_n_scat; break;
        default: FAIL("unknown track_origin flag");
        }
        G.x_min = in.x_min; G.x_max = in.x_max; G.y_min = in.y_min; G.y_max = in.y_max;
        G.ap_min = in.ap_min; G.ap_max = in.ap_max;
        G.log10_nu_min = std::log10(in.nu_min); G.log10_nu_max = std::log10(in.nu_max);
        if (in.compute_sed) { G.log10_ap_min = std::log10(in.ap_min); G.log10_ap_max = std::log10(in.ap_max); }
        G.d_min = in.d_min; G.d_max = in.d_max;
        G.inside_observer = in.inside_observer ? 1 : 0;
        if (in.inside_observer && G.d_min < 0.0) G.d_min = 0.0;
        for (int k = 0; k < 3; k++) G.origin[k] = in.peeloff_origin[k];
        std::vector<double> view((size_t)in.n_view * 4);
        for (int v = 0; v < in.n_view; v++) {
            double t = in.theta[v] * HYP_PI / 180.0, f = in.phi[v] * HYP_PI / 180.0;
            view[4 * v + 0] = std::cos(t); view[4 * v + 1] = std::sin(t);
            view[4 * v + 2] = std::cos(f); view[4 * v + 3] = std::sin(f);
        }
        poff[g].view = B.put(view);
        if (in.use_filters) {       // image_type.f90:173-181,285-291; images_peeled.f90:349-351
            if (pr->config.monochromatic) FAIL("cannot use filters in monochromatic mode");
            if (pr->config.raytracing && g < pr->n_peeled) FAIL("filter convolution cannot be used with raytracing");
            if (!in.filt_n || !in.filt_nu || !in.filt_tr) FAIL("filter tables are missing");
            std::vector<double> off(in.n_nu + 1, 0.0);
            for (int i = 0; i < in.n_nu; i++) {
                if (in.filt_n[i] < 2) FAIL("a filter needs at least two points");
                off[i + 1] = off[i] + in.filt_n[i];
            }
            poff[g].filt_off = B.put(off);
            poff[g].filt_nu = B.put(in.filt_nu, (int)off[in.n_nu]);
            poff[g].filt_tr = B.put(in.filt_tr, (int)off[in.n_nu]);
            G.use_filters = 1;
        }
        if (pr->config.raytracing && g < pr->n_peeled) ray_groups.push_back(g);
        if (in.compute_sed) {
            h->sed_n[g] = (size_t)G.n_stokes * G.n_orig * in.n_view * in.n_ap * in.n_nu;
            h->sed_off[g] = img_total; img_total += 2 * h->sed_n[g];
        }
        if (in.compute_image) {
            h->img_n[g] = (size_t)G.n_stokes * G.n_orig * in.n_view * in.n_y * in.n_x * in.n_nu;
            h->img_off[g] = img_total; img_total += 2 * h->img_n[g];
        }
    }

    // Raytracing caches (images_peeled.f90:422-538): source spectra, dust emissivities and opacities
    // binned on each group's frequency grid with get_spectrum_binned (source_type.f90:1118-1172),
    // get_j_nu_binned and get_chi_nu_binned (dust_type_4elem.f90:722-750, 793-818).
    int nj_stride = 1;
    for (int d = 0; d < pr->n_dust; d++) nj_stride = std::max(nj_stride, pr->dust[d].n_jnu);
    if (!ray_groups.empty() && pr->config.monochromatic) {
        // use_exact_nu: get_spectrum_interp (source_type.f90:1098-1116), get_j_nu_interp / get_chi_nu_interp
        // (dust_type_4elem.f90:708-720, 780-791) at the group's own frequencies
        for (int g : ray_groups) {
            const int nn = h->h_peeled[g].n_nu;
            const double *nu = pr->config.frequencies + (pdesc[g].inu_min - 1);
            std::vector<double> spec((size_t)pr->n_sources * nn, 0.0), em((size_t)pr->n_dust * nj_stride * nn, 0.0), chi((size_t)pr->n_dust * nn, 0.0);
            for (int is = 0; is < pr->n_sources; is++) {
                const hyp_source_desc &src = pr->sources[is];
                for (int i = 0; i < nn; i++)
                    spec[(size_t)is * nn + i] = src.spectrum_type == 1 ? interp_log_pdf(src.spec_nu, src.spec_fnu, 1, src.n_spec, nu[i])
                                              : src.spectrum_type == 2 ? normalized_B_nu(nu[i], src.temperature) : 0.0;     // lte: the packets carry the dust emissivity
            }
            for (int d = 0; d < pr->n_dust; d++) {
                const hyp_dust_desc &in = pr->dust[d];
                for (int j = 0; j < in.n_jnu; j++)
                    for (int i = 0; i < nn; i++)
                        em[((size_t)d * nj_stride + j) * nn + i] = std::log10(interp_log_pdf(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu, nu[i]));
                for (int i = 0; i < nn; i++) {
                    double c = 0.0;
                    if (nu[i] >= in.nu[0] && nu[i] <= in.nu[in.n_nu - 1]) {
                        int j = in.n_nu - 2;
                        if (nu[i] != in.nu[in.n_nu - 1]) { int jl = 0, ju = in.n_nu - 1; while (ju - jl > 1) { int jm = (ju + jl) >> 1; if (nu[i] >= in.nu[jm]) jl = jm; else ju = jm; } j = jl; }
                        const double y1 = in.chi[j], y2 = in.chi[j + 1];
                        if (y1 > 0.0 && y2 > 0.0) {
                            const double f = (std::log10(nu[i]) - std::log10(in.nu[j])) / (std::log10(in.nu[j + 1]) - std::log10(in.nu[j]));
                            c = std::pow(10.0, std::log10(y1) + f * (std::log10(y2) - std::log10(y1)));
                        } else c = y1 + (nu[i] - in.nu[j]) / (in.nu[j + 1] - in.nu[j]) * (y2 - y1);
                    }
                    chi[(size_t)d * nn + i] = c;
                }
            }
            poff[g].src_spec = B.put(spec); poff[g].dust_em = B.put(em); poff[g].dust_chi = B.put(chi);
        }
    } else if (!ray_groups.empty()) {
        const double l0 = std::log10(3.e9), l1 = std::log10(3.e16);
        const int nb = (int)std::ceil((l1 - l0) * 100000);
        std::vector<double> bnu, bfnu;
        std::vector<std::vector<double>> lo(n_groups), hi(n_groups), spec(n_groups), em(n_groups), chi(n_groups);
        for (int g : ray_groups) {
            const DPeeled &G = h->h_peeled[g];
            const int nn = G.n_nu;
            lo[g].resize(nn); hi[g].resize(nn);
            for (int i = 0; i < nn; i++) {
                lo[g][i] = std::pow(10.0, G.log10_nu_min + (G.log10_nu_max - G.log10_nu_min) * (double)i / (double)nn);
                hi[g][i] = std::pow(10.0, G.log10_nu_min + (G.log10_nu_max - G.log10_nu_min) * (double)(i + 1) / (double)nn);
            }
            spec[g].assign((size_t)pr->n_sources * nn, 0.0);
            em[g].assign((size_t)pr->n_dust * nj_stride * nn, 0.0);
            chi[g].assign((size_t)pr->n_dust * nn, 0.0);
        }
        for (int is = 0; is < pr->n_sources; is++) {
            const hyp_source_desc &src = pr->sources[is];
            const double *x, *y; int n;
            if (src.spectrum_type == 3) continue;       // lte: the packets carry the dust emissivity
            if (src.spectrum_type == 1) { x = src.spec_nu; y = src.spec_fnu; n = src.n_spec; }
            else {
                // blackbody on 100000 points per decade between 3e9 and 3e16 Hz, normalized_B_nu :1088-1096
                if (bnu.empty()) {
                    bnu.resize(nb); bfnu.resize(nb);
                    for (int k = 0; k < nb; k++) bnu[k] = std::pow(10.0, (double)k / (double)(nb - 1) * (l1 - l0) + l0);
                }
                const double a = 2.0 * HYP_H_CGS / HYP_C_CGS / HYP_C_CGS / HYP_STEF_BOLTZ * HYP_PI, b = HYP_H_CGS / HYP_K_CGS;
                const double T = src.temperature, T4 = T * T * T * T;
                for (int k = 0; k < nb; k++) bfnu[k] = a * bnu[k] * bnu[k] * bnu[k] / (std::exp(b * bnu[k] / T) - 1.0) / T4;
                x = bnu.data(); y = bfnu.data(); n = nb;
            }
            const double tot = integral_loglog_all(x, y, 1, n);
            for (int g : ray_groups) {
                const int nn = h->h_peeled[g].n_nu;
                for (int i = 0; i < nn; i++) spec[g][(size_t)is * nn + i] = integral_loglog_range(x, y, 1, n, lo[g][i], hi[g][i]) / tot;
            }
        }
        for (int d = 0; d < pr->n_dust; d++) {
            const hyp_dust_desc &in = pr->dust[d];
            for (int j = 0; j < in.n_jnu; j++) {
                const double tot = integral_loglog_all(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu);
                for (int g : ray_groups) {
                    const int nn = h->h_peeled[g].n_nu;
                    for (int i = 0; i < nn; i++)
                        em[g][((size_t)d * nj_stride + j) * nn + i] =
                            std::log10(integral_loglog_range(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu, lo[g][i], hi[g][i]) / tot);
                }
            }
            for (int g : ray_groups) {
                const int nn = h->h_peeled[g].n_nu;
                for (int i = 0; i < nn; i++)
                    chi[g][(size_t)d * nn + i] = integral_loglog_range(in.nu, in.chi, 1, in.n_nu, lo[g][i], hi[g][i]) / (hi[g][i] - lo[g][i]);
            }
        }
        for (int g : ray_groups) { poff[g].src_spec = B.put(spec[g]); poff[g].dust_em = B.put(em[g]); poff[g].dust_chi = B.put(chi[g]); }
    }

    // monochromatic mode: emission probability of every source and of every emissivity row at the run's frequencies
    size_t mono_src_off = 0;
    if (pr->config.monochromatic) {
        const int nf = pr->config.n_frequencies;
        const double *fr = pr->config.frequencies;
        h->frequencies.assign(fr, fr + nf);
        std::vector<double> sp((size_t)pr->n_sources * nf);
        for (int is = 0; is < pr->n_sources; is++) {
            const hyp_source_desc &src = pr->sources[is];
            for (int i = 0; i < nf; i++)
                sp[(size_t)is * nf + i] = src.spectrum_type == 1 ? interp_log_pdf(src.spec_nu, src.spec_fnu, 1, src.n_spec, fr[i])
                                        : src.spectrum_type == 2 ? normalized_B_nu(fr[i], src.temperature) : 0.0;
        }
        mono_src_off = B.put(sp);
        for (int d = 0; d < pr->n_dust; d++) {
            const hyp_dust_desc &in = pr->dust[d];
            std::vector<double> lp((size_t)in.n_jnu * nf);
            for (int j = 0; j < in.n_jnu; j++)
                for (int i = 0; i < nf; i++) {
                    const double pv = interp_log_pdf(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu, fr[i]);
                    lp[(size_t)j * nf + i] = pv == 0.0 ? -INFINITY : std::log10(pv);
                }
            doff[d].mono_prob = B.put(lp);
        }
    }

    // ---- device allocations ----
    HIPC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPC(hipEventCreate(&h->ev0)); HIPC(hipEventCreate(&h->ev1));
    HIPC(hipEventCreate(&h->ev2)); HIPC(hipEventCreate(&h->ev3));
    HIPC(hipMalloc(&h->d_blob, sizeof(double) * B.h.size()));
    HIPC(hipMemcpy(h->d_blob, B.h.data(), sizeof(double) * B.h.size(), hipMemcpyHostToDevice));
    const double *db = h->d_blob;
    for (int a = 0; a < 3 && is_car; a++) { P.w[a] = db + w_off[a]; P.ew[a] = db + ew_off[a]; }
    if (pr->config.monochromatic) {
        P.n_frequencies = pr->config.n_frequencies; P.mono_threshold = pr->config.monochromatic_energy_threshold;
        P.mono_src_prob = db + mono_src_off; P.mono_which = 0; P.mono_inu = 0;
    }
    if (is_polar) {
        P.wr2 = db + polar_off[0]; P.wtanp = db + polar_off[4];
        if (is_sph) { P.wtant = db + polar_off[1]; P.wtant2 = db + polar_off[2]; P.wcost = db + polar_off[3]; }
        P.midplane = midplane; P.n_dim = n[2] == 1 ? 2 : 3;
    }
    if (is_vor) {
        const size_t nc = h->n_cells, nn = (size_t)pr->grid.vor_idx[nc];
        std::vector<double> vol(nc);
        for (size_t i = 0; i < nc; i++) vol[i] = pr->grid.vor_volume[i] < 0.0 ? 0.0 : pr->grid.vor_volume[i];
        HIPC(hipMalloc(&h->d_vor_sites, sizeof(double) * 3 * nc));
        HIPC(hipMemcpy(h->d_vor_sites, pr->grid.vor_sites, sizeof(double) * 3 * nc, hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_vor_volume, sizeof(double) * nc));
        HIPC(hipMemcpy(h->d_vor_volume, vol.data(), sizeof(double) * nc, hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_vor_idx, sizeof(int) * (nc + 1)));
        HIPC(hipMemcpy(h->d_vor_idx, pr->grid.vor_idx, sizeof(int) * (nc + 1), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_vor_neigh, sizeof(int) * (nn ? nn : 1)));
        HIPC(hipMemcpy(h->d_vor_neigh, pr->grid.vor_neighs, sizeof(int) * nn, hipMemcpyHostToDevice));
        {
            std::vector<VorWall> walls(nn ? nn : 1);
            for (size_t k = 0; k < nn; k++) {
                const int nb = pr->grid.vor_neighs[k];
                VorWall &w = walls[k];
                w.nb = nb; w.loc = 0; w.x = w.y = w.z = 0.0;
                if (nb >= 0) { w.x = pr->grid.vor_sites[3 * (size_t)nb]; w.y = pr->grid.vor_sites[3 * (size_t)nb + 1]; w.z = pr->grid.vor_sites[3 * (size_t)nb + 2]; }
            }
            HIPC(hipMalloc(&h->d_vor_walls, sizeof(VorWall) * walls.size()));
            HIPC(hipMemcpy(h->d_vor_walls, walls.data(), sizeof(VorWall) * walls.size(), hipMemcpyHostToDevice));
        }
        HIPC(hipMalloc(&h->d_vor_seed, sizeof(int) * vor_seed.size()));
        HIPC(hipMemcpy(h->d_vor_seed, vor_seed.data(), sizeof(int) * vor_seed.size(), hipMemcpyHostToDevice));
        if (pr->grid.vor_bb) {
            HIPC(hipMalloc(&h->d_vor_bb, sizeof(double) * 6 * nc));
            HIPC(hipMemcpy(h->d_vor_bb, pr->grid.vor_bb, sizeof(double) * 6 * nc, hipMemcpyHostToDevice));
        }
        h->h_vor_sites.assign(pr->grid.vor_sites, pr->grid.vor_sites + 3 * nc);
        h->h_vor_idx.assign(pr->grid.vor_idx, pr->grid.vor_idx + nc + 1);
        h->h_vor_neigh.assign(pr->grid.vor_neighs, pr->grid.vor_neighs + nn);
        P.vor_bb = h->d_vor_bb;
        P.vor_sites = h->d_vor_sites; P.vor_volume = h->d_vor_volume; P.vor_idx = h->d_vor_idx;
        P.vor_neigh = h->d_vor_neigh; P.vor_seed = h->d_vor_seed; P.vor_g = vor_g; P.vor_walls = h->d_vor_walls;
        for (int k = 0; k < 6; k++) P.vor_box[k] = pr->grid.vor_box[k];
    }
    if (is_amr) {
        HIPC(hipMalloc(&h->d_amr_grids, sizeof(AmrGrid) * amr_grids.size()));
        HIPC(hipMemcpy(h->d_amr_grids, amr_grids.data(), sizeof(AmrGrid) * amr_grids.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_amr_go, sizeof(int) * amr_go.size()));
        HIPC(hipMemcpy(h->d_amr_go, amr_go.data(), sizeof(int) * amr_go.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_amr_walls, sizeof(double) * amr_walls.size()));
        HIPC(hipMemcpy(h->d_amr_walls, amr_walls.data(), sizeof(double) * amr_walls.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_amr_cell_grid, sizeof(int) * amr_cell_grid.size()));
        HIPC(hipMemcpy(h->d_amr_cell_grid, amr_cell_grid.data(), sizeof(int) * amr_cell_grid.size(), hipMemcpyHostToDevice));
        P.amr_grids = h->d_amr_grids; P.amr_go = h->d_amr_go; P.amr_walls = h->d_amr_walls; P.amr_cell_grid = h->d_amr_cell_grid;
        P.amr_eps = amr_eps; P.n_amr_grids = (int)amr_grids.size(); P.n_amr_level1 = amr_level1;
        h->h_amr_grids = amr_grids; h->h_amr_go = amr_go;
    }
    if (is_oct) {
        HIPC(hipMalloc(&h->d_oct_cells, sizeof(OctCell) * oct_cells.size()));
        HIPC(hipMemcpy(h->d_oct_cells, oct_cells.data(), sizeof(OctCell) * oct_cells.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_oct_children, sizeof(int) * oct_children.size()));
        HIPC(hipMemcpy(h->d_oct_children, oct_children.data(), sizeof(int) * oct_children.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_oct_neigh, sizeof(int) * oct_neigh.size()));
        HIPC(hipMemcpy(h->d_oct_neigh, oct_neigh.data(), sizeof(int) * oct_neigh.size(), hipMemcpyHostToDevice));
        P.oct_cells = h->d_oct_cells; P.oct_children = h->d_oct_children; P.oct_neigh = h->d_oct_neigh;
        h->h_oct_cells = oct_cells; h->h_oct_children = oct_children; h->h_oct_neigh = oct_neigh;
    }
    for (int d = 0; d < pr->n_dust; d++) {
        DDust &D = P.dust[d]; const DustOffsets &O = doff[d];
        D.nu = db + O.nu; D.log10_nu = db + O.log10_nu; D.chi = db + O.chi; D.albedo = db + O.albedo;
        D.log10_chi = db + O.log10_chi; D.log10_albedo = db + O.log10_albedo; D.mu = db + O.mu;
        D.P1 = db + O.P1; D.P2 = db + O.P2; D.P3 = db + O.P3; D.P4 = db + O.P4;
        D.P1_cdf = db + O.P1_cdf; D.P2_cdf = db + O.P2_cdf;
        D.emiss_x = db + O.emiss_x; D.emiss_cdf = db + O.emiss_cdf; D.emiss_bp1 = db + O.emiss_bp1;
        D.emiss_coarse = db + O.emiss_coarse; D.n_ecoarse = (D.n_enu + HYP_COARSE - 1) / HYP_COARSE; D.pad2 = 0;
        D.jnu_var = db + O.jnu_var; D.log10_jnu_var = db + O.log10_jnu_var;
        D.mo_e = O.have_mo_e ? db + O.mo_e : nullptr;
        D.mo_chi_ross = O.have_mo_chi ? db + O.mo_chi_ross : nullptr;
        D.mo_kappa_planck = (O.have_mrw || O.have_pda) ? db + O.mo_kappa_planck : nullptr;
        D.mo_chi_inv_planck = O.have_mrw ? db + O.mo_chi_inv_planck : nullptr;
        D.mono_log10_prob = pr->config.monochromatic ? db + O.mono_prob : nullptr;
        D.bnu_cdf = O.have_mrw ? db + O.bnu_cdf : nullptr; D.bnu_bp1 = O.have_mrw ? db + O.bnu_bp1 : nullptr;
        D.bnu_coarse = O.have_mrw ? db + O.bnu_coarse : nullptr;
    }
    P.mrw = pr->config.mrw ? 1 : 0; P.pad5 = 0;
    P.n_inter_mrw_max = pr->config.n_inter_mrw_max; P.mrw_gamma = pr->config.mrw_gamma;
    P.mrw_x = P.mrw ? db + mrw_x_off : nullptr; P.mrw_y = P.mrw ? db + mrw_y_off : nullptr;
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have) { hs[i].spec_x = db + soff[i].x; hs[i].spec_cdf = db + soff[i].cdf; hs[i].spec_bp1 = db + soff[i].bp1; }
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have_points) { hs[i].points = db + soff[i].points; hs[i].point_cdf = db + soff[i].point_cdf; }
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have_map) hs[i].map_cdf = db + soff[i].map_cdf;
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have_spots) { hs[i].spot_tab = db + soff[i].spot_tab; hs[i].spot_blob = db; }
    HIPC(hipMalloc(&h->d_sources, sizeof(DSource) * hs.size()));
    HIPC(hipMemcpy(h->d_sources, hs.data(), sizeof(DSource) * hs.size(), hipMemcpyHostToDevice));
    P.sources = h->d_sources;

    const size_t ne = h->n_elem;
    HIPC(hipMalloc(&h->d_density, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_specific_energy, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_scratch, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_jnu_id, sizeof(int) * ne));
    HIPC(hipMalloc(&h->d_jnu_frac, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_energy_abs_tot, sizeof(double) * HYP_MAXD));
    HIPC(hipMalloc(&h->d_counter, sizeof(unsigned long long)));
    HIPC(hipMalloc(&h->d_err, sizeof(int)));
    HIPC(hipMalloc(&h->d_err_data, sizeof(double) * 4));
    HIPC(hipMemset(h->d_err, 0, sizeof(int)));
    // The accumulator block that the ranks all-reduce: [sums | tail | n_photons as doubles | spectrum sums]
    h->count_photons = pr->config.count_photons || pr->config.pda;
    h->pda = pr->config.pda != 0;
    h->n_bins = pr->config.n_spectrum_bins > 0 ? pr->config.n_spectrum_bins : 0;
    if (h->pda && !is_car) { /* grid_pda_disabled.f90: nothing to solve, but the counters are kept */ }
    h->ext_nphot = ne + TAIL_SIZE;
    h->ext_spec = h->ext_nphot + (h->count_photons ? h->n_cells : 0);
    h->block_doubles = h->ext_spec + (size_t)h->n_bins * ne;
    h->accum_stride = ((h->block_doubles + 31) / 32) * 32;
    h->accum_copies_alloc = h->n_bins ? 1 : 8;
    if (h->n_bins) h->accum_copies = 1;        // the spectrum planes are not replicated; their atomics dominate anyway
    if (h->count_photons) {
        HIPC(hipMalloc(&h->d_nphot, sizeof(unsigned int) * h->n_cells));
        HIPC(hipMalloc(&h->d_nphot_inexact, sizeof(int)));
        HIPC(hipMemset(h->d_nphot, 0, sizeof(unsigned int) * h->n_cells));
        HIPC(hipMemset(h->d_nphot_inexact, 0, sizeof(int)));
        P.n_photons = h->d_nphot; P.visit_tab = nullptr; P.nphot_inexact = h->d_nphot_inexact; P.count_photons = 1;
    }
    if (h->n_bins) {     // grid_physics_3d.f90:124-143,269-282,326-348
        const int nb = h->n_bins;
        if (!pr->config.spectrum_bin_edges) FAIL("specific_energy_spectrum_bin_edges should be present in the input when output_specific_energy_spectrum is enabled");
        h->spectrum_edges.assign(pr->config.spectrum_bin_edges, pr->config.spectrum_bin_edges + nb + 1);
        std::vector<double> le(nb + 1);
        for (int b = 0; b <= nb; b++) {
            if (b && !(h->spectrum_edges[b] > h->spectrum_edges[b - 1])) FAIL("specific_energy_spectrum_bin_edges should be strictly increasing");
            le[b] = std::log10(h->spectrum_edges[b]);
        }
        HIPC(hipMalloc(&h->d_log_edges, sizeof(double) * (nb + 1)));
        HIPC(hipMemcpy(h->d_log_edges, le.data(), sizeof(double) * (nb + 1), hipMemcpyHostToDevice));
        h->nj_max = 1;
        for (int d = 0; d < pr->n_dust; d++) h->nj_max = std::max(h->nj_max, pr->dust[d].n_jnu);
        // get_j_nu_bin_fractions (dust_type_4elem.f90:752-778): share of each emissivity row in each bin
        std::vector<double> frac((size_t)pr->n_dust * h->nj_max * nb, 0.0);
        for (int d = 0; d < pr->n_dust; d++) {
            const hyp_dust_desc &in = pr->dust[d];
            for (int iv = 0; iv < in.n_jnu; iv++) {
                double *f = frac.data() + ((size_t)d * h->nj_max + iv) * nb;
                for (int b = 0; b < nb; b++)
                    f[b] = integral_loglog_range(in.emiss_nu, in.emiss_jnu + iv, in.n_jnu, in.n_enu, h->spectrum_edges[b], h->spectrum_edges[b + 1]);
                const double norm = integral_loglog_all(in.emiss_nu, in.emiss_jnu + iv, in.n_jnu, in.n_enu);
                if (norm > 0.0) for (int b = 0; b < nb; b++) f[b] /= norm;
            }
        }
        HIPC(hipMalloc(&h->d_bin_frac, sizeof(double) * frac.size()));
        HIPC(hipMemcpy(h->d_bin_frac, frac.data(), sizeof(double) * frac.size(), hipMemcpyHostToDevice));
        // specific_energy_spectrum starts at the minimum specific energy unless an initial specific energy was given
        // (then it starts at 0): grid_physics_3d.f90:143,215-253
        std::vector<double> sp((size_t)nb * ne, 0.0);
        if (!pr->specific_energy || pr->config.specific_energy_type == 1)
            for (int b = 0; b < nb; b++) for (size_t ic = 0; ic < h->n_cells; ic++) for (int d = 0; d < pr->n_dust; d++)
                sp[((size_t)b * h->n_cells + ic) * pr->n_dust + d] = pr->dust[d].minimum_specific_energy;
        HIPC(hipMalloc(&h->d_spec, sizeof(double) * sp.size()));
        HIPC(hipMemcpy(h->d_spec, sp.data(), sizeof(double) * sp.size(), hipMemcpyHostToDevice));
        P.n_bins = nb; P.nj_max = h->nj_max; P.log_nu_edges = h->d_log_edges; P.jnu_bin_frac = h->d_bin_frac;
    }
    HIPC(hipMalloc(&h->d_accum, sizeof(double) * h->accum_stride * h->accum_copies_alloc));
    HIPC(hipMemset(h->d_accum, 0, sizeof(double) * h->accum_stride * h->accum_copies_alloc));

    if (img_total > 0) {
        h->img_accum_n = img_total + TAIL_SIZE;
        HIPC(hipMalloc(&h->d_img_accum, sizeof(double) * h->img_accum_n));
        HIPC(hipMemset(h->d_img_accum, 0, sizeof(double) * h->img_accum_n));
    }
    for (int g = 0; g < n_groups; g++) {
        DPeeled &G = h->h_peeled[g];
        G.view = db + poff[g].view;
        if (G.use_filters) { G.filt_off = db + poff[g].filt_off; G.filt_nu = db + poff[g].filt_nu; G.filt_tr = db + poff[g].filt_tr; }
        if (pr->config.raytracing) {
            G.src_spec = db + poff[g].src_spec; G.dust_log10_em = db + poff[g].dust_em; G.dust_chi = db + poff[g].dust_chi;
            G.nj_stride = nj_stride;
        }
        if (h->sed_n[g]) { G.sed = h->d_img_accum + h->sed_off[g]; G.sed2 = G.sed + h->sed_n[g]; }
        if (h->img_n[g]) { G.img = h->d_img_accum + h->img_off[g]; G.img2 = G.img + h->img_n[g]; }
    }
    {
        bool plain = !pr->config.mrw && !pr->config.monochromatic && !pr->binned;
        for (int i = 0; i < pr->n_sources; i++) plain = plain && pr->sources[i].type == 1 && (pr->sources[i].spectrum_type == 1 || pr->sources[i].spectrum_type == 2);
        // (filters are the peel kernel's / deposit_images' business; inside observers are the peel kernel's, not the inline plain kernel's)
        h->plain_imaging = plain && h->n_dust <= 4;      // five to eight species: the general kernel only (hyp_geom.hip)
        {
            bool md = pr->config.monochromatic && !pr->binned && h->n_dust <= 4;      // (the modified random walk is not made in monochromatic launches: iter_final_mono.f90 has none)
            for (int i = 0; i < pr->n_sources; i++) md = md && pr->sources[i].type == 1 && (pr->sources[i].spectrum_type == 1 || pr->sources[i].spectrum_type == 2);
            h->mono_defer = md;
        }
        {
            // sources with a surface (spheres, limb darkening and re-absorption included; no spots) next to points: the imaging iteration
            // on the deferred schedule with the GEN kernels (hyp_defer.h) instead of the general kernel with inline peel-off
            // (any sources: the GEN kernels emit with the general emitter; what stays on final_kernel is MRW, binned images, inside
            // observers together with such sources, and more than four species)
            bool gd = !plain && !pr->config.monochromatic && !pr->binned && h->n_dust <= 4 && pr->n_sources > 0;      // (with the modified random walk: the MRWF instance)
            for (int g = 0; g < pr->n_peeled; g++) gd = gd && !pr->peeled[g].inside_observer;
            h->gen_defer = gd;
            // ... and the same sources in a monochromatic run (the Pascucci / Pinte benchmark models: a stellar sphere)
            bool mg = pr->config.monochromatic && !h->mono_defer && !pr->binned && h->n_dust <= 4 && pr->n_sources > 0;
            for (int g = 0; g < pr->n_peeled; g++) mg = mg && !pr->peeled[g].inside_observer;
            h->mono_gen_defer = mg;
        }
        h->inside_observers = false;
        for (int g = 0; g < pr->n_peeled; g++) h->inside_observers = h->inside_observers || pr->peeled[g].inside_observer;
        {
            bool lean = !pr->config.mrw && !pr->config.monochromatic && !pr->binned;
            for (int g = 0; g < pr->n_peeled; g++) lean = lean && !pr->peeled[g].inside_observer;
            (void)lean;
            h->lean_imaging = false;      // round 4: the lean kernel's problems image on the deferred schedule (GEN kernels); with gen_defer = 0 they run on the general kernel
        }
        bool simple = pr->n_sources > 0, ext = pr->n_sources > 0;
        for (int i = 0; i < pr->n_sources; i++) {
            const bool spec = pr->sources[i].spectrum_type == 1 || pr->sources[i].spectrum_type == 2;
            simple = simple && pr->sources[i].type == 1 && spec;
            ext = ext && (pr->sources[i].type == 1 || pr->sources[i].type == 5 || pr->sources[i].type == 6) && spec;
        }
        h->simple_sources = simple; h->ext_sources = ext;
    }
    P.n_views_total = views_total;
    P.binned = pr->binned ? pr->n_peeled : -1; P.n_bin_theta = pr->n_binned_theta; P.n_bin_phi = pr->n_binned_phi;
    if (n_groups > 0) {
        HIPC(hipMalloc(&h->d_peeled, sizeof(DPeeled) * n_groups));
        HIPC(hipMemcpy(h->d_peeled, h->h_peeled.data(), sizeof(DPeeled) * n_groups, hipMemcpyHostToDevice));
    }
    P.peeled = h->d_peeled;

    // density / specific energy: reference layout -> cell-major device layout
    {
        std::vector<double> dens(pr->density, pr->density + ne);
        if (is_oct)   // density is reset to zero in masked (refined) cells: grid_physics_3d.f90:152-160
            for (int d = 0; d < h->n_dust; d++)
                for (size_t ic = 0; ic < h->n_cells; ic++)
                    if (oct_cells[ic].refined) dens[(size_t)d * h->n_cells + ic] = 0.0;
        if (is_amr)   // mask = cells not covered by a finer grid: grid_geometry_amr.f90:489-496
            for (size_t ic = 0; ic < h->n_cells; ic++) {
                const AmrGrid &g = amr_grids[amr_cell_grid[ic]];
                const size_t l = ic - g.start;
                const int i1 = (int)(l % g.n[0]), i2 = (int)((l / g.n[0]) % g.n[1]), i3 = (int)(l / ((size_t)g.n[0] * g.n[1]));
                if (amr_go[g.go_off + ((size_t)(i3 + 1) * (g.n[1] + 2) + (i2 + 1)) * (g.n[0] + 2) + (i1 + 1)] != 0)
                    for (int d = 0; d < h->n_dust; d++) dens[(size_t)d * h->n_cells + ic] = 0.0;
            }
        if (is_vor)   // mask = volume > 0: grid_geometry_voronoi.f90:161-173
            for (int d = 0; d < h->n_dust; d++)
                for (size_t ic = 0; ic < h->n_cells; ic++)
                    if (!(pr->grid.vor_volume[ic] > 0.0)) dens[(size_t)d * h->n_cells + ic] = 0.0;
        HIPC(hipMemcpy(h->d_scratch, dens.data(), sizeof(double) * ne, hipMemcpyHostToDevice));
    }
    to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_density, h->n_cells, h->n_dust);
    HIPC(hipStreamSynchronize(h->stream));
    // grid_physics_3d.f90:176-253
    std::vector<double> se(ne);
    if (pr->specific_energy) {
        if (pr->config.specific_energy_type == 1) {
            HIPC(hipMalloc(&h->d_additional, sizeof(double) * ne));
            HIPC(hipMemcpy(h->d_scratch, pr->specific_energy, sizeof(double) * ne, hipMemcpyHostToDevice));
            to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_additional, h->n_cells, h->n_dust);
            HIPC(hipStreamSynchronize(h->stream));
            for (int d = 0; d < h->n_dust; d++)
                for (size_t ic = 0; ic < h->n_cells; ic++) se[(size_t)d * h->n_cells + ic] = pr->dust[d].minimum_specific_energy;
        } else {
            std::memcpy(se.data(), pr->specific_energy, sizeof(double) * ne);
        }
    } else {
        if (pr->config.specific_energy_type == 1) FAIL("cannot specify specific_energy_type since specific_energy was not given");
        for (int d = 0; d < h->n_dust; d++)
            for (size_t ic = 0; ic < h->n_cells; ic++) se[(size_t)d * h->n_cells + ic] = pr->dust[d].minimum_specific_energy;
    }
    HIPC(hipMemcpy(h->d_scratch, se.data(), sizeof(double) * ne, hipMemcpyHostToDevice));
    to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_specific_energy, h->n_cells, h->n_dust);
    HIPC(hipStreamSynchronize(h->stream));

    P.density = h->d_density;
    P.sum = h->d_accum;
    P.copy_stride = h->accum_stride;
    P.n_copies = 1;
    P.tail = h->d_accum + ne;
    P.sum_spec = h->n_bins ? h->d_accum + h->ext_spec : nullptr;
    P.jnu_id = h->d_jnu_id; P.jnu_frac = h->d_jnu_frac;
    P.specific_energy = h->d_specific_energy; P.energy_abs_tot = h->d_energy_abs_tot;
    P.energy_total = h->energy_total; P.peel_scattered_only = pr->config.raytracing ? 1 : 0;
    {   // geo%mask_map: cartesian_3d.f90:101, octree.f90:214-225, amr.f90:489-505, voronoi.f90:161-173
        std::vector<unsigned int> mask;
        mask.reserve(h->n_cells);
        for (size_t ic = 0; ic < h->n_cells; ic++) {
            bool valid = true;
            if (is_oct) valid = !oct_cells[ic].refined;
            else if (is_vor) valid = pr->grid.vor_volume[ic] > 0.0;
            else if (is_amr) {
                const AmrGrid &g = amr_grids[amr_cell_grid[ic]];
                const size_t l = ic - g.start;
                const int i1 = (int)(l % g.n[0]), i2 = (int)((l / g.n[0]) % g.n[1]), i3 = (int)(l / ((size_t)g.n[0] * g.n[1]));
                valid = amr_go[g.go_off + ((size_t)(i3 + 1) * (g.n[1] + 2) + (i2 + 1)) * (g.n[0] + 2) + (i1 + 1)] == 0;
            }
            if (valid) mask.push_back((unsigned int)ic);
        }
        P.n_masked = mask.size();
        if (pr->config.raytracing) {
            HIPC(hipMalloc(&h->d_mask_map, sizeof(unsigned int) * (mask.size() ? mask.size() : 1)));
            HIPC(hipMemcpy(h->d_mask_map, mask.data(), sizeof(unsigned int) * mask.size(), hipMemcpyHostToDevice));
            P.mask_map = h->d_mask_map;
        }
    }
    P.counter = h->d_counter; P.err = h->d_err; P.err_data = h->d_err_data;
    HIPC(hipMalloc(&h->d_problem, sizeof(DProblem)));
    HIPC(hipMemcpy(h->d_problem, &P, sizeof(DProblem), hipMemcpyHostToDevice));

    // check_energy_abs at set-up (grid_physics_3d.f90:277) + first jnu_var
    if (run_finish_kernel(h, 1, 1.0, nullptr)) { g_error = h->err; hyp_destroy(h); return 1; }
    HIPC(hipStreamSynchronize(h->stream));
#undef FAIL
#undef HIPC
    *out = h;
    return 0;
}

// Clusters of Voronoi cells for the tiled schedule (hyp_vtile.h): recursive coordinate bisection of the sites into groups
// of equal cell count whose tables (VtInfo in hyp_device.h: sites of the cluster's cells and of the cells across its
// boundary, one FP32 record and one link word per wall, one header word per cell), densities and accumulators fit the LDS
// budget of one walk workgroup.
static size_t vt_blob16(size_t n_own, size_t n_site, size_t n_wall)
{
    return 3 * ((n_site + 1) / 2) + n_wall + (n_wall + 3) / 4 + 2 * ((n_own + 3) / 4) + 3 * ((n_site - n_own + 3) / 4);
}

static int build_vor_clusters(hyp_handle h)
{
    const int nd = h->n_dust;
    if (h->vt_built_for == nd && h->d_vt_cluster) return 0;
    const size_t nc = h->n_cells;
    const double *S = h->h_vor_sites.data();
    const int *idx = h->h_vor_idx.data(), *nei = h->h_vor_neigh.data();
    if (h->h_vor_sites.size() != 3 * nc) return h->set_error("voronoi tables missing for the cluster builder");
    for (size_t i = 0; i < nc; i++) if (idx[i + 1] - idx[i] > 255) return h->set_error("a voronoi cell has more than 255 walls: no cluster-tiled schedule");
    const size_t budget = (size_t)h->vt_lds_kb * 1024;
    std::vector<int> perm(nc), cl_of(nc), cell_off;
    std::vector<int> local(nc, -1);          // index of a cell in the site table of the cluster being laid out (-1: not in it)
    struct Layout { std::vector<int> ghosts; size_t n_wall = 0; };
    std::vector<Layout> lay;
    int n_cl = 0;
    size_t max_lds = 0;
    // sites + per wall 20 bytes + header, densities, accumulators, and about as many ghost sites as own cells at these sizes
    double target = h->vt_cells > 0 ? (double)h->vt_cells : std::max(8.0, (double)budget / (24.0 * 2 + 20.0 * 16.5 + 4 + 16.0 * nd));
    for (int attempt = 0;; attempt++) {
        n_cl = (int)std::max<double>(1.0, std::ceil((double)nc / target));
        if (n_cl > HYP_TILE_MAX_BRICKS) return h->set_error("voronoi grid has too many cells for the cluster-tiled schedule");
        for (size_t i = 0; i < nc; i++) perm[i] = (int)i;
        cell_off.assign(n_cl + 1, 0);
        // iterative bisection: (first cell, number of cells, first cluster, number of clusters)
        struct Part { size_t lo, n; int c0, k; };
        std::vector<Part> stack{{0, nc, 0, n_cl}};
        while (!stack.empty()) {
            const Part p = stack.back(); stack.pop_back();
            if (p.k == 1) { cell_off[p.c0 + 1] = (int)p.n; for (size_t i = p.lo; i < p.lo + p.n; i++) cl_of[perm[i]] = p.c0; continue; }
            double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
            for (size_t i = p.lo; i < p.lo + p.n; i++)
                for (int a = 0; a < 3; a++) { const double x = S[3 * (size_t)perm[i] + a]; lo[a] = std::min(lo[a], x); hi[a] = std::max(hi[a], x); }
            int ax = 0;
            for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
            const int k1 = p.k / 2;
            const size_t n1 = (size_t)((double)p.n * k1 / p.k + 0.5);
            std::nth_element(perm.begin() + p.lo, perm.begin() + p.lo + n1, perm.begin() + p.lo + p.n,
                             [&](int a, int b) { const double xa = S[3 * (size_t)a + ax], xb = S[3 * (size_t)b + ax]; return xa < xb || (xa == xb && a < b); });
            stack.push_back({p.lo, n1, p.c0, k1});
            stack.push_back({p.lo + n1, p.n - n1, p.c0 + k1, p.k - k1});
        }
        for (int c = 0; c < n_cl; c++) cell_off[c + 1] += cell_off[c];
        // ghosts (cells of other clusters across a wall, each once, in the order met) and the LDS each cluster needs
        lay.assign(n_cl, Layout());
        std::vector<int> seen(nc, -1);
        for (size_t i = 0; i < nc; i++) {
            Layout &Lc = lay[cl_of[i]];
            Lc.n_wall += (size_t)(idx[i + 1] - idx[i]);
            for (int k = idx[i]; k < idx[i + 1]; k++) {
                const int nb = nei[k];
                if (nb >= 0 && cl_of[nb] != cl_of[i] && seen[nb] != cl_of[i]) { seen[nb] = cl_of[i]; Lc.ghosts.push_back(nb); }
            }
        }
        max_lds = 0;
        bool fits = true;
        for (int c = 0; c < n_cl; c++) {
            const size_t n_own = (size_t)(cell_off[c + 1] - cell_off[c]), n_site = n_own + lay[c].ghosts.size();
            const size_t lds = 16 * vt_blob16(n_own, n_site, lay[c].n_wall) + sizeof(double) * 2 * n_own * nd;
            max_lds = std::max(max_lds, lds);
            if (n_site > 65535 || lay[c].n_wall >= (1u << 20)) fits = false;
        }
        if (fits && max_lds <= budget) break;
        if (fits && h->vt_cells > 0 && max_lds <= (size_t)156 * 1024) break;       // a forced size may take a whole CU's LDS
        if (h->vt_cells > 0 || attempt > 60) return h->set_error("voronoi clusters do not fit in LDS");
        target *= std::min(0.95, 0.98 * (double)budget / (double)max_lds);
        if (target < 1.0) target = 1.0;
    }
    // members of each cluster in ascending cell order
    std::vector<int> members(nc), cursor(cell_off.begin(), cell_off.end() - 1), packed(nc);
    for (size_t i = 0; i < nc; i++) {
        const int c = cl_of[i], l = cursor[c]++ - cell_off[c];
        members[cell_off[c] + l] = (int)i;
        packed[i] = (c << 16) | l;
    }
    std::vector<VtInfo> info(n_cl);
    std::vector<VtGhost> ghosts;
    std::vector<int> adj((size_t)n_cl * VT_MAX_ADJ, -1);
    size_t total16 = 0;
    for (int c = 0; c < n_cl; c++) {
        VtInfo &I = info[c];
        I.n_own = cell_off[c + 1] - cell_off[c]; I.n_site = I.n_own + (int)lay[c].ghosts.size(); I.n_wall = (int)lay[c].n_wall;
        I.cell0 = cell_off[c]; I.ghost0 = (int)ghosts.size();
        if (total16 > 0x7fffffffull) return h->set_error("voronoi cluster tables too large");
        I.blob16 = (int)total16;
        total16 += vt_blob16((size_t)I.n_own, (size_t)I.n_site, (size_t)I.n_wall);
        int *ad = adj.data() + (size_t)c * VT_MAX_ADJ;
        for (int nb : lay[c].ghosts) {
            const int cn = cl_of[nb];
            int s = 0;
            while (s < VT_MAX_ADJ && ad[s] != cn && ad[s] != -1) s++;
            if (s < VT_MAX_ADJ) ad[s] = cn;
            ghosts.push_back(VtGhost{nb, s});
        }
    }
    std::vector<float4> blob(total16 ? total16 : 1, make_float4(0.f, 0.f, 0.f, 0.f));
    const double *B = h->hp.vor_box;
    for (int c = 0; c < n_cl; c++) {
        VtInfo &I = info[c];
        for (int j = 0; j < I.n_own; j++) local[members[I.cell0 + j]] = j;
        for (int gI = 0; gI < I.n_site - I.n_own; gI++) local[lay[c].ghosts[gI]] = I.n_own + gI;
        const int ns = (I.n_site + 1) & ~1;
        double *sx = (double *)(blob.data() + I.blob16), *sy = sx + ns, *sz = sy + ns;
        float4 *wrec = (float4 *)(sz + ns);
        uint32_t *wlink = (uint32_t *)(wrec + I.n_wall), *hdr = wlink + ((I.n_wall + 3) & ~3);
        const int ng = I.n_site - I.n_own, ngp = (ng + 3) & ~3;
        int *mem = (int *)(hdr + ((I.n_own + 3) & ~3)), *gcell = mem + ((I.n_own + 3) & ~3), *gpacked = gcell + ngp, *gadj = gpacked + ngp;
        for (int j = 0; j < I.n_own; j++) mem[j] = members[I.cell0 + j];
        for (int gI = 0; gI < ng; gI++) {
            const VtGhost &gh = ghosts[(size_t)I.ghost0 + gI];
            gcell[gI] = gh.cell; gpacked[gI] = packed[gh.cell]; gadj[gI] = gh.adj;
        }
        double rmax = 0.0, len_sum = 0.0; size_t len_n = 0;
        for (int j = 0; j < I.n_site; j++) {
            const int cell = j < I.n_own ? members[I.cell0 + j] : lay[c].ghosts[j - I.n_own];
            sx[j] = S[3 * (size_t)cell]; sy[j] = S[3 * (size_t)cell + 1]; sz[j] = S[3 * (size_t)cell + 2];
            for (int a = 0; a < 3; a++) rmax = std::max(rmax, std::fabs(S[3 * (size_t)cell + a]));
        }
        for (int a = 0; a < 6; a++) rmax = std::max(rmax, std::fabs(B[a]));
        // first pass: the scale (a power of two that brings the mean |n| to order one)
        for (int j = 0; j < I.n_own; j++) {
            const int cell = members[I.cell0 + j];
            for (int k = idx[cell]; k < idx[cell + 1]; k++) {
                const int nb = nei[k];
                double n[3];
                if (nb >= 0) for (int a = 0; a < 3; a++) n[a] = S[3 * (size_t)nb + a] - S[3 * (size_t)cell + a];
                else { const int iw = -nb - 1, ax = iw >> 1; n[0] = n[1] = n[2] = 0.0; n[ax] = 2.0 * (B[iw] - S[3 * (size_t)cell + ax]); }
                const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                if (len > 0.0 && std::isfinite(len)) { len_sum += len; len_n++; }
            }
        }
        int e2 = 0;
        if (len_n) (void)std::frexp(len_sum / (double)len_n, &e2);
        const double scale = std::ldexp(1.0, -e2);
        I.scale = (float)scale;
        I.abs_eps = (float)(std::ldexp(1.0, -49) * rmax * scale * (1.0 + 1e-6));
        int kw = 0;
        for (int j = 0; j < I.n_own; j++) {
            const int cell = members[I.cell0 + j];
            const int k0 = kw;
            bool exact = false;
            for (int k = idx[cell]; k < idx[cell + 1]; k++, kw++) {
                const int nb = nei[k];
                double n[3];
                uint32_t link;
                if (nb >= 0) {
                    for (int a = 0; a < 3; a++) n[a] = S[3 * (size_t)nb + a] - S[3 * (size_t)cell + a];
                    int back = VT_NO_BACK;
                    for (int q = idx[nb]; q < idx[nb + 1]; q++) if (nei[q] == cell) { if (q - idx[nb] < VT_FIND_BACK) back = q - idx[nb]; break; }
                    link = (uint32_t)local[nb] | ((uint32_t)back << 16);
                    for (int q = idx[cell]; q < k; q++) if (nei[q] == nb) exact = true;      // a neighbour listed twice
                } else {
                    // a face of the box: the bisector plane with the site's mirror image in it (FP32 filter only)
                    const int iw = -nb - 1, ax = iw >> 1;
                    n[0] = n[1] = n[2] = 0.0; n[ax] = 2.0 * (B[iw] - S[3 * (size_t)cell + ax]);
                    link = 0xffffu | ((uint32_t)VT_NO_BACK << 16) | ((uint32_t)(iw + 1) << 24);
                }
                const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) * scale * (1.0 + 4.0 * 5.9604645e-8);
                wrec[kw] = make_float4((float)(n[0] * scale), (float)(n[1] * scale), (float)(n[2] * scale), nb >= 0 ? (float)len : -(float)len);
                wlink[kw] = link;
            }
            hdr[j] = (uint32_t)k0 | ((uint32_t)(kw - k0) << 20) | (exact ? VT_HDR_EXACT : 0u);
        }
        for (int j = 0; j < I.n_own; j++) local[members[I.cell0 + j]] = -1;
        for (int gI = 0; gI < I.n_site - I.n_own; gI++) local[lay[c].ghosts[gI]] = -1;
    }
    // unused adjacency slots point at the cluster itself (the walk adds a zero count there)
    for (int c = 0; c < n_cl; c++) for (int s = 0; s < VT_MAX_ADJ; s++) if (adj[(size_t)c * VT_MAX_ADJ + s] < 0) adj[(size_t)c * VT_MAX_ADJ + s] = c;
    if (ghosts.empty()) ghosts.push_back(VtGhost{0, VT_MAX_ADJ});
    free_dev(h->d_vt_cluster); free_dev(h->d_vt_info); free_dev(h->d_vt_blob); free_dev(h->d_vt_members); free_dev(h->d_vt_adj); free_dev(h->d_vt_ghost);
    auto up = [&](auto *&dst, const auto &v) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (hipMalloc((void **)&dst, sizeof(T) * v.size()) != hipSuccess) return 1;
        return hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    };
    if (up(h->d_vt_cluster, packed) || up(h->d_vt_info, info) || up(h->d_vt_blob, blob) || up(h->d_vt_members, members) ||
        up(h->d_vt_adj, adj) || up(h->d_vt_ghost, ghosts))
        return h->set_error("cannot allocate the cluster tables of the tiled Voronoi schedule");
    DProblem &P = h->hp;
    P.vt_cluster = h->d_vt_cluster; P.vt_info = h->d_vt_info; P.vt_blob = h->d_vt_blob; P.vt_members = h->d_vt_members;
    P.vt_adj = h->d_vt_adj; P.vt_ghost = h->d_vt_ghost;
    int max_cells = 0;
    for (int c = 0; c < n_cl; c++) max_cells = std::max(max_cells, info[c].n_own);
    h->vt_clusters = n_cl; h->vt_max_cells = max_cells; h->vt_max_lds = max_lds; h->vt_built_for = nd;
    return 0;
}

// Bricks of AMR grids for the tiled schedule (hyp_atile.h): every grid is cut into bricks of at most b0 x b1 x b2 cells, the
// shape of the Cartesian schedule for the number of species (16^3 for one), shrunk along z until densities + accumulators,
// walls and the brick's slice of the goto table (ghost layer included, 16 bits per entry) fit the LDS budget.
static int build_amr_slabs(hyp_handle h)
{
    const int nd = h->n_dust;
    if (h->at_built_for == nd && h->d_at_slabs) return 0;
    const std::vector<AmrGrid> &G = h->h_amr_grids;
    const std::vector<int> &GO = h->h_amr_go;
    if (G.empty()) return h->set_error("amr tables missing for the brick builder");
    if (G.size() >= 32767) return h->set_error("too many amr grids for the 16-bit goto slices of the tiled schedule");
    const size_t budget = (size_t)h->at_lds_kb * 1024;
    int b[3] = {h->at_lds_kb > 100 ? 32 : 16, nd <= 2 ? 16 : 8, nd == 1 ? 16 : 8};      // one 1024-thread workgroup per CU (option at_lds_kb <= 100: 16-cell bricks, two fit a CU)
    if (h->at_cells > 0)         // option: smaller bricks (tests)
        while ((long long)b[0] * b[1] * b[2] > h->at_cells && (b[0] > 1 || b[1] > 1 || b[2] > 1)) {
            int a = b[2] >= b[1] && b[2] >= b[0] ? 2 : (b[1] >= b[0] ? 1 : 0);
            b[a] = (b[a] + 1) / 2;
        }
    auto lds_of = [&](const int n[3]) { return amr_slab_lds((size_t)n[0] * n[1] * n[2], (size_t)(n[0] + 2) * (n[1] + 2) * (n[2] + 2), (size_t)n[0] + n[1] + n[2] + 3, nd); };
    while (lds_of(b) > budget && (b[0] > 1 || b[1] > 1 || b[2] > 1)) {
        int a = b[2] >= b[1] && b[2] >= b[0] ? 2 : (b[1] >= b[0] ? 1 : 0);
        b[a]--;
    }
    if (lds_of(b) > budget) return h->set_error("the LDS budget of the tiled amr schedule is too small");
    std::vector<AtSlab> bricks;
    std::vector<short> go;
    std::vector<int> c0(G.size()), gnb(2 * G.size());
    int max_cells = 0, max_go = 0, max_walls = 0;
    for (size_t k = 0; k < G.size(); k++) {
        const AmrGrid &g = G[k];
        const int nb[3] = {(g.n[0] + b[0] - 1) / b[0], (g.n[1] + b[1] - 1) / b[1], (g.n[2] + b[2] - 1) / b[2]};
        c0[k] = (int)bricks.size(); gnb[2 * k] = nb[0]; gnb[2 * k + 1] = nb[1];
        for (int kz = 0; kz < nb[2]; kz++) for (int ky = 0; ky < nb[1]; ky++) for (int kx = 0; kx < nb[0]; kx++) {
            AtSlab s; std::memset(&s, 0, sizeof s);
            s.grid = (int)k;
            s.o[0] = kx * b[0]; s.o[1] = ky * b[1]; s.o[2] = kz * b[2];
            for (int a = 0; a < 3; a++) s.n[a] = std::min(b[a], g.n[a] - s.o[a]);
            s.go_off = (int)go.size();
            // goto entries of the brick's cells and one layer around them: 1-based positions o .. o + n + 1 of the grid's table
            for (int z = 0; z < s.n[2] + 2; z++) for (int y = 0; y < s.n[1] + 2; y++) for (int x = 0; x < s.n[0] + 2; x++)
                go.push_back((short)GO[(size_t)g.go_off + ((size_t)(s.o[2] + z) * (g.n[1] + 2) + (s.o[1] + y)) * (g.n[0] + 2) + (s.o[0] + x)]);
            bricks.push_back(s);
            max_cells = std::max(max_cells, s.n[0] * s.n[1] * s.n[2]);
            max_go = std::max(max_go, (s.n[0] + 2) * (s.n[1] + 2) * (s.n[2] + 2));
            max_walls = std::max(max_walls, s.n[0] + s.n[1] + s.n[2] + 3);
        }
    }
    if (bricks.size() > HYP_TILE_MAX_BRICKS) return h->set_error("amr grid has too many cells for the brick-tiled schedule");
    if (go.size() > 2000000000ull) return h->set_error("amr goto slices too large");
    free_dev(h->d_at_slabs); free_dev(h->d_at_go); free_dev(h->d_at_grid_c0); free_dev(h->d_at_grid_nz);
    auto up = [&](auto *&dst, const auto &v) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (hipMalloc((void **)&dst, sizeof(T) * std::max<size_t>(v.size(), 1)) != hipSuccess) return 1;
        return hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    };
    if (up(h->d_at_slabs, bricks) || up(h->d_at_go, go) || up(h->d_at_grid_c0, c0) || up(h->d_at_grid_nz, gnb))
        return h->set_error("cannot allocate the brick tables of the tiled AMR schedule");
    DProblem &P = h->hp;
    P.at_slabs = h->d_at_slabs; P.at_go = h->d_at_go; P.at_grid_c0 = h->d_at_grid_c0; P.at_grid_nb = h->d_at_grid_nz;
    for (int a = 0; a < 3; a++) P.at_b[a] = b[a];
    h->at_slabs_n = (int)bricks.size(); h->at_max_cells = max_cells; h->at_max_go = max_go; h->at_max_walls = max_walls; h->at_built_for = nd;
    return 0;
}

// Clusters of octree cells for the tiled schedule (hyp_otile.h).  Cells are numbered depth first
// (grid_geometry_octree.f90:206-246), so a subtree is a contiguous range of ids and so is a run of consecutive siblings.
// Top down: a subtree that fits the LDS budget is a unit; the children of one that does not are grouped, in order, into
// runs that fit; the cells above the units belong to no cluster (they are refined, a packet is never in one of them).
static int build_oct_clusters(hyp_handle h)
{
    const int nd = h->n_dust;
    if (h->ot_built_for == nd && h->d_ot_cluster) return 0;
    const size_t nc = h->n_cells;
    const std::vector<OctCell> &C = h->h_oct_cells;
    const std::vector<int> &CH = h->h_oct_children, &NB = h->h_oct_neigh;
    if (C.size() != nc || NB.size() != 6 * nc) return h->set_error("octree tables missing for the cluster builder");
    // subtree sizes (cells, refined cells): children have larger ids than their parent
    std::vector<int> size(nc, 1), nref(nc, 0);
    for (size_t i = nc; i-- > 0;) {
        if (C[i].refined) nref[i] += 1;
        if (i > 0) { size[C[i].parent] += size[i]; nref[C[i].parent] += nref[i]; }
    }
    const size_t budget = (size_t)h->ot_lds_kb * 1024;
    const int cap = h->ot_cells > 0 ? h->ot_cells : 32767;
    auto fits = [&](long long n, long long k) { return n <= cap && n <= 32767 && oct_cluster_lds((size_t)n, (size_t)k, nd) <= budget; };
    std::vector<int> cl_of(nc, -1), c0v, ncv, kid_off{0};
    auto emit = [&](int first, int n, int k) {
        const int c = (int)c0v.size();
        c0v.push_back(first); ncv.push_back(n); kid_off.push_back(kid_off.back() + k);
        for (int i = first; i < first + n; i++) cl_of[i] = c;
    };
    std::vector<int> stack{0};
    if (fits(size[0], nref[0])) { emit(0, size[0], nref[0]); stack.clear(); }
    while (!stack.empty()) {
        const int p = stack.back(); stack.pop_back();      // a refined cell whose subtree does not fit
        int first = -1, n = 0, k = 0;
        std::vector<int> deeper;
        for (int s = 0; s < 8; s++) {
            const int c = CH[(size_t)p * 8 + s];
            if (!fits(size[c], nref[c])) {
                if (!C[c].refined) return h->set_error("octree cluster budget too small for a single cell");
                if (n) emit(first, n, k);
                n = 0; k = 0; deeper.push_back(c);
                continue;
            }
            if (n && !fits(n + size[c], k + nref[c])) { emit(first, n, k); n = 0; k = 0; }
            if (!n) first = c;
            n += size[c]; k += nref[c];
        }
        if (n) emit(first, n, k);
        for (size_t i = deeper.size(); i-- > 0;) stack.push_back(deeper[i]);
    }
    const int n_cl = (int)c0v.size();
    if (n_cl > HYP_TILE_MAX_BRICKS) return h->set_error("octree has too many cells for the cluster-tiled schedule");
    int max_cells = 0, max_kids = 0;
    for (int c = 0; c < n_cl; c++) { max_cells = std::max(max_cells, ncv[c]); max_kids = std::max(max_kids, kid_off[c + 1] - kid_off[c]); }
    // per-cluster images: records with the row of a refined cell's children in `parent`, children and neighbours as local indices
    std::vector<OctCell> rec(C);
    std::vector<short> kid((size_t)std::max(1, kid_off[n_cl]) * 8, (short)-1), nb(6 * nc, (short)-2);
    for (int c = 0; c < n_cl; c++) {
        int row = 0;
        for (int i = c0v[c]; i < c0v[c] + ncv[c]; i++) {
            if (C[i].refined) {
                rec[i].parent = row;
                for (int s = 0; s < 8; s++) kid[((size_t)kid_off[c] + row) * 8 + s] = (short)(CH[(size_t)i * 8 + s] - c0v[c]);
                row++;
            }
            for (int f = 0; f < 6; f++) {
                const int n = NB[(size_t)i * 6 + f];
                nb[(size_t)i * 6 + f] = (size_t)n == nc ? (short)-1 : (cl_of[n] == c ? (short)(n - c0v[c]) : (short)-2);
            }
        }
    }
    free_dev(h->d_ot_cluster); free_dev(h->d_ot_c0); free_dev(h->d_ot_nc); free_dev(h->d_ot_kid_off); free_dev(h->d_ot_rec); free_dev(h->d_ot_kid); free_dev(h->d_ot_nb);
    auto up = [&](auto *&dst, const auto &v) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (hipMalloc((void **)&dst, sizeof(T) * v.size()) != hipSuccess) return 1;
        return hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    };
    if (up(h->d_ot_cluster, cl_of) || up(h->d_ot_c0, c0v) || up(h->d_ot_nc, ncv) || up(h->d_ot_kid_off, kid_off) || up(h->d_ot_rec, rec) ||
        up(h->d_ot_kid, kid) || up(h->d_ot_nb, nb))
        return h->set_error("cannot allocate the cluster tables of the tiled octree schedule");
    DProblem &P = h->hp;
    P.ot_cluster = h->d_ot_cluster; P.ot_c0 = h->d_ot_c0; P.ot_nc = h->d_ot_nc; P.ot_kid_off = h->d_ot_kid_off;
    P.ot_rec = h->d_ot_rec; P.ot_kid = h->d_ot_kid; P.ot_nb = h->d_ot_nb;
    h->ot_clusters = n_cl; h->ot_max_cells = max_cells; h->ot_max_kids = max_kids; h->ot_built_for = nd;
    return 0;
}

static int sync_problem(hyp_handle h)
{
    hipError_t e = hipMemcpyAsync(h->d_problem, &h->hp, sizeof(DProblem), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(problem): ") + hipGetErrorString(e));
    return 0;
}

static int run_finish_kernel(hyp_handle h, int mode, double scale, double *d_out_ref)
{
    FinishParams F;
    F.scale = scale; F.enforce_energy_range = h->cfg.enforce_energy_range;
    F.additional = (h->d_additional != nullptr); F.write_out = d_out_ref != nullptr; F.pad = 0;
    hipError_t e = hipMemsetAsync(h->d_energy_abs_tot, 0, sizeof(double) * HYP_MAXD, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    int blocks = h->n_cu * 8;
    size_t need = (h->n_elem + 255) / 256;
    if ((size_t)blocks > need) blocks = (int)need;
    if (blocks < 1) blocks = 1;
    finish_kernel<<<blocks, 256, 0, h->stream>>>(h->d_problem, F, mode, h->d_specific_energy, h->d_density,
                                                 h->d_additional, h->d_jnu_id, h->d_jnu_frac, h->d_energy_abs_tot, d_out_ref,
                                                 h->d_spec, h->n_bins);
    e = hipGetLastError();
    if (e != hipSuccess) return h->set_error(std::string("finish_kernel launch: ") + hipGetErrorString(e));
    return 0;
}

// prepare_mrw + update_alpha_inv_planck at the start of an iteration (iter_lucy.f90:109-112,
// iter_final.f90:93-96); must run before sync_problem (it sets table pointers of the problem)
static int mrw_prepare(hyp_handle h)
{
    DProblem &P = h->hp;
    if (!P.mrw) return 0;
    if (P.grid_type == 3) return h->set_error("distance_to_closest_wall: not implemented for Voronoi grid");
    if (!h->d_mrw_alpha) {
        if (hipMalloc(&h->d_mrw_alpha, sizeof(double) * h->n_cells) != hipSuccess ||
            hipMalloc(&h->d_mrw_diff, sizeof(double) * h->n_cells) != hipSuccess ||
            hipMalloc(&h->d_mrw_kp, sizeof(double) * h->n_elem) != hipSuccess)
            return h->set_error("cannot allocate the MRW tables");
    }
    P.mrw_alpha = h->d_mrw_alpha; P.mrw_diff = h->d_mrw_diff; P.mrw_kp = h->d_mrw_kp;     // reach the device with the caller's sync_problem
    unsigned blocks = (unsigned)std::min<size_t>((h->n_cells + 255) / 256, 65535);
    mrw_prepare_kernel<<<blocks, 256, 0, h->stream>>>(h->d_problem, h->d_specific_energy, h->d_density,
                                                      h->d_mrw_alpha, h->d_mrw_diff, h->d_mrw_kp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return h->set_error(std::string("mrw_prepare_kernel launch: ") + hipGetErrorString(e));
    return 0;
}

static int check_device_error(hyp_handle h)
{
    int code = 0;
    double data[3] = {0, 0, 0};
    if (hipMemcpy(&code, h->d_err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return h->set_error("cannot read device error flag");
    if (code == ERR_NONE) return 0;
    (void)hipMemcpy(data, h->d_err_data, sizeof(data), hipMemcpyDeviceToHost);
    (void)hipMemset(h->d_err, 0, sizeof(int));
    char buf[512];
    if (code == ERR_NU_RANGE) {
        // message of src/dust/dust.f90:71
        std::snprintf(buf, sizeof buf,
                      "photon frequency (%10.4E Hz) is outside the range defined for the dust optical properties (%10.4E to %10.4E Hz)",
                      data[0], data[1], data[2]);
    } else if (code == ERR_NOT_IN_CELL) {
        // message of src/sources/source.f90:177
        std::snprintf(buf, sizeof buf,
                      "photon was not emitted inside a cell - this usually indicates that a source is not inside the grid");
    } else if (code == ERR_NEGATIVE_T) {
        // error("find_wall","negative t"), src/grid/grid_geometry_amr.f90:829
        std::snprintf(buf, sizeof buf, "negative t");
    } else if (code == ERR_RAY_GRID) {
        std::snprintf(buf, sizeof buf, "raytracing of dust emission is not available for this grid type");
    } else if (code == ERR_INTERNAL) {
        std::snprintf(buf, sizeof buf, "internal error: a work list of the tiled Lucy iteration overflowed (%g entries)", data[0]);
    } else std::snprintf(buf, sizeof buf, "device error %d", code);
    return h->set_error(buf);
}

// solve_pda (src/grid/grid_pda_3d.f90:84-172) on the device, after update_energy_abs.  The reference solves the
// diffusion equation for the mean intensity in the cells that saw fewer than max(30, 0.5 % of the mean) packets:
// with fewer than 10 000 such cells by Gaussian elimination, otherwise by Gauss-Seidel sweeps in cell order down to
// a relative change of 1e-4 per sweep, and repeats with the updated Rosseland means until the specific energy moves
// by less than 1e-5 / 1e-4.  Here: the Gauss pivot branch is a dense elimination on the device (pda_dense_* kernels;
// rows are diagonally dominant, no pivoting, zero rows skipped), the iterative branch Gauss-Seidel sweeps ordered by
// hyperplanes (pda_gs_kernel), which reproduce the reference's sequential sweeps exactly.
static int solve_pda(hyp_handle h)
{
    h->pda_last_cells = 0; h->pda_last_outer = 0; h->pda_last_sweeps = 0;
    const DProblem &P = h->hp;
    if (!(P.grid_type == 1 || P.grid_type == 5 || P.grid_type == 6)) return 0;      // grid_pda_disabled.f90
    const size_t nc = h->n_cells;
    const int n_hp = P.n1 + P.n2 + P.n3 - 2;       // i1 + i2 + i3 = 0 .. n1 + n2 + n3 - 3
    if (!h->d_pda_ctl) {
        if (hipMalloc(&h->d_pda_ctl, sizeof(PdaCtl)) != hipSuccess || hipMalloc(&h->d_pda_mask, nc) != hipSuccess ||
            hipMalloc(&h->d_pda_cells, sizeof(unsigned int) * nc) != hipSuccess ||
            hipMalloc(&h->d_pda_hp, sizeof(unsigned int) * 3 * (n_hp + 1)) != hipSuccess ||
            hipMalloc(&h->d_pda_emean, sizeof(double) * nc) != hipSuccess)
            return h->set_error("cannot allocate the PDA work arrays");
    }
    unsigned int *hp_count = h->d_pda_hp, *hp_off = h->d_pda_hp + (n_hp + 1), *hp_cursor = h->d_pda_hp + 2 * (n_hp + 1);
    const double *nphot = h->d_accum + h->ext_nphot;
    const int blocks = h->n_cu * 4;
    PdaCtl ctl;
    (void)hipMemsetAsync(h->d_pda_ctl, 0, sizeof(PdaCtl), h->stream);
    (void)hipMemsetAsync(h->d_pda_hp, 0, sizeof(unsigned int) * 3 * (n_hp + 1), h->stream);
    pda_total_kernel<<<blocks, 256, 0, h->stream>>>(nphot, nc, h->d_pda_ctl);
    if (hipMemcpyAsync(&ctl, h->d_pda_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("PDA: cannot read the packet total");
    // mean_n_photons = sum(n_photons) / size(n_photons) is an INTEGER division (:99); threshold max(30, ceiling(0.005 mean))
    const double mean_n = (double)((long long)ctl.total_photons / (long long)nc);
    const double threshold = std::max(30.0, std::ceil(0.005 * mean_n));
    pda_mask_kernel<<<blocks, 256, 0, h->stream>>>(h->d_problem, nphot, threshold, h->d_specific_energy, h->d_density, h->d_pda_mask,
                                                   h->d_pda_emean, hp_count, h->d_pda_ctl);
    if (hipMemcpyAsync(&ctl, h->d_pda_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("PDA: cannot read the cell count");
    const unsigned int n_pda = ctl.n_pda;
    h->pda_last_cells = (int)n_pda;
    if (n_pda == 0) return 0;        // " [pda] not necessary for this iteration"
    pda_scan_kernel<<<1, 64, 0, h->stream>>>(hp_count, hp_off, hp_cursor, n_hp);
    pda_list_kernel<<<blocks, 256, 0, h->stream>>>(h->d_problem, h->d_pda_mask, hp_off, hp_cursor, h->d_pda_cells);
    if ((size_t)n_pda * 6 > h->pda_coef_alloc) {
        free_dev(h->d_pda_coef);
        if (hipMalloc(&h->d_pda_coef, sizeof(double) * 6 * n_pda) != hipSuccess) return h->set_error("cannot allocate the PDA coefficients");
        h->pda_coef_alloc = (size_t)n_pda * 6;
    }
    const bool exact = n_pda < 10000;
    const double tolerance = exact ? 1.e-5 : 1.e-4, gs_tol = 1.e-4;
    const int cb = (int)std::min<size_t>((n_pda + 255) / 256, (size_t)h->n_cu * 4);
    if (exact) {
        if (!h->d_pda_id && hipMalloc(&h->d_pda_id, sizeof(unsigned int) * nc) != hipSuccess) return h->set_error("cannot allocate the PDA index");
        if ((size_t)n_pda > h->pda_dense_alloc) {
            free_dev(h->d_pda_a); free_dev(h->d_pda_b); free_dev(h->d_pda_f);
            if (hipMalloc(&h->d_pda_a, sizeof(double) * (size_t)n_pda * n_pda) != hipSuccess || hipMalloc(&h->d_pda_b, sizeof(double) * n_pda) != hipSuccess ||
                hipMalloc(&h->d_pda_f, sizeof(double) * n_pda) != hipSuccess) return h->set_error("cannot allocate the dense PDA system");
            h->pda_dense_alloc = n_pda;
        }
        (void)hipMemsetAsync(h->d_pda_id, 0xff, sizeof(unsigned int) * nc, h->stream);
        pda_id_kernel<<<cb, 256, 0, h->stream>>>(h->d_pda_cells, n_pda, h->d_pda_id);
    }
    for (int outer = 1; outer <= 10000; outer++) {
        h->pda_last_outer = outer;
        pda_coef_kernel<<<cb, 256, 0, h->stream>>>(h->d_problem, h->d_pda_cells, n_pda, h->d_specific_energy, h->d_density, h->d_pda_emean,
                                                  h->d_pda_coef, exact ? 1 : 0);
        if (exact) {
            (void)hipMemsetAsync(h->d_pda_a, 0, sizeof(double) * (size_t)n_pda * n_pda, h->stream);
            pda_dense_build_kernel<<<cb, 256, 0, h->stream>>>(h->d_problem, h->d_pda_cells, n_pda, h->d_pda_id, h->d_pda_coef, h->d_pda_emean,
                                                             h->d_pda_a, h->d_pda_b);
            for (unsigned int k = 0; k + 1 < n_pda; k++) {
                const unsigned int rows = n_pda - k - 1;
                pda_pivot_kernel<<<1, 1024, 0, h->stream>>>(h->d_pda_a, n_pda, k, (unsigned int *)h->d_pda_f);       // f[0 .. k] is free: the pivot row's index lives in f[0]
                pda_swap_kernel<<<std::min(64u, (rows + 256) / 256), 256, 0, h->stream>>>(h->d_pda_a, h->d_pda_b, n_pda, k, (const unsigned int *)h->d_pda_f);
                pda_elim_factor_kernel<<<(rows + 255) / 256, 256, 0, h->stream>>>(h->d_pda_a, n_pda, k, h->d_pda_f);
                pda_elim_update_kernel<<<dim3(std::min(8u, (rows + 255) / 256), rows), 256, 0, h->stream>>>(h->d_pda_a, h->d_pda_b, n_pda, k, h->d_pda_f);
            }
            pda_backsub_kernel<<<1, 1024, 0, h->stream>>>(h->d_pda_a, h->d_pda_b, n_pda);
            pda_scatter_solution_kernel<<<cb, 256, 0, h->stream>>>(h->d_pda_cells, n_pda, h->d_pda_b, h->d_pda_emean);
        } else
            pda_gs_kernel<<<1, 1024, 0, h->stream>>>(h->d_problem, h->d_pda_cells, hp_off, n_hp, h->d_pda_coef, h->d_pda_emean, gs_tol,
                                                     20000000, h->d_pda_ctl);
        (void)hipMemsetAsync(&h->d_pda_ctl->maxdiff_bits, 0, sizeof(unsigned long long), h->stream);
        pda_update_kernel<<<cb, 256, 0, h->stream>>>(h->d_problem, h->d_pda_cells, n_pda, h->d_pda_emean, h->d_specific_energy, h->d_spec,
                                                    h->n_bins, h->d_pda_ctl);
        if (hipMemcpyAsync(&ctl, h->d_pda_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error(std::string("PDA solve failed: ") + hipGetErrorString(hipGetLastError()));
        h->pda_last_sweeps += ctl.sweeps;
        double maxdiff;
        std::memcpy(&maxdiff, &ctl.maxdiff_bits, sizeof maxdiff);
        if (maxdiff < tolerance) return 0;      // " [pda] converged"
    }
    return h->set_error("PDA did not converge");
}

int hyp_get_n_photons(hyp_handle h, double *out)
{
    if (!h || !out) return 1;
    if (!h->count_photons) return h->set_error("n_photons array is not allocated");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    // after hyp_lucy_accumulators (and the all-reduce) the block holds the whole-job counts
    hipError_t e = hipMemcpy(out, h->d_accum + h->ext_nphot, sizeof(double) * h->n_cells, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(n_photons): ") + hipGetErrorString(e));
    return 0;
}

int hyp_get_specific_energy_spectrum(hyp_handle h, double *out, double *bin_edges_out)
{
    if (!h) return 1;
    if (!h->n_bins) return h->set_error("specific_energy_spectrum array is not allocated");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    if (bin_edges_out) std::memcpy(bin_edges_out, h->spectrum_edges.data(), sizeof(double) * (h->n_bins + 1));
    if (!out) return 0;
    double *tmp = nullptr;
    const size_t n = (size_t)h->n_bins * h->n_elem;
    if (hipMalloc(&tmp, sizeof(double) * n) != hipSuccess) return h->set_error("cannot allocate the spectrum staging buffer");
    spectrum_to_ref_kernel<<<h->n_cu * 8, 256, 0, h->stream>>>(h->d_spec, tmp, h->n_cells, h->n_dust, h->n_bins);
    hipError_t e = hipMemcpyAsync(out, tmp, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return h->set_error(std::string("copy out failed: ") + hipGetErrorString(e));
    return 0;
}

// specific_energy_converged (grid_physics_3d.f90:637-689): the `percentile` quantile of max(a/b, b/a) between the
// specific energy at the previous call and now.  status 0: value computed; 1: nothing changed (value 0); 2: could not
// check (only cells that were or became zero changed); 3: first call (no previous state).  fortranlib's quantile
// (source absent) is restated as the element of rank nint(percentile / 100 * (n - 1)) of the sorted sample; it is
// found by a search over the bit patterns of the (positive) ratios: 63 counting passes, no sort.
int hyp_convergence_value(hyp_handle h, double percentile, double *value, int *status)
{
    if (!h || !value || !status) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    const size_t n = h->n_elem;
    if (!h->d_prev_se) {
        if (hipMalloc(&h->d_prev_se, sizeof(double) * n) != hipSuccess || hipMalloc(&h->d_ratio, sizeof(double) * n) != hipSuccess ||
            hipMalloc(&h->d_conv_ctl, sizeof(ConvCtl)) != hipSuccess) return h->set_error("cannot allocate the convergence work arrays");
    }
    *value = 0.0;
    if (!h->have_prev) {
        (void)hipMemcpyAsync(h->d_prev_se, h->d_specific_energy, sizeof(double) * n, hipMemcpyDeviceToDevice, h->stream);
        (void)hipStreamSynchronize(h->stream);
        h->have_prev = true; *status = 3;
        return 0;
    }
    const int blocks = h->n_cu * 8;
    ConvCtl c;
    (void)hipMemsetAsync(h->d_conv_ctl, 0, sizeof(ConvCtl), h->stream);
    conv_ratio_kernel<<<blocks, 256, 0, h->stream>>>(h->d_prev_se, h->d_specific_energy, n, h->d_ratio, h->d_conv_ctl);
    (void)hipMemcpyAsync(h->d_prev_se, h->d_specific_energy, sizeof(double) * n, hipMemcpyDeviceToDevice, h->stream);
    if (hipMemcpyAsync(&c, h->d_conv_ctl, sizeof c, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("convergence check failed");
    if (c.n_changed == 0) { *status = 1; return 0; }
    if (c.n_changed_nonzero == 0 || c.n_valid == 0) { *status = 2; return 0; }
    long long rank = (long long)std::floor(percentile / 100.0 * (double)(c.n_valid - 1) + 0.5);
    if (rank < 0) rank = 0;
    if ((unsigned long long)rank > c.n_valid - 1) rank = (long long)(c.n_valid - 1);
    // largest bit pattern v with #(ratios < v) <= rank is the ratio of that rank
    unsigned long long prefix = 0;
    for (int bit = 62; bit >= 0; bit--) {
        const unsigned long long cand = prefix | (1ull << bit);
        (void)hipMemsetAsync(&h->d_conv_ctl->count, 0, sizeof(unsigned long long), h->stream);
        conv_count_kernel<<<blocks, 256, 0, h->stream>>>(h->d_ratio, n, cand, h->d_conv_ctl);
        if (hipMemcpyAsync(&c.count, &h->d_conv_ctl->count, sizeof c.count, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("convergence check failed");
        if (c.count <= (unsigned long long)rank) prefix = cand;
    }
    std::memcpy(value, &prefix, sizeof(double));
    *status = 0;
    return 0;
}

int hyp_lucy_launch(hyp_handle h, uint64_t first_id, uint64_t n_local, int iteration)
{
    if (!h) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    if (P.n_sources == 0) return h->set_error("no sources set up - need sources for initial iteration(s)");      // setup_rt.f90:230
    int copies = h->accum_copies;
    if (copies < 1) copies = 1;
    if (copies > 256) copies = 256;
    if (copies > h->accum_copies_alloc) {   // grow the replica pool on demand
        double *nb = nullptr;
        if (hipMalloc(&nb, sizeof(double) * h->accum_stride * copies) != hipSuccess)
            return h->set_error("cannot allocate accumulator replicas");
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(h->d_accum);
        h->d_accum = nb; h->accum_copies_alloc = copies;
    }
    P.sum = h->d_accum; P.tail = h->d_accum + h->n_elem; P.n_copies = copies; P.copy_stride = h->accum_stride;
    P.sum_spec = h->n_bins ? h->d_accum + h->ext_spec : nullptr;
    if (h->count_photons) {      // grid_reset_energy: grid_generic.f90:21-27
        (void)hipMemsetAsync(h->d_nphot, 0, sizeof(unsigned int) * h->n_cells, h->stream);
        (void)hipMemsetAsync(h->d_nphot_inexact, 0, sizeof(int), h->stream);
        // the visited sets are sized by hyp_lucy_launch below, once the grid of the persistent kernel is known
    }
    if (mrw_prepare(h)) return 1;
    // The brick-tiled iteration pays off once the grid has many bricks and the
    // iteration is long enough to amortise its generations (measured: profiles/r01c_*).
    // (the per-cell packet counter and the spectrum planes live in global memory: those runs use the persistent kernel)
    bool tile_ok = false, tile_auto = false;
    if (P.grid_type == 1) {
        tile_ok = h->n_dust <= 4 && tile_bricks(P, h->n_dust) <= HYP_TILE_MAX_BRICKS && !h->count_photons && !h->n_bins;
        tile_auto = tile_ok && tile_bricks(P, h->n_dust) >= 32 && n_local >= 1500000ull;      // (128^3: 14.2 against 18.0 ms at 2e6 packets, even at 1e6; tools/small_probe.py)
    } else if (P.grid_type == 3) {
        // Voronoi: clusters of cells in LDS (hyp_vtile.h); the modified random walk does not exist on these grids
        tile_ok = h->n_dust <= 4 && !h->count_photons && !h->n_bins && !P.mrw;
        tile_auto = tile_ok && h->n_cells >= 8192 && n_local >= 2000000ull;
    }
    else if (P.grid_type == 2) {
        // octree: clusters of sibling subtrees in LDS (hyp_otile.h)
        tile_ok = h->n_dust <= 4 && !h->count_photons && !h->n_bins && h->oct_neighbours;
        tile_auto = tile_ok && h->n_cells >= 4096 && n_local >= 2000000ull;
    }
    else if (P.grid_type == 5 || P.grid_type == 6) {
        // spherical / cylindrical polar grids: index bricks in LDS (hyp_ptile.h)
        tile_ok = h->n_dust <= 4 && polar_tile_bricks(P, h->n_dust, h->pt_lds_kb) > 0 && !h->count_photons && !h->n_bins;
        tile_auto = tile_ok && h->n_cells >= 4096 && n_local >= 3000000ull;      // (400 x 200: 91 against 81 ms at 2e6 packets, 140 against 150 at 4e6)
    }
    else if (P.grid_type == 4) {
        // AMR: bricks of the grids in LDS (hyp_atile.h)
        tile_ok = h->n_dust <= 4 && !h->count_photons && !h->n_bins;
        tile_auto = tile_ok && h->n_cells >= 32768 && n_local >= 2000000ull;
    }
    bool tiled = tile_ok && (h->lucy_mode == 1 || (h->lucy_mode < 0 && tile_auto && !h->tile_unbuildable));
    if (tiled && (P.grid_type == 2 || P.grid_type == 3 || P.grid_type == 4)) {
        // the builders have limits of their own (HYP_TILE_MAX_BRICKS clusters / bricks, the LDS budget, 16-bit grid numbers):
        // a grid beyond them runs on the persistent kernel as before; only a FORCED tiled iteration (lucy_mode = 1) reports the limit
        const int rc = P.grid_type == 4 ? build_amr_slabs(h) : P.grid_type == 3 ? build_vor_clusters(h) : build_oct_clusters(h);
        if (rc) {
            if (h->lucy_mode == 1) return 1;
            h->tile_unbuildable = true;
            h->err.clear();
            tiled = false;
        }
    }
    if (sync_problem(h)) return 1;
    hipError_t e = hipMemsetAsync(h->d_accum, 0, sizeof(double) * h->accum_stride * copies, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(accum): ") + hipGetErrorString(e));
    unsigned long long first = first_id;
    e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));

    if (tiled) {
        if (launch_tiled(h, first_id, n_local, (uint32_t)iteration)) return 1;
        h->last_lucy_mode = 1;
        h->lucy_pending = true;
        h->pending_packets = n_local;
        return 0;
    }
    h->last_lucy_mode = 0;
    LucyKernel k = pick_lucy_kernel(h->n_dust, h->hp.grid_type);
    const size_t lds = lds_bytes(P);
    int bpc = h->blocks_per_cu;
    if (bpc <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, lds) != hipSuccess || occ <= 0) occ = 2;
        bpc = occ;
    }
    long long blocks = (long long)h->n_cu * bpc;
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    if (h->count_photons) {
        // one visited set per lane of THIS launch (HYP_VISIT_SLOTS words each); with less memory than that, fewer workgroups
        for (;;) {
            const size_t lanes = (size_t)blocks * 256;
            if (h->visit_lanes >= lanes) break;
            free_dev(h->d_visit);
            h->visit_lanes = 0;
            if (hipMalloc((void **)&h->d_visit, lanes * HYP_VISIT_SLOTS * sizeof(unsigned long long)) == hipSuccess) { h->visit_lanes = lanes; break; }
            (void)hipGetLastError();
            h->d_visit = nullptr;
            if (blocks <= 1) return h->set_error("no memory for the per-lane visited sets of the n_photons counter");
            blocks = (blocks + 1) / 2;
        }
        (void)hipMemsetAsync(h->d_visit, 0, (size_t)blocks * 256 * HYP_VISIT_SLOTS * sizeof(unsigned long long), h->stream);
        if (P.visit_tab != h->d_visit) { P.visit_tab = h->d_visit; if (sync_problem(h)) return 1; }
    }
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = (uint32_t)iteration;
    int chunk = h->chunk;
    if (chunk <= 0) {
        unsigned long long waves = (unsigned long long)blocks * 4ull;
        unsigned long long c = n_local / (waves * 8ull);
        if (c < 64) c = 64;
        if (c > 4096) c = 4096;
        chunk = (int)c;
    }
    L.chunk = chunk;
    L.interact_threshold = h->interact_threshold; L.emit_threshold = h->emit_threshold;
    (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("lucy_kernel launch: ") + hipGetErrorString(e));
    h->lucy_pending = true;
    h->pending_packets = n_local;
    return 0;
}

int hyp_lucy_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->lucy_pending) return h->set_error("hyp_lucy_accumulators called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    if (h->hp.n_copies > 1) {
        int blocks = h->n_cu * 8;
        reduce_copies_kernel<<<blocks, 256, 0, h->stream>>>(h->d_accum, h->n_elem + TAIL_SIZE, h->accum_stride, h->hp.n_copies);
    }
    if (h->count_photons) nphot_to_block_kernel<<<h->n_cu * 4, 256, 0, h->stream>>>(h->d_nphot, h->d_accum + h->ext_nphot, h->n_cells);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("propagation failed: ") + hipGetErrorString(e));
    (void)hipEventElapsedTime(&h->last_propagate_ms, h->ev0, h->ev1);
    if (h->count_photons) (void)hipMemcpy(&h->nphot_inexact, h->d_nphot_inexact, sizeof(int), hipMemcpyDeviceToHost);
    if (check_device_error(h)) { h->lucy_pending = false; return 1; }
    if (device_ptr) *device_ptr = h->d_accum;
    if (n_doubles) *n_doubles = h->block_doubles;
    return 0;
}

int hyp_lucy_finish(hyp_handle h, double *specific_energy_out, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->lucy_pending) return h->set_error("hyp_lucy_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->lucy_pending = false;
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->d_accum + h->n_elem, sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st;
    std::memset(&st, 0, sizeof st);
    st.energy_current = tail[TAIL_ENERGY];
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS]; st.interactions = (uint64_t)tail[TAIL_INTERACTIONS];
    st.n_packets = h->pending_packets;
    if (!(st.energy_current > 0.0)) return h->set_error("no energy emitted");
    // update_energy_abs(energy_total/energy_current): iter_lucy.f90:224
    (void)hipEventRecord(h->ev2, h->stream);
    double *d_out = (specific_energy_out && h->n_dust > 1) ? h->d_scratch : nullptr;
    const double scale = h->energy_total / st.energy_current;
    if (h->n_bins) {
        spectrum_update_kernel<<<h->n_cu * 8, 256, 0, h->stream>>>(h->d_problem, h->d_accum + h->ext_spec, h->d_spec, scale, h->n_bins);
        if (hipGetLastError() != hipSuccess) return h->set_error("spectrum_update_kernel launch failed");
    }
    if (h->pda) {
        // update_energy_abs, then solve_pda, then sublimate_dust: iter_lucy.f90:224-235
        if (run_finish_kernel(h, 2, scale, nullptr)) return 1;
        if (solve_pda(h)) return 1;
        if (run_finish_kernel(h, 3, scale, d_out)) return 1;
    } else if (run_finish_kernel(h, 0, scale, d_out)) return 1;
    (void)hipEventRecord(h->ev3, h->stream);
    double tot[HYP_MAXD];
    e = hipMemcpyAsync(tot, h->d_energy_abs_tot, sizeof(tot), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && specific_energy_out)
        e = hipMemcpyAsync(specific_energy_out, h->n_dust > 1 ? h->d_scratch : h->d_specific_energy,
                           sizeof(double) * h->n_elem, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("finish failed: ") + hipGetErrorString(e));
    (void)hipEventElapsedTime(&h->last_finish_ms, h->ev2, h->ev3);
    for (int d = 0; d < h->n_dust; d++) st.energy_abs_tot[d] = tot[d];
    h->last_stats = st;
    if (stats) *stats = st;
    return 0;
}

int hyp_lucy_iteration(hyp_handle h, uint64_t n_packets, int iteration, double *specific_energy_out, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (n_packets == 0) return 0;   // "Skipping": iter_lucy.f90:87-94
    if (hyp_lucy_launch(h, 0, n_packets, iteration)) return 1;
    if (hyp_lucy_accumulators(h, nullptr, nullptr)) return 1;
    return hyp_lucy_finish(h, specific_energy_out, stats);
}

static int copy_out_ref_layout(hyp_handle h, const double *d_src, double *out)
{
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    const double *src = d_src;
    if (h->n_dust > 1) {
        to_ref_layout_kernel<<<1024, 256, 0, h->stream>>>(d_src, h->d_scratch, h->n_cells, h->n_dust);
        src = h->d_scratch;
    }
    hipError_t e = hipMemcpyAsync(out, src, sizeof(double) * h->n_elem, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("copy out failed: ") + hipGetErrorString(e));
    return 0;
}

int hyp_get_specific_energy(hyp_handle h, double *out)
{
    if (!h || !out) return 1;
    return copy_out_ref_layout(h, h->d_specific_energy, out);
}

int hyp_get_density(hyp_handle h, double *out)
{
    if (!h || !out) return 1;
    return copy_out_ref_layout(h, h->d_density, out);
}

int hyp_set_specific_energy(hyp_handle h, const double *in)
{
    if (!h || !in) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    hipError_t e = hipMemcpyAsync(h->d_scratch, in, sizeof(double) * h->n_elem, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_specific_energy, h->n_cells, h->n_dust);
    if (run_finish_kernel(h, 1, 1.0, nullptr)) return 1;
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("set_specific_energy failed: ") + hipGetErrorString(e));
    return 0;
}

int hyp_last_kernel_ms(hyp_handle h, float *propagate_ms, float *finish_ms)
{
    if (!h) return 1;
    if (propagate_ms) *propagate_ms = h->last_propagate_ms;
    if (finish_ms) *finish_ms = h->last_finish_ms;
    return 0;
}

int hyp_set_option(hyp_handle h, const char *name, int64_t value)
{
    if (!h || !name) return 1;
    std::string n(name);
    if (n == "interact_threshold") h->interact_threshold = (int)value;
    else if (n == "emit_threshold") h->emit_threshold = (int)value;
    else if (n == "accum_copies") h->accum_copies = (int)value;
    else if (n == "blocks_per_cu") h->blocks_per_cu = (int)value;
    else if (n == "chunk") h->chunk = (int)value;
    else if (n == "lucy_mode") h->lucy_mode = (int)value;       // -1 auto, 0 persistent atomics kernel, 1 brick-tiled
    else if (n == "tile_slots") h->tile_slots = (int)value;
    else if (n == "tile_task") h->tile_task = (int)value;
    else if (n == "tile_pools") h->tile_pools = (int)value;
    else if (n == "tile_time_walk") h->tile_time_walk = (int)value;
    else if (n == "plain_imaging") h->plain_imaging = value != 0 && h->plain_imaging;      // can only be switched off
    else if (n == "defer_peel") h->defer_peel = value < 0 ? 0 : value > 3 ? 3 : (int)value;
    else if (n == "mono_defer") h->mono_defer_opt = value ? 1 : 0;
    else if (n == "gen_defer") h->gen_defer_opt = value ? 1 : 0;
    else if (n == "direct_memo") h->direct_memo = value ? 1 : 0;
    else if (n == "peel_sort") h->peel_sort = value != 0;
    else if (n == "ff_prepass") h->ff_prepass = value != 0;
    else if (n == "oct_neighbours") { h->oct_neighbours = value != 0; h->hp.oct_neigh = h->oct_neighbours ? h->d_oct_neigh : nullptr; }
    else if (n == "peel_events") {
        if (value < 1) return h->set_error("peel_events must be positive");
        if (h->peel_events != value) {      // the buffers are allocated by the next imaging iteration
            free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
            h->peel_cap = 0;
        }
        h->peel_events = value; h->peel_events_exact = true;
    }
    else if (n == "vt_cells") { h->vt_cells = (int)value; h->vt_built_for = -1; h->tile_unbuildable = false; }
    // (the tests' handles on the tiled schedules: small clusters / bricks, early drain, a look at the device after every generation)
    else if (n == "at_cells") { h->at_cells = (int)value; h->at_built_for = -1; h->tile_unbuildable = false; }
    else if (n == "ot_cells") { h->ot_cells = (int)value; h->ot_built_for = -1; h->tile_unbuildable = false; }
    else if (n == "pt_lds_kb") h->pt_lds_kb = (int)std::max<int64_t>(1, std::min<int64_t>(150, value));
    else if (n == "tile_drain") h->tile_drain = (int)value;
    else if (n == "tile_poll") h->tile_poll = value < 1 ? 1 : (int)value;
    else return h->set_error("unknown option: " + n);
    return 0;
}

int hyp_get_option(hyp_handle h, const char *name, int64_t *value)
{
    if (!h || !name || !value) return 1;
    std::string n(name);
    if (n == "interact_threshold") *value = h->interact_threshold;
    else if (n == "emit_threshold") *value = h->emit_threshold;
    else if (n == "accum_copies") *value = h->accum_copies;
    else if (n == "blocks_per_cu") *value = h->blocks_per_cu;
    else if (n == "chunk") *value = h->chunk;
    else if (n == "lucy_mode") *value = h->lucy_mode;
    else if (n == "tile_slots") *value = h->tile_slots;
    else if (n == "tile_task") *value = h->tile_task;
    else if (n == "tile_pools") *value = h->tile_pools;
    else if (n == "lucy_block_doubles") *value = (int64_t)h->block_doubles;
    else if (n == "lucy_flag_index") *value = (int64_t)(h->n_elem + TAIL_RANK_ERROR);
    else if (n == "image_block_doubles") *value = (int64_t)(h->d_img_accum ? h->img_accum_n : (size_t)TAIL_SIZE);
    else if (n == "image_flag_index") *value = (int64_t)((h->d_img_accum ? h->img_accum_n - TAIL_SIZE : (size_t)0) + TAIL_RANK_ERROR);
    else if (n == "last_walk_us") *value = (int64_t)(h->last_walk_ms * 1000.0);
    else if (n == "last_walk_launches") *value = h->last_walk_launches;
    else if (n == "pda_last_cells") *value = h->pda_last_cells;
    else if (n == "defer_peel") *value = h->defer_peel;
    else if (n == "mono_defer") *value = h->mono_defer_opt;
    else if (n == "gen_defer") *value = h->gen_defer_opt && h->gen_defer ? 1 : 0;
    else if (n == "direct_memo") *value = h->direct_memo;
    else if (n == "last_direct_memo") *value = h->last_direct_memo;
    else if (n == "last_mono_deferred") *value = h->last_mono_deferred;
    else if (n == "last_tiled_imaging") *value = h->last_tiled_imaging;
    else if (n == "peel_sort") *value = h->peel_sort;
    else if (n == "ff_prepass") *value = h->ff_prepass;
    else if (n == "last_ff_prepass") *value = h->last_ff_prepass;
    else if (n == "n_photons_inexact") *value = h->nphot_inexact;
    else if (n == "peel_events") *value = h->peel_events;
    else if (n == "plain_imaging") *value = h->plain_imaging ? 1 : 0;
    else if (n == "last_defer_rounds") *value = h->last_defer_rounds;
    else if (n == "last_defer_events") *value = (int64_t)h->last_defer_events;
    else if (n == "at_slabs") *value = h->at_slabs_n;
    else if (n == "ot_clusters") *value = h->ot_clusters;
    else if (n == "vt_cells") *value = h->vt_cells;
    else if (n == "vt_clusters") *value = h->vt_clusters;
    else if (n == "vt_max_cells") *value = h->vt_max_cells;
    else if (n == "last_vt_exact_steps") *value = h->h_ctl ? (int64_t)h->h_ctl->dbg[38] : 0;      // steps of the Voronoi walk that ran the reference's loop
    else if (n.rfind("last_walk_why", 0) == 0 && n.size() == 14 && n[13] >= '0' && n[13] <= '7') *value = h->h_ctl ? (int64_t)h->h_ctl->dbg[30 + (n[13] - '0')] : 0;
    else if (n == "last_lucy_mode") *value = h->last_lucy_mode;         // schedule the last Lucy iteration ran with
    else if (n == "last_generations") *value = h->last_generations;   // generations of the last tiled iteration
    else return h->set_error("unknown option: " + n);
    return 0;
}

// ---- imaging iteration -------------------------------------------------------

// Buffers of the deferred peel-off, sized for `lanes` lanes of the propagation grid.  Returns nonzero when they cannot be had
// (the caller then peels off inline).
static int defer_buffers(hyp_handle h, const DeferKernels &dk, size_t lanes, uint64_t n_local)
{
    const size_t waves = lanes / 64;
    // `peel_events` is the ceiling; a small iteration does not need it (8 events per packet in one round, more rounds
    // beyond that) and the buffer only grows
    size_t cap = (size_t)h->peel_events;
    // (at least 4 Mi slots, ~0.9 GB: a packet of an optically thick run leaves thousands of events, and every round costs a host
    // synchronisation and three sort launches -- 2e4 packets with 1e4 events each took 1 413 rounds with the 8-per-packet rule alone)
    const size_t want = n_local > (1ull << 40) ? cap : std::max<size_t>((size_t)n_local * 8, (size_t)1 << 22);
    if (want < cap && !h->peel_events_exact) cap = want;
    cap = (cap + HYP_PEEL_CHUNK - 1) / HYP_PEEL_CHUNK * HYP_PEEL_CHUNK;
    if (h->d_peel_events && h->peel_event_bytes == dk.event_bytes && h->peel_lanes >= lanes &&
        (h->peel_events_exact ? h->peel_cap == cap : h->peel_cap >= cap)) return 0;
    free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
    h->peel_cap = 0;
    bool ok = hipMalloc(&h->d_peel_events, cap * dk.event_bytes) == hipSuccess;
    while (!ok && !h->peel_events_exact && cap > ((size_t)1 << 20)) {        // a smaller buffer means more rounds, not another schedule
        (void)hipGetLastError();
        cap = (cap / 2 + HYP_PEEL_CHUNK - 1) / HYP_PEEL_CHUNK * HYP_PEEL_CHUNK;
        ok = hipMalloc(&h->d_peel_events, cap * dk.event_bytes) == hipSuccess;
    }
    for (int i = 0; i < 2 && ok; i++)
        ok = hipMalloc(&h->d_peel_susp[i], lanes * dk.susp_bytes) == hipSuccess &&
             hipMalloc((void **)&h->d_peel_ret[i], waves * 2 * sizeof(unsigned long long)) == hipSuccess;
    if (ok && !h->d_peel_ctl)
        ok = hipMalloc((void **)&h->d_peel_ctl, sizeof(PeelCtl)) == hipSuccess && hipHostMalloc((void **)&h->h_peel_ctl, sizeof(PeelCtl)) == hipSuccess &&
             hipHostMalloc((void **)&h->h_peel_counter, sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
        return 1;
    }
    h->peel_cap = cap; h->peel_event_bytes = dk.event_bytes; h->peel_lanes = lanes;
    return 0;
}

// Rounds of {propagate, peel} until every packet id has been used and no packet is left set aside (hyp_defer.h).
// the event buffer and, where the memory is there, the tables of the sorted peel-off
static void defer_setup_buffers(hyp_handle h, const DeferKernels &dk, DeferBuf &B)
{
    B.events = h->d_peel_events; B.cap = h->peel_cap; B.ctl = h->d_peel_ctl;
    B.susp[0] = h->d_peel_susp[0]; B.susp[1] = h->d_peel_susp[1]; B.ret[0] = h->d_peel_ret[0]; B.ret[1] = h->d_peel_ret[1];
    B.order = nullptr; B.keys = nullptr; B.bins = nullptr; B.n_bins = 0; B.ff = nullptr; B.cur = 0; B.direct = nullptr;
    h->last_direct_memo = 0;
    if (h->direct_memo && dk.direct && h->hp.n_sources > 0 && h->hp.n_views_total > 0 && !h->hp.peel_scattered_only) {
        // direct light of the point sources: one walk per (source, view) instead of one per packet (hyp_defer.h: direct_column_kernel)
        const size_t n = (size_t)h->hp.n_sources * (size_t)h->hp.n_views_total;
        if (h->direct_cap < n) {
            free_dev(h->d_direct);
            h->direct_cap = 0;
            if (hipMalloc((void **)&h->d_direct, n * sizeof(DirectCol)) == hipSuccess) h->direct_cap = n;
            else { (void)hipGetLastError(); h->d_direct = nullptr; }
        }
        if (h->direct_cap >= n) {
            hipLaunchKernelGGL(dk.direct, dim3((unsigned)std::min<size_t>((n + 63) / 64, 1024)), dim3(64), lds_bytes(h->hp), h->stream, (const DProblem *)h->d_problem, h->d_direct);
            B.direct = h->d_direct;
            h->last_direct_memo = 1;
        }
    }
    if (h->peel_sort && dk.sort_hist && h->peel_cap < 0xffffffffull) {
        // sorted peel-off: order + keys per event slot, counts | offsets per bin; without the memory the events are taken as written
        if (h->peel_sort_cap < h->peel_cap) {
            free_dev(h->d_peel_order); free_dev(h->d_peel_keys);
            h->peel_sort_cap = 0;
            if (hipMalloc((void **)&h->d_peel_order, sizeof(unsigned int) * h->peel_cap) == hipSuccess &&
                hipMalloc((void **)&h->d_peel_keys, sizeof(unsigned int) * h->peel_cap) == hipSuccess) h->peel_sort_cap = h->peel_cap;
            else { (void)hipGetLastError(); free_dev(h->d_peel_order); free_dev(h->d_peel_keys); }
        }
        if (!h->d_peel_bins && hipMalloc((void **)&h->d_peel_bins, sizeof(unsigned int) * 2 * HYP_SORT_MAX_BINS) != hipSuccess) { (void)hipGetLastError(); h->d_peel_bins = nullptr; }
        if (h->peel_sort_cap >= h->peel_cap && h->d_peel_bins) {
            B.order = h->d_peel_order; B.keys = h->d_peel_keys; B.bins = h->d_peel_bins;
            B.n_bins = (unsigned int)std::max<unsigned long long>(1ull, std::min<unsigned long long>(HYP_SORT_MAX_BINS, h->hp.n_cells));
        }
    }
}

// forced first interaction: every packet's emission, escape walk and first optical depth ahead of the rounds, one record per id
// (128 bytes at one dust species; without the memory the propagation kernel does it all itself).  Sets B.ff where it ran.
static void defer_ff_prepass(hyp_handle h, const DeferKernels &dk, const LaunchParams &L, DeferBuf &B, size_t lds)
{
    B.ff = nullptr;
    h->last_ff_prepass = 0;
    const unsigned long long n_ids = L.end_id - L.first_id;
    if (!(h->ff_prepass && h->hp.forced_first && dk.ff_walk && n_ids > 0)) return;
    const size_t want = (size_t)n_ids * dk.ff_bytes;
    if (h->ff_cap < want) {
        free_dev(h->d_ff);
        h->ff_cap = 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want < free_b / 2 && hipMalloc(&h->d_ff, want) == hipSuccess) h->ff_cap = want;
        else { (void)hipGetLastError(); h->d_ff = nullptr; }
    }
    if (h->ff_cap < want) return;
    B.ff = h->d_ff;
    (void)hipMemsetAsync(&h->d_peel_ctl->ff_cursor, 0, sizeof(unsigned long long), h->stream);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)dk.ff_walk, 256, lds) != hipSuccess || occ <= 0) occ = 2;
    const unsigned long long need = (n_ids + 255) / 256;
    const unsigned ff_blocks = (unsigned)std::min<unsigned long long>((unsigned long long)h->n_cu * occ, need);
    hipLaunchKernelGGL(dk.ff_walk, dim3(ff_blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L, B);
    h->last_ff_prepass = 1;
}

// the events in the buffer: ordered by cell (where the tables are there), then every (event, view) pair walked to the observer
static void defer_peel_events(hyp_handle h, const DeferKernels &dk, const DeferBuf &B, unsigned peel_blocks, size_t lds, uint32_t iter_tag)
{
    if (B.order) {
        const unsigned sort_blocks = (unsigned)((h->peel_cap + HYP_SORT_PER_WG - 1) / HYP_SORT_PER_WG);
        (void)hipMemsetAsync(B.bins, 0, sizeof(unsigned int) * B.n_bins, h->stream);
        hipLaunchKernelGGL(dk.sort_hist, dim3(sort_blocks), dim3(256), 0, h->stream, (const DProblem *)h->d_problem, B);
        hipLaunchKernelGGL(dk.sort_scan, dim3(1), dim3(1024), 0, h->stream, B);
        hipLaunchKernelGGL(dk.sort_scatter, dim3(sort_blocks), dim3(256), 0, h->stream, (const DProblem *)h->d_problem, B);
    }
    hipLaunchKernelGGL(h->inside_observers ? dk.peel_inside : dk.peel, dim3(peel_blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, B, iter_tag);
}

static int run_deferred_rounds(hyp_handle h, const DeferKernels &dk, const LaunchParams &L, unsigned blocks, size_t lds, bool ff_ahead = true)
{
    DeferBuf B;
    defer_setup_buffers(h, dk, B);
    if (ff_ahead) defer_ff_prepass(h, dk, L, B, lds);
    else h->last_ff_prepass = 0;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)(h->inside_observers ? dk.peel_inside : dk.peel), 256, lds) != hipSuccess || occ <= 0) occ = 2;
    const unsigned peel_blocks = (unsigned)(h->n_cu * occ);
    int idle_rounds = 0;
    for (int round = 0;; round++) {
        B.cur = round & 1;
        hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, B.cur, round == 0 ? 1 : 0);
        hipLaunchKernelGGL(B.ff ? dk.propagate_pre : dk.propagate, dim3(blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L, B);
        defer_peel_events(h, dk, B, peel_blocks, lds, L.iter_tag);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return h->set_error(std::string("deferred imaging launch: ") + hipGetErrorString(e));
        (void)hipMemcpyAsync(h->h_peel_ctl, h->d_peel_ctl, sizeof(PeelCtl), hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(h->h_peel_counter, h->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
        e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("propagation failed: ") + hipGetErrorString(e));
        const PeelCtl &C = *h->h_peel_ctl;
        h->last_defer_rounds = round + 1;
        h->last_defer_events += C.written;
        if (C.n_susp[B.cur] == 0 && C.n_ret[B.cur] == 0 && *h->h_peel_counter >= L.end_id) break;
        // a round without a single event can happen (all packets in flight left the grid), a long run of them cannot
        idle_rounds = C.written == 0 ? idle_rounds + 1 : 0;
        if (idle_rounds > 64) return h->set_error("deferred peel-off makes no progress (event buffer too small?)");
        int err = 0;
        if (hipMemcpy(&err, h->d_err, sizeof err, hipMemcpyDeviceToHost) != hipSuccess || err != 0) break;     // reported by hyp_final_accumulators
    }
    return 0;
}

// The imaging iteration with its propagation half on the slot-pool schedule of the Lucy iteration (hyp_tiled.h: IMG kernels):
// emission and the forced first interaction ahead of everything (ff_walk_kernel), then generations of interact / emit / sort /
// WALK FROM LDS -- the packets' own walks start at interaction points in random directions, which is what made them slow in
// final_defer_kernel (scattered loads) --, events appended to the buffer and peeled (sorted) when it could overflow and at the
// end.  Returns 0 done, 1 error, 2 not applicable (no tiled schedule for the grid, tables too large, forced first interaction
// without room for its records): the caller runs the rounds of hyp_defer.h instead.
static int run_tiled_imaging(hyp_handle h, const DeferKernels &dk, const LaunchParams &L, size_t lds, uint64_t n_local)
{
    const DProblem &P = h->hp;
    const TileKernels K = pick_tile_kernels(h->n_dust, P.grid_type);
    if (!K.walk || !K.interact_img || !K.emit_img || K.event_bytes != dk.event_bytes) return 2;
    if (P.grid_type == 1 && tile_bricks(P, h->n_dust) > HYP_TILE_MAX_BRICKS) return 2;
    if ((P.grid_type == 5 || P.grid_type == 6) && polar_tile_bricks(P, h->n_dust, h->pt_lds_kb) < 0) return 2;
    if (P.grid_type == 2 && !h->oct_neighbours) return 2;
    if (P.grid_type == 2 || P.grid_type == 3 || P.grid_type == 4) {
        const int rc = P.grid_type == 4 ? build_amr_slabs(h) : P.grid_type == 3 ? build_vor_clusters(h) : build_oct_clusters(h);
        if (rc) { h->err.clear(); return 2; }
        if (sync_problem(h)) return 1;
    }
    DeferBuf B;
    defer_setup_buffers(h, dk, B);
    {
        // the event buffer must hold a few generations' worth of events (one per slot and generation at most); decided BEFORE the
        // pre-pass runs: it counts its crossings and kills, and the caller's fall-back runs it again
        const long long want_slots = h->tile_slots > 0 ? h->tile_slots : ((P.grid_type == 2 || P.grid_type == 4) ? 3ll << 22 : 3ll << 21);
        const unsigned long long slots = (unsigned long long)std::min<long long>(want_slots, (long long)n_local) + 4096ull;
        if (B.cap < 3ull * (slots + slots / 8)) return 2;
    }
    defer_ff_prepass(h, dk, L, B, lds);
    if (P.forced_first && !B.ff) return 2;          // (no room for the records: the pre-pass did not run)
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)(h->inside_observers ? dk.peel_inside : dk.peel), 256, lds) != hipSuccess || occ <= 0) occ = 2;
    const unsigned peel_blocks = (unsigned)(h->n_cu * occ);
    hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, 0, 1);
    const uint32_t iter_tag = L.iter_tag;
    auto flush = [&]() -> int {
        // (every pool's stream is idle here)
        defer_peel_events(h, dk, B, peel_blocks, lds, iter_tag);
        (void)hipMemcpyAsync(h->h_peel_ctl, h->d_peel_ctl, sizeof(PeelCtl), hipMemcpyDeviceToHost, h->stream);
        hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, 0, 0);
        const hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("tiled imaging: peel-off failed: ") + hipGetErrorString(e));
        h->last_defer_rounds++;
        h->last_defer_events += B.order ? h->h_peel_ctl->n_sorted : h->h_peel_ctl->reserved;
        return 0;
    };
    const int rc = launch_tiled(h, L.first_id, n_local, iter_tag, &B, flush);
    h->last_tiled_imaging = rc == 0 ? 1 : 0;
    return rc ? 1 : 0;
}

int hyp_final_launch(hyp_handle h, uint64_t first_id, uint64_t n_local)
{
    if (!h) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    if (P.n_sources == 0 && n_local > 0) return h->set_error("no sources set up - need sources for last iteration");      // setup_rt.f90:236
    double *tail;
    if (h->d_img_accum) {
        hipError_t e = hipMemsetAsync(h->d_img_accum, 0, sizeof(double) * h->img_accum_n, h->stream);
        if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(images): ") + hipGetErrorString(e));
        tail = h->d_img_accum + (h->img_accum_n - TAIL_SIZE);
    } else {
        hipError_t e = hipMemsetAsync(h->d_accum + h->n_elem, 0, sizeof(double) * TAIL_SIZE, h->stream);
        if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(tail): ") + hipGetErrorString(e));
        tail = h->d_accum + h->n_elem;
    }
    P.tail = tail; P.sum = h->d_accum; P.n_copies = 1;
    if (mrw_prepare(h)) return 1;
    if (sync_problem(h)) return 1;
    unsigned long long first = first_id;
    hipError_t e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));
    LucyKernel k = pick_final_kernel(h->n_dust, h->hp.grid_type, h->plain_imaging && !h->inside_observers && !h->hp.mono_which ? 1 : h->lean_imaging && !h->hp.mono_which ? 2 : 0);
    // deferred peel-off where the plain kernel applies and there is something to peel into (hyp_defer.h)
    const bool gen = !h->plain_imaging && h->gen_defer && h->gen_defer_opt;
    bool deferred = (h->plain_imaging || gen) && !h->hp.mono_which && h->defer_peel && P.n_peeled > 0 && P.n_views_total > 0;
    DeferKernels dk;
    std::memset(&dk, 0, sizeof dk);
    if (deferred) dk = pick_defer_kernels(h->n_dust, h->hp.grid_type);
    if (deferred && gen) { dk.propagate = h->cfg.mrw ? dk.propagate_gen_mrw : dk.propagate_gen; dk.peel = dk.peel_gen; dk.direct = nullptr; }      // (a source may stand in the way of another's direct light)
    if (deferred && (!dk.propagate || !dk.peel)) deferred = false;
    const size_t lds = lds_bytes(P);
    int bpc = h->blocks_per_cu;
    if (bpc <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, deferred ? (const void *)dk.propagate : (const void *)k, 256, lds) != hipSuccess || occ <= 0) occ = 2;
        bpc = occ;
    }
    long long blocks = (long long)h->n_cu * bpc;
    if (deferred && defer_buffers(h, dk, (size_t)blocks * 256, n_local)) deferred = false;      // no memory for the buffers: peel off inline
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = 0x10000u;
    int chunk = h->chunk;
    if (chunk <= 0) {
        unsigned long long c = n_local / ((unsigned long long)blocks * 32ull);
        if (c < 64) c = 64;
        if (c > 4096) c = 4096;
        chunk = (int)c;
    }
    L.chunk = chunk;
    L.interact_threshold = h->final_interact_threshold >= 0 ? h->final_interact_threshold : (deferred ? 16 : 32);
    L.emit_threshold = h->final_emit_threshold >= 0 ? h->final_emit_threshold : (!deferred ? 32 : h->hp.grid_type == 1 ? 16 : 48);
    h->last_defer_rounds = 0; h->last_defer_events = 0;
    h->last_tiled_imaging = 0;
    if (deferred) {
        (void)hipEventRecord(h->ev0, h->stream);
        // large launches of problems whose grid has a tiled schedule: the propagation half on it (defer_peel = 2 forces, 3 forbids)
        int rc = 2;
        if (!gen && !h->inside_observers && h->defer_peel != 3 && (h->defer_peel == 2 || n_local >= 4000000ull)) rc = run_tiled_imaging(h, dk, L, lds, n_local);
        if (rc == 1) return 1;
        if (rc == 0) {
            (void)hipEventRecord(h->ev1, h->stream);
            h->final_pending = true;
            h->pending_packets = n_local;
            return 0;
        }
        if (run_deferred_rounds(h, dk, L, (unsigned)blocks, lds, !gen)) return 1;
        (void)hipEventRecord(h->ev1, h->stream);
        h->final_pending = true;
        h->pending_packets = n_local;
        return 0;
    }
    (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("final_kernel launch: ") + hipGetErrorString(e));
    h->final_pending = true;
    h->pending_packets = n_local;
    return 0;
}

int hyp_final_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->final_pending) return h->set_error("hyp_final_accumulators called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("propagation failed: ") + hipGetErrorString(e));
    (void)hipEventElapsedTime(&h->last_propagate_ms, h->ev0, h->ev1);
    if (check_device_error(h)) { h->final_pending = false; return 1; }
    if (h->d_img_accum) {
        if (device_ptr) *device_ptr = h->d_img_accum;
        if (n_doubles) *n_doubles = h->img_accum_n;
    } else {
        if (device_ptr) *device_ptr = h->d_accum + h->n_elem;
        if (n_doubles) *n_doubles = TAIL_SIZE;
    }
    return 0;
}

int hyp_final_finish(hyp_handle h, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->final_pending) return h->set_error("hyp_final_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->final_pending = false;
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->hp.tail, sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st;
    std::memset(&st, 0, sizeof st);
    st.energy_current = tail[TAIL_ENERGY];
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS]; st.interactions = (uint64_t)tail[TAIL_INTERACTIONS];
    st.n_packets = h->pending_packets;
    // peeled_images_adjust_scale(energy_total/energy_current): iter_final.f90:142-143
    if (st.energy_current > 0.0) {
        double scale = h->energy_total / st.energy_current;
        for (size_t g = 0; g < h->h_peeled.size(); g++) {
            // binned_images_adjust_scale (images_binned.f90:34-38): x n_theta x n_phi
            const double sc = (int)g == h->hp.binned ? scale * (double)h->hp.n_bin_theta * (double)h->hp.n_bin_phi : scale;
            if (h->sed_n[g]) image_scale_kernel<<<256, 256, 0, h->stream>>>(h->d_img_accum + h->sed_off[g], h->sed_n[g], sc);
            if (h->img_n[g]) image_scale_kernel<<<1024, 256, 0, h->stream>>>(h->d_img_accum + h->img_off[g], h->img_n[g], sc);
        }
        e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("image scaling failed: ") + hipGetErrorString(e));
    }
    h->last_stats = st;
    if (stats) *stats = st;
    return 0;
}

int hyp_final_iteration(hyp_handle h, uint64_t n_packets, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (n_packets == 0) return 0;   // "Skipping": iter_final.f90:78-85
    if (hyp_final_launch(h, 0, n_packets)) return 1;
    if (hyp_final_accumulators(h, nullptr, nullptr)) return 1;
    return hyp_final_finish(h, stats);
}

// ---- raytracing iteration (iter_raytracing.f90) --------------------------------------------

int hyp_raytracing_launch(hyp_handle h, int which, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first)
{
    if (!h) return 1;
    if (!h->cfg.raytracing) return h->set_error("raytracing was not requested in the configuration");
    if (which < 0 || which > 1) return h->set_error("hyp_raytracing_launch: which must be 0 (sources) or 1 (dust)");
    if (!h->d_img_accum) return h->set_error("no peeled images set up");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    double *tail = h->d_img_accum + (h->img_accum_n - TAIL_SIZE);
    hipError_t e;
    if (zero_first) e = hipMemsetAsync(h->d_img_accum, 0, sizeof(double) * h->img_accum_n, h->stream);
    else if (!h->ray_pending) e = hipMemsetAsync(tail, 0, sizeof(double) * TAIL_SIZE, h->stream);
    else e = hipSuccess;
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(images): ") + hipGetErrorString(e));
    P.tail = tail; P.sum = h->d_accum; P.n_copies = 1;
    if (sync_problem(h)) return 1;
    if (!h->ray_pending) h->ray_ms = 0.f;        // hyp_last_kernel_ms after hyp_raytracing_finish: the launches of this iteration
    h->ray_pending = true;
    if (which == 0 && P.n_sources == 0) n_local = 0;       // n_raytracing_photons_sources = 0: setup_rt.f90:238
    if (n_local == 0 || n_total == 0) return 0;
    unsigned long long first = first_id;
    e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));
    RayKernel k = pick_ray_kernel(h->n_dust, h->hp.grid_type);
    const size_t lds = lds_bytes(P);
    long long blocks = (long long)h->n_cu * 2;
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = which == 0 ? 0x20000u : 0x30000u;
    unsigned long long c = n_local / ((unsigned long long)blocks * 32ull);
    if (c < 64) c = 64;
    if (c > 4096) c = 4096;
    L.chunk = (int)c;
    L.interact_threshold = h->interact_threshold; L.emit_threshold = h->emit_threshold;
    (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L, which, (double)n_total);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("ray_kernel launch: ") + hipGetErrorString(e));
    // the two parts share the id dispenser: finish this launch before the next one resets it
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("raytracing failed: ") + hipGetErrorString(e));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->ray_ms += ms; }
    if (check_device_error(h)) { h->ray_pending = false; return 1; }
    return 0;
}

int hyp_raytracing_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->ray_pending) return h->set_error("hyp_raytracing_accumulators called without a launched iteration");
    if (device_ptr) *device_ptr = h->d_img_accum;
    if (n_doubles) *n_doubles = h->img_accum_n;
    return 0;
}

int hyp_raytracing_finish(hyp_handle h, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->ray_pending) return h->set_error("hyp_raytracing_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->ray_pending = false;
    h->last_propagate_ms = h->ray_ms; h->last_finish_ms = 0.f;
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->d_img_accum + (h->img_accum_n - TAIL_SIZE), sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st;
    std::memset(&st, 0, sizeof st);
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS];
    if (stats) *stats = st;
    return 0;
}

int hyp_raytracing_iteration(hyp_handle h, uint64_t n_sources, uint64_t n_dust, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (hyp_raytracing_launch(h, 0, 0, n_sources, n_sources, 0)) return 1;
    if (hyp_raytracing_launch(h, 1, 0, n_dust, n_dust, 0)) return 1;
    hyp_iter_stats st;
    if (hyp_raytracing_finish(h, &st)) return 1;
    st.n_packets = n_sources + n_dust;
    if (stats) *stats = st;
    return 0;
}

// ---- monochromatic final iteration (iter_final_mono.f90) ---------------------------------------

int hyp_mono_launch(hyp_handle h, int which, int inu, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first)
{
    if (!h) return 1;
    if (!h->cfg.monochromatic) return h->set_error("monochromatic mode was not requested in the configuration");
    if (which < 0 || which > 1) return h->set_error("hyp_mono_launch: which must be 0 (sources) or 1 (dust)");
    if (inu < 0 || inu >= (int)h->frequencies.size()) return h->set_error("incorrect inu");
    if (!h->d_img_accum) return h->set_error("no peeled images set up");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    double *tail = h->d_img_accum + (h->img_accum_n - TAIL_SIZE);
    hipError_t e = hipSuccess;
    if (zero_first) e = hipMemsetAsync(h->d_img_accum, 0, sizeof(double) * h->img_accum_n, h->stream);
    else if (!h->mono_pending) e = hipMemsetAsync(tail, 0, sizeof(double) * TAIL_SIZE, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(images): ") + hipGetErrorString(e));
    if (!h->mono_pending) { std::memset(&h->mono_stats, 0, sizeof h->mono_stats); h->ray_ms = 0.f; }
    h->mono_pending = true;
    P.tail = tail; P.sum = h->d_accum; P.n_copies = 1;
    P.mono_which = 0; P.mono_inu = inu; P.mono_nu = h->frequencies[inu]; P.mono_n_total = (double)n_total;
    if (which == 0 && h->hp.n_sources == 0) n_local = 0;       // n_last_photons_sources = 0: setup_rt.f90:232
    if (n_local == 0 || n_total == 0) return sync_problem(h);
    if (which == 1) {
        // setup_monochromatic_grid_pdfs: precompute_jnu_var ran in the last finish step (jnu_id / jnu_frac are current)
        const size_t nc = h->n_cells;
        if (!h->d_mono_cdf) {
            if (hipMalloc(&h->d_mono_cdf, sizeof(double) * nc * h->n_dust) != hipSuccess ||
                hipMalloc(&h->d_mono_mean, sizeof(double) * 2 * HYP_MAXD) != hipSuccess) return h->set_error("hipMalloc(monochromatic pdfs) failed");
        }
        P.mono_which = 2;       // dust_emit_probability reads mono_inu
        if (sync_problem(h)) return 1;
        mono_weight_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>((const DProblem *)h->d_problem, h->d_mono_cdf);
        mono_scan_kernel<<<dim3(h->n_dust), dim3(1024), 0, h->stream>>>(h->d_mono_cdf, nc, h->d_mono_mean);
        double mean[2 * HYP_MAXD];
        e = hipMemcpyAsync(mean, h->d_mono_mean, sizeof mean, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("monochromatic emission pdfs: ") + hipGetErrorString(e));
        double tot = 0.0;
        for (int d = 0; d < h->n_dust; d++) { P.mono_mean_prob[d] = mean[d]; tot += mean[d]; }
        P.mono_cdf = h->d_mono_cdf;
        if (tot == 0.0) { P.mono_which = 0; return sync_problem(h); }      // "No emission at this frequency"
    }
    P.mono_which = which + 1;
    if (sync_problem(h)) return 1;
    unsigned long long first = first_id;
    e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));
    LucyKernel k = pick_final_kernel(h->n_dust, h->hp.grid_type, 0);
    const size_t lds = lds_bytes(P);
    // problems that are plain apart from being monochromatic: the launch on the deferred schedule (hyp_defer.h: the propagation
    // kernel writes events, the peel kernel walks them sorted by cell into the launch's frequency plane); option mono_defer = 0: inline
    const bool mgen = h->mono_gen_defer && h->gen_defer_opt;
    bool deferred = (h->mono_defer || mgen) && h->mono_defer_opt && h->defer_peel && P.n_peeled > 0 && P.n_views_total > 0;
    DeferKernels dk;
    std::memset(&dk, 0, sizeof dk);
    if (deferred) dk = pick_defer_kernels(h->n_dust, h->hp.grid_type);
    if (deferred && mgen) { dk.propagate_mono = dk.propagate_mono_gen; dk.peel = dk.peel_gen; dk.direct = nullptr; }
    if (deferred && (!dk.propagate_mono || !dk.peel)) deferred = false;
    long long blocks = (long long)h->n_cu * 2;
    if (deferred) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)dk.propagate_mono, 256, lds) != hipSuccess || occ <= 0) occ = 2;
        blocks = (long long)h->n_cu * occ;
        if (defer_buffers(h, dk, (size_t)blocks * 256, n_local * 4)) deferred = false;      // (a packet leaves tens of events: fewer rounds)
    }
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = (which == 0 ? 0x40000u : 0x50000u) + (uint32_t)inu;
    unsigned long long c = n_local / ((unsigned long long)blocks * 32ull);
    if (c < 64) c = 64;
    if (c > 4096) c = 4096;
    L.chunk = (int)c;
    // the monochromatic iteration is final_kernel with inline peel-off: the imaging iteration's batch sizes
    L.interact_threshold = h->final_interact_threshold >= 0 ? h->final_interact_threshold : 32;
    L.emit_threshold = h->final_emit_threshold >= 0 ? h->final_emit_threshold : 48;
    h->last_mono_deferred = deferred ? 1 : 0;
    (void)hipEventRecord(h->ev0, h->stream);
    if (deferred) {
        L.interact_threshold = h->final_interact_threshold >= 0 ? h->final_interact_threshold : 16;
        L.emit_threshold = h->final_emit_threshold >= 0 ? h->final_emit_threshold : 48;
        dk.propagate = dk.propagate_mono;
        if (run_deferred_rounds(h, dk, L, (unsigned)blocks, lds, false)) { P.mono_which = 0; h->mono_pending = false; return 1; }
    } else
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("final_kernel (monochromatic) launch: ") + hipGetErrorString(e));
    // the launches share the id dispenser and the problem block: finish this one before the next changes them
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("monochromatic iteration failed: ") + hipGetErrorString(e));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->ray_ms += ms; }
    P.mono_which = 0;
    if (sync_problem(h)) return 1;
    h->mono_stats.n_packets += n_local;
    if (check_device_error(h)) { h->mono_pending = false; return 1; }
    return 0;
}

int hyp_mono_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->mono_pending) return h->set_error("hyp_mono_accumulators called without a launched iteration");
    if (device_ptr) *device_ptr = h->d_img_accum;
    if (n_doubles) *n_doubles = h->img_accum_n;
    return 0;
}

int hyp_mono_finish(hyp_handle h, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->mono_pending) return h->set_error("hyp_mono_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->mono_pending = false;
    h->last_propagate_ms = h->ray_ms; h->last_finish_ms = 0.f;       // hyp_last_kernel_ms: the propagation kernels of all launches of this iteration
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->d_img_accum + (h->img_accum_n - TAIL_SIZE), sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st = h->mono_stats;
    st.energy_current = tail[TAIL_ENERGY];
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS]; st.interactions = (uint64_t)tail[TAIL_INTERACTIONS];
    if (stats) *stats = st;
    return 0;
}

int hyp_mono_iteration(hyp_handle h, uint64_t n_sources, uint64_t n_dust, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->cfg.monochromatic) return h->set_error("monochromatic mode was not requested in the configuration");
    bool first = true;
    for (int which = 0; which < 2; which++) {
        const uint64_t n = which == 0 ? n_sources : n_dust;
        for (int inu = 0; inu < (int)h->frequencies.size(); inu++) {
            if (hyp_mono_launch(h, which, inu, 0, n, n, first ? 1 : 0)) return 1;
            first = false;
        }
    }
    return hyp_mono_finish(h, stats);
}

int hyp_peeled_n_orig(hyp_handle h, int g) { return (h && g >= 0 && g < (int)h->h_peeled.size()) ? h->h_peeled[g].n_orig : -1; }

int hyp_peeled_get(hyp_handle h, int g, int which, double *out, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (g < 0 || g >= (int)h->h_peeled.size() || which < 0 || which > 3) return h->set_error("hyp_peeled_get: bad group or selector");
    size_t n = (which < 2) ? h->sed_n[g] : h->img_n[g];
    size_t off = ((which < 2) ? h->sed_off[g] : h->img_off[g]) + ((which & 1) ? n : 0);
    if (n_doubles) *n_doubles = n;
    if (!out || n == 0) return 0;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    hipError_t e = hipMemcpy(out, h->d_img_accum + off, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(images): ") + hipGetErrorString(e));
    return 0;
}

}  // extern "C"
