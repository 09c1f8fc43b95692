// hyp_engine.h -- what the translation units of the host side share: the engine's state (struct hyp_engine behind hyp_handle), the
// entry points of the kernels of every geometry (hyp_pick.h), and the few functions one unit calls in another.
//   hyp_engine.hip   handle life cycle, problem digest, getters / setters, options
//   hyp_create.hip   hyp_create: tables, device residency; the cluster / brick builders of the tiled schedules
//   hyp_lucy.hip     Lucy iteration: persistent kernel or generations of the tiled schedule; epilogue (update_energy_abs, PDA, MRW tables,
//                    convergence, n_photons, spectrum)
//   hyp_imaging.hip  final iteration (inline, deferred, tiled), raytracing and monochromatic iterations
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#pragma once
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

extern std::string g_error;   // message of a failed hyp_create (hyp_engine.hip)

#define HIP_TRY(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_));                  \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

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
    double lucy_cross_per_flight = 0.0;      // of the last Lucy iteration (0: none ran); hyp_final_launch's choice of schedule

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
    int lucy_mode = -1, tile_slots = 0 /* 0: 3 << 23 slots, 3 << 24 on Cartesian grids from 1024 bricks (imaging: 3 << 22, trees 3 << 23): launch_tiled */, tile_task = 0 /* 0: 8192 packets per task, 16384 on Voronoi grids (Lucy iteration) */, tile_pools = 3, tile_drain = -1 /* -1: 1 000 000 packets in flight (profiles/r04_tiled_log.md) */, tile_park = 48 /* round 6, 25e6-slot pools: 16 -> 48: spherical 1 517 -> 1 458 ms, configs[1] 196.2 -> 194.1, octree 85.3 -> 84.6, tessellation 325 -> 323 */;
    int img_end_game = 1;           // option: the tiled imaging iteration hands its last packets to the deferred rounds (0: generations to the end, as until round 5)
    int reproducible = 0;           // option: every persistent kernel runs as ONE wave (one workgroup of 64 threads), no tiled / deferred schedule, one accumulator copy:
                                    // the order of every floating-point sum is the program order of that wave -- a seed gives the same bits on every run (tests; ~1000x slower)
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
    long long last_tile_slots = 0;  // slots of the last tiled iteration's pools together (get-only option)
    bool tiled_img_gen = false;     // run_tiled_imaging of a problem with general sources is under way: the GEN instances of the IMG kernels (run_tiled_generations)
    bool gen_defer = false;         // sources with a surface: the imaging iteration on the deferred schedule (final_defer_kernel<.., GEN>, peel_kernel<.., GEN>)
    bool mono_defer = false;        // a monochromatic run of a problem that is plain otherwise: its launches run on the deferred schedule (final_defer_kernel<.., true, true>)
    bool lean_imaging = false;      // final_kernel<.., false, LEAN>: any sources, but no MRW / monochromatic / binned images / inside observers
    bool simple_sources = false;    // every source is a point source with a tabulated / blackbody spectrum (tile_emit_kernel<.., SIMPLE>)
    // deferred peel-off (hyp_defer.h): event buffer, control block, packets / id ranges carried between rounds
    int defer_peel = 1;             // option: 1 = deferred peel-off where plain_imaging holds (hyp_defer.h; large launches: propagation on the tiled schedule),
                                    //   2 = always on the tiled schedule where there is one, 3 = never, 0 = inline peel-off
    int last_tiled_imaging = 0;
    long long last_end_game = 0;      // packets the last tiled imaging iteration handed to the deferred rounds at its end
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


inline int set_error(const std::string &m) { g_error = m; return 1; }

template <typename T>
inline void free_dev(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

inline size_t lds_bytes(const DProblem &P) { return P.grid_type != 1 ? 0 : sizeof(double) * 2 * ((size_t)P.n1 + P.n2 + P.n3 + 3); }

inline RayKernel pick_ray_kernel(int nd, int grid_type)
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

inline LucyKernel pick_lucy_kernel(int nd, int grid_type)
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

inline DeferKernels pick_defer_kernels(int nd, int grid_type)
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

inline LucyKernel pick_final_kernel(int nd, int grid_type, int mode)
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


// hyp_lucy.hip
TileKernels pick_tile_kernels(int nd, int grid_type);
int tile_bricks(const DProblem &P, int nd);
long long polar_tile_bricks(const DProblem &P, int nd, int lds_kb);
long long car_tile_bricks(const DProblem &P, int nd);
long long tiled_imaging_slots(const DProblem &P);
size_t amr_slab_lds(size_t n, size_t g, size_t w, int nd);
size_t oct_cluster_lds(size_t n, size_t k, int nd);
// the imaging iteration's end-game on the tiled schedule: at most max_packets live packets become SuspRec of the deferred schedule
// (tile_to_susp_kernel), then run() finishes them in its rounds (hyp_imaging.hip)
struct TiledEndGame { uint64_t max_packets; std::function<int()> run; };
int launch_tiled(hyp_handle h, uint64_t first_id, uint64_t n_local, uint32_t iter_tag, const DeferBuf *img = nullptr,
                 const std::function<int()> &flush = std::function<int()>(), const TiledEndGame *end_game = nullptr);
int run_finish_kernel(hyp_handle h, int mode, double scale, double *d_out_ref);
int solve_pda(hyp_handle h);
int sync_problem(hyp_handle h);
int check_device_error(hyp_handle h);
int mrw_prepare(hyp_handle h);
// hyp_create.hip
int build_vor_clusters(hyp_handle h);
int build_oct_clusters(hyp_handle h);
int build_amr_slabs(hyp_handle h);
