// hyp_geom.hip -- instantiates the propagation kernels of ONE grid geometry and ONE family of kernels
// (-DHYP_GEOM_TU=GEOM_xxx -DHYP_PART=n; hyperion_amd/build.py compiles this file once per pair, in
// parallel, and links the objects with hyp_engine.hip into libhyperion_amd.so).  Every unit includes only the
// headers of its family, so that an edit of one schedule recompiles the units that carry it and nothing else.
//   HYP_PART 0: lucy_kernel      1: the tiled Lucy schedules (hyp_tiled.h + hyp_vtile.h / hyp_otile.h / hyp_atile.h)
//            2: final_kernel, general     5: final_kernel, plain and lean     3: deferred imaging (hyp_defer.h)     4: ray_kernel
// Species counts: 1-4 are compile-time (registers hold the per-species state); 5-8 run on the HYP_MAXD instances that
// read the count from the problem -- only the persistent Lucy kernel, the general imaging kernel and the raytracing
// kernel exist in that form (the engine does not pick a plain / lean / deferred / tiled schedule above four species).
#ifndef HYP_GEOM_TU
#define HYP_GEOM_TU 0   // GEOM_CAR
#endif
#ifndef HYP_PART
#error "compile with -DHYP_PART=0..5 (hyperion_amd/build.py)"
#endif
#include "hyp_kernels.h"
#if HYP_PART == 1
#include "hyp_tiled.h"
#if HYP_GEOM_TU == 2
#include "hyp_vtile.h"
#elif HYP_GEOM_TU == 1
#include "hyp_otile.h"
#elif HYP_GEOM_TU == 3
#include "hyp_atile.h"
#elif HYP_GEOM_TU == 4 || HYP_GEOM_TU == 5
#include "hyp_ptile.h"
#endif
#endif
#if HYP_PART == 3
#include "hyp_defer.h"
#endif
#include "hyp_pick.h"
#include <cstring>

#if HYP_PART == 0
template <int GEOM>
LucyKernel pick_lucy_kernel_g(int nd)
{
#ifdef HYP_ONLY_ND   // tuning builds (tools/variants.py) instantiate one species count only
    (void)nd;
    return lucy_kernel<HYP_ONLY_ND, GEOM>;
#else
    switch (nd) {
    case 1: return lucy_kernel<1, GEOM>;
    case 2: return lucy_kernel<2, GEOM>;
    case 3: return lucy_kernel<3, GEOM>;
    case 4: return lucy_kernel<4, GEOM>;
    default: return lucy_kernel<HYP_MAXD, GEOM>;
    }
#endif
}

#endif

#if HYP_PART == 2
template <int GEOM>
LucyKernel pick_final_kernel_g(int nd)      // the general imaging kernel final_kernel<nd, GEOM, false>
{
#ifdef HYP_ONLY_ND
    (void)nd;
    return final_kernel<HYP_ONLY_ND, GEOM, false>;
#else
    switch (nd) {
    case 1: return final_kernel<1, GEOM, false>;
    case 2: return final_kernel<2, GEOM, false>;
    case 3: return final_kernel<3, GEOM, false>;
    case 4: return final_kernel<4, GEOM, false>;
    default: return final_kernel<HYP_MAXD, GEOM, false>;      // five to eight species: the general kernel only
    }
#endif
}
#endif

#if HYP_PART == 5
template <int GEOM>
LucyKernel pick_final_special_g(int nd, int mode)      // mode 1: plain (final_kernel<.., true>); one to four species
{
#define FINAL_PICK(N) ((void)mode, final_kernel<N, GEOM, true>)      // (mode 2, the lean specialisation of round 3, is gone: its problems run on the deferred schedule's GEN kernels)
#ifdef HYP_ONLY_ND
    (void)nd;
    return FINAL_PICK(HYP_ONLY_ND);
#else
    switch (nd) {
    case 1: return FINAL_PICK(1);
    case 2: return FINAL_PICK(2);
    case 3: return FINAL_PICK(3);
    default: return FINAL_PICK(4);
    }
#endif
#undef FINAL_PICK
}
#endif

#if HYP_PART == 4
template <int GEOM>
RayKernel pick_ray_kernel_g(int nd)
{
#ifdef HYP_ONLY_ND
    (void)nd;
    return ray_kernel<HYP_ONLY_ND, GEOM>;
#else
    switch (nd) {
    case 1: return ray_kernel<1, GEOM>;
    case 2: return ray_kernel<2, GEOM>;
    case 3: return ray_kernel<3, GEOM>;
    case 4: return ray_kernel<4, GEOM>;
    default: return ray_kernel<HYP_MAXD, GEOM>;
    }
#endif
}

#endif

#if HYP_PART == 3
template <int NDT, int GEOM>
static DeferKernels defer_kernels()
{
    DeferKernels k;
    k.propagate = final_defer_kernel<NDT, GEOM, true>; k.propagate_mono = final_defer_kernel<NDT, GEOM, true, true>; k.propagate_gen = final_defer_kernel<NDT, GEOM, true, false, true>; k.propagate_mono_gen = final_defer_kernel<NDT, GEOM, true, true, true>; k.propagate_gen_mrw = final_defer_kernel<NDT, GEOM, true, false, true, true>; k.peel_gen = peel_kernel<NDT, GEOM, false, true>; k.propagate_pre = final_defer_kernel<NDT, GEOM, false>; k.ff_walk = ff_walk_kernel<NDT, GEOM>; k.peel = peel_kernel<NDT, GEOM, false>; k.peel_inside = peel_kernel<NDT, GEOM, true>; k.reset = defer_reset_kernel<GEOM>;
    k.event_bytes = sizeof(PeelEvent<NDT, GEOM>); k.susp_bytes = sizeof(SuspRec<NDT, GEOM>); k.ff_bytes = sizeof(EmitRec<NDT>);
    k.direct = direct_column_kernel<NDT, GEOM>; k.sort_hist = peel_sort_hist_kernel<NDT, GEOM>; k.sort_scatter = peel_sort_scatter_kernel<NDT, GEOM>; k.sort_scan = peel_sort_scan_kernel;
    return k;
}

template <int GEOM>
DeferKernels pick_defer_kernels_g(int nd)
{
#ifdef HYP_ONLY_ND
    (void)nd;
    return defer_kernels<HYP_ONLY_ND, GEOM>();
#else
    switch (nd) {
    case 1: return defer_kernels<1, GEOM>();
    case 2: return defer_kernels<2, GEOM>();
    case 3: return defer_kernels<3, GEOM>();
    case 4: return defer_kernels<4, GEOM>();
    default: { DeferKernels k; memset(&k, 0, sizeof k); return k; }      // five to eight species: inline peel-off (the engine checks .propagate)
    }
#endif
}

#endif

#if HYP_PART == 1
template <int NDT, int GEOM>
static TileKernels tile_kernels()
{
    TileKernels k;
    memset(&k, 0, sizeof k);
    k.nd = NDT;
    {
        k.interact[0][0] = tile_interact_kernel<NDT, false, false, GEOM>; k.interact[1][0] = tile_interact_kernel<NDT, true, false, GEOM>;
        k.drain[0][0] = tile_drain_kernel<NDT, false, false, GEOM>; k.drain[1][0] = tile_drain_kernel<NDT, true, false, GEOM>;
        k.emit = tile_emit_kernel<NDT, GEOM, 0>; k.emit_simple = tile_emit_kernel<NDT, GEOM, 1>; k.emit_ext = tile_emit_kernel<NDT, GEOM, 2>;
        k.hot_bytes = sizeof(HotRec<NDT>); k.cold_bytes = sizeof(ColdRec<NDT>);
        k.interact_img = tile_interact_kernel<NDT, false, false, GEOM, true>; k.emit_img = tile_emit_kernel<NDT, GEOM, 1, true>;
        k.interact_img_gen = tile_interact_kernel<NDT, true, false, GEOM, true>; k.emit_img_gen = tile_emit_kernel<NDT, GEOM, 0, true>;
        k.event_bytes = sizeof(PeelEvent<NDT, GEOM>);
        k.to_susp = tile_to_susp_kernel<NDT, GEOM>;
    }
#if HYP_GEOM_TU == 3
    {
        k.interact[0][1] = tile_interact_kernel<NDT, false, true, GEOM>; k.interact[1][1] = tile_interact_kernel<NDT, true, true, GEOM>;
        k.drain[0][1] = tile_drain_kernel<NDT, false, true, GEOM>; k.drain[1][1] = tile_drain_kernel<NDT, true, true, GEOM>;
        k.walk = atile_walk_kernel<NDT>;
        k.walk_threads = HYP_ATILE_WG;
    }
#elif HYP_GEOM_TU == 1
    {
        k.interact[0][1] = tile_interact_kernel<NDT, false, true, GEOM>; k.interact[1][1] = tile_interact_kernel<NDT, true, true, GEOM>;
        k.drain[0][1] = tile_drain_kernel<NDT, false, true, GEOM>; k.drain[1][1] = tile_drain_kernel<NDT, true, true, GEOM>;
        k.walk = otile_walk_kernel<NDT>;
        k.walk_threads = HYP_OTILE_WG;
    }
#elif HYP_GEOM_TU == 0
    {
        k.interact[0][1] = tile_interact_kernel<NDT, false, true, GEOM>; k.interact[1][1] = tile_interact_kernel<NDT, true, true, GEOM>;
        k.drain[0][1] = tile_drain_kernel<NDT, false, true, GEOM>; k.drain[1][1] = tile_drain_kernel<NDT, true, true, GEOM>;
        k.walk = tile_walk_kernel<NDT, TileShape<NDT>::X, TileShape<NDT>::Y, TileShape<NDT>::Z>;
        k.walk_threads = HYP_TILE_WG;
        k.bx = TileShape<NDT>::X; k.by = TileShape<NDT>::Y; k.bz = TileShape<NDT>::Z;
    }
#elif HYP_GEOM_TU == 4 || HYP_GEOM_TU == 5
    {
        k.interact[0][1] = tile_interact_kernel<NDT, false, true, GEOM>; k.interact[1][1] = tile_interact_kernel<NDT, true, true, GEOM>;
        k.drain[0][1] = tile_drain_kernel<NDT, false, true, GEOM>; k.drain[1][1] = tile_drain_kernel<NDT, true, true, GEOM>;
        k.walk = ptile_walk_kernel<NDT, GEOM>;
        k.walk_threads = ptile_wg<NDT>();
    }
#elif HYP_GEOM_TU == 2
    {      // the modified random walk is not defined on Voronoi grids (the engine refuses it)
        k.walk = vtile_walk_kernel<NDT>;
        k.walk_threads = HYP_VTILE_WG;
    }
#endif
    return k;
}

template <int GEOM>
TileKernels pick_tile_kernels_g(int nd)
{
#ifdef HYP_ONLY_ND
    (void)nd;
    return tile_kernels<HYP_ONLY_ND, GEOM>();
#else
    switch (nd) {
    case 1: return tile_kernels<1, GEOM>();
    case 2: return tile_kernels<2, GEOM>();
    case 3: return tile_kernels<3, GEOM>();
    case 4: return tile_kernels<4, GEOM>();
    default: { TileKernels k; memset(&k, 0, sizeof k); return k; }
    }
#endif
}

#endif

#if HYP_PART == 0
template LucyKernel pick_lucy_kernel_g<HYP_GEOM_TU>(int);
#elif HYP_PART == 1
template TileKernels pick_tile_kernels_g<HYP_GEOM_TU>(int);
#elif HYP_PART == 2
template LucyKernel pick_final_kernel_g<HYP_GEOM_TU>(int);
#elif HYP_PART == 5
template LucyKernel pick_final_special_g<HYP_GEOM_TU>(int, int);
#elif HYP_PART == 3
template DeferKernels pick_defer_kernels_g<HYP_GEOM_TU>(int);
#else
template RayKernel pick_ray_kernel_g<HYP_GEOM_TU>(int);
#endif
