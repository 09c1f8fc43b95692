// hyp_pick.h -- the propagation kernels are instantiated per geometry in separate translation
// units (hyp_geom.hip compiled once per GEOM_*, in parallel); the host side of the C-ABI
// (hyp_engine.hip) obtains their entry points through these functions.
#pragma once
#include "hyp_device.h"

using LucyKernel = void (*)(const DProblem *, LaunchParams);
using RayKernel = void (*)(const DProblem *, LaunchParams, int, double);

template <int GEOM> LucyKernel pick_lucy_kernel_g(int nd);    // lucy_kernel<nd, GEOM>
template <int GEOM> LucyKernel pick_final_kernel_g(int nd);               // final_kernel<nd, GEOM, false>: the general imaging kernel
template <int GEOM> LucyKernel pick_final_special_g(int nd, int mode);    // its specialisations, one to four species: 1 plain, 2 lean
template <int GEOM> RayKernel pick_ray_kernel_g(int nd);      // ray_kernel<nd, GEOM>

// deferred peel-off (hyp_defer.h): final_defer_kernel<nd, GEOM> / peel_kernel<nd, GEOM> and their record sizes
using DeferKernel = void (*)(const DProblem *, LaunchParams, DeferBuf);
using PeelKernel = void (*)(const DProblem *, DeferBuf, uint32_t);
using PeelSortK = void (*)(const DProblem *, DeferBuf);
struct DeferKernels {
    DeferKernel propagate, propagate_pre, ff_walk;       // propagate_pre: with the forced-first walks made ahead (ff_walk)
    DeferKernel propagate_gen, propagate_mono_gen, propagate_gen_mrw; PeelKernel peel_gen;      // sources with a surface (final_defer_kernel<.., true, false, true>, peel_kernel<.., false, true>)
    DeferKernel propagate_mono;                          // a launch of the monochromatic iteration (final_defer_kernel<.., true, true>)
    PeelKernel peel, peel_inside; void (*reset)(PeelCtl *, int, int); size_t event_bytes, susp_bytes, ff_bytes;
    void (*direct)(const DProblem *, DirectCol *);       // direct_column_kernel<nd, GEOM>
    PeelSortK sort_hist, sort_scatter; void (*sort_scan)(DeferBuf);  // sorted peel-off: keys + histogram, scatter (peel_sort_scan_kernel between them)
};
template <int GEOM> DeferKernels pick_defer_kernels_g(int nd);

// brick- / cluster-tiled Lucy iteration (hyp_tiled.h, hyp_vtile.h): the kernels of one geometry and species count.
// Slot records travel as void * (HotRec<nd> / ColdRec<nd>).
struct TileGeom; struct TileCtl; struct TileTask; struct TileCount;
using TileInteractK = void (*)(const DProblem *, TileGeom, TileCtl *, void *, void *, int *, const TileTask *, const int *, int *, TileCount *,
                               unsigned int *, int *, DeferBuf);
using TileEmitK = void (*)(const DProblem *, TileGeom, TileCtl *, void *, void *, int *, const TileTask *, const int *, const TileCount *,
                           unsigned int *, int *, DeferBuf);
using TileDrainK = void (*)(const DProblem *, TileGeom, TileCtl *, void *, void *, int *);
using TileToSuspK = void (*)(const DProblem *, TileGeom, TileCtl *, void *, void *, int *, DeferBuf);
using TileWalkK = void (*)(const DProblem *, TileGeom, TileCtl *, void *, void *, const int *, const TileTask *, int *, int *, int *, TileCount *,
                           unsigned int *);
struct TileKernels {
    TileInteractK interact[2][2];       // [sources can re-absorb][modified random walk]
    TileEmitK emit, emit_simple, emit_ext;      // emit_simple: point sources with tabulated / blackbody spectra only; emit_ext: those + external sources
    TileToSuspK to_susp;                // the imaging iteration's end-game: live slots -> SuspRec of the deferred schedule (tile_to_susp_kernel)
    TileInteractK interact_img; TileEmitK emit_img;      // the imaging iteration on this schedule (IMG kernels of hyp_tiled.h); event_bytes = sizeof(PeelEvent)
    TileInteractK interact_img_gen; TileEmitK emit_img_gen;      // ... of problems with general sources (a surface that emits with limb darkening and re-absorbs)
    size_t event_bytes;
    TileDrainK drain[2][2];
    TileWalkK walk;
    size_t hot_bytes, cold_bytes;
    int walk_threads;
    int bx, by, bz;                     // Cartesian: brick shape; LDS of the walk = walls + 2 x 8 B x bx by bz nd
    int nd;
};
template <int GEOM> TileKernels pick_tile_kernels_g(int nd);      // .walk == nullptr: no tiled schedule for this geometry
