// hyp_pick.h -- the propagation kernels are instantiated per geometry in separate translation
// units (hyp_geom.hip compiled once per GEOM_*, in parallel); the host side of the C-ABI
// (hyp_engine.hip) obtains their entry points through these functions.
#pragma once
#include "hyp_device.h"

using LucyKernel = void (*)(const DProblem *, LaunchParams);
using RayKernel = void (*)(const DProblem *, LaunchParams, int, double);

template <int GEOM> LucyKernel pick_lucy_kernel_g(int nd);    // lucy_kernel<nd, GEOM>
template <int GEOM> LucyKernel pick_final_kernel_g(int nd, bool plain);   // final_kernel<nd, GEOM, plain>
template <int GEOM> RayKernel pick_ray_kernel_g(int nd);      // ray_kernel<nd, GEOM>

// deferred peel-off (hyp_defer.h): final_defer_kernel<nd, GEOM> / peel_kernel<nd, GEOM> and their record sizes
using DeferKernel = void (*)(const DProblem *, LaunchParams, DeferBuf);
using PeelKernel = void (*)(const DProblem *, DeferBuf, uint32_t);
struct DeferKernels { DeferKernel propagate; PeelKernel peel; void (*reset)(PeelCtl *, int, int); size_t event_bytes, susp_bytes; };
template <int GEOM> DeferKernels pick_defer_kernels_g(int nd);
