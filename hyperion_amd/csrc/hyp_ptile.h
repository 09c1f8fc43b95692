// hyp_ptile.h -- brick-tiled Lucy iteration for spherical and cylindrical polar grids (gfx950).
//
// The polar grids of AnalyticalYSOModel (grid_geometry_spherical_3d.f90, grid_geometry_cylindrical_3d.f90) number their cells
// (i1, i2, i3) = (r, theta, phi) or (w, z, phi) exactly like a Cartesian grid, so a BRICK is a box of indices whose densities
// and accumulators live in LDS while one workgroup walks, with ds_add_f64, the packets that sit in it -- the slot-pool schedule
// of hyp_tiled.h (interaction / emission / sort / drain kernels shared through TileCellIO<GEOM>).  The persistent kernel made one
// memory-side atomic per crossing and carried emission and interaction code next to a wall search of two to four quadratics
// (256 VGPRs + 143 spilled on the spherical grid); this kernel only walks.
//
// The step is find_wall of the geometry (hyp_polar.h: geo_find_wall, the reference's operations in the reference's order, so
// every packet's history is the persistent kernel's and the oracle's), the density of the cell from LDS, the deposit into LDS,
// next_cell (phi periodic).  A packet whose next cell is outside the brick goes back to its slot record; the propagation check
// runs in the service phase with the general function.
#pragma once

#include "hyp_tiled.h"

#ifndef HYP_PTILE_WG
#define HYP_PTILE_WG 768         // threads per workgroup (one workgroup per task, one per CU)
#endif
#ifndef HYP_PTILE_OCC
#define HYP_PTILE_OCC 3          // waves per SIMD the register budget is set for
#endif
#ifndef HYP_PTILE_SERVICE
#define HYP_PTILE_SERVICE 16     // lanes that must wait before a wave runs its service phase
#endif
#ifndef HYP_PTILE_STEPS
#define HYP_PTILE_STEPS 4        // cell steps between two scheduling decisions of a wave
#endif
#define PT_HIST 256              // bricks whose packet counts a task collects in LDS (the others: global atomics)


// ---------------------------------------------------------------------------------------------------------------------
// Mixed-precision wall search on spherical grids (round 4).  find_wall of the reference (spherical_3d.f90:741-1073) solves up
// to four quadratics per step in FP64 -- two spheres, two cones: a square root and one or two IEEE divisions each -- and the
// walk kernel is bound by issuing them (profiles/r04_sph_summary.md: 84 % VALU busy; without the cones the kernel is 1.9x as
// fast).  Here every candidate root is first bounded in FP32: positions scaled by the grid's outer radius, cones written with
// cos^2 / sin^2 of the wall (no tan^2 -> infinity at the mid-plane), each root with a first-order error bound
//     e_t = 4 (t^2 E_a + |t| E_b + E_c) / sqrt(delta) + U |t|,     E_a = U, E_b = U rho, E_c = U rho^2,  U = 32 x 2^-24,
// (rho = |r|; the coefficient bounds cover the rounding of the inputs to FP32 and of every operation with a factor > 4 to
// spare; the bound is only used where it is small against the distance between the two roots).  If ONE candidate's upper
// bound lies below the lower bounds of all others by more than the largest merge epsilon of the brick's walls, that wall is
// the reference's answer: no other candidate can win or be merged with it by insert_t, whatever the order.  Its quadratic is
// then solved ONCE in FP64 with the reference's expressions (quad_full with a = 1 is quad_reduced bit for bit), the side test
// of a cone and the on-the-wall rule included, and the root is checked against the FP32 interval.  Anything else -- a sign or
// a side that the bounds cannot decide, a grazing ray, a direction along a cone, two candidates closer than the bounds, phi
// walls (3-D grids) -- returns false and the caller runs geo_find_wall.  -DHYP_PTILE_VERIFY runs both and counts disagreements.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef HYP_PTILE_FAST
#define HYP_PTILE_FAST 0       // measured SLOWER than the reference's search (623 against 493 ms at 2e7 packets: ~2000 instructions of bounds and selects per step, and 8 % of the wave-steps still run the reference's search for one lane); kept for the record, profiles/r04_tiled_log.md
#endif
#define PT_U (16.0f * 0x1p-24f)
// v_rcp_f32 / v_sqrt_f32: one ulp each, inside the U |t| term of the bounds (an IEEE FP32 division is ten instructions)
#define PT_RCP(x) __builtin_amdgcn_rcpf(x)
#define PT_SQRT(x) __builtin_amdgcn_sqrtf(x)

struct PtCand { float best_hi, best_lo, other_lo; int best; bool amb; int why; };      // why: first reason for amb (verify builds)

// a root t known to e: a candidate of insert_t if t > 0.  T: nothing beyond it can matter any more (3e38: not known yet)
__device__ __forceinline__ void pt_add(PtCand &C, float t, float e, int id, float T)
{
    const float lo = t - e, hi = t + e;
    if (hi <= 0.0f || lo > T) return;            // certainly not positive, or certainly behind the walls already found
    if (!(lo > 0.0f)) { C.amb = true; if (!C.why) C.why = 1; return; }  // the sign of t cannot be told (NaN ends here too)
    if (hi < C.best_hi) { C.other_lo = fminf(C.other_lo, C.best_lo); C.best_hi = hi; C.best_lo = lo; C.best = id; }
    else C.other_lo = fminf(C.other_lo, lo);
}

// The candidates of one curved wall a t^2 + b t + c = 0 whose coefficients are known to (Ea, Eb, Ec).  on_it: the packet sits on the
// wall and insert_pair takes the root of larger magnitude; cone: a root counts only on the nappe of the wall (z + v_z t on the
// side of cw), otherwise it is +huge.  Roots behind T are left out whatever their sign; a root whose first-order bound
// e = 4 (t^2 Ea + |t| Eb + Ec) / sqrt(delta) is not small against the distance of the roots is bounded from below through
// t1 + t2 = -b / a instead.
__device__ __forceinline__ void pt_curved(PtCand &C, float a, float b, float c, float Ea, float Eb, float Ec, bool on_it, bool cone, float z, float vz, float cw,
                                          int id, float T)
{
    const float ac4 = 4.0f * a * c, bb = b * b;
    const float delta = bb - ac4;
    const float Ed = 2.0f * fabsf(b) * Eb + 4.0f * (fabsf(a) * Ec + fabsf(c) * Ea) + PT_U * (bb + fabsf(ac4));
    if (delta < -2.0f * Ed) return;                                      // certainly no real root
    if (!(delta > 4.0f * Ed)) { C.amb = true; if (!C.why) C.why = 4; return; }
    const float sd = PT_SQRT(delta), isd4 = 4.0f * PT_RCP(sd);
    const float q = -0.5f * (b + copysignf(sd, b));
    // t2 = c / q is the root of smaller magnitude, t1 = q / a the other one (|q| >= sd / 2 > 0; a may be anything)
    const float t2 = c * PT_RCP(q), e2 = (t2 * t2 * Ea + fabsf(t2) * Eb + Ec) * isd4 + PT_U * fabsf(t2);
    const bool ok2 = 8.0f * fabsf(a) * e2 < sd;
    if (!ok2) { C.amb = true; if (!C.why) C.why = 5; return; }
    const float t1 = q * PT_RCP(a), e1 = (t1 * t1 * Ea + fabsf(t1) * Eb + Ec) * isd4 + PT_U * fabsf(t1);
    const bool ok1 = fabsf(a) > 4.0f * Ea && 8.0f * fabsf(a) * e1 < sd;
    // |t1| >= |b / a| - |t2|
    const float far_lo = (fabsf(b) - Eb) * PT_RCP(fabsf(a) + Ea) * (1.0f - PT_U) - (fabsf(t2) + e2);
    const bool far_away = far_lo > T && far_lo > fabsf(t2) + e2;         // behind everything, and the larger one of the two
    if (!ok1 && !far_away) { C.amb = true; if (!C.why) C.why = 6; return; }
    int k1 = 1, k2 = 1;
    if (cone) {
        const float z2 = z + vz * t2, ez2 = PT_U * (fabsf(z) + fabsf(vz * t2)) + fabsf(vz) * e2;
        const bool rel2 = on_it || (t2 + e2 > 0.0f && t2 - e2 <= T);      // does the side of this root matter?
        if (rel2 && !(fabsf(z2) > ez2)) { C.amb = true; if (!C.why) C.why = 3; return; }
        k2 = ((z2 > 0.0f) == (cw > 0.0f)) ? 1 : 0;
        if (ok1 && !far_away) {
            const float z1 = z + vz * t1, ez1 = PT_U * (fabsf(z) + fabsf(vz * t1)) + fabsf(vz) * e1;
            const bool rel1 = on_it || (t1 + e1 > 0.0f && t1 - e1 <= T);
            if (rel1 && !(fabsf(z1) > ez1)) { C.amb = true; if (!C.why) C.why = 3; return; }
            k1 = ((z1 > 0.0f) == (cw > 0.0f)) ? 1 : 0;
        }
    }
    if (on_it) {
        // insert_pair: the root of larger magnitude, +huge (wrong nappe) included; +huge changes nothing
        if (!k2) return;
        if (far_away) return;                    // t1 is the larger one and lies behind T (or is negative, or +huge): nothing to insert
        if (!k1) return;
        if (!(fabsf(t1) - e1 > fabsf(t2) + e2)) { C.amb = true; if (!C.why) C.why = 2; return; }
        pt_add(C, t1, e1, id, T);
    } else {
        if (k2) pt_add(C, t2, e2, id, T);
        if (k1 && !far_away) pt_add(C, t1, e1, id, T);
    }
}

// wr2f[i - x0] = (float)(w1[i]^2 S^2), cf[j - y0] = (float)cos(theta_j), s2f[j - y0] = (float)sin^2(theta_j) for the brick's walls
__device__ __forceinline__ bool sph_fast_wall(const DProblem &P, const double r[3], const double v[3], const Cell<GEOM_SPH> &c,
                                              const float *__restrict__ wr2f, const float *__restrict__ cf, const float *__restrict__ s2f,
                                              int x0, int y0, double S, float mrg, double &tnear, int im[3], int *why = nullptr)
{
    const float x = (float)(r[0] * S), y = (float)(r[1] * S), z = (float)(r[2] * S);
    const float vx = (float)v[0], vy = (float)v[1], vz = (float)v[2];
    const float r2xy = x * x + y * y, r2z = z * z, rvxy = x * vx + y * vy, rvz = z * vz, v2xy = vx * vx + vy * vy, v2z = vz * vz;
    const float arvxy = fabsf(x * vx) + fabsf(y * vy), arvz = fabsf(rvz);      // (the magnitudes the rounding errors scale with)
    const float r2 = r2xy + r2z;
    PtCand C; C.best_hi = 3.0e38f; C.best_lo = 3.0e38f; C.other_lo = 3.0e38f; C.best = -1; C.amb = false; C.why = 0;
    const int i1 = c.ic[0], i2 = c.ic[1];
    // spheres: t^2 + b t + (|r|^2 - R^2) = 0
    const float bs = 2.0f * (rvxy + rvz), Ebs = 2.0f * PT_U * (arvxy + arvz);
    if (!c.radial) {
        const float R2 = wr2f[i1 - x0];
        pt_curved(C, 1.0f, bs, r2 - R2, 0.0f, Ebs, PT_U * (r2 + R2), c.ow[0] == -1, false, 0.0f, 0.0f, 0.0f, 0, 3.0e38f);
    }
    {
        const float R2 = wr2f[i1 + 1 - x0];
        pt_curved(C, 1.0f, bs, r2 - R2, 0.0f, Ebs, PT_U * (r2 + R2), c.ow[0] == +1, false, 0.0f, 0.0f, 0.0f, 1, 3.0e38f);
    }
    if (C.amb) { if (why) *why = C.why ? C.why : 4; return false; }
    // cones: (v_xy^2 c^2 - v_z^2 s^2) t^2 + 2 (r.v_xy c^2 - r.v_z s^2) t + (r_xy^2 c^2 - r_z^2 s^2) = 0 on the wall's side of the mid-plane.
    // T: behind the nearest sphere nothing matters.  A nearly radial ray (every packet before its first interaction) makes all three
    // coefficients small together and the roots meaningless (the apex, far behind): such a wall is dismissed WITHOUT solving -- the
    // parabola keeps its sign on [0, T] when both ends have the same sign by more than their error and by more than its bulge |a| T^2 / 4.
    const float T = C.best_hi + mrg;
#pragma unroll
    for (int side = 0; side < 2; side++) {
        if (side == 0 ? i2 > 0 : i2 < P.n2 - 1) {
            const int iw = i2 + side, dir = side ? +1 : -1;
            const bool on_it = c.ow[1] == dir;
            if (iw == P.midplane && v[2] != 0.0) {
                if (on_it && fabsf(vz) < 1.0e-6f) { C.amb = true; if (!C.why) C.why = 7; }      // (a direction in the mid-plane: the reference's iext rule may apply)
                if (!on_it) { const float t = -z * PT_RCP(vz); pt_add(C, t, 8.0f * PT_U * fabsf(t) + PT_U * fabsf(z), 2 + side, T); }
            } else {
                const float cw = cf[iw - y0], c2 = cw * cw, s2 = s2f[iw - y0];
                const float a = v2xy * c2 - v2z * s2, b = 2.0f * (rvxy * c2 - rvz * s2), cc = r2xy * c2 - r2z * s2;
                const float Ea = PT_U * (v2xy * c2 + v2z * s2), Eb = 2.0f * PT_U * (arvxy * c2 + arvz * s2), Ec = PT_U * (r2xy * c2 + r2z * s2);
                bool skip = false;
                if (!on_it && T < 1.0e37f) {
                    const float pT = (a * T + b) * T + cc, EpT = 1.5f * ((Ea * T + Eb) * T + Ec) + PT_U * (fabsf(a) * T * T + fabsf(b) * T + fabsf(cc));
                    const float m0 = fabsf(cc) - 1.5f * Ec, mT = fabsf(pT) - EpT;
                    skip = ((cc > 0.0f) == (pT > 0.0f)) && fminf(m0, mT) > 0.25f * (fabsf(a) + Ea) * T * T;
                }
                if (!skip) pt_curved(C, a, b, cc, Ea, Eb, Ec, on_it, true, z, vz, cw, 2 + side, T);
            }
        }
    }
    if (C.amb || C.best < 0 || !(C.other_lo > C.best_hi + mrg)) { if (why) *why = C.amb ? (C.why ? C.why : 4) : (C.best < 0 ? 5 : 6); return false; }
    // ---- the winner's wall, exactly: the reference's expressions (geo_find_wall / sph_wall_cone in hyp_polar.h) ----
    const int w = C.best;
    const double v2_xy = v[0] * v[0] + v[1] * v[1], v2_z = v[2] * v[2];
    const double rv_xy = r[0] * v[0] + r[1] * v[1], rv_z = r[2] * v[2];
    const double r2_xy = r[0] * r[0] + r[1] * r[1], r2_z = r[2] * r[2];
    double tw;
    if (w >= 2 && i2 + (w - 2) == P.midplane && v[2] != 0.0) tw = -r[2] / v[2];
    else {
        double qa, qb, qc;
        bool on_it;
        double tt = 1.0;
        if (w < 2) {
            double pB = rv_xy + rv_z; pB = pB + pB;
            const double pC = r2_xy + r2_z;
            qa = 1.0; qb = pB; qc = pC - P.wr2[i1 + w];
            on_it = c.ow[0] == (w ? +1 : -1);
        } else {
            const int iw = i2 + (w - 2);
            const double tt2 = P.wtant2[iw];
            tt = P.wtant[iw];
            qa = v2_xy - v2_z * tt2;
            double pB = rv_xy - rv_z * tt2; pB = pB + pB;
            qb = pB; qc = r2_xy - r2_z * tt2;
            on_it = c.ow[1] == (w == 3 ? +1 : -1);
            if (!(fabs(qa) > 0.0)) return false;
        }
        double x1, x2;
        quad_full(qa, qb, qc, x1, x2);
        if (w >= 2) {
            const double z1 = r[2] + v[2] * x1;
            if ((z1 > 0.0) != (tt > 0.0)) x1 = HYP_DBL_MAX;
            const double z2 = r[2] + v[2] * x2;
            if ((z2 > 0.0) != (tt > 0.0)) x2 = HYP_DBL_MAX;
        }
        if (on_it) tw = fabs(x1) < fabs(x2) ? x2 : x1;
        else {
            // both roots go through insert_t; they are further apart than any epsilon (checked above), so the smaller positive one stays
            const double p1 = x1 > 0.0 ? x1 : HYP_DBL_MAX, p2 = x2 > 0.0 ? x2 : HYP_DBL_MAX;
            tw = p1 < p2 ? p1 : p2;
        }
    }
    const double ts = tw * S;
    if (!(tw > 0.0) || !(ts >= (double)C.best_lo) || !(ts <= (double)C.best_hi)) { if (why) *why = 7; return false; }      // not what the bounds promised: the reference's loop decides
    tnear = tw;
    im[0] = w == 0 ? -1 : (w == 1 ? +1 : 0); im[1] = w == 2 ? -1 : (w == 3 ? +1 : 0); im[2] = 0;
    return true;
}

// TileGeom: bx, by, bz = brick size in cells (the last brick of an axis is ragged), nbx, nby, nbz = bricks per axis
template <int ND, int GEOM>
__global__ __launch_bounds__(HYP_PTILE_WG, HYP_PTILE_OCC) void ptile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                                  void *__restrict__ hot_v, void *__restrict__ cold_v,
                                                                  const int *__restrict__ order,
                                                                  const TileTask *__restrict__ tasks, int *__restrict__ slot_brick,
                                                                  int *__restrict__ ilist, int *__restrict__ dlist,
                                                                  TileCount *__restrict__ tcount, unsigned int *__restrict__ counts)
{
    extern __shared__ double lds[];
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    const DProblem &P = *Pp;
    if (blockIdx.x >= ctl->n_tasks[T.pool]) return;
    const TileTask tk = tasks[blockIdx.x];
    const int vs = T.vsplit > 1 ? T.vsplit : 1;
    const bool reach_task = vs > 1 && tk.brick % vs == 0;      // packets that have not interacted yet: cone walls dismissed without solving where the test allows
    const int cl = tk.brick / vs;       // (vsplit: the task's packets are all of one kind, tk.brick % vs)
    const int x0 = (cl % T.nbx) * T.bx, y0 = ((cl / T.nbx) % T.nby) * T.by, z0 = (cl / (T.nbx * T.nby)) * T.bz;
    const int x1 = min(x0 + T.bx, P.n1), y1 = min(y0 + T.by, P.n2), z1 = min(z0 + T.bz, P.n3);
    const int bx = x1 - x0, by = y1 - y0, bz = z1 - z0, nc = bx * by * bz;
    double *dens = lds;
    double *accum = dens + (size_t)T.bx * T.by * T.bz * ND;
    __shared__ int next_pkt, n_int_l, n_dead_l, pub_base[2];
    __shared__ unsigned int nb_cnt[PT_HIST];
    __shared__ double red[TILE_RED_N];
    // tables of the mixed-precision wall search (spherical grids, sph_fast_wall): behind the accumulators
    float *wr2f = (float *)(accum + (size_t)T.bx * T.by * T.bz * ND), *cf = wr2f + (T.bx + 2), *s2f = cf + (T.by + 2);
    __shared__ unsigned int mrg_bits;
    double S_scale = 0.0;
    bool fast_ok = false;
    if constexpr (GEOM == GEOM_SPH && HYP_PTILE_FAST) {
        const double r_out = P.w[0][P.n1];
        S_scale = 1.0 / r_out;
        fast_ok = P.n_dim != 3 && r_out > 0.0 && S_scale > 0.0 && r_out < HYP_DBL_MAX && S_scale < HYP_DBL_MAX;
        if (threadIdx.x == 0) mrg_bits = 0u;
        __syncthreads();
        float m = 0.0f;
        for (int i = threadIdx.x; i <= bx; i += blockDim.x) {
            wr2f[i] = (float)((P.wr2[x0 + i] * S_scale) * S_scale);
            m = fmaxf(m, (float)(P.ew[0][x0 + i] * S_scale));
        }
        for (int j = threadIdx.x; j <= by; j += blockDim.x) {
            const double ct = P.wcost[y0 + j];
            cf[j] = (float)ct; s2f[j] = (float)(1.0 - ct * ct);
            m = fmaxf(m, (float)(P.ew[1][y0 + j] * S_scale));
        }
        if (m > 0.0f) atomicMax(&mrg_bits, __float_as_uint(m));      // (positive floats order like their bit patterns)
    }
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        const int lx = c % bx, ly = (c / bx) % by, lz = c / (bx * by);
        const size_t gid = ((size_t)(z0 + lz) * P.n2 + (y0 + ly)) * P.n1 + (x0 + lx);
#pragma unroll
        for (int d = 0; d < ND; d++) { dens[c * ND + d] = P.density[gid * ND + d]; accum[c * ND + d] = 0.0; }
    }
    for (int i = threadIdx.x; i < PT_HIST; i += blockDim.x) nb_cnt[i] = 0;
    if (threadIdx.x >= 256 && threadIdx.x < 256 + TILE_RED_N) red[threadIdx.x - 256] = 0.0;
    if (threadIdx.x == 320) { next_pkt = 0; n_int_l = 0; n_dead_l = 0; }
    __syncthreads();

    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    // lane state: the walking part of a packet (the rest stays in its ColdRec)
    double r[3] = {0.0, 0.0, 0.0}, v[3] = {1.0, 0.0, 0.0}, tau_req = 0.0, tau_ach = 0.0, energy = 0.0, chi[ND], kappa[ND];
    double t_src = HYP_INF, t_ach = 0.0;     // re-absorption by sources (P.any_intersect): see Packet
    Cell<GEOM> cell;
    cell.ic[0] = x0; cell.ic[1] = y0; cell.ic[2] = z0; cell.ow[0] = cell.ow[1] = cell.ow[2] = 0;
    if constexpr (GEOM == GEOM_SPH) cell.radial = 0;
    Rng g; g.key0 = P.seed_key; g.key1 = T.iter_tag; g.blk_a = 0; g.have_a = 0; g.buf_a = 0.0;
    g.id_lo = g.id_hi = 0; g.blk_b = 0; g.countdown = 0;
    int slot = -1;                            // (bit 30: HotRec::pad, the kind of the packet's next interaction: TileGeom::presort, hyp_tiled.h)
#define SLOT (slot & 0x3fffffff)
    int st = LS_IDLE;
    bool exhausted = false;
#pragma unroll
    for (int d = 0; d < ND; d++) { chi[d] = 0.0; kappa[d] = 0.0; }
    const Walls W = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {0, 0, 0}};
    const float mrg = (GEOM == GEOM_SPH && HYP_PTILE_FAST) ? 8.0f * __uint_as_float(mrg_bits) + 1.0e-30f : 0.0f;
#ifdef HYP_PTILE_VERIFY
    unsigned long long dbg_fast = 0, dbg_slow = 0, dbg_bad = 0;
#endif

    bool queue_empty = false;        // wave-uniform: some lane found the task's queue empty
    for (;;) {
        if (queue_empty && st == LS_IDLE) exhausted = true;
        const unsigned long long m_walk = __ballot(st == LS_WALK);
        const unsigned long long m_out = __ballot(st >= LS_LEFT);      // anything the service phase must look at
        const unsigned long long m_idle = __ballot(st == LS_IDLE && !exhausted);
        if (!(m_walk | m_out | m_idle)) break;
        // tail of a task: the last few walking packets of a wave go back to their slots and continue next generation
        const bool park = !m_idle && queue_empty && __popcll(m_walk) <= T.park;
        // ---- service phase: write finished visits back, take new packets ----
        if (park || ((m_out | m_idle) && (__popcll(m_out | m_idle) >= HYP_PTILE_SERVICE || !m_walk))) {
            // propagation check (grid_propagate_3d.f90:112-120), then the step goes on as usual
            if (st == LS_CHECK) {
                const int gap = rng_check_gap(g, P.check_p, P.check_log1mp);
                g.countdown = gap < 0x7fffffff ? gap + 1 : gap;      // the step below takes one off again
                if (geo_check_cell(P, W, r, v, cell)) st = LS_WALK;
                else { cnt.killed_geo++; st = LS_DEAD; }
            }
            if (st == LS_DEAD) {
                hot[SLOT].state = TS_DEAD; slot_brick[SLOT] = TILE_NEEDS_PREPARE;
                dlist[tk.start + atomicAdd(&n_dead_l, 1)] = SLOT;
                finished++; st = LS_IDLE;
            } else if (st == LS_LEFT || st == LS_HIT || st == LS_REABS || (park && st == LS_WALK)) {
                HotRec<ND> &H = hot[SLOT];
                if (P.any_intersect) cold[SLOT].t_ach = t_ach;
                int state = TS_WALK;
                if (st == LS_REABS) { state = TS_REEMIT; slot_brick[SLOT] = TILE_NEEDS_REEMIT; }
                else if (st == LS_HIT) { state = TS_INTERACT; slot_brick[SLOT] = TILE_NEEDS_INTERACT; }
                else {
                    const int nb = st == LS_LEFT ? brick_of(T, cell.ic) * vs + tk.brick % vs : tk.brick;           // parked: same brick again
                    if (st == LS_LEFT) slot_brick[SLOT] = nb;
                    if (nb < PT_HIST) atomicAdd(&nb_cnt[nb], 1u); else atomicAdd(&counts[nb], 1u);
                }
                if (st == LS_REABS || st == LS_HIT) ilist[tk.start + atomicAdd(&n_int_l, 1)] = (T.presort && st == LS_HIT) ? slot : SLOT;
                // the 64-byte line a visit changes, as four 16-byte stores (r | r, tau_ach | ic, ow | countdown, blk_b, state, pad)
                int ow = pack_ow(cell.ow);
                if constexpr (GEOM == GEOM_SPH) ow |= cell.radial << 6;
                double2 *line = (double2 *)&H;
                line[0] = make_double2(r[0], r[1]);
                line[1] = make_double2(r[2], tau_ach);
                ((int4 *)line)[2] = make_int4(cell.ic[0], cell.ic[1], cell.ic[2], ow);
                ((int4 *)line)[3] = make_int4(g.countdown, (int)g.blk_b, state, (slot >> 30) & 1);
                st = LS_IDLE;
            }
            if (park) break;
            if (st == LS_IDLE && !exhausted) {
                const int j = atomicAdd(&next_pkt, 1);
                if (j >= tk.len) exhausted = true;
                else {
                    slot = order[tk.start + j];
                    const HotRec<ND> &H = hot[SLOT];
#pragma unroll
                    for (int a = 0; a < 3; a++) { r[a] = H.r[a]; v[a] = H.v[a]; cell.ic[a] = H.ic[a]; }
                    const int ow = H.ow;
                    unpack_ow(ow & 63, cell.ow);
                    if constexpr (GEOM == GEOM_SPH) cell.radial = (ow >> 6) & 1;
                    tau_req = H.tau_req; tau_ach = H.tau_ach; energy = H.energy;
#pragma unroll
                    for (int d = 0; d < ND; d++) { chi[d] = H.chi[d]; kappa[d] = H.kappa[d]; }
                    const unsigned long long id = H.id;
                    g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
                    g.countdown = H.countdown; g.blk_b = H.blk_b;
                    slot |= (H.pad & 1) << 30;
                    if (P.any_intersect) { t_src = cold[SLOT].t_src; t_ach = cold[SLOT].t_ach; }
                    st = LS_WALK;
                }
            }
            if (__ballot(exhausted)) queue_empty = true;
        }
        // ---- a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232) ----
#pragma unroll 1
        for (int q = 0; q < HYP_PTILE_STEPS; q++) {
            if (st == LS_WALK) {
                if (g.countdown == 0) st = LS_CHECK;
                else {
                    g.countdown--;
                    double tmin; int im[3];
                    bool found;
                    if constexpr (GEOM == GEOM_SPH && HYP_PTILE_FAST) {
#ifdef HYP_PTILE_VERIFY
                        int why = 0;
                        found = fast_ok && sph_fast_wall(P, r, v, cell, wr2f, cf, s2f, x0, y0, S_scale, mrg, tmin, im, &why);
                        if (!found) atomicAdd(&ctl->dbg[30 + (why & 7)], 1ull);
#else
                        found = fast_ok && sph_fast_wall(P, r, v, cell, wr2f, cf, s2f, x0, y0, S_scale, mrg, tmin, im);
#endif
#ifdef HYP_PTILE_VERIFY
                        if (found) {
                            double t_ref; int im_ref[3];
                            const bool f_ref = geo_find_wall(P, W, r, v, cell, t_ref, im_ref);
                            dbg_fast++;
                            if (!f_ref || t_ref != tmin || im_ref[0] != im[0] || im_ref[1] != im[1] || im_ref[2] != im[2]) { dbg_bad++; tmin = t_ref; im[0] = im_ref[0]; im[1] = im_ref[1]; im[2] = im_ref[2]; found = f_ref; }
                        } else { dbg_slow++; found = geo_find_wall(P, W, r, v, cell, tmin, im); }
#else
                        if (!found) found = geo_find_wall(P, W, r, v, cell, tmin, im);
#endif
                    } else if constexpr (GEOM == GEOM_SPH) found = sph_find_wall(P, W, r, v, cell, tmin, im, reach_task);
                    else found = geo_find_wall(P, W, r, v, cell, tmin, im);
                    if (!found) { cnt.killed_geo++; st = LS_DEAD; }
                    else {
                        const int loc = ((cell.ic[2] - z0) * by + (cell.ic[1] - y0)) * bx + (cell.ic[0] - x0);
                        double rho[ND], chi_rho = 0.0;
#pragma unroll
                        for (int d = 0; d < ND; d++) { rho[d] = dens[loc * ND + d]; chi_rho += chi[d] * rho[d]; }
                        const double tau_cell = chi_rho * tmin;
                        const double tau_needed = tau_req - tau_ach;
                        cnt.crossings++;
                        if (tau_cell < tau_needed) {
                            bool reabs = false;
                            if (P.any_intersect) { t_ach += tmin; reabs = t_ach > t_src; }      // :139-143
                            if (reabs) st = LS_REABS;
                            else {
#pragma unroll
                                for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
                                tau_ach += tau_cell;
#pragma unroll
                                for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tmin * kappa[d] * energy);
                                geo_advance(P, r, cell, im);
                                if (geo_escaped(P, cell)) st = LS_DEAD;         // left the grid: the packet ends here
                                else if (cell.ic[0] < x0 || cell.ic[0] >= x1 || cell.ic[1] < y0 || cell.ic[1] >= y1 || cell.ic[2] < z0 || cell.ic[2] >= z1) st = LS_LEFT;
                            }
                        } else {
                            // the interaction happens inside this cell: grid_propagate_3d.f90:170-200
                            const double tact = tmin * (tau_needed / tau_cell);
                            bool reabs = false;
                            if (P.any_intersect) { t_ach += tact; reabs = t_ach > t_src; }      // :184-188
                            if (reabs) st = LS_REABS;
                            else {
#pragma unroll
                                for (int a = 0; a < 3; a++) r[a] = r[a] + tact * v[a];
                                tau_ach += tau_needed;
                                geo_clear_wall(cell);
#pragma unroll
                                for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tact * kappa[d] * energy);
                                st = LS_HIT;
                            }
                        }
                    }
                }
            }
        }
    }
#ifdef HYP_PTILE_VERIFY
    atomicAdd(&ctl->dbg[37], dbg_fast); atomicAdd(&ctl->dbg[38], dbg_slow); atomicAdd(&ctl->dbg[39], dbg_bad);
#endif
    __syncthreads();
    for (int i = threadIdx.x; i < PT_HIST; i += blockDim.x) if (nb_cnt[i]) atomicAdd(&counts[i], nb_cnt[i]);
    tile_walk_publish_lists(T, ctl, tk, ilist, dlist, n_int_l, n_dead_l, pub_base);
    // flush the brick's accumulators (replica chosen like in the persistent kernel)
    double *sum = P.sum;
    if (P.n_copies > 1) {
        unsigned c = xcc_id();
        if (P.n_copies > 8) c += 8u * ((blockIdx.x >> 3) % (unsigned)(P.n_copies >> 3));
        sum += (size_t)(c % (unsigned)P.n_copies) * P.copy_stride;
    }
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        const int lx = c % bx, ly = (c / bx) % by, lz = c / (bx * by);
        const size_t gid = ((size_t)(z0 + lz) * P.n2 + (y0 + ly)) * P.n1 + (x0 + lx);
#pragma unroll
        for (int d = 0; d < ND; d++) {
            const double val = accum[c * ND + d];
            if (val != 0.0) hyp_atomic_add_g(&sum[gid * ND + d], val);
        }
    }
#undef SLOT
    block_tally_flush(P, ctl, red, cnt, finished);
}
