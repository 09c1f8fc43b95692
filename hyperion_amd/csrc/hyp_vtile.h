// hyp_vtile.h -- cluster-tiled Lucy iteration for Voronoi grids (gfx950).
//
// The persistent kernel reads, per cell crossing, one 32-byte wall record per neighbour (~15) from wherever it lives in
// the memory hierarchy (49 MB at 100 000 sites: L2 misses, ~500 B of L2<->fabric traffic per crossing) and makes one
// memory-side atomic per species.  Here the cells are grouped at set-up into spatially compact CLUSTERS (recursive
// coordinate bisection of the sites) whose tables, densities and accumulators fit in LDS.  The slot-pool schedule of
// hyp_tiled.h does the rest with "brick" = cluster: packets wait in slot records, are sorted by cluster every generation,
// and one workgroup per task walks the packets of one cluster from LDS (ds_add_f64 deposits) until they leave the
// cluster, interact or die.
//
// The wall search is grid_geometry_voronoi.f90:322-402: the nearest bisector plane ahead, t = n.(m - r) / n.v over the
// neighbours, one IEEE division each -- ~45 FP64 instructions and 32 bytes of LDS per wall, ~16 walls per crossing, which
// is what bound round 3's kernel (VALU issue 62 % and the LDS pipe 75 % busy).  Round 4 keeps the reference's arithmetic
// for the wall that WINS and finds that wall in FP32:
//
//   * per wall one 16-byte record (n.x, n.y, n.z, |n|) x scale, n = neighbour's site - own site.  With q = r - site (formed
//     in FP64, then rounded) num = |n|^2 / 2 - n.q and den = n.v are seven FP32 operations; t32 = num / den with v_rcp_f32.
//   * every t32 carries a rigorous bound eps on its distance from the reference's FP64 quotient (vt_filter below: rounding
//     of the inputs, of the seven operations, of the reciprocal, and the reference's own rounding of m - r at the cluster's
//     coordinate magnitude), so each wall has an interval [lo, hi] that contains its true t.
//   * U = the smallest hi among walls that are certainly ahead (lo > 0).  A wall can be the reference's minimum only if it
//     may be ahead (hi > 0) and lo <= U.  The loop tracks the two smallest lo of such walls: when the second one is above
//     U the first is the ONLY candidate and the step evaluates the reference's expression for that wall alone -- the
//     same t, bit for bit, and the same wall.  Anything else -- two candidates (ties on lattices, a corner clipped within
//     1e-6 of a cell width), a wall nearly parallel to the flight, non-finite FP32 values at absurd scales, a cell whose
//     list names a neighbour twice -- runs the reference's loop over all walls in FP64 (vt_find_wall_exact).
//
// So results are those of the reference's loop whatever the filter does (tests/test_gpu_voronoi.py: tallies equal to the
// oracle's and to the persistent kernel's; -DHYP_VTILE_VERIFY runs both searches on every step and counts disagreements),
// and a crossing costs ~20 FP32 instructions and 16 bytes of LDS per wall plus ONE FP64 evaluation.  Sites are stored once
// per cell (own cells + the "ghost" cells across the cluster's boundary) instead of once per wall, which makes the tables of a
// cell ~400 bytes instead of ~580: clusters of ~380 cells on a whole CU's LDS (16-bit cell indices), longer visits.
#pragma once

#include "hyp_tiled.h"

constexpr int HYP_VTILE_WG = 1024;        // threads per workgroup (one workgroup per task, one per CU with vt_lds_kb = 156)
constexpr int HYP_VTILE_OCC = 4;          // waves per SIMD the register budget is set for (128 VGPRs, nothing spilled at one and two species)
#ifndef HYP_VTILE_SERVICE_N
#define HYP_VTILE_SERVICE_N 16
#endif
#ifndef HYP_VTILE_STEPS_N
#define HYP_VTILE_STEPS_N 2
#endif
constexpr int HYP_VTILE_SERVICE = HYP_VTILE_SERVICE_N;     // lanes that must wait before a wave runs its service phase
#ifndef HYP_VTILE_UNROLL
#define HYP_VTILE_UNROLL 4                // unrolling of the filter loop (1: 361.5, 2: 355.6, 3: 354.1, 4: 351.2 ms on one box; no spills at 128 VGPRs)
#endif
constexpr int HYP_VTILE_STEPS = HYP_VTILE_STEPS_N;        // cell steps between two scheduling decisions of a wave

// the cluster's tables in LDS (layout of the blob: hyp_device.h, VtInfo)
struct VtLds {
    const double *sx, *sy, *sz;
    const float4 *wrec;
    const uint32_t *wlink, *hdr;
    const float *lmax;                       // per own cell: its longest |n| x scale
    const int *members;                      // cell ids of the own cells
    const int *gcell, *gpacked, *gadj;       // per ghost: cell id, its vt_cluster word, slot of its cluster in the adjacency list
    int blob16;                              // size of all of it in units of 16 bytes
};

__device__ __forceinline__ void vt_lds_view(const VtInfo &I, const float4 *base, VtLds &L)
{
    const int ns = (I.n_site + 1) & ~1;
    L.sx = (const double *)base; L.sy = L.sx + ns; L.sz = L.sy + ns;
    L.wrec = (const float4 *)(L.sz + ns);
    L.wlink = (const uint32_t *)(L.wrec + I.n_wall);
    L.hdr = L.wlink + ((I.n_wall + 3) & ~3);
    const int ng = I.n_site - I.n_own, ngp = (ng + 3) & ~3;
    L.lmax = (const float *)(L.hdr + ((I.n_own + 3) & ~3));
    L.members = (const int *)(L.lmax + ((I.n_own + 3) & ~3));
    L.gcell = L.members + ((I.n_own + 3) & ~3); L.gpacked = L.gcell + ngp; L.gadj = L.gpacked + ngp;
    L.blob16 = (int)(((const char *)(L.gadj + ngp) - (const char *)base) >> 4);
}

// the reference's expression for ONE wall (grid_geometry_voronoi.f90:357-393); returns `ahead` for a face of the box
// (:362-371), true for a bisector plane (the caller excludes the wall the packet came through)
__device__ __forceinline__ bool vt_exact_t(const DProblem &P, const VtLds &L, uint32_t link, double s0, double s1, double s2,
                                           double r0, double r1, double r2, double v0, double v1, double v2, double &t)
{
    const int box = VT_LINK_BOX(link);
    if (box) {
        // (position and direction are scalars, selected by comparisons: an indexed array would live in scratch memory)
        const int iw = box - 1, up = iw & 1;
        const double va = iw < 2 ? v0 : iw < 4 ? v1 : v2;
        const double ra = iw < 2 ? r0 : iw < 4 ? r1 : r2;
        const double wall = iw < 2 ? (up ? P.vor_box[1] : P.vor_box[0]) : iw < 4 ? (up ? P.vor_box[3] : P.vor_box[2]) : (up ? P.vor_box[5] : P.vor_box[4]);
        t = (wall - ra) / va;
        return up ? (va > 0.0) : (va < 0.0);
    }
    const int j = VT_LINK_LOC(link);
    const double wx = L.sx[j], wy = L.sy[j], wz = L.sz[j];
    const double n0 = wx - s0, n1 = wy - s1, n2 = wz - s2;
    const double m0 = 0.5 * (wx + s0), m1 = 0.5 * (wy + s1), m2 = 0.5 * (wz + s2);
    t = (n0 * (m0 - r0) + n1 * (m1 - r1) + n2 * (m2 - r2)) / (n0 * v0 + n1 * v1 + n2 * v2);
    return true;
}

// grid_geometry_voronoi.f90:357-393 over the cluster's tables: the reference's loop, wall by wall in FP64
__device__ __forceinline__ bool vt_find_wall_exact(const DProblem &P, const VtLds &L, int k0, int nk, int prev_k, double s0, double s1, double s2,
                                                   double r0, double r1, double r2, double v0, double v1, double v2, double &tnear, int &kmin)
{
    // `ahead = neighbour /= previous cell`: compared through the neighbours' places in the site table (one per cell id)
    const int prev_site = prev_k < nk ? VT_LINK_LOC(L.wlink[k0 + prev_k]) : -1;      // (a face of the box, mark_face_behind: 0xffff, nobody's index)
    double tmin = HYP_DBL_MAX; int imin = -1;
    for (int k = 0; k < nk; k++) {
        const uint32_t link = L.wlink[k0 + k];
        double t;
        bool ahead = vt_exact_t(P, L, link, s0, s1, s2, r0, r1, r2, v0, v1, v2, t);
        if (!VT_LINK_BOX(link)) ahead = VT_LINK_LOC(link) != prev_site;
        if (ahead && t > 0.0 && t < tmin) { tmin = t; imin = k; }
    }
    tnear = tmin; kmin = imin;
    return imin >= 0;
}

// The FP32 filter.  Inputs: q = (r - site) x scale and v rounded to FP32, Q >= |q|, L >= the cell's longest |n| x scale.  With
// u = 2^-24 and a wall record (n, h) -- n within u of N x scale componentwise, h within u of |N|^2 / 2 x scale^2 -- the computed
//   num32 = h - n.q    obeys   |num32 - num| <= dn = L (12u L + 10u Q + abs_eps)   (three fma for n.q: 5.5u |n| Q; h: u |n|^2 / 2; the
//                                                                                  subtraction), every term bounded with |n| <= L
//   den32 = n.v        obeys   |den32 - den| <= dd = 8u L                          (three fma, |v| = 1: 5.5u |n|)
// where num / den is the real-number quotient and the reference's FP64 value lies within abs_eps |n| of num (its m - r is
// formed from absolute coordinates).  For |den32| > 4 dd:  |num32 / den32 - num / den| <= 2 (dn + |t| dd) / |den32|, and
// v_rcp_f32 (1 ulp) times one multiplication adds 4u |t|.  The constants below are those with a margin of 1.4 - 2.
//
// Round 6: the bounds take the cell's longest wall for every wall (they grow with |n|, so they hold; a short wall is "undecided" a
// little more often) -- dn and dd become constants of the step, the record carries |n|^2 / 2 instead of |n| -- and a
// classification that needs neither |t| nor a rule for the faces of the box:
//   P: den32 >  4 dd  (the wall is surely approached)     N: den32 < -4 dd  (surely receding)      M: num32 > dn  (the packet is
//   surely on the near side of the plane, i.e. inside as far as this wall is concerned)
//   P & M   a candidate: t > 0, its interval [t - eps, t + eps] holds the reference's t;
//   N & M   the reference's t is negative: never its minimum (a face of the box the packet moves away from is of this kind:
//           its den is the single product that decides the reference's `ahead`, grid_geometry_voronoi.f90:362-371);
//   else    undecided (within ~1e-6 of a plane other than the one it came through, a wall nearly parallel to the flight,
//           non-finite values): the step runs the reference's loop.
// With lo1 <= lo2 the two smallest lower ends among the candidates and hi1 = lo1 + 2 eps1 the upper end of the first: lo2 > hi1
// says every other candidate's t exceeds the first one's, which is then the reference's minimum -- the only wall whose t the
// reference would keep.  eps = 2.5 (t dd + dn) / den + 16u t is formed as t (2.5 dd rcp + 16u) + 2.5 dn rcp.  Which wall was
// the first is read off a history of the comparisons (one add-with-carry per wall: bit b = "wall nk - 1 - b became the first").
// 23 VALU instructions per wall (38 in round 5).  Measured and rejected (profiles/r06_tiled_log.md): the loop on pairwise-SoA
// records with v_pk_fma/mul/add_f32 -- 17.4 instead of 25.3 VALU wave-instructions per crossing, and 3 % SLOWER than the scalar
// form of the same classification (365 against 352 ms): a packed instruction is not cheaper here than the two it replaces.
// Returns the index of that wall, or -1 when there is more than one candidate or none.
#define VT_U 5.9604645e-8f
__device__ __forceinline__ int vt_filter(const VtLds &L, int k0, int nk, int prev_k, float q0, float q1, float q2, float Q, float abs_eps, float Lm,
                                         float v0, float v1, float v2)
{
    const float cq = fmaf(10.0f * VT_U, Q, abs_eps);
    const float dn4 = Lm * fmaf(Lm, 48.0f * VT_U, 4.0f * cq);      // 4 dn
    const float dd4 = (32.0f * VT_U) * Lm;                          // 4 dd
    const float ea = 0.625f * dd4, eb = 0.625f * dn4, dn1 = 0.25f * dn4;
    const float inf = __builtin_inff();
    float lo1 = inf, lo2 = inf, eps1 = 0.0f;
    uint32_t hist = 0;
    bool unc = false;
    // written without branches: a wave's lanes are in different cells, every `if` here would be a pair of exec-mask updates
#pragma unroll HYP_VTILE_UNROLL
    for (int k = 0; k < nk; k++) {
        const float4 w = L.wrec[k0 + k];
        const float dq = fmaf(w.z, q2, fmaf(w.y, q1, w.x * q0));
        const float den = fmaf(w.z, v2, fmaf(w.y, v1, w.x * v0));
        const float num = w.w - dq;
        const float rcp = __builtin_amdgcn_rcpf(den);
        const float t = num * rcp;
        const float eps = fmaf(t, fmaf(ea, rcp, 16.0f * VT_U), eb * rcp);
        const float lo = t - eps;
        const bool pos = den > dd4, neg = den < -dd4, inside = num > dn1;
        unc |= (k != prev_k) & !(inside & (pos | neg));
        // (the wall the packet came through is never `pos`: it moves away from it)
        const float c = (pos & inside) ? lo : inf;
        const bool first = c < lo1;
        lo2 = __builtin_amdgcn_fmed3f(lo1, lo2, c);         // the second smallest of three
        lo1 = first ? c : lo1;
        eps1 = first ? eps : eps1;
        hist = (hist + hist) + (first ? 1u : 0u);          // (one v_addc_co_u32: the comparison's mask is the carry)
    }
    const float hi1 = fmaf(2.000001f, eps1, lo1);
    return (!unc && hist != 0u && lo2 > hi1) ? nk - 1 - (int)__builtin_ctz(hist) : -1;
}


// the search of one step: filter, then the reference's expression for the wall it names, or the reference's loop
__device__ __forceinline__ bool vt_find_wall(const DProblem &P, const VtLds &L, const VtInfo &I, int loc, int k0, int nk, bool exact_only, int prev_k,
                                             double r0, double r1, double r2, double v0, double v1, double v2, double &tnear, int &kmin,
                                             unsigned int &n_exact, int k1_coop = -2, unsigned long long *dbg_t = nullptr)
{
#ifdef HYP_TILE_STATS
    const long long dbg_c0 = clock64();
#endif
    const double s0 = L.sx[loc], s1 = L.sy[loc], s2 = L.sz[loc];       // the cell's site (three ds_read_b64 per step: registers are scarcer)
    int k1 = -1;
    if (k1_coop != -2) k1 = exact_only ? -1 : k1_coop;
    else if (!exact_only) {
        const float q0 = (float)((r0 - s0) * (double)I.scale), q1 = (float)((r1 - s1) * (double)I.scale), q2 = (float)((r2 - s2) * (double)I.scale);
        const float Q = (fabsf(q0) + fabsf(q1) + fabsf(q2)) * 1.000001f;        // >= |q| (two additions instead of the expansion of sqrtf)
#ifdef HYP_TILE_STATS
        const long long dbg_c1 = clock64();
#endif
        k1 = vt_filter(L, k0, nk, prev_k, q0, q1, q2, Q, I.abs_eps, L.lmax[loc], (float)v0, (float)v1, (float)v2);
#ifdef HYP_TILE_STATS
        if (dbg_t) { dbg_t[0] += (unsigned long long)(dbg_c1 - dbg_c0); dbg_t[1] += (unsigned long long)(clock64() - dbg_c1); }
#endif
    }
    bool ok = false;
    if (k1 >= 0) {
        double t;
        const bool ahead = vt_exact_t(P, L, L.wlink[k0 + k1], s0, s1, s2, r0, r1, r2, v0, v1, v2, t);
        if (ahead && t > 0.0 && t < HYP_DBL_MAX) { tnear = t; kmin = k1; ok = true; }
    }
#ifdef HYP_VTILE_VERIFY     // both searches on every step: a disagreement is counted in TileCtl::dbg[39] (tests/test_gpu_voronoi.py)
    {
        double te; int ke;
        const bool fe = vt_find_wall_exact(P, L, k0, nk, prev_k, s0, s1, s2, r0, r1, r2, v0, v1, v2, te, ke);
        if (ok && (!fe || ke != kmin || te != tnear)) n_exact |= 0x80000000u;
        if (!ok) n_exact++;
        tnear = te; kmin = ke;
        return fe;
    }
#endif
    if (ok) return true;
    n_exact++;
    return vt_find_wall_exact(P, L, k0, nk, prev_k, s0, s1, s2, r0, r1, r2, v0, v1, v2, tnear, kmin);
}

// the propagation check's in_correct_cell (grid_geometry_voronoi.f90:274-283) walks the neighbour graph in global memory: a
// real call, once per ~1000 steps, so that its registers are not the walk's (inlined it cost the kernel 66 spilled VGPRs)
__device__ __attribute__((noinline)) bool vt_in_correct_cell(const DProblem *Pp, double r0, double r1, double r2, int cell)
{
    const Walls W = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {0, 0, 0}};
    const double r[3] = {r0, r1, r2};
    Cell<GEOM_VOR> c; c.id = cell; c.ow[0] = 0; c.ow[1] = 0; c.ow[2] = 0;      // (only the cell is looked at)
    return geo_in_correct_cell(*Pp, W, r, c);
}

// TileGeom for this schedule: n_bricks = number of clusters; the LDS of the launch holds the largest cluster
template <int ND>
__global__ __launch_bounds__(HYP_VTILE_WG, HYP_VTILE_OCC) void vtile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                                  void *__restrict__ hot_v, void *__restrict__ cold_v,
                                                                  const int *__restrict__ order,
                                                                  const TileTask *__restrict__ tasks, int *__restrict__ slot_brick,
                                                                  int *__restrict__ ilist, int *__restrict__ dlist,
                                                                  TileCount *__restrict__ tcount, unsigned int *__restrict__ counts)
{
    extern __shared__ float4 lds16[];        // 16-byte aligned base: the wall records are read with ds_read_b128
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    const DProblem &P = *Pp;
    if (blockIdx.x >= ctl->n_tasks[T.pool]) return;
    const TileTask tk = tasks[blockIdx.x];
    const int cl = tk.brick;
    const VtInfo I = P.vt_info[cl];
    const int nc = I.n_own;
    VtLds L;
    vt_lds_view(I, lds16, L);
    const int blob16 = L.blob16;
    double *dens = (double *)(lds16 + blob16);
    double *accum = dens + (size_t)nc * ND;
    __shared__ int next_pkt, n_int_l, n_dead_l, pub_base[2];
    __shared__ unsigned int nb_cnt[VT_MAX_ADJ + 1];      // packets that move on to each adjacent cluster; [VT_MAX_ADJ]: parked here
    __shared__ int adj[VT_MAX_ADJ];
    __shared__ double red[TILE_RED_N];
    {
        const float4 *src = P.vt_blob + I.blob16;
        for (int i = threadIdx.x; i < blob16; i += blockDim.x) lds16[i] = src[i];
    }
    for (int i = threadIdx.x; i < nc * ND; i += blockDim.x) {
        const int cell = P.vt_members[I.cell0 + i / ND];       // (the LDS copy is still on its way)
        dens[i] = P.density[(size_t)cell * ND + i % ND];
        accum[i] = 0.0;
    }
    if (threadIdx.x < VT_MAX_ADJ) adj[threadIdx.x] = P.vt_adj[(size_t)cl * VT_MAX_ADJ + threadIdx.x];
    if (threadIdx.x <= VT_MAX_ADJ) nb_cnt[threadIdx.x] = 0;
    if (threadIdx.x >= 128 && threadIdx.x < 128 + TILE_RED_N) red[threadIdx.x - 128] = 0.0;
    if (threadIdx.x == 192) { next_pkt = 0; n_int_l = 0; n_dead_l = 0; }
    __syncthreads();

    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0, n_exact = 0;
    // lane state: the walking part of a packet (the rest stays in its ColdRec)
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, v0 = 1.0, v1 = 0.0, v2 = 0.0, tau_req = 0.0, tau_ach = 0.0, energy = 0.0, chi[ND], kappa[ND];
    double t_src = HYP_INF, t_ach = 0.0;     // re-absorption by sources (P.any_intersect): see Packet
    int hk0 = 0, hnk = 0;                    // its wall records [hk0, hk0 + hnk)
    bool hexact = false;                     // VT_HDR_EXACT
    int loc = 0;                             // the current cell's index in the cluster
    int prev_k = VT_NO_BACK;                 // position, in the current cell's list, of the wall the packet came through
    int prev_loc = -2;                       // the cell it came from, if that happened in this visit (else prev_g, the record's ic[1])
    int prev_g = -1;
    int left_ghost = 0;
    Rng g; g.key0 = P.seed_key; g.key1 = T.iter_tag; g.blk_a = 0; g.have_a = 0; g.buf_a = 0.0;
    g.id_lo = g.id_hi = 0; g.blk_b = 0; g.countdown = 0;
    int slot = -1;
    int st = LS_IDLE;
    bool exhausted = false;
#pragma unroll
    for (int d = 0; d < ND; d++) { chi[d] = 0.0; kappa[d] = 0.0; }

#ifdef HYP_TILE_STATS
    unsigned long long dbg_outer = 0, dbg_wsteps = 0, dbg_lsteps = 0, dbg_service = 0, dbg_nservice = 0, dbg_visits = 0, dbg_wb = 0, dbg_claim = 0, dbg_nclaim = 0, dbg_nwb = 0, dbg_find = 0, dbg_nfind = 0, dbg_walls = 0, dbg_sect[2] = {0, 0};
    const long long dbg_t0 = clock64();
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
        if (park || ((m_out | m_idle) && (__popcll(m_out | m_idle) >= HYP_VTILE_SERVICE || !m_walk))) {
#ifdef HYP_TILE_STATS
            const long long dbg_ts = clock64();
#endif
            // propagation check (grid_propagate_3d.f90:112-120), then the step goes on as usual
            // (a lane whose check is due waits until four are, or nobody walks any more: tile_walk_kernel, hyp_tiled.h)
            if (st == LS_CHECK && (__popcll(__ballot(st == LS_CHECK)) >= 4 || !m_walk || park)) {
                const int gap = rng_check_gap(g, P.check_p, P.check_log1mp);
                g.countdown = gap < 0x7fffffff ? gap + 1 : gap;      // the step below takes one off again
                // in_correct_cell (:274-283) first asks whether the nearest-site walk from the packet's cell stays there, i.e. whether
                // no neighbour's site is closer than the cell's own: answered from the cluster's tables with vor_dist2's operations;
                // only a packet that fails this goes through the reference's whole test on the global tables
                bool stays = true;
                {
                    const double ox = L.sx[loc] - r0, oy = L.sy[loc] - r1, oz = L.sz[loc] - r2;
                    const double d_own = ox * ox + oy * oy + oz * oz;
                    for (int k = 0; k < hnk; k++) {
                        const uint32_t link = L.wlink[hk0 + k];
                        if (VT_LINK_BOX(link)) continue;
                        const int j2 = VT_LINK_LOC(link);
                        const double dx = L.sx[j2] - r0, dy = L.sy[j2] - r1, dz = L.sz[j2] - r2;
                        if (dx * dx + dy * dy + dz * dz < d_own) stays = false;
                    }
                }
                if (stays || vt_in_correct_cell(Pp, r0, r1, r2, L.members[loc])) st = LS_WALK;
                else { cnt.killed_geo++; st = LS_DEAD; }
            }
            if (st == LS_DEAD) {
                hot[slot].state = TS_DEAD; slot_brick[slot] = TILE_NEEDS_PREPARE;
                dlist[tk.start + atomicAdd(&n_dead_l, 1)] = slot;
                finished++; st = LS_IDLE;
            } else if (st == LS_LEFT || st == LS_HIT || st == LS_REABS || (park && st == LS_WALK)) {
                HotRec<ND> &H = hot[slot];
                if (P.any_intersect) cold[slot].t_ach = t_ach;
                const int here = L.members[loc];
                // the 64-byte line a visit changes goes out as four 16-byte stores (r | r, tau_ach | ic, ow | countdown, blk_b,
                // state, pad): a store instruction per field was a third of the service phase
                int c0, c1, c2, c3, state = TS_WALK;
                if (st == LS_LEFT) {
                    const int packed = L.gpacked[left_ghost], ga = L.gadj[left_ghost];
                    c0 = L.gcell[left_ghost]; c1 = here; c2 = packed; c3 = prev_k;
                    slot_brick[slot] = packed >> 16;
                    if (ga < VT_MAX_ADJ) atomicAdd(&nb_cnt[ga], 1u);
                    else atomicAdd(&counts[packed >> 16], 1u);
                } else {
                    c0 = here; c2 = (cl << 16) | loc;
                    if (st == LS_HIT) { c1 = -1; c3 = VT_NO_BACK; }                    // geo_clear_wall
                    else { c1 = prev_loc >= 0 ? L.members[prev_loc] : prev_g; c3 = prev_k; }
                    if (st == LS_REABS) { state = TS_REEMIT; slot_brick[slot] = TILE_NEEDS_REEMIT; }
                    else if (st == LS_HIT) { state = TS_INTERACT; slot_brick[slot] = TILE_NEEDS_INTERACT; }
                    else atomicAdd(&nb_cnt[VT_MAX_ADJ], 1u);                          // parked: same cluster again
                    if (st == LS_REABS || st == LS_HIT) ilist[tk.start + atomicAdd(&n_int_l, 1)] = slot;
                }
                static_assert(offsetof(HotRec<ND>, tau_ach) == 24 && offsetof(HotRec<ND>, ic) == 32 && offsetof(HotRec<ND>, ow) == 44 &&
                              offsetof(HotRec<ND>, countdown) == 48 && offsetof(HotRec<ND>, blk_b) == 52 && offsetof(HotRec<ND>, state) == 56, "HotRec layout");
                double2 *line = (double2 *)&H;
                line[0] = make_double2(r0, r1);
                line[1] = make_double2(r2, tau_ach);
                ((int4 *)line)[2] = make_int4(c0, c1, c2, c3);
                ((int4 *)line)[3] = make_int4(g.countdown, (int)g.blk_b, state, 0);
                st = LS_IDLE;
            }
#ifdef HYP_TILE_STATS
            const long long dbg_tw = clock64();
            dbg_wb += (unsigned long long)(dbg_tw - dbg_ts); dbg_nwb += __popcll(m_out);
            dbg_nclaim += __popcll(__ballot(st == LS_IDLE && !exhausted));
#endif
            if (park) break;
            if (st == LS_IDLE && !exhausted) {
                const int j = atomicAdd(&next_pkt, 1);
                if (j >= tk.len) exhausted = true;
                else {
                    slot = order[tk.start + j];
                    const HotRec<ND> &H = hot[slot];
                    r0 = H.r[0]; r1 = H.r[1]; r2 = H.r[2]; v0 = H.v[0]; v1 = H.v[1]; v2 = H.v[2];
                    loc = H.ic[2] & 0xffff; prev_k = H.ow & 255; prev_loc = -2; prev_g = H.ic[1];
                    tau_req = H.tau_req; tau_ach = H.tau_ach; energy = H.energy;
#pragma unroll
                    for (int d = 0; d < ND; d++) { chi[d] = H.chi[d]; kappa[d] = H.kappa[d]; }
                    const unsigned long long id = H.id;
                    g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
                    g.countdown = H.countdown; g.blk_b = H.blk_b;
                    if (P.any_intersect) { t_src = cold[slot].t_src; t_ach = cold[slot].t_ach; }
                    const uint32_t hh = L.hdr[loc];
                    hk0 = (int)(hh & 0xfffffu); hnk = (int)((hh >> 20) & 0xffu); hexact = (hh & VT_HDR_EXACT) != 0;
                    if (prev_k == VT_FIND_BACK) {        // a record written without the wall's position: look the previous cell up
                        const int want = prev_g;
                        prev_k = VT_NO_BACK;
                        for (int k = 0; k < hnk; k++) {
                            const uint32_t link = L.wlink[hk0 + k];
                            if (VT_LINK_BOX(link)) continue;
                            const int j2 = VT_LINK_LOC(link);
                            const int idn = j2 < nc ? L.members[j2] : L.gcell[j2 - nc];
                            if (idn == want && prev_k == VT_NO_BACK) prev_k = k;
                        }
                    }
                    st = LS_WALK;
                }
            }
            if (__ballot(exhausted)) queue_empty = true;
#ifdef HYP_TILE_STATS
            dbg_service += (unsigned long long)(clock64() - dbg_ts); dbg_nservice++; dbg_claim += (unsigned long long)(clock64() - dbg_tw);
            dbg_visits += __popcll(__ballot(st == LS_WALK)) ;
#endif
        }
#ifdef HYP_TILE_STATS
        dbg_outer++;
        { unsigned long long mw = __ballot(st == LS_WALK); if (mw) { dbg_wsteps++; dbg_lsteps += __popcll(mw); } }
#endif
        // ---- a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232) ----
#pragma unroll 1
        for (int q = 0; q < HYP_VTILE_STEPS; q++) {
            const int k1_coop = -2;
            if (st == LS_WALK) {
                if (g.countdown == 0) st = LS_CHECK;
                else {
                    g.countdown--;
                    double tmin; int kmin;
#ifdef HYP_TILE_STATS
                    const long long dbg_tf = clock64();
                    const bool fw_ok = vt_find_wall(P, L, I, loc, hk0, hnk, hexact, prev_k, r0, r1, r2, v0, v1, v2, tmin, kmin, n_exact, k1_coop, dbg_sect);
                    dbg_find += (unsigned long long)(clock64() - dbg_tf); dbg_nfind++; dbg_walls += (unsigned long long)hnk;
                    if (!fw_ok) { cnt.killed_geo++; st = LS_DEAD; }
#else
                    if (!vt_find_wall(P, L, I, loc, hk0, hnk, hexact, prev_k, r0, r1, r2, v0, v1, v2, tmin, kmin, n_exact, k1_coop)) { cnt.killed_geo++; st = LS_DEAD; }
#endif
                    else {
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
                                r0 = r0 + tmin * v0; r1 = r1 + tmin * v1; r2 = r2 + tmin * v2;
                                tau_ach += tau_cell;
#pragma unroll
                                for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tmin * kappa[d] * energy);
                                const uint32_t link = L.wlink[hk0 + kmin];
                                if (VT_LINK_BOX(link)) st = LS_DEAD;      // left the grid: the packet ends here
                                else {
                                    const int nloc = VT_LINK_LOC(link);
                                    prev_k = VT_LINK_BACK(link);
                                    if (nloc < nc) {
                                        prev_loc = loc; loc = nloc;
                                        const uint32_t hh = L.hdr[loc];
                                        hk0 = (int)(hh & 0xfffffu); hnk = (int)((hh >> 20) & 0xffu); hexact = (hh & VT_HDR_EXACT) != 0;
                                                        } else { left_ghost = nloc - nc; st = LS_LEFT; }
                                }
                            }
                        } else {
                            // the interaction happens inside this cell: grid_propagate_3d.f90:170-200
                            const double tact = tmin * (tau_needed / tau_cell);
                            bool reabs = false;
                            if (P.any_intersect) { t_ach += tact; reabs = t_ach > t_src; }      // :184-188
                            if (reabs) st = LS_REABS;
                            else {
                                r0 = r0 + tact * v0; r1 = r1 + tact * v1; r2 = r2 + tact * v2;
                                tau_ach += tau_needed;
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
#ifdef HYP_TILE_STATS
    if (__lane_id() == 0) {
        atomicAdd(&ctl->dbg[0], dbg_outer); atomicAdd(&ctl->dbg[1], dbg_wsteps); atomicAdd(&ctl->dbg[2], dbg_lsteps);
        atomicAdd(&ctl->dbg[3], 1ull); atomicAdd(&ctl->dbg[10], dbg_wb); atomicAdd(&ctl->dbg[11], dbg_claim); atomicAdd(&ctl->dbg[12], dbg_nwb); atomicAdd(&ctl->dbg[13], dbg_nclaim);
        atomicAdd(&ctl->dbg[6], dbg_service); atomicAdd(&ctl->dbg[7], dbg_nservice); atomicAdd(&ctl->dbg[8], (unsigned long long)(clock64() - dbg_t0));
        if (threadIdx.x == 0) { atomicAdd(&ctl->dbg[4], 1ull); atomicAdd(&ctl->dbg[5], (unsigned long long)tk.len); }
    }
    {   // the wall search: clocks of the lane that searched most often, its searches, the walls of all lanes' searches
        const double wsum = wave_sum((double)dbg_walls), nsum = wave_sum((double)dbg_nfind);
        unsigned long long fmax = dbg_find, nmax = dbg_nfind;
        for (int o = 32; o; o >>= 1) { const unsigned long long f2 = __shfl_xor(fmax, o, 64), n2 = __shfl_xor(nmax, o, 64); if (n2 > nmax) { nmax = n2; fmax = f2; } }
        unsigned long long s0m = dbg_sect[0], s1m = dbg_sect[1], n3 = dbg_nfind;
        for (int o = 32; o; o >>= 1) { const unsigned long long a2 = __shfl_xor(s0m, o, 64), b2 = __shfl_xor(s1m, o, 64), n2 = __shfl_xor(n3, o, 64); if (n2 > n3) { n3 = n2; s0m = a2; s1m = b2; } }
        if (__lane_id() == 0) { atomicAdd(&ctl->dbg[16], fmax); atomicAdd(&ctl->dbg[17], nmax); atomicAdd(&ctl->dbg[18], (unsigned long long)wsum); atomicAdd(&ctl->dbg[19], (unsigned long long)nsum);
                                atomicAdd(&ctl->dbg[20], s0m); atomicAdd(&ctl->dbg[21], s1m); }
    }
    const long long dbg_te = clock64();
#endif
    __syncthreads();
#ifdef HYP_TILE_STATS
    if (__lane_id() == 0) atomicAdd(&ctl->dbg[9], (unsigned long long)(clock64() - dbg_te));      // waiting for the workgroup's last wave
#endif
    if (threadIdx.x < VT_MAX_ADJ && nb_cnt[threadIdx.x]) atomicAdd(&counts[adj[threadIdx.x]], nb_cnt[threadIdx.x]);
    if (threadIdx.x == VT_MAX_ADJ && nb_cnt[VT_MAX_ADJ]) atomicAdd(&counts[cl], nb_cnt[VT_MAX_ADJ]);
    tile_walk_publish_lists(T, ctl, tk, ilist, dlist, n_int_l, n_dead_l, pub_base);
    // flush the cluster's accumulators (replica chosen like in the persistent kernel)
    double *sum = P.sum;
    if (P.n_copies > 1) {
        unsigned c = xcc_id();
        if (P.n_copies > 8) c += 8u * ((blockIdx.x >> 3) % (unsigned)(P.n_copies >> 3));
        sum += (size_t)(c % (unsigned)P.n_copies) * P.copy_stride;
    }
    for (int i = threadIdx.x; i < nc * ND; i += blockDim.x) {
        const double val = accum[i];
        if (val != 0.0) hyp_atomic_add_g(&sum[(size_t)L.members[i / ND] * ND + i % ND], val);
    }
    // how often the search ran the reference's loop (option last_vt_exact_steps); bit 31: the two searches disagreed (-DHYP_VTILE_VERIFY)
    {
        const unsigned long long bad = __ballot((n_exact & 0x80000000u) != 0);
        const double ne = wave_sum((double)(n_exact & 0x7fffffffu));
        if (__lane_id() == 0) {
            if (ne != 0.0) atomicAdd(&ctl->dbg[38], (unsigned long long)ne);
            if (bad) atomicAdd(&ctl->dbg[39], (unsigned long long)__popcll(bad));
        }
    }
    block_tally_flush(P, ctl, red, cnt, finished);
}
