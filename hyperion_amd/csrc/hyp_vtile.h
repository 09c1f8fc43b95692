// hyp_vtile.h -- cluster-tiled Lucy iteration for Voronoi grids (gfx950).
//
// The persistent kernel reads, per cell crossing, one 32-byte wall record per neighbour (~15) from wherever it lives in
// the memory hierarchy (49 MB at 100 000 sites: L2 misses, ~500 B of L2<->fabric traffic per crossing) and makes one
// memory-side atomic per species.  Here the cells are grouped at set-up into spatially compact CLUSTERS (recursive
// coordinate bisection of the sites, ~100 cells each) whose wall records, sites, densities and accumulators fit in LDS.
// The slot-pool schedule of hyp_tiled.h does the rest with "brick" = cluster: packets wait in slot records, are sorted by
// cluster every generation, and one workgroup per task walks the packets of one cluster from LDS (ds_read_b128 for the
// records, ds_add_f64 for the deposits) until they leave the cluster, interact or die.
//
// The wall search is grid_geometry_voronoi.f90:322-402 (geo_find_wall<GEOM_VOR>): the nearest bisector plane ahead,
// t = n.(m - r) / n.v per neighbour, one IEEE division per neighbour like the reference.  A division-free variant is kept
// behind -DHYP_VTILE_CROSSMUL: numerator and denominator of every candidate formed with the reference's operations, the
// minimum found by cross-multiplication with a guard band of 2^-48, ONE quotient (the winner's) per step, and the
// reference's loop for any step with a second candidate inside the guard band (bit-identical results, tested).  It is
// SLOWER (1e8 packets, one species: 627 ms against 564 ms): the kernel waits on LDS (56 % of its wave cycles are
// s_waitcnt, two thirds of the LDS cycles are bank conflicts of the scattered 32-byte record reads), not on VALU issue,
// and the extra state of the cross-multiplication costs 16 spilled VGPRs at 4 waves per SIMD.
#pragma once

#include "hyp_tiled.h"

#ifndef HYP_VTILE_WG
#define HYP_VTILE_WG 512         // threads per workgroup (one workgroup per task)
#endif
#ifndef HYP_VTILE_OCC
#define HYP_VTILE_OCC 4         // waves per SIMD the register budget is set for (two 512-thread workgroups per CU)
#endif
#ifndef HYP_VTILE_SERVICE
#define HYP_VTILE_SERVICE 16     // lanes that must wait before a wave runs its service phase
#endif
#ifndef HYP_VTILE_UNROLL
#define HYP_VTILE_UNROLL 1       // unrolling of the wall loop (more LDS reads in flight per lane, more registers)
#endif
#ifndef HYP_VTILE_STEPS
#define HYP_VTILE_STEPS 1        // cell steps between two scheduling decisions of a wave (measured: 1 beats 2 and 4)
#endif

// grid_geometry_voronoi.f90:357-393 over the cluster's copy of the wall records; kmin = index of the nearest wall ahead
__device__ __forceinline__ bool vt_find_wall_exact(const DProblem &P, const VorWall *walls, int k0, int k1, double s0, double s1, double s2,
                                                   const double r[3], const double v[3], int prev, double &tnear, int &kmin)
{
    double tmin = HYP_DBL_MAX; int imin = -1;
#pragma unroll HYP_VTILE_UNROLL
    for (int k = k0; k < k1; k++) {
        const VorWall w = walls[k];
        double t; bool ahead;
        if (w.nb < 0) {
            const int iw = -w.nb - 1, a = iw >> 1, up = iw & 1;
            const double va = a == 0 ? v[0] : a == 1 ? v[1] : v[2];
            const double ra = a == 0 ? r[0] : a == 1 ? r[1] : r[2];
            ahead = up ? (va > 0.0) : (va < 0.0);
            t = (P.vor_box[iw] - ra) / va;
        } else {
            const double n0 = w.x - s0, n1 = w.y - s1, n2 = w.z - s2;
            const double m0 = 0.5 * (w.x + s0), m1 = 0.5 * (w.y + s1), m2 = 0.5 * (w.z + s2);
            t = (n0 * (m0 - r[0]) + n1 * (m1 - r[1]) + n2 * (m2 - r[2])) / (n0 * v[0] + n1 * v[1] + n2 * v[2]);
            ahead = w.nb != prev;
        }
        if (ahead && t > 0.0 && t < tmin) { tmin = t; imin = k; }
    }
    tnear = tmin; kmin = imin;
    return imin >= 0;
}

__device__ __forceinline__ bool vt_find_wall(const DProblem &P, const VorWall *walls, int k0, int k1, double s0, double s1, double s2,
                                             const double r[3], const double v[3], int prev, double &tnear, int &kmin)
{
#ifndef HYP_VTILE_CROSSMUL       // default: the reference's loop, one division per wall (measured faster, see the header comment)
    return vt_find_wall_exact(P, walls, k0, k1, s0, s1, s2, r, v, prev, tnear, kmin);
#else
    double bn = 0.0, bd = 1.0; int bk = -1; bool amb = false;
    for (int k = k0; k < k1; k++) {
        const VorWall w = walls[k];
        double num, den; bool ahead;
        if (w.nb < 0) {
            const int iw = -w.nb - 1, a = iw >> 1, up = iw & 1;
            const double va = a == 0 ? v[0] : a == 1 ? v[1] : v[2];
            const double ra = a == 0 ? r[0] : a == 1 ? r[1] : r[2];
            ahead = up ? (va > 0.0) : (va < 0.0);
            num = P.vor_box[iw] - ra; den = va;
        } else {
            const double n0 = w.x - s0, n1 = w.y - s1, n2 = w.z - s2;
            const double m0 = 0.5 * (w.x + s0), m1 = 0.5 * (w.y + s1), m2 = 0.5 * (w.z + s2);
            num = n0 * (m0 - r[0]) + n1 * (m1 - r[1]) + n2 * (m2 - r[2]);
            den = n0 * v[0] + n1 * v[1] + n2 * v[2];
            ahead = w.nb != prev;
        }
        // the quotient is positive exactly when both have the same sign (and neither is zero or NaN)
        const bool pos = (num > 0.0 && den > 0.0) || (num < 0.0 && den < 0.0);
        if (ahead && pos) {
            const double an = fabs(num), ad = fabs(den);
            if (bk < 0) { bn = an; bd = ad; bk = k; }
            else {
                // an / ad < bn / bd  <=>  an bd < bn ad; within 2^-48 of equality (or not finite) the reference's loop decides
                const double lhs = an * bd, rhs = bn * ad;
                if (lhs < rhs * (1.0 - 0x1p-48)) { bn = an; bd = ad; bk = k; }
                else if (!(lhs > rhs * (1.0 + 0x1p-48))) amb = true;
            }
        }
    }
    if (bk < 0) { tnear = HYP_DBL_MAX; kmin = -1; return false; }
    const double t = bn / bd;       // |num| / |den| rounds like num / den
    if (amb || !(t > 0.0 && t < HYP_DBL_MAX)) return vt_find_wall_exact(P, walls, k0, k1, s0, s1, s2, r, v, prev, tnear, kmin);
    tnear = t; kmin = bk;
    return true;
#endif
}

// TileGeom for this schedule: n_bricks = number of clusters; bx = most cells, by = most wall records of a cluster (LDS
// layout); the other brick fields are unused.
template <int ND>
__global__ __launch_bounds__(HYP_VTILE_WG, HYP_VTILE_OCC) void vtile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                                  void *__restrict__ hot_v, void *__restrict__ cold_v,
                                                                  const int *__restrict__ order,
                                                                  const TileTask *__restrict__ tasks, int *__restrict__ slot_brick,
                                                                  int *__restrict__ ilist, int *__restrict__ dlist,
                                                                  TileCount *__restrict__ tcount, unsigned int *__restrict__ counts)
{
    extern __shared__ float4 lds16[];        // 16-byte aligned base: headers and wall records are read with ds_read_b128
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    const DProblem &P = *Pp;
    if (blockIdx.x >= ctl->n_tasks[T.pool]) return;
    const TileTask tk = tasks[blockIdx.x];
    const int cl = tk.brick;
    const int c0 = P.vt_cell_off[cl], nc = P.vt_cell_off[cl + 1] - c0;
    const int w0 = P.vt_wall_off[cl], nw = P.vt_wall_off[cl + 1] - w0;
    VtHdr *hdr = (VtHdr *)lds16;
    VorWall *walls = (VorWall *)(hdr + T.bx);
    double *dens = (double *)(walls + T.by);
    double *accum = dens + (size_t)T.bx * ND;
    __shared__ int next_pkt, n_int_l, n_dead_l, pub_base[2];
    __shared__ unsigned int nb_cnt[VT_MAX_ADJ + 1];      // packets that move on to each adjacent cluster; [VT_MAX_ADJ]: parked here
    __shared__ int adj[VT_MAX_ADJ];
    __shared__ double red[TILE_RED_N];
    {
        const float4 *src = (const float4 *)(P.vt_hdr + c0);
        float4 *dst = (float4 *)hdr;
        for (int i = threadIdx.x; i < nc * 2; i += blockDim.x) dst[i] = src[i];
        src = (const float4 *)(P.vt_walls + w0); dst = (float4 *)walls;
        for (int i = threadIdx.x; i < nw * 2; i += blockDim.x) dst[i] = src[i];
    }
    for (int i = threadIdx.x; i < nc * ND; i += blockDim.x) {
        const int cell = P.vt_members[c0 + i / ND];
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
    unsigned int finished = 0;
    // lane state: the walking part of a packet (the rest stays in its ColdRec)
    double r[3] = {0.0, 0.0, 0.0}, v[3] = {1.0, 0.0, 0.0}, tau_req = 0.0, tau_ach = 0.0, energy = 0.0, chi[ND], kappa[ND];
    double t_src = HYP_INF, t_ach = 0.0;     // re-absorption by sources (P.any_intersect): see Packet
    double hs0 = 0.0, hs1 = 0.0, hs2 = 0.0;  // site of the current cell and its wall records [hk0, hk1)
    int hk0 = 0, hk1 = 0;
    int cell = 0, prev = -1, loc = 0, left_adj = 0;
    Rng g; g.key0 = P.seed_key; g.key1 = T.iter_tag; g.blk_a = 0; g.have_a = 0; g.buf_a = 0.0;
    g.id_lo = g.id_hi = 0; g.blk_b = 0; g.countdown = 0;
    int slot = -1;
    int st = LS_IDLE;
    bool exhausted = false;
#pragma unroll
    for (int d = 0; d < ND; d++) { chi[d] = 0.0; kappa[d] = 0.0; }
    const Walls W = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {0, 0, 0}};

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
            // propagation check (grid_propagate_3d.f90:112-120), then the step goes on as usual
            if (st == LS_CHECK) {
                const int gap = rng_check_gap(g, P.check_p, P.check_log1mp);
                g.countdown = gap < 0x7fffffff ? gap + 1 : gap;      // the step below takes one off again
                Cell<GEOM_VOR> c; c.id = cell; c.ow[0] = 0; c.ow[1] = -(prev + 1); c.ow[2] = 0;
                if (geo_in_correct_cell(P, W, r, c)) st = LS_WALK;
                else { cnt.killed_geo++; st = LS_DEAD; }
            }
            if (st == LS_DEAD) {
                hot[slot].state = TS_DEAD; slot_brick[slot] = TILE_NEEDS_PREPARE;
                dlist[tk.start + atomicAdd(&n_dead_l, 1)] = slot;
                finished++; st = LS_IDLE;
            } else if (st == LS_LEFT || st == LS_HIT || st == LS_REABS || (park && st == LS_WALK)) {
                HotRec<ND> &H = hot[slot];
#pragma unroll
                for (int a = 0; a < 3; a++) H.r[a] = r[a];
                H.ic[0] = cell; H.ic[1] = prev;
                H.tau_ach = tau_ach; H.countdown = g.countdown; H.blk_b = g.blk_b;
                if (P.any_intersect) cold[slot].t_ach = t_ach;
                if (st == LS_LEFT) {                                                  // H.state stays TS_WALK
                    const int packed = P.vt_cluster[cell];
                    H.ic[2] = packed;
                    slot_brick[slot] = packed >> 8;
                    if (left_adj < VT_MAX_ADJ) atomicAdd(&nb_cnt[left_adj], 1u);
                    else atomicAdd(&counts[packed >> 8], 1u);
                } else {
                    H.ic[2] = (cl << 8) | loc;
                    if (st == LS_REABS) { H.state = TS_REEMIT; slot_brick[slot] = TILE_NEEDS_REEMIT; }
                    else if (st == LS_HIT) { H.state = TS_INTERACT; slot_brick[slot] = TILE_NEEDS_INTERACT; }
                    else atomicAdd(&nb_cnt[VT_MAX_ADJ], 1u);                          // parked: same cluster again
                    if (st == LS_REABS || st == LS_HIT) ilist[tk.start + atomicAdd(&n_int_l, 1)] = slot;
                }
                st = LS_IDLE;
            }
            if (park) break;
            if (st == LS_IDLE && !exhausted) {
                const int j = atomicAdd(&next_pkt, 1);
                if (j >= tk.len) exhausted = true;
                else {
                    slot = order[tk.start + j];
                    const HotRec<ND> &H = hot[slot];
#pragma unroll
                    for (int a = 0; a < 3; a++) { r[a] = H.r[a]; v[a] = H.v[a]; }
                    cell = H.ic[0]; prev = H.ic[1]; loc = H.ic[2] & 255;
                    tau_req = H.tau_req; tau_ach = H.tau_ach; energy = H.energy;
#pragma unroll
                    for (int d = 0; d < ND; d++) { chi[d] = H.chi[d]; kappa[d] = H.kappa[d]; }
                    const unsigned long long id = H.id;
                    g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
                    g.countdown = H.countdown; g.blk_b = H.blk_b;
                    if (P.any_intersect) { t_src = cold[slot].t_src; t_ach = cold[slot].t_ach; }
                    const VtHdr hh = hdr[loc];
                    hs0 = hh.x; hs1 = hh.y; hs2 = hh.z; hk0 = hh.k0; hk1 = hh.k1;
                    st = LS_WALK;
                }
            }
            if (__ballot(exhausted)) queue_empty = true;
        }
        // ---- a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232) ----
#pragma unroll 1
        for (int q = 0; q < HYP_VTILE_STEPS; q++) {
            if (st == LS_WALK) {
                if (g.countdown == 0) st = LS_CHECK;
                else {
                    g.countdown--;
                    double tmin; int kmin;
                    if (!vt_find_wall(P, walls, hk0, hk1, hs0, hs1, hs2, r, v, prev, tmin, kmin)) { cnt.killed_geo++; st = LS_DEAD; }
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
#pragma unroll
                                for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
                                tau_ach += tau_cell;
#pragma unroll
                                for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tmin * kappa[d] * energy);
                                const int nb = walls[kmin].nb, nloc = walls[kmin].loc;
                                prev = cell;
                                if (nb < 0) { cell = (int)P.n_cells; st = LS_DEAD; }      // left the grid: the packet ends here
                                else {
                                    cell = nb;
                                    if (nloc >= 0) {
                                        loc = nloc;
                                        const VtHdr hh = hdr[loc];
                                        hs0 = hh.x; hs1 = hh.y; hs2 = hh.z; hk0 = hh.k0; hk1 = hh.k1;
                                    } else { left_adj = -nloc - 2; st = LS_LEFT; }
                                }
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
                                prev = -1;       // geo_clear_wall
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
    __syncthreads();
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
        if (val != 0.0) hyp_atomic_add_g(&sum[(size_t)P.vt_members[c0 + i / ND] * ND + i % ND], val);
    }
    block_tally_flush(P, ctl, red, cnt, finished);
}
