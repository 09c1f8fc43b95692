// hyp_otile.h -- cluster-tiled Lucy iteration for octree grids (gfx950).
//
// The persistent kernel makes one memory-side FP64 atomic per cell crossing (configs[3]: 73 % of the chip's scattered-atomic
// ceiling, profiles/r02_extra_summary.md) and reads cell records and the neighbour table through L2.  Here the cells are
// grouped at set-up into CLUSTERS: runs of sibling subtrees, i.e. contiguous ranges of the depth-first cell ids of
// grid_geometry_octree.f90:206-246, small enough that the cell records, the children of the refined cells, the neighbour
// table, the densities and the accumulators of a cluster fit the LDS share of one workgroup.  The slot-pool schedule of
// hyp_tiled.h does the rest with "brick" = cluster: packets wait in slot records, are sorted by cluster every generation, and
// one workgroup per task walks the packets of one cluster from LDS (ds_read for records / neighbours / density, ds_add_f64 for
// the deposits, one flush of the non-zero accumulators per task) until they leave the cluster, interact or die.
//
// The walk is grid_geometry_octree.f90:438-537 (find_wall: nearest of the three faces ahead) and :328-347 (next_cell) through
// the neighbour table of geo_advance<GEOM_OCT> (hyp_kernels.h), bit for bit: the three quotients (wall - r) / v are formed
// from one reciprocal per visit and corrected to the IEEE quotient (Markstein, see find_wall_ahead in hyp_tiled.h); the
// descent below a same-level neighbour makes the reference's comparisons on the LDS copy of the records.  Everything rare
// -- the propagation check, a step within 1e-6 of a cell edge (the reference's climb), a direction component below 2^-400,
// a neighbour in another cluster -- is handled in the service phase with the general functions on global memory.
#pragma once

#include "hyp_tiled.h"

constexpr int HYP_OTILE_WG = 1024;         // threads per workgroup (one workgroup per task; one per CU with clusters of up to 156 KB: 104.5 -> 97.3 ms on configs[3] against two 512-thread workgroups on 78 KB clusters)
constexpr int HYP_OTILE_OCC = 4;          // waves per SIMD the register budget is set for (16 waves per CU)
#ifndef HYP_OTILE_SERVICE_N
#define HYP_OTILE_SERVICE_N 24
#endif
#ifndef HYP_OTILE_STEPS_N
#define HYP_OTILE_STEPS_N 8
#endif
constexpr int HYP_OTILE_SERVICE = HYP_OTILE_SERVICE_N;      // lanes that must wait before a wave runs its service phase (8 / 16 / 24 / 32: 118.9 / 119.1 / 114.2 / - ms at 4 steps)
constexpr int HYP_OTILE_STEPS = HYP_OTILE_STEPS_N;         // cell steps between two scheduling decisions of a wave (2 / 4 / 8: 125.5 / 119.1 / 108.9 ms; 8 with 24 lanes: 103.2)
#define OT_HIST 256               // clusters whose packet counts a task collects in LDS (the others: global atomics)

// lane states beyond those of tile_walk_kernel: LS_LEFT = the neighbour is in another cluster (found through the global
// tables in the service phase); LS_SLOW = this step needs the general find_wall / advance
enum { LS_OSLOW = 8 };

// TileGeom for this schedule: n_bricks = number of clusters; bx = most cells, by = most refined cells of a cluster.
template <int ND>
__global__ __launch_bounds__(HYP_OTILE_WG, HYP_OTILE_OCC) void otile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                                  void *__restrict__ hot_v, void *__restrict__ cold_v,
                                                                  const int *__restrict__ order,
                                                                  const TileTask *__restrict__ tasks, int *__restrict__ slot_brick,
                                                                  int *__restrict__ ilist, int *__restrict__ dlist,
                                                                  TileCount *__restrict__ tcount, unsigned int *__restrict__ counts)
{
    extern __shared__ float4 lds16[];        // 16-byte aligned base: the records are read with ds_read_b128
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    const DProblem &P = *Pp;
    if (blockIdx.x >= ctl->n_tasks[T.pool]) return;
    const TileTask tk = tasks[blockIdx.x];
    const int cl = tk.brick;
    const int c0 = P.ot_c0[cl], nc = P.ot_nc[cl];
    const int k0 = P.ot_kid_off[cl], nk = P.ot_kid_off[cl + 1] - k0;
    OctCell *rec = (OctCell *)lds16;
    double *dens = (double *)(rec + T.bx);
    double *accum = dens + (size_t)T.bx * ND;
    short *kid = (short *)(accum + (size_t)T.bx * ND);
    short *nbt = kid + (size_t)T.by * 8;
    __shared__ int next_pkt, n_int_l, n_dead_l, pub_base[2];
    __shared__ unsigned int nb_cnt[OT_HIST];
    __shared__ double red[TILE_RED_N];
    {
        const float4 *src = (const float4 *)(P.ot_rec + c0);
        float4 *dst = (float4 *)rec;
        for (int i = threadIdx.x; i < nc * 2; i += blockDim.x) dst[i] = src[i];
        src = (const float4 *)(P.ot_kid + (size_t)k0 * 8); dst = (float4 *)kid;
        for (int i = threadIdx.x; i < nk; i += blockDim.x) dst[i] = src[i];
        const int *s3 = (const int *)(P.ot_nb + (size_t)c0 * 6); int *d3 = (int *)nbt;
        for (int i = threadIdx.x; i < nc * 3; i += blockDim.x) d3[i] = s3[i];
    }
    for (int i = threadIdx.x; i < nc * ND; i += blockDim.x) {
        dens[i] = P.density[(size_t)c0 * ND + i];
        accum[i] = 0.0;
    }
    for (int i = threadIdx.x; i < OT_HIST; i += blockDim.x) nb_cnt[i] = 0;
    if (threadIdx.x >= 256 && threadIdx.x < 256 + TILE_RED_N) red[threadIdx.x - 256] = 0.0;
    if (threadIdx.x == 320) { next_pkt = 0; n_int_l = 0; n_dead_l = 0; }
    __syncthreads();

    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    // lane state: the walking part of a packet (the rest stays in its ColdRec)
    double r[3] = {0.0, 0.0, 0.0}, v[3] = {1.0, 1.0, 1.0}, inv[3] = {1.0, 1.0, 1.0}, tau_req = 0.0, tau_ach = 0.0, energy = 0.0, chi[ND], kappa[ND];
    double cc[3] = {0.0, 0.0, 0.0};          // centre of the current cell
    double t_src = HYP_INF, t_ach = 0.0;     // re-absorption by sources (P.any_intersect): see Packet
    // the record's last word as it is (subcell | level << 8 | refined << 16 | pad << 24): the level, and in `pad` of the cluster's copy the axes
    // on which the cell passes the size half of geo_advance's edge test (build_oct_clusters) -- one register for both
    int meta = 0;
#define OT_LEVEL ((meta >> 8) & 255)
#define OT_EDGE_OK ((unsigned)meta >> 24)
    auto meta_of = [](const OctCell &o) { int m; __builtin_memcpy(&m, &o.subcell, 4); return m; };
    static_assert(offsetof(OctCell, subcell) == 28 && offsetof(OctCell, level) == 29 && offsetof(OctCell, pad) == 31, "OctCell layout");
    int loc = 0, ow_axis = 0;     // ow_axis: (axis + 1) * sign of the wall the packet sits on (0: none), the packed on_wall_id
    int next_cell = 0;                       // LS_LEFT: where the neighbour table points (a cell of another cluster)
    bool v_ok = true;
    Rng g; g.key0 = P.seed_key; g.key1 = T.iter_tag; g.blk_a = 0; g.have_a = 0; g.buf_a = 0.0;
    g.id_lo = g.id_hi = 0; g.blk_b = 0; g.countdown = 0;
    int slot = -1, kind = 0;                  // kind: HotRec::pad, the kind of the packet's next interaction (store_records, TileGeom::presort)
    int st = LS_IDLE;
    bool exhausted = false;
#pragma unroll
    for (int d = 0; d < ND; d++) { chi[d] = 0.0; kappa[d] = 0.0; }
    const Walls W = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {0, 0, 0}};

    // the general cell record of the lane's packet (for the functions of hyp_kernels.h)
    auto full_cell = [&](Cell<GEOM_OCT> &c) {
        c.id = c0 + loc; c.c[0] = cc[0]; c.c[1] = cc[1]; c.c[2] = cc[2]; c.level = OT_LEVEL;
        c.parent = P.oct_cells[c.id].parent; c.subcell = rec[loc].subcell;
        c.ow[0] = c.ow[1] = c.ow[2] = 0;
        if (ow_axis > 0) c.ow[ow_axis - 1] = 1; else if (ow_axis < 0) c.ow[-ow_axis - 1] = -1;
    };

    bool queue_empty = false;        // wave-uniform: some lane found the task's queue empty
#ifdef HYP_TILE_STATS
    unsigned long long dbg_wsteps = 0, dbg_lsteps = 0, dbg_outer = 0;
    const long long dbg_t0 = clock64();
#endif
    for (;;) {
#ifdef HYP_TILE_STATS
        dbg_outer++;
#endif
        if (queue_empty && st == LS_IDLE) exhausted = true;
        const unsigned long long m_walk = __ballot(st == LS_WALK);
        const unsigned long long m_out = __ballot(st >= LS_LEFT);      // anything the service phase must look at
        const unsigned long long m_idle = __ballot(st == LS_IDLE && !exhausted);
        if (!(m_walk | m_out | m_idle)) break;
        // tail of a task: the last few walking packets of a wave go back to their slots and continue next generation
        const bool park = !m_idle && queue_empty && __popcll(m_walk) <= T.park;
        // ---- service phase: rare events, write finished visits back, take new packets ----
        if (park || ((m_out | m_idle) && (__popcll(m_out | m_idle) >= HYP_OTILE_SERVICE || !m_walk))) {
            int left_cell = -1;              // >= 0: the packet moves on to this leaf (of another cluster)
            // propagation check (grid_propagate_3d.f90:112-120), then the step goes on as usual
            // (a lane whose check is due waits until four are, or nobody walks any more: tile_walk_kernel, hyp_tiled.h)
            if (st == LS_CHECK && (__popcll(__ballot(st == LS_CHECK)) >= 4 || !m_walk || park)) {
                const int gap = rng_check_gap(g, P.check_p, P.check_log1mp);
                g.countdown = gap < 0x7fffffff ? gap + 1 : gap;      // the step below takes one off again
                Cell<GEOM_OCT> c; full_cell(c);
                if (geo_in_correct_cell(P, W, r, c)) st = LS_WALK;
                else { cnt.killed_geo++; st = LS_DEAD; }
            }
            // a whole step with the general functions: wall search with true divisions, the reference's climb near an edge
            if (st == LS_OSLOW) {
                Cell<GEOM_OCT> c; full_cell(c);
                double tmin; int im[3];
                if (g.countdown == 0) st = LS_CHECK;       // (cannot happen: the step loop tests it first)
                else if (!geo_find_wall(P, W, r, v, c, tmin, im)) { cnt.killed_geo++; st = LS_DEAD; }
                else {
                    g.countdown--;
                    double rho[ND], chi_rho = 0.0;
#pragma unroll
                    for (int d = 0; d < ND; d++) { rho[d] = dens[loc * ND + d]; chi_rho += chi[d] * rho[d]; }
                    const double tau_cell = chi_rho * tmin;
                    const double tau_needed = tau_req - tau_ach;
                    cnt.crossings++;
                    if (tau_cell < tau_needed) {
                        bool reabs = false;
                        if (P.any_intersect) { t_ach += tmin; reabs = t_ach > t_src; }
                        if (reabs) st = LS_REABS;
                        else {
#pragma unroll
                            for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
                            tau_ach += tau_cell;
#pragma unroll
                            for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tmin * kappa[d] * energy);
                            geo_advance(P, r, c, im);
                            ow_axis = c.ow[0] ? c.ow[0] : c.ow[1] ? 2 * c.ow[1] : 3 * c.ow[2];
                            if (geo_escaped(P, c)) st = LS_DEAD;
                            else if (c.id >= c0 && c.id < c0 + nc) {
                                loc = c.id - c0; cc[0] = c.c[0]; cc[1] = c.c[1]; cc[2] = c.c[2]; meta = meta_of(rec[loc]); st = LS_WALK;
                            } else { left_cell = c.id; st = LS_LEFT; }
                        }
                    } else {
                        const double tact = tmin * (tau_needed / tau_cell);
                        bool reabs = false;
                        if (P.any_intersect) { t_ach += tact; reabs = t_ach > t_src; }
                        if (reabs) st = LS_REABS;
                        else {
#pragma unroll
                            for (int a = 0; a < 3; a++) r[a] = r[a] + tact * v[a];
                            tau_ach += tau_needed;
                            ow_axis = 0;       // geo_clear_wall
#pragma unroll
                            for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tact * kappa[d] * energy);
                            st = LS_HIT;
                        }
                    }
                }
            } else if (st == LS_LEFT) {
                // the neighbour table pointed outside the cluster: descend to the leaf through the global records
                Cell<GEOM_OCT> c;
                oct_descend(P, r, next_cell, c);
                left_cell = c.id;
            }
            if (st == LS_DEAD) {
                hot[slot].state = TS_DEAD; slot_brick[slot] = TILE_NEEDS_PREPARE;
                dlist[tk.start + atomicAdd(&n_dead_l, 1)] = slot;
                finished++; st = LS_IDLE;
            } else if (st == LS_LEFT || st == LS_HIT || st == LS_REABS || (park && st == LS_WALK)) {
                HotRec<ND> &H = hot[slot];
#pragma unroll
                for (int a = 0; a < 3; a++) H.r[a] = r[a];
                // pack_ow: (ow0 + 1) | (ow1 + 1) << 2 | (ow2 + 1) << 4
                H.ow = ow_axis == 0 ? 21 : (ow_axis > 0 ? 21 + (1 << (2 * (ow_axis - 1))) : 21 - (1 << (2 * (-ow_axis - 1))));
                H.tau_ach = tau_ach; H.countdown = g.countdown; H.blk_b = g.blk_b;
                if (P.any_intersect) cold[slot].t_ach = t_ach;
                if (st == LS_LEFT) {                                                  // H.state stays TS_WALK
                    const int ncl = P.ot_cluster[left_cell];
                    H.ic[0] = left_cell; H.ic[2] = ncl;
                    slot_brick[slot] = ncl;
                    if (ncl < OT_HIST) atomicAdd(&nb_cnt[ncl], 1u); else atomicAdd(&counts[ncl], 1u);
                } else {
                    H.ic[0] = c0 + loc; H.ic[2] = cl;
                    if (st == LS_REABS) { H.state = TS_REEMIT; slot_brick[slot] = TILE_NEEDS_REEMIT; }
                    else if (st == LS_HIT) { H.state = TS_INTERACT; slot_brick[slot] = TILE_NEEDS_INTERACT; }
                    else if (cl < OT_HIST) atomicAdd(&nb_cnt[cl], 1u);                // parked: same cluster again
                    else atomicAdd(&counts[cl], 1u);
                    if (st == LS_REABS || st == LS_HIT) ilist[tk.start + atomicAdd(&n_int_l, 1)] = (T.presort && st == LS_HIT) ? (slot | (kind << 30)) : slot;
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
                    v_ok = true;
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        r[a] = H.r[a]; v[a] = H.v[a];
                        inv[a] = 1.0 / v[a];
                        v_ok = v_ok & ((v[a] == 0.0) | (fabs(v[a]) >= 0x1p-400));      // no short circuit: its branches split the record's loads into batches with a wait each
                    }
                    loc = H.ic[0] - c0;
                    int ow[3]; unpack_ow(H.ow, ow);
                    ow_axis = ow[0] ? ow[0] : ow[1] ? 2 * ow[1] : 3 * ow[2];
                    tau_req = H.tau_req; tau_ach = H.tau_ach; energy = H.energy;
#pragma unroll
                    for (int d = 0; d < ND; d++) { chi[d] = H.chi[d]; kappa[d] = H.kappa[d]; }
                    const unsigned long long id = H.id;
                    g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
                    g.countdown = H.countdown; g.blk_b = H.blk_b;
                    kind = H.pad;
                    if (P.any_intersect) { t_src = cold[slot].t_src; t_ach = cold[slot].t_ach; }
                    const OctCell o = rec[loc];
                    cc[0] = o.x; cc[1] = o.y; cc[2] = o.z; meta = meta_of(o);
                    st = LS_WALK;
                }
            }
            if (__ballot(exhausted)) queue_empty = true;
        }
        // ---- a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232) ----
#pragma unroll 1
        for (int q = 0; q < HYP_OTILE_STEPS; q++) {
#ifdef HYP_TILE_STATS
            { const unsigned long long mw = __ballot(st == LS_WALK); if (mw) { dbg_wsteps++; dbg_lsteps += __popcll(mw); } }
#endif
            if (st == LS_WALK) {
                // find_wall :438-537 -- the face ahead on each axis, t = (c +- h - r) / v as the correctly rounded quotient
                double t[3], h[3];
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    h[a] = ldexp(P.oct_half[a], -OT_LEVEL);
                    // (c - h is c + (-h), bit for bit: the face ahead is c + copysign(h, v) -- one v_bfi_b32 and one addition instead of
                    // two additions, a comparison and a 64-bit select; v = 0 is overridden below)
                    const double wall = cc[a] + __builtin_copysign(h[a], v[a]);
                    const double d = wall - r[a];
                    const double q0 = d * inv[a];
                    const double tq = __builtin_fma(__builtin_fma(-q0, v[a], d), inv[a], q0);
                    t[a] = v[a] == 0.0 ? HYP_DBL_MAX : tq;
                }
                int a;
                if (t[0] < t[2]) a = (t[0] < t[1]) ? 0 : 1;
                else a = (t[2] < t[1]) ? 2 : 1;
                double tmin = a == 0 ? t[0] : a == 1 ? t[1] : t[2];
                const double va = a == 0 ? v[0] : a == 1 ? v[1] : v[2];
                const int up = va > 0.0 ? 1 : 0;
                bool found = true;
                if (tmin < 0.0) {
                    if (tmin > -10.0 * P.oct_eps) tmin = 0.0;
                    else found = false;
                }
                // geo_advance's test: inside the cell's extent on the two other axes, not within 1e-6 of an edge -- evaluated
                // at the end point of a full step
                if (g.countdown == 0) st = LS_CHECK;
                else if (!v_ok) st = LS_OSLOW;
                else if (!found) { cnt.killed_geo++; st = LS_DEAD; }
                else {
                    double rho[ND], chi_rho = 0.0;
#pragma unroll
                    for (int d = 0; d < ND; d++) { rho[d] = dens[loc * ND + d]; chi_rho += chi[d] * rho[d]; }
                    const double tau_cell = chi_rho * tmin;
                    const double tau_needed = tau_req - tau_ach;
                    if (tau_cell < tau_needed) {
                        double rn[3];
#pragma unroll
                        for (int b = 0; b < 3; b++) rn[b] = r[b] + tmin * v[b];
                        bool fast = ((OT_EDGE_OK | (1u << a)) & 7u) == 7u;      // h 1e-6 > 1e-14 (|c| + h) on the two other axes: the builder's bits
#pragma unroll
                        for (int b = 0; b < 3; b++) {
                            const double d = fabs(rn[b] - cc[b]);
                            if (b != a && !(d < h[b] * (1.0 - 1e-6))) fast = false;
                        }
                        if (!fast) st = LS_OSLOW;       // nothing of this step has been applied yet
                        else {
                            g.countdown--;
                            cnt.crossings++;
                            bool reabs = false;
                            if (P.any_intersect) { t_ach += tmin; reabs = t_ach > t_src; }      // :139-143
                            if (reabs) st = LS_REABS;
                            else {
#pragma unroll
                                for (int b = 0; b < 3; b++) r[b] = rn[b];
                                tau_ach += tau_cell;
#pragma unroll
                                for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tmin * kappa[d] * energy);
                                ow_axis = up ? -(a + 1) : (a + 1);           // opposite_wall
                                int n = nbt[loc * 6 + 2 * a + up];
                                if (n == -1) st = LS_DEAD;                    // left the grid: the packet ends here
                                else if (n == -2) { next_cell = P.oct_neigh[6 * (size_t)(c0 + loc) + 2 * a + up]; st = LS_LEFT; }
                                else {
                                    // locate_cell :135-146 from the neighbour, on the cluster's copy of the records
                                    OctCell o = rec[n];
                                    while (o.refined) {
                                        const int sub = (r[0] < o.x ? 0 : 1) | (r[1] < o.y ? 0 : 2) | (r[2] < o.z ? 0 : 4);
                                        n = kid[o.parent * 8 + sub];
                                        o = rec[n];
                                    }
                                    loc = n; cc[0] = o.x; cc[1] = o.y; cc[2] = o.z; meta = meta_of(o);
                                }
                            }
                        }
                    } else {
                        // the interaction happens inside this cell: grid_propagate_3d.f90:170-200
                        g.countdown--;
                        cnt.crossings++;
                        const double tact = tmin * (tau_needed / tau_cell);
                        bool reabs = false;
                        if (P.any_intersect) { t_ach += tact; reabs = t_ach > t_src; }      // :184-188
                        if (reabs) st = LS_REABS;
                        else {
#pragma unroll
                            for (int b = 0; b < 3; b++) r[b] = r[b] + tact * v[b];
                            tau_ach += tau_needed;
                            ow_axis = 0;       // geo_clear_wall
#pragma unroll
                            for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tact * kappa[d] * energy);
                            st = LS_HIT;
                        }
                    }
                }
            }
        }
    }
#ifdef HYP_TILE_STATS
    if (__lane_id() == 0) {
        atomicAdd(&ctl->dbg[0], dbg_outer); atomicAdd(&ctl->dbg[1], dbg_wsteps); atomicAdd(&ctl->dbg[2], dbg_lsteps); atomicAdd(&ctl->dbg[3], 1ull);
        atomicAdd(&ctl->dbg[8], (unsigned long long)(clock64() - dbg_t0));
        if (threadIdx.x == 0) { atomicAdd(&ctl->dbg[4], 1ull); atomicAdd(&ctl->dbg[5], (unsigned long long)tk.len); }
    }
#endif
    __syncthreads();
    for (int i = threadIdx.x; i < OT_HIST; i += blockDim.x) if (nb_cnt[i]) atomicAdd(&counts[i], nb_cnt[i]);
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
        if (val != 0.0) hyp_atomic_add_g(&sum[(size_t)c0 * ND + i], val);
    }
    block_tally_flush(P, ctl, red, cnt, finished);
}
#undef OT_LEVEL
#undef OT_EDGE_OK
