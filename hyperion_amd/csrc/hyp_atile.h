// hyp_atile.h -- brick-tiled Lucy iteration for AMR grids (gfx950).
//
// An AMR grid (type_grid_amr.f90:12-21) is a Cartesian block of n1 x n2 x n3 equal cells with a goto table that says where
// a position continues when it enters a cell covered by a finer grid or steps out of the grid (grid_geometry_amr.f90:
// 357-486).  Every grid is cut into BRICKS of at most 16^3 cells (fewer with several species) whose densities and
// accumulators, walls and slice of the goto table -- ghost layer included, 16 bits per entry -- fit the LDS share of one
// workgroup.  The slot-pool schedule of hyp_tiled.h does the rest: packets wait in slot records, are sorted by brick every
// generation, and one workgroup per task walks the packets of one brick from LDS (ds_add_f64 deposits, one flush per
// task) until they leave the brick, change grid, interact or die.
//
// The walk is grid_geometry_amr.f90:775-871 (find_wall: nearest of the three faces ahead; the quotients (wall - r) / v from
// one reciprocal per visit, corrected to the IEEE quotient as in find_wall_ahead, hyp_tiled.h) and :599-655 (next_cell: goto
// lookup).  A step into another grid needs find_position_in_grid on the global tables (:521-545, the position nudged by
// half the smallest cell width): that, the propagation check, a negative distance (the reference's fatal `negative t`) and
// direction components below 2^-400 are handled in the service phase with the general functions of hyp_kernels.h.
#pragma once

#include "hyp_tiled.h"

constexpr int HYP_ATILE_WG = 768;          // threads per workgroup (one workgroup per task; one per CU with bricks of 32 x 16 x 16 cells: two of 512 threads ->
                                  // one of 1024: 216.1 -> 200.9 ms on the three-level nest; 768 threads at 167 VGPRs, nothing spilled: 204.0 -> 177.9 ms)
constexpr int HYP_ATILE_OCC = 3;          // waves per SIMD the register budget is set for (12 waves per CU; at 4 the walk spilled 40 VGPRs in the step loop)
constexpr int HYP_ATILE_SERVICE = 24;      // lanes that must wait before a wave runs its service phase (16 with 4 steps: 235.5 ms, 24 with 8: 224.2)
constexpr int HYP_ATILE_STEPS = 8;         // cell steps between two scheduling decisions of a wave
#define AT_HIST 256               // slabs whose packet counts a task collects in LDS (the others: global atomics)

enum { LS_ASLOW = 8, LS_AGRID = 9 };      // a whole step / the arrival in another grid through the general functions

// TileGeom for this schedule: n_bricks = number of bricks; bx = most cells, by = most goto entries, bz = most walls of a brick.
template <int ND>
__global__ __launch_bounds__(HYP_ATILE_WG, HYP_ATILE_OCC) void atile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                                  void *__restrict__ hot_v, void *__restrict__ cold_v,
                                                                  const int *__restrict__ order,
                                                                  const TileTask *__restrict__ tasks, int *__restrict__ slot_brick,
                                                                  int *__restrict__ ilist, int *__restrict__ dlist,
                                                                  TileCount *__restrict__ tcount, unsigned int *__restrict__ counts)
{
    extern __shared__ float4 lds16[];
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    const DProblem &P = *Pp;
    if (blockIdx.x >= ctl->n_tasks[T.pool]) return;
    const TileTask tk = tasks[blockIdx.x];
    const int cl = tk.brick;
    const AtSlab S = P.at_slabs[cl];
    const AmrGrid G = P.amr_grids[S.grid];
    const int n0 = G.n[0], n1 = G.n[1], n2 = G.n[2];             // the grid
    const int x0 = S.o[0], y0 = S.o[1], z0 = S.o[2], bx = S.n[0], by = S.n[1], bz = S.n[2];      // the brick
    const int nc = bx * by * bz, ngo = (bx + 2) * (by + 2) * (bz + 2);
    double *dens = (double *)lds16;
    double *accum = dens + (size_t)T.bx * ND;
    double *wx = accum + (size_t)T.bx * ND, *wy = wx + (bx + 1), *wz = wy + (by + 1);      // wx[k] = wall x0 + k of the grid, ...
    short *go = (short *)(accum + (size_t)T.bx * ND + T.bz);
    __shared__ int next_pkt, n_int_l, n_dead_l, pub_base[2];
    __shared__ unsigned int nb_cnt[AT_HIST];
    __shared__ double red[TILE_RED_N];
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        const int lx = c % bx, ly = (c / bx) % by, lz = c / (bx * by);
        const size_t gid = (size_t)G.start + ((size_t)(z0 + lz) * n1 + (y0 + ly)) * n0 + (x0 + lx);
#pragma unroll
        for (int d = 0; d < ND; d++) { dens[c * ND + d] = P.density[gid * ND + d]; accum[c * ND + d] = 0.0; }
    }
    for (int i = threadIdx.x; i <= bx; i += blockDim.x) wx[i] = P.amr_walls[G.w_off[0] + x0 + i];
    for (int i = threadIdx.x; i <= by; i += blockDim.x) wy[i] = P.amr_walls[G.w_off[1] + y0 + i];
    for (int i = threadIdx.x; i <= bz; i += blockDim.x) wz[i] = P.amr_walls[G.w_off[2] + z0 + i];
    for (int i = threadIdx.x; i < ngo; i += blockDim.x) go[i] = P.at_go[S.go_off + i];
    for (int i = threadIdx.x; i < AT_HIST; i += blockDim.x) nb_cnt[i] = 0;
    if (threadIdx.x >= 256 && threadIdx.x < 256 + TILE_RED_N) red[threadIdx.x - 256] = 0.0;
    if (threadIdx.x == 320) { next_pkt = 0; n_int_l = 0; n_dead_l = 0; }
    __syncthreads();

    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    // lane state: the walking part of a packet (the rest stays in its ColdRec)
    double r[3] = {0.0, 0.0, 0.0}, v[3] = {1.0, 1.0, 1.0}, inv[3] = {1.0, 1.0, 1.0}, tau_req = 0.0, tau_ach = 0.0, energy = 0.0, chi[ND], kappa[ND];
    double t_src = HYP_INF, t_ach = 0.0;     // re-absorption by sources (P.any_intersect): see Packet
    int i0 = 0, i1 = 0, i2 = 0;              // 0-based position in the grid
    int ow_axis = 0;                         // (axis + 1) * sign of the wall the packet sits on (0: none)
    int go_grid = 0, go_axis = 0;            // LS_AGRID: the grid the goto table named (+ 1), the axis and sense of the step
    bool v_ok = true;
    Rng g; g.key0 = P.seed_key; g.key1 = T.iter_tag; g.blk_a = 0; g.have_a = 0; g.buf_a = 0.0;
    g.id_lo = g.id_hi = 0; g.blk_b = 0; g.countdown = 0;
    int slot = -1, kind = 0;                  // kind: HotRec::pad, the kind of the packet's next interaction (store_records, TileGeom::presort)
    int st = LS_IDLE;
    bool exhausted = false;
#pragma unroll
    for (int d = 0; d < ND; d++) { chi[d] = 0.0; kappa[d] = 0.0; }
    const Walls W = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {0, 0, 0}};

    auto full_cell = [&](Cell<GEOM_AMR> &c) {
        c.grid = S.grid; c.i[0] = i0; c.i[1] = i1; c.i[2] = i2;
        c.id = (int)(G.start + (unsigned)((i2 * n1 + i1) * n0 + i0));
        c.ow[0] = c.ow[1] = c.ow[2] = 0;
        if (ow_axis > 0) c.ow[ow_axis - 1] = 1; else if (ow_axis < 0) c.ow[-ow_axis - 1] = -1;
    };
    auto local_index = [&]() { return ((i2 - z0) * by + (i1 - y0)) * bx + (i0 - x0); };
    auto in_brick = [&](const int i[3]) { return i[0] >= x0 && i[0] < x0 + bx && i[1] >= y0 && i[1] < y0 + by && i[2] >= z0 && i[2] < z0 + bz; };

    bool queue_empty = false;        // wave-uniform: some lane found the task's queue empty
    for (;;) {
        if (queue_empty && st == LS_IDLE) exhausted = true;
        const unsigned long long m_walk = __ballot(st == LS_WALK);
        const unsigned long long m_out = __ballot(st >= LS_LEFT);      // anything the service phase must look at
        const unsigned long long m_idle = __ballot(st == LS_IDLE && !exhausted);
        if (!(m_walk | m_out | m_idle)) break;
        const bool park = !m_idle && queue_empty && __popcll(m_walk) <= T.park;
        // ---- service phase: rare events, write finished visits back, take new packets ----
        if (park || ((m_out | m_idle) && (__popcll(m_out | m_idle) >= HYP_ATILE_SERVICE || !m_walk))) {
            int left_cell = -1, left_grid = -1, left_i[3] = {0, 0, 0};      // LS_LEFT: where the packet goes on
            // (a lane whose check is due waits until four are, or nobody walks any more: tile_walk_kernel, hyp_tiled.h)
            if (st == LS_CHECK && (__popcll(__ballot(st == LS_CHECK)) >= 4 || !m_walk || park)) {
                const int gap = rng_check_gap(g, P.check_p, P.check_log1mp);
                g.countdown = gap < 0x7fffffff ? gap + 1 : gap;      // the step below takes one off again
                Cell<GEOM_AMR> c; full_cell(c);
                if (geo_in_correct_cell(P, W, r, c)) st = LS_WALK;
                else { cnt.killed_geo++; st = LS_DEAD; }
            }
            // a whole step with the general functions (true divisions; a negative distance raises the reference's error)
            if (st == LS_ASLOW) {
                Cell<GEOM_AMR> c; full_cell(c);
                double tmin; int im[3];
                if (!geo_find_wall(P, W, r, v, c, tmin, im)) { cnt.killed_geo++; st = LS_DEAD; }
                else {
                    g.countdown--;
                    const int loc = local_index();
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
                            if (geo_invalid(P, c)) { cnt.killed_geo++; st = LS_DEAD; }
                            else if (geo_escaped(P, c)) st = LS_DEAD;
                            else if (c.grid == S.grid && in_brick(c.i)) { i0 = c.i[0]; i1 = c.i[1]; i2 = c.i[2]; st = LS_WALK; }
                            else { left_cell = c.id; left_grid = c.grid; left_i[0] = c.i[0]; left_i[1] = c.i[1]; left_i[2] = c.i[2]; st = LS_LEFT; }
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
            } else if (st == LS_AGRID) {
                // next_cell_int :629-654: the goto table named another grid; the position, nudged by eps along the step, is
                // located there (and in the grids it points on to)
                double rr[3] = {r[0], r[1], r[2]};
                const int ax = (go_axis < 0 ? -go_axis : go_axis) - 1;
                const double e = go_axis > 0 ? P.amr_eps : -P.amr_eps;
                if (ax == 0) rr[0] += e; else if (ax == 1) rr[1] += e; else rr[2] += e;
                Cell<GEOM_AMR> c;
                amr_find_position(P, rr, go_grid - 1, c);
                if (c.id < 0) { cnt.killed_geo++; st = LS_DEAD; }          // invalid_cell
                else if (c.grid == S.grid && in_brick(c.i)) { i0 = c.i[0]; i1 = c.i[1]; i2 = c.i[2]; st = LS_WALK; }
                else { left_cell = c.id; left_grid = c.grid; left_i[0] = c.i[0]; left_i[1] = c.i[1]; left_i[2] = c.i[2]; st = LS_LEFT; }
            } else if (st == LS_LEFT) {
                // the neighbouring brick of this grid
                left_cell = (int)(G.start + (unsigned)((i2 * n1 + i1) * n0 + i0)); left_grid = S.grid; left_i[0] = i0; left_i[1] = i1; left_i[2] = i2;
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
                    const int ncl = amr_brick_of(P, left_grid, left_i);
                    H.ic[0] = left_cell; H.ic[1] = left_grid; H.ic[2] = ncl;
                    slot_brick[slot] = ncl;
                    if (ncl < AT_HIST) atomicAdd(&nb_cnt[ncl], 1u); else atomicAdd(&counts[ncl], 1u);
                } else {
                    H.ic[0] = (int)(G.start + (unsigned)((i2 * n1 + i1) * n0 + i0)); H.ic[1] = S.grid; H.ic[2] = cl;
                    if (st == LS_REABS) { H.state = TS_REEMIT; slot_brick[slot] = TILE_NEEDS_REEMIT; }
                    else if (st == LS_HIT) { H.state = TS_INTERACT; slot_brick[slot] = TILE_NEEDS_INTERACT; }
                    else if (cl < AT_HIST) atomicAdd(&nb_cnt[cl], 1u);                // parked: same brick again
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
                    const int local = H.ic[0] - (int)G.start;    // index inside the grid
                    i0 = local % n0; i1 = (local / n0) % n1; i2 = local / (n0 * n1);
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
                    st = LS_WALK;
                }
            }
            if (__ballot(exhausted)) queue_empty = true;
        }
        // ---- a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232) ----
#pragma unroll 1
        for (int q = 0; q < HYP_ATILE_STEPS; q++) {
            if (st == LS_WALK) {
                // find_wall :775-871 -- the face ahead on each axis, t = (wall - r) / v as the correctly rounded quotient
                const int ii[3] = {i0 - x0, i1 - y0, i2 - z0};
                const double *ww[3] = {wx, wy, wz};
                double t[3];
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const double wall = ww[a][ii[a] + (v[a] > 0.0 ? 1 : 0)];
                    const double d = wall - r[a];
                    const double q0 = d * inv[a];
                    const double tq = __builtin_fma(__builtin_fma(-q0, v[a], d), inv[a], q0);
                    t[a] = v[a] == 0.0 ? HYP_DBL_MAX : tq;
                }
                int a;
                if (t[0] < t[2]) a = (t[0] < t[1]) ? 0 : 1;
                else a = (t[2] < t[1]) ? 2 : 1;
                const double tmin = a == 0 ? t[0] : a == 1 ? t[1] : t[2];
                const double va = a == 0 ? v[0] : a == 1 ? v[1] : v[2];
                const int dir = va > 0.0 ? 1 : -1;
                if (g.countdown == 0) st = LS_CHECK;
                else if (!v_ok || fmin(t[0], fmin(t[1], t[2])) < 0.0) st = LS_ASLOW;      // negative t: the general function raises the error
                else {
                    const int loc = local_index();
                    double rho[ND], chi_rho = 0.0;
#pragma unroll
                    for (int d = 0; d < ND; d++) { rho[d] = dens[loc * ND + d]; chi_rho += chi[d] * rho[d]; }
                    const double tau_cell = chi_rho * tmin;
                    const double tau_needed = tau_req - tau_ach;
                    g.countdown--;
                    cnt.crossings++;
                    if (tau_cell < tau_needed) {
                        bool reabs = false;
                        if (P.any_intersect) { t_ach += tmin; reabs = t_ach > t_src; }      // :139-143
                        if (reabs) st = LS_REABS;
                        else {
#pragma unroll
                            for (int b = 0; b < 3; b++) r[b] = r[b] + tmin * v[b];
                            tau_ach += tau_cell;
#pragma unroll
                            for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[loc * ND + d], tmin * kappa[d] * energy);
                            ow_axis = dir > 0 ? -(a + 1) : (a + 1);           // opposite_wall
                            // next_cell_int :599-627: the goto table at the new position (1-based, ghost layer 0 and n + 1)
                            if (a == 0) i0 += dir; else if (a == 1) i1 += dir; else i2 += dir;
                            const int gidx = ((i2 - z0 + 1) * (by + 2) + (i1 - y0 + 1)) * (bx + 2) + (i0 - x0 + 1);
                            const int gg = go[gidx];
                            if (gg != 0) { go_grid = gg; go_axis = dir * (a + 1); st = LS_AGRID; }
                            else if (i0 < 0 || i0 >= n0 || i1 < 0 || i1 >= n1 || i2 < 0 || i2 >= n2) st = LS_DEAD;      // outside every grid: the packet ends here
                            else if (i0 < x0 || i0 >= x0 + bx || i1 < y0 || i1 >= y0 + by || i2 < z0 || i2 >= z0 + bz) st = LS_LEFT;
                        }
                    } else {
                        // the interaction happens inside this cell: grid_propagate_3d.f90:170-200
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
    __syncthreads();
    for (int i = threadIdx.x; i < AT_HIST; i += blockDim.x) if (nb_cnt[i]) atomicAdd(&counts[i], nb_cnt[i]);
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
        const size_t gid = (size_t)G.start + ((size_t)(z0 + lz) * n1 + (y0 + ly)) * n0 + (x0 + lx);
#pragma unroll
        for (int d = 0; d < ND; d++) {
            const double val = accum[c * ND + d];
            if (val != 0.0) hyp_atomic_add_g(&sum[gid * ND + d], val);
        }
    }
    block_tally_flush(P, ctl, red, cnt, finished);
}
