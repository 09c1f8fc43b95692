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

// Round 6 (same-box A/Bs on the 400 x 200 grids, one species): one 1024-thread workgroup per CU at the 4-waves budget (128 VGPRs, 39 spilled) with 8 steps
// per scheduling decision against 768 threads at the 3-waves budget (168 VGPRs) with 4 -- spherical 548.0 -> 528.0 ms per 3e7 packets, cylindrical 140.4 ->
// 129.5 ms per 2e7; 8 steps alone 538.9 / 135.9; 1024 threads alone 538.1; 2 steps / 8 lanes and 6 / 24: 566-568.  Two to four species keep the shape
// they were measured with.
#ifndef HYP_PTILE_WIDE_ND
#define HYP_PTILE_WIDE_ND 1      // species counts up to this one walk in the 1024 / 4 / 8 shape
#endif
template <int ND> constexpr int ptile_wg() { return ND <= HYP_PTILE_WIDE_ND ? 1024 : 768; }       // threads per workgroup (one workgroup per task, one per CU)
template <int ND> constexpr int ptile_occ() { return ND <= HYP_PTILE_WIDE_ND ? 4 : 3; }           // waves per SIMD the register budget is set for
constexpr int HYP_PTILE_SERVICE = 16;     // lanes that must wait before a wave runs its service phase
template <int ND> constexpr int ptile_steps() { return ND <= HYP_PTILE_WIDE_ND ? 8 : 4; }         // cell steps between two scheduling decisions of a wave
#define PT_HIST 256              // bricks whose packet counts a task collects in LDS (the others: global atomics)


// (A mixed-precision wall search for spherical grids -- every candidate root bounded in FP32, the winner solved once in FP64 -- was built
// and measured in round 4: 623 against 493 ms at 2e7 packets, ~2000 instructions of bounds and selects per step with 8 % of the wave-steps
// still running the reference's search for one lane.  profiles/r04_tiled_log.md has the numbers; the code is in the history.)

// TileGeom: bx, by, bz = brick size in cells (the last brick of an axis is ragged), nbx, nby, nbz = bricks per axis
template <int ND, int GEOM>
__global__ __launch_bounds__(ptile_wg<ND>(), ptile_occ<ND>()) void ptile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
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
    double cyl_v2 = 1.0, cyl_inv_v2 = 1.0, cyl_inv_vz = 1.0;      // cylindrical grids: v_xy^2 and the reciprocals of the flight
    bool cyl_ok = false;
    const bool grid_tame = P.w[0][P.n1] >= 0x1p-250 && P.w[0][P.n1] <= 0x1p250;
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

    bool queue_empty = false;        // wave-uniform: some lane found the task's queue empty
#ifdef HYP_TILE_STATS
    unsigned long long dbg_outer = 0, dbg_wsteps = 0, dbg_lsteps = 0, dbg_service = 0, dbg_nservice = 0, dbg_step = 0, dbg_q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
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
        // ---- service phase: write finished visits back, take new packets ----
        if (park || ((m_out | m_idle) && (__popcll(m_out | m_idle) >= HYP_PTILE_SERVICE || !m_walk))) {
#ifdef HYP_TILE_STATS
            const long long dbg_ts = clock64();
#endif
            // propagation check (grid_propagate_3d.f90:112-120), then the step goes on as usual
            // (a lane whose check is due waits until four are, or nobody walks any more: tile_walk_kernel, hyp_tiled.h)
            if (st == LS_CHECK && (__popcll(__ballot(st == LS_CHECK)) >= 4 || !m_walk || park)) {
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
                    if constexpr (GEOM == GEOM_CYL) {      // the flight's reciprocals (cyl_find_wall_inv)
                        cyl_v2 = v[0] * v[0] + v[1] * v[1];
                        cyl_ok = grid_tame && cyl_v2 >= 0x1p-200 && fabs(v[2]) >= 0x1p-200;
                        cyl_inv_v2 = 1.0 / cyl_v2; cyl_inv_vz = 1.0 / v[2];
                    }
                    st = LS_WALK;
                }
            }
            if (__ballot(exhausted)) queue_empty = true;
#ifdef HYP_TILE_STATS
            dbg_service += clock64() - dbg_ts; dbg_nservice++;
#endif
        }
        // ---- a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232) ----
#pragma unroll 1
        for (int q = 0; q < ptile_steps<ND>(); q++) {
#ifdef HYP_TILE_STATS
            long long dbg_tf = 0;
            {
                const unsigned long long mw = __ballot(st == LS_WALK);
                if (mw) { dbg_wsteps++; dbg_lsteps += __popcll(mw); }
#ifdef HYP_TILE_STATS_Q
                if constexpr (GEOM == GEOM_SPH) {      // which quadratics this step solves (the conditions of sph_find_wall / sph_wall_cone)
                    const bool wk = st == LS_WALK && g.countdown != 0;
                    const double v2_xy = v[0] * v[0] + v[1] * v[1], v2_z = v[2] * v[2], rv_xy = r[0] * v[0] + r[1] * v[1], rv_z = r[2] * v[2];
                    const double r2_xy = r[0] * r[0] + r[1] * r[1], r2_z = r[2] * r[2];
                    const double pB = 2.0 * (rv_xy + rv_z), pC = r2_xy + r2_z;
                    const bool inner = wk && !cell.radial && !(pB >= 0.0 && pC - P.wr2[cell.ic[0]] >= 0.0);
                    bool cone[2], onw[2];
                    for (int side = 0; side < 2; side++) {
                        const int iw = cell.ic[1] + side;
                        const double tt2 = P.wtant2[iw];
                        const double pA = v2_xy - v2_z * tt2, pB2 = 2.0 * (rv_xy - rv_z * tt2), pC2 = r2_xy - r2_z * tt2;
                        const bool there = wk && (side ? cell.ic[1] < P.n2 - 1 : cell.ic[1] > 0);
                        cone[side] = there && iw != P.midplane && !((pA > 0.0 && pB2 > 0.0 && pC2 > 0.0) || (pA < 0.0 && pB2 < 0.0 && pC2 < 0.0));
                        onw[side] = there && cell.ow[1] == (side ? +1 : -1);
                    }
                    const unsigned long long b0 = __ballot(inner), b1 = __ballot(cone[0]), b2 = __ballot(cone[1]), b3 = __ballot(onw[0] || onw[1]);
                    dbg_q[0] += b0 != 0; dbg_q[1] += __popcll(b0); dbg_q[2] += (b1 != 0) + (b2 != 0); dbg_q[3] += __popcll(b1) + __popcll(b2);
                    dbg_q[4] += b3 != 0; dbg_q[5] += __popcll(b3);
                    const unsigned long long b4 = __ballot(wk && cell.ow[0] != 0);
                    dbg_q[6] += __popcll(b4);
                }
#endif
                dbg_tf = clock64();
            }
#endif
            if (st == LS_WALK) {
                if (g.countdown == 0) st = LS_CHECK;
                else {
                    g.countdown--;
                    double tmin; int im[3];
                    bool found;
                    if constexpr (GEOM == GEOM_SPH) found = sph_find_wall(P, W, r, v, cell, tmin, im, reach_task);
                    else found = cyl_ok ? cyl_find_wall_inv(P, r, v, cell, cyl_v2, cyl_inv_v2, cyl_inv_vz, tmin, im) : geo_find_wall(P, W, r, v, cell, tmin, im);
#ifdef HYP_TILE_STATS
                    dbg_q[7] += clock64() - dbg_tf;      // (lanes of the wave agree on the clock: the busiest lane's path)
#endif
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
#ifdef HYP_TILE_STATS
            dbg_step += clock64() - dbg_tf;
#endif
        }
    }
#ifdef HYP_TILE_STATS
    if (__lane_id() == 0) {
        atomicAdd(&ctl->dbg[32], dbg_step);
        atomicAdd(&ctl->dbg[0], dbg_outer); atomicAdd(&ctl->dbg[1], dbg_wsteps); atomicAdd(&ctl->dbg[2], dbg_lsteps); atomicAdd(&ctl->dbg[3], 1ull);
        atomicAdd(&ctl->dbg[6], dbg_service); atomicAdd(&ctl->dbg[7], dbg_nservice); atomicAdd(&ctl->dbg[8], (unsigned long long)(clock64() - dbg_t0));
        for (int i = 0; i < 8; i++) atomicAdd(&ctl->dbg[24 + i], dbg_q[i]);
        if (threadIdx.x == 0) { atomicAdd(&ctl->dbg[4], 1ull); atomicAdd(&ctl->dbg[5], (unsigned long long)tk.len); }
    }
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
