// hyp_tiled.h -- brick-tiled Lucy iteration for Cartesian grids (gfx950).
//
// Why: one global FP64 atomic per cell crossing caps the straightforward kernel
// at the chip's memory-side atomic rate (2.38e10/s, profiles/r01_atomic_rate_
// ubench.md).  Here the grid is cut into bricks of BX*BY*BZ cells whose density
// and accumulators live in LDS while a workgroup walks, with ds_add_f64, every
// packet that currently sits in that brick; a packet that leaves the brick is
// written back (128-byte "hot" record) and continues in the next generation.
// Per generation:  tile_prepare (interactions + emission, one lane per slot)
//   -> tile_count / tile_scan / tile_scatter (counting sort of slots by brick)
//   -> tile_walk (one workgroup per task = up to TASK packets of one brick).
// Per-packet physics and random streams are the ones of hyp_kernels.h, so the
// result equals the persistent kernel's up to FP64 summation order.
#pragma once

#include "hyp_kernels.h"

enum { TS_DEAD = 0, TS_WALK = 1, TS_INTERACT = 2, TS_DONE = 3 };

#define HYP_TILE_MAX_BRICKS 8192

template <int ND>
struct alignas(16) HotRec {      // what the walk needs (128 B for ND = 1)
    double r[3], v[3];
    double tau_req, tau_ach, energy;
    double chi[ND], kappa[ND];
    unsigned long long id;
    int ic[3];
    int ow;                      // (ow0+1) | (ow1+1)<<2 | (ow2+1)<<4
    int countdown;
    unsigned int blk_b;
    int state;
    int pad;
};

template <int ND>
struct alignas(16) ColdRec {     // only touched at interactions / emission
    Angle a;
    double s[4];
    double nu;
    double albedo[ND];
    double buf_a;
    unsigned int blk_a;
    int have_a, inter, pad;
};

// slot_brick[] values besides a brick index
#define TILE_IDLE (-1)            // slot retired (no packet ids left)
#define TILE_NEEDS_PREPARE (-2)   // packet awaits an interaction, or the slot is free for a new packet
#define HYP_PREP_CHUNK 2048
// build-time shape of tile_walk_kernel (tools/variants.py sweeps these)
#ifndef HYP_TILE_WG
#define HYP_TILE_WG 512          // threads per workgroup (one workgroup per task)
#endif
#ifndef HYP_TILE_FUSE
#define HYP_TILE_FUSE 0          // 1: interactions in place inside the walk kernel
#endif
#ifndef HYP_TILE_DENS_LDS
#define HYP_TILE_DENS_LDS 1      // 1: brick densities staged in LDS, 0: read through L1/L2
#endif
// lane states of tile_walk_kernel
enum { LS_IDLE = 0, LS_WALK = 1, LS_INTERACT = 2 };

struct TileCtl {
    unsigned long long next_id, end_id, n_finished;
    unsigned int n_tasks, pad;
};

struct TileGeom {
    int bx, by, bz;              // brick size in cells
    int nbx, nby, nbz, n_bricks;
    int n_slots, task_size;
    uint32_t iter_tag, pad;
};

struct TileTask { int brick, start, len, pad; };

__device__ __forceinline__ int pack_ow(const int ow[3]) { return (ow[0] + 1) | ((ow[1] + 1) << 2) | ((ow[2] + 1) << 4); }
__device__ __forceinline__ void unpack_ow(int w, int ow[3]) { ow[0] = (w & 3) - 1; ow[1] = ((w >> 2) & 3) - 1; ow[2] = ((w >> 4) & 3) - 1; }

__device__ __forceinline__ int brick_of(const TileGeom &T, const int ic[3])
{
    return ((ic[2] / T.bz) * T.nby + (ic[1] / T.by)) * T.nbx + (ic[0] / T.bx);
}

// ---------------------------------------------------------------------------
// tile_prepare: interactions and (re-)emission, one lane per slot; writes the
// brick of every walking packet.
// ---------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(256) void tile_prepare_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                         HotRec<ND> *__restrict__ hot, ColdRec<ND> *__restrict__ cold,
                                                         int *__restrict__ slot_brick)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM_CAR>(P, lds, W);
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    // Each workgroup scans a chunk of slot_brick[] (coalesced), gathers the slots marked
    // TILE_NEEDS_PREPARE into an LDS list and then works through that list with full waves.
    __shared__ int list[HYP_PREP_CHUNK];
    __shared__ int n_list;
    const int n_chunks = (T.n_slots + HYP_PREP_CHUNK - 1) / HYP_PREP_CHUNK;
    for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    __syncthreads();
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < HYP_PREP_CHUNK; k += blockDim.x) {
        const int s = ch * HYP_PREP_CHUNK + k;
        if (s < T.n_slots && slot_brick[s] == TILE_NEEDS_PREPARE) list[atomicAdd(&n_list, 1)] = s;
    }
    __syncthreads();
    const int nl = n_list;
    for (int k0 = 0; k0 < nl; k0 += (int)blockDim.x) {
        const int k = k0 + (int)threadIdx.x;
        const bool valid = k < nl;
        const int slot = valid ? list[k] : 0;
        int state = valid ? hot[slot].state : TS_DONE;
        Packet<ND, GEOM_CAR> p;
        Rng g;
        unsigned long long id = 0;
        bool touched = false;
        if (state == TS_INTERACT) {
            touched = true;
            const HotRec<ND> &H = hot[slot];
            const ColdRec<ND> &C = cold[slot];
#pragma unroll
            for (int a = 0; a < 3; a++) { p.r[a] = H.r[a]; p.v[a] = H.v[a]; p.cell.ic[a] = H.ic[a]; }
            unpack_ow(H.ow, p.cell.ow);
            p.a = C.a;
            p.s[0] = C.s[0]; p.s[1] = C.s[1]; p.s[2] = C.s[2]; p.s[3] = C.s[3];
            p.nu = C.nu; p.energy = H.energy; p.tau_req = H.tau_req; p.tau_ach = H.tau_ach;
#pragma unroll
            for (int d = 0; d < ND; d++) { p.chi[d] = H.chi[d]; p.kappa[d] = H.kappa[d]; p.albedo[d] = C.albedo[d]; }
            p.inter = C.inter;
            id = H.id;
            g.key0 = P.seed_key; g.key1 = T.iter_tag; g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
            g.blk_a = C.blk_a; g.blk_b = H.blk_b; g.buf_a = C.buf_a; g.have_a = C.have_a; g.countdown = H.countdown;
            if ((long long)p.inter == P.n_inter_max + 1) {
                cnt.killed_int++; state = TS_DEAD; finished++;
            } else {
                int scattered, dust_id;
                bool ok = interact<ND, GEOM_CAR>(P, p, g, cnt, scattered, dust_id);
                bool killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered);
                if (killed) { state = TS_DEAD; finished++; }
                else {
                    p.inter++;
                    p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                    state = (p.tau_req == 0.0) ? TS_INTERACT : TS_WALK;
                }
            }
        }
        // (re-)emission into free slots while packet ids remain
        bool want = state == TS_DEAD && valid;
        unsigned long long m = __ballot(want);
        if (m) {
            const unsigned lane = __lane_id();
            unsigned long long base = 0;
            if (lane == (unsigned)(__ffsll((long long)m) - 1)) base = atomicAdd(&ctl->next_id, (unsigned long long)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1, 64);
            if (want) {
                touched = true;
                id = base + __popcll(m & ((1ull << lane) - 1ull));
                if (id >= ctl->end_id) state = TS_DONE;
                else {
                    rng_init(g, P.seed_key, T.iter_tag, id);
                    int source_id; Angle src_normal;
                    bool ok = emit_packet<ND, GEOM_CAR>(P, W, p, g, cnt, source_id, src_normal);
                    if (!ok || geo_escaped(P, p.cell)) { state = TS_DEAD; finished++; }
                    else {
                        p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                        state = (p.tau_req == 0.0) ? TS_INTERACT : TS_WALK;
                    }
                }
            }
        }
        if (valid && touched) {
            if (state == TS_WALK || state == TS_INTERACT) {
                HotRec<ND> &H = hot[slot];
                ColdRec<ND> &C = cold[slot];
#pragma unroll
                for (int a = 0; a < 3; a++) { H.r[a] = p.r[a]; H.v[a] = p.v[a]; H.ic[a] = p.cell.ic[a]; }
                H.ow = pack_ow(p.cell.ow);
                H.tau_req = p.tau_req; H.tau_ach = p.tau_ach; H.energy = p.energy;
#pragma unroll
                for (int d = 0; d < ND; d++) { H.chi[d] = p.chi[d]; H.kappa[d] = p.kappa[d]; C.albedo[d] = p.albedo[d]; }
                H.id = id; H.countdown = g.countdown; H.blk_b = g.blk_b; H.state = state;
                C.a = p.a; C.s[0] = p.s[0]; C.s[1] = p.s[1]; C.s[2] = p.s[2]; C.s[3] = p.s[3];
                C.nu = p.nu; C.buf_a = g.buf_a; C.blk_a = g.blk_a; C.have_a = g.have_a; C.inter = p.inter;
                // zero optical depth drawn: interact again in the next generation
                slot_brick[slot] = state == TS_WALK ? brick_of(T, p.cell.ic) : TILE_NEEDS_PREPARE;
            } else {
                hot[slot].state = state;
                // a new packet that left the grid at once frees the slot again; TS_DONE retires it
                slot_brick[slot] = state == TS_DEAD ? TILE_NEEDS_PREPARE : TILE_IDLE;
            }
        }
    }
    }
    double e = wave_sum(cnt.energy_current);
    double kg = wave_sum((double)cnt.killed_geo);
    double ki = wave_sum((double)cnt.killed_int);
    double ni = wave_sum((double)cnt.interactions);
    double nf = wave_sum((double)finished);
    if (__lane_id() == 0) {
        if (e != 0.0) unsafeAtomicAdd(&P.tail[TAIL_ENERGY], e);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
        if (ki != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], ki);
        if (ni != 0.0) unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], ni);
        if (nf != 0.0) atomicAdd(&ctl->n_finished, (unsigned long long)nf);
    }
}

// ---------------------------------------------------------------------------
// counting sort of the walking slots by brick
// ---------------------------------------------------------------------------
#define HYP_SORT_PER_THREAD 8

__global__ __launch_bounds__(256) void tile_count_kernel(TileGeom T, const int *__restrict__ slot_brick, unsigned int *__restrict__ counts)
{
    __shared__ unsigned int hist[HYP_TILE_MAX_BRICKS];
    for (int b = threadIdx.x; b < T.n_bricks; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * blockDim.x * HYP_SORT_PER_THREAD;
    for (int k = 0; k < HYP_SORT_PER_THREAD; k++) {
        int slot = base + k * blockDim.x + threadIdx.x;
        if (slot < T.n_slots) { int b = slot_brick[slot]; if (b >= 0) atomicAdd(&hist[b], 1u); }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < T.n_bricks; b += blockDim.x) if (hist[b]) atomicAdd(&counts[b], hist[b]);
}

// exclusive scan of the brick counts, task list, reset of the cursors
__global__ __launch_bounds__(1024) void tile_scan_kernel(TileGeom T, unsigned int *__restrict__ counts, unsigned int *__restrict__ offsets,
                                                        unsigned int *__restrict__ cursor, TileTask *__restrict__ tasks,
                                                        TileCtl *__restrict__ ctl)
{
    __shared__ unsigned int part_c[1024], part_t[1024];
    const int per = (T.n_bricks + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(b0 + per, T.n_bricks);
    unsigned int sc = 0, stt = 0;
    for (int b = b0; b < b1; b++) { sc += counts[b]; stt += (counts[b] + T.task_size - 1) / T.task_size; }
    part_c[threadIdx.x] = sc; part_t[threadIdx.x] = stt;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int ac = 0, at = 0;
        for (int i = 0; i < 1024; i++) { unsigned int c = part_c[i], t = part_t[i]; part_c[i] = ac; part_t[i] = at; ac += c; at += t; }
        ctl->n_tasks = at;
    }
    __syncthreads();
    unsigned int oc = part_c[threadIdx.x], ot = part_t[threadIdx.x];
    for (int b = b0; b < b1; b++) {
        unsigned int c = counts[b];
        offsets[b] = oc; cursor[b] = 0;
        for (unsigned int s = 0; s < c; s += T.task_size) {
            TileTask tk; tk.brick = b; tk.start = (int)(oc + s); tk.len = (int)min((unsigned int)T.task_size, c - s); tk.pad = 0;
            tasks[ot++] = tk;
        }
        oc += c;
        counts[b] = 0;          // ready for the next generation
    }
}

__global__ __launch_bounds__(256) void tile_scatter_kernel(TileGeom T, const int *__restrict__ slot_brick, const unsigned int *__restrict__ offsets,
                                                          unsigned int *__restrict__ cursor, int *__restrict__ order)
{
    __shared__ unsigned int hist[HYP_TILE_MAX_BRICKS];
    for (int b = threadIdx.x; b < T.n_bricks; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * blockDim.x * HYP_SORT_PER_THREAD;
    int br[HYP_SORT_PER_THREAD]; unsigned int rank[HYP_SORT_PER_THREAD];
#pragma unroll
    for (int k = 0; k < HYP_SORT_PER_THREAD; k++) {
        int slot = base + k * blockDim.x + threadIdx.x;
        br[k] = slot < T.n_slots ? slot_brick[slot] : -1;
        rank[k] = br[k] >= 0 ? atomicAdd(&hist[br[k]], 1u) : 0u;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < T.n_bricks; b += blockDim.x)
        if (hist[b]) hist[b] = atomicAdd(&cursor[b], hist[b]);      // hist[] now holds this workgroup's base
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HYP_SORT_PER_THREAD; k++) {
        int slot = base + k * blockDim.x + threadIdx.x;
        if (br[k] >= 0) order[offsets[br[k]] + hist[br[k]] + rank[k]] = slot;
    }
}

// ---------------------------------------------------------------------------
// tile_walk: one workgroup per task; density and accumulators of the brick in LDS
// ---------------------------------------------------------------------------
template <int ND, int BX, int BY, int BZ>
__global__ __launch_bounds__(HYP_TILE_WG) void tile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                      HotRec<ND> *__restrict__ hot, ColdRec<ND> *__restrict__ cold,
                                                      const int *__restrict__ order,
                                                      const TileTask *__restrict__ tasks, int *__restrict__ slot_brick)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    if (blockIdx.x >= ctl->n_tasks) return;
    const TileTask tk = tasks[blockIdx.x];
    constexpr int NC = BX * BY * BZ;
    Walls W;
    stage_walls<GEOM_CAR>(P, lds, W);
#if HYP_TILE_DENS_LDS
    double *dens = lds + 2 * ((size_t)P.n1 + P.n2 + P.n3 + 3);
    double *accum = dens + (size_t)NC * ND;
#else
    double *accum = lds + 2 * ((size_t)P.n1 + P.n2 + P.n3 + 3);
#endif
    __shared__ int next_pkt;
    const int bi = tk.brick % T.nbx, bj = (tk.brick / T.nbx) % T.nby, bk = tk.brick / (T.nbx * T.nby);
    const int x0 = bi * BX, y0 = bj * BY, z0 = bk * BZ;
    for (int c = threadIdx.x; c < NC; c += blockDim.x) {
        int lx = c % BX, ly = (c / BX) % BY, lz = c / (BX * BY);
        int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
        bool in = gx < P.n1 && gy < P.n2 && gz < P.n3;
        size_t gidx = ((size_t)gz * P.n2 + gy) * P.n1 + gx;
        for (int d = 0; d < ND; d++) {
#if HYP_TILE_DENS_LDS
            dens[c * ND + d] = in ? P.density[gidx * ND + d] : 0.0;
#else
            (void)in; (void)gidx;
#endif
            accum[c * ND + d] = 0.0;
        }
    }
    if (threadIdx.x == 0) next_pkt = 0;
    __syncthreads();

    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    // lane state: the walking part of a packet (the rest stays in its ColdRec until an interaction)
    double r[3], v[3], tau_req = 0.0, tau_ach = 0.0, energy = 0.0, chi[ND], kappa[ND];
    Cell<GEOM_CAR> cell;
    Rng g; g.key0 = P.seed_key; g.key1 = T.iter_tag; g.blk_a = 0; g.have_a = 0; g.buf_a = 0.0;
    g.id_lo = g.id_hi = 0; g.blk_b = 0; g.countdown = 0;
    int slot = -1;
    int st = LS_IDLE;
    bool exhausted = false, dirty = false;
#pragma unroll
    for (int a = 0; a < 3; a++) { r[a] = 0.0; v[a] = 1.0; cell.ic[a] = 0; cell.ow[a] = 0; }
#pragma unroll
    for (int d = 0; d < ND; d++) { chi[d] = 0.0; kappa[d] = 0.0; }

    for (;;) {
        unsigned long long m_walk = __ballot(st == LS_WALK);
        unsigned long long m_int = __ballot(st == LS_INTERACT);
        unsigned long long m_idle = __ballot(st == LS_IDLE && !exhausted);
        if (!(m_walk | m_int | m_idle)) break;
        // refill idle lanes when enough of them wait (or nobody walks)
        if (m_idle && (__popcll(m_idle) >= 16 || !m_walk)) {
            if (st == LS_IDLE && !exhausted) {
                int j = atomicAdd(&next_pkt, 1);
                if (j >= tk.len) exhausted = true;
                else {
                    slot = order[tk.start + j];
                    const HotRec<ND> &H = hot[slot];
#pragma unroll
                    for (int a = 0; a < 3; a++) { r[a] = H.r[a]; v[a] = H.v[a]; cell.ic[a] = H.ic[a]; }
                    unpack_ow(H.ow, cell.ow);
                    tau_req = H.tau_req; tau_ach = H.tau_ach; energy = H.energy;
#pragma unroll
                    for (int d = 0; d < ND; d++) { chi[d] = H.chi[d]; kappa[d] = H.kappa[d]; }
                    unsigned long long id = H.id;
                    g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
                    g.countdown = H.countdown; g.blk_b = H.blk_b;
                    st = LS_WALK; dirty = false;
                }
            }
            m_walk = __ballot(st == LS_WALK);
        }
        // interactions in place (interact_with_dust, iter_lucy.f90:165-208): the packet stays in
        // this brick, so it keeps walking here afterwards instead of going through another sort
#if HYP_TILE_FUSE
        if (m_int && (__popcll(m_int) >= 16 || !m_walk)) {
            if (st == LS_INTERACT) {
                ColdRec<ND> &C = cold[slot];
                Packet<ND, GEOM_CAR> p;
#pragma unroll
                for (int a = 0; a < 3; a++) { p.r[a] = r[a]; p.v[a] = v[a]; }
                p.cell = cell;
                p.a = C.a;
                p.s[0] = C.s[0]; p.s[1] = C.s[1]; p.s[2] = C.s[2]; p.s[3] = C.s[3];
                p.nu = C.nu; p.energy = energy; p.tau_req = tau_req; p.tau_ach = tau_ach;
#pragma unroll
                for (int d = 0; d < ND; d++) { p.chi[d] = chi[d]; p.kappa[d] = kappa[d]; p.albedo[d] = C.albedo[d]; }
                p.inter = C.inter;
                g.blk_a = C.blk_a; g.buf_a = C.buf_a; g.have_a = C.have_a;
                bool killed;
                if ((long long)p.inter == P.n_inter_max + 1) { cnt.killed_int++; killed = true; }
                else {
                    int scattered, dust_id;
                    bool ok = interact<ND, GEOM_CAR>(P, p, g, cnt, scattered, dust_id);
                    killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered);
                }
                if (killed) {
                    hot[slot].state = TS_DEAD; slot_brick[slot] = TILE_NEEDS_PREPARE;
                    finished++; st = LS_IDLE;
                } else {
                    p.inter++;
                    p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                    C.a = p.a; C.s[0] = p.s[0]; C.s[1] = p.s[1]; C.s[2] = p.s[2]; C.s[3] = p.s[3];
                    C.nu = p.nu; C.buf_a = g.buf_a; C.blk_a = g.blk_a; C.have_a = g.have_a; C.inter = p.inter;
#pragma unroll
                    for (int d = 0; d < ND; d++) { C.albedo[d] = p.albedo[d]; chi[d] = p.chi[d]; kappa[d] = p.kappa[d]; }
#pragma unroll
                    for (int a = 0; a < 3; a++) { r[a] = p.r[a]; v[a] = p.v[a]; }
                    cell = p.cell;
                    energy = p.energy; tau_req = p.tau_req; tau_ach = 0.0;
                    dirty = true;
                    st = (tau_req == 0.0) ? LS_INTERACT : LS_WALK;
                }
            }
        }
#else
        (void)m_int; (void)cold;
#endif
        // a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232)
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            if (st == LS_WALK) {
                int new_state = TS_WALK;
                bool done = false;
                if (g.countdown == 0) {
                    g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp);
                    if (!geo_in_correct_cell(P, W, r, cell)) { cnt.killed_geo++; new_state = TS_DEAD; done = true; }
                } else g.countdown--;
                if (!done) {
                    double tmin; int im[3];
                    if (!geo_find_wall(P, W, r, v, cell, tmin, im)) { cnt.killed_geo++; new_state = TS_DEAD; done = true; }
                    else {
                        const int lc = ((cell.ic[2] - z0) * BY + (cell.ic[1] - y0)) * BX + (cell.ic[0] - x0);
                        double rho[ND], chi_rho = 0.0;
#pragma unroll
                        for (int d = 0; d < ND; d++) {
#if HYP_TILE_DENS_LDS
                            rho[d] = dens[lc * ND + d];
#else
                            rho[d] = P.density[(((size_t)cell.ic[2] * P.n2 + cell.ic[1]) * P.n1 + cell.ic[0]) * ND + d];
#endif
                            chi_rho += chi[d] * rho[d];
                        }
                        double tau_cell = chi_rho * tmin;
                        double tau_needed = tau_req - tau_ach;
                        cnt.crossings++;
                        if (tau_cell < tau_needed) {
#pragma unroll
                            for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
                            tau_ach += tau_cell;
#pragma unroll
                            for (int d = 0; d < ND; d++) if (rho[d] > 0.0) unsafeAtomicAdd(&accum[lc * ND + d], tmin * kappa[d] * energy);
                            geo_advance(P, r, cell, im);
                            if (geo_escaped(P, cell)) { new_state = TS_DEAD; done = true; }
                            else if (cell.ic[0] < x0 || cell.ic[0] >= x0 + BX || cell.ic[1] < y0 || cell.ic[1] >= y0 + BY ||
                                     cell.ic[2] < z0 || cell.ic[2] >= z0 + BZ) done = true;       // left the brick
                        } else {
                            double tact = tmin * (tau_needed / tau_cell);
#pragma unroll
                            for (int a = 0; a < 3; a++) r[a] = r[a] + tact * v[a];
                            tau_ach += tau_needed;
                            geo_clear_wall(cell);
#pragma unroll
                            for (int d = 0; d < ND; d++) if (rho[d] > 0.0) unsafeAtomicAdd(&accum[lc * ND + d], tact * kappa[d] * energy);
#if HYP_TILE_FUSE
                            st = LS_INTERACT;
#else
                            new_state = TS_INTERACT; done = true;    // tile_prepare does the interaction
#endif
                        }
                    }
                }
                if (done) {
                    HotRec<ND> &H = hot[slot];
                    slot_brick[slot] = new_state == TS_WALK ? brick_of(T, cell.ic) : TILE_NEEDS_PREPARE;
                    if (new_state == TS_DEAD) { H.state = TS_DEAD; finished++; }
                    else {
#pragma unroll
                        for (int a = 0; a < 3; a++) { H.r[a] = r[a]; H.ic[a] = cell.ic[a]; }
                        H.ow = pack_ow(cell.ow);
                        H.tau_ach = tau_ach; H.countdown = g.countdown; H.blk_b = g.blk_b; H.state = new_state;
                        if (dirty) {     // an interaction changed direction, frequency-dependent opacities and tau_req
#pragma unroll
                            for (int a = 0; a < 3; a++) H.v[a] = v[a];
                            H.tau_req = tau_req; H.energy = energy;
#pragma unroll
                            for (int d = 0; d < ND; d++) { H.chi[d] = chi[d]; H.kappa[d] = kappa[d]; }
                        }
                    }
                    st = LS_IDLE;
                }
            }
        }
    }
    __syncthreads();
    // flush the brick's accumulators (replica chosen like in the persistent kernel)
    double *sum = P.sum;
    if (P.n_copies > 1) {
        unsigned c = xcc_id();
        if (P.n_copies > 8) c += 8u * ((blockIdx.x >> 3) % (unsigned)(P.n_copies >> 3));
        sum += (size_t)(c % (unsigned)P.n_copies) * P.copy_stride;
    }
    for (int c = threadIdx.x; c < NC; c += blockDim.x) {
        int lx = c % BX, ly = (c / BX) % BY, lz = c / (BX * BY);
        int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
        if (gx < P.n1 && gy < P.n2 && gz < P.n3) {
            size_t gidx = ((size_t)gz * P.n2 + gy) * P.n1 + gx;
            for (int d = 0; d < ND; d++) {
                double val = accum[c * ND + d];
                if (val != 0.0) unsafeAtomicAdd(&sum[gidx * ND + d], val);
            }
        }
    }
    double cr = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    double ki = wave_sum((double)cnt.killed_int);
    double ni = wave_sum((double)cnt.interactions);
    double nf = wave_sum((double)finished);
    if (__lane_id() == 0) {
        unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], cr);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
        if (ki != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], ki);
        if (ni != 0.0) unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], ni);
        if (nf != 0.0) atomicAdd(&ctl->n_finished, (unsigned long long)nf);
    }
}
