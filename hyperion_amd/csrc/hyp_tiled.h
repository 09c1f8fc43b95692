// hyp_tiled.h -- brick-tiled Lucy iteration for Cartesian grids (gfx950).
//
// Why: one global FP64 atomic per cell crossing caps the straightforward kernel
// at the chip's memory-side atomic rate (2.38e10/s, profiles/r01_atomic_rate_
// ubench.md).  Here the grid is cut into bricks of BX*BY*BZ cells whose density
// and accumulators live in LDS while a workgroup walks, with ds_add_f64, every
// packet that currently sits in that brick; a packet that leaves the brick is
// written back (128-byte "hot" record) and continues in the next generation.
// Per generation and slot pool:  tile_interact (the packets that reached their interaction point) -> tile_emit (new packets
//   into the freed slots) -> tile_sort (counting sort of the walking slots by brick, task list)
//   -> tile_walk (one workgroup per task = up to TASK packets of one brick).
// Per-packet physics and random streams are the ones of hyp_kernels.h, so the
// result equals the persistent kernel's up to FP64 summation order.
#pragma once

#include "hyp_kernels.h"
#include "hyp_defer.h"      // PeelEvent, EmitRec: the imaging iteration on this schedule (IMG kernels below)

enum { TS_DEAD = 0, TS_WALK = 1, TS_INTERACT = 2, TS_DONE = 3, TS_REEMIT = 4 };   // TS_REEMIT: re-absorbed by a source

// bricks (clusters, slabs) per grid on the tiled schedules: tile_sort_kernel keeps two 4-byte tables of them in LDS (2 x 4 B x 18 432 + 2 KB = 146 KB of the
// CU's 160 KB) -- a 512^3 Cartesian grid has 16 384 bricks of 32 x 16 x 16 cells; larger grids run on the persistent kernel
constexpr int HYP_TILE_MAX_BRICKS = 18432;

template <int ND>
struct alignas(64) HotRec {      // what the walk needs (128 B for ND = 1)
    // first 64 bytes: what a brick visit changes -- tile_walk writes back one aligned half cache line
    double r[3];
    double tau_ach;
    int ic[3];
    int ow;                      // (ow0+1) | (ow1+1)<<2 | (ow2+1)<<4
    int countdown;
    unsigned int blk_b;
    int state;
    int pad;
    // fixed between two interactions
    double v[3];
    double tau_req, energy;
    unsigned long long id;
    double chi[ND], kappa[ND];
};

template <int ND>
struct alignas(64) ColdRec {     // only touched at interactions / emission
    Angle a;
    double s[4];
    double nu;
    double albedo[ND];
    double buf_a;
    unsigned int blk_a;
    int have_a, inter, pad;
    // re-absorption by sources (only used when P.any_intersect): see Packet
    double t_src, t_ach;
    int reabs_id, reabs;
    // imaging iteration (IMG kernels): the origin flags of peeloff_photon of problems whose sources can absorb packets (P.any_intersect); the
    // others keep them in the re-absorption block above, which they never use, so that an interaction touches two of the record's three
    // 64-byte lines (configs[3] imaging: 292 against 302 ms).  The packet's event counter rides in `pad`.  (In the alignment padding of the
    // record: 136 + 8 ND bytes of fields in 192.)
    int img_f[5];
};
static_assert(sizeof(ColdRec<1>) == 192 && sizeof(ColdRec<4>) == 192, "ColdRec: three cache-line halves");

template <int ND>
__device__ __forceinline__ void cold_flags_store(ColdRec<ND> &C, const PeelFlags &f, unsigned int peel_seq, bool own_bytes)
{
    int *q = own_bytes ? C.img_f : (int *)&C.t_src;
    q[0] = f.scattered; q[1] = f.reprocessed; q[2] = f.n_scat; q[3] = f.dust_id; q[4] = f.source_id;
    C.pad = (int)peel_seq;
}
template <int ND>
__device__ __forceinline__ void cold_flags_load(const ColdRec<ND> &C, PeelFlags &f, unsigned int &peel_seq, bool own_bytes)
{
    const int *q = own_bytes ? C.img_f : (const int *)&C.t_src;
    f.scattered = q[0]; f.reprocessed = q[1]; f.n_scat = q[2]; f.dust_id = q[3]; f.source_id = q[4];
    peel_seq = (unsigned int)C.pad;
}
template <int ND, int GEOM>
__device__ __forceinline__ void write_peel_event(PeelEvent<ND, GEOM> &E, const Packet<ND, GEOM> &p, const Rng &g, const Angle &a_prev, const double s_prev[4],
                                                 const PeelFlags &f, unsigned int peel_seq, int last, bool last_iso)
{
    E.r[0] = p.r[0]; E.r[1] = p.r[1]; E.r[2] = p.r[2]; E.nu = p.nu; E.energy = p.energy;
    E.a_prev = a_prev;
    E.s_prev[0] = s_prev[0]; E.s_prev[1] = s_prev[1]; E.s_prev[2] = s_prev[2]; E.s_prev[3] = s_prev[3];
#pragma unroll
    for (int d = 0; d < ND; d++) E.chi[d] = p.chi[d];
    E.id = ((unsigned long long)g.id_hi << 32) | g.id_lo;
    E.peel_seq = peel_seq;
    E.code = 1 | (last << 1) | ((last_iso ? 1 : 0) << 3);
    E.f = f;
    E.cell = p.cell;
}

// slot_brick[] values besides a brick index
#define TILE_IDLE (-1)            // slot retired (no packet ids left)
#define TILE_NEEDS_PREPARE (-2)   // the slot is free for a new packet
#define TILE_NEEDS_INTERACT (-3)  // the packet in the slot awaits an interaction
#define TILE_NEEDS_REEMIT (-4)    // the packet was re-absorbed by a source and awaits its re-emission
constexpr int HYP_PREP_CHUNK = 2048;
// build-time shape of tile_walk_kernel (tools/variants.py sweeps these)
constexpr int HYP_TILE_WG = 1024;         // threads per workgroup (one workgroup per task; with 32 x 16 x 16 bricks one workgroup per CU)
constexpr int HYP_TILE_OCC = 4;           // waves per SIMD the walk kernel's registers are budgeted for (16 waves per CU)
#ifndef HYP_TILE_SERVICE_N
#define HYP_TILE_SERVICE_N 16
#endif
#ifndef HYP_TILE_STEPS_N
#define HYP_TILE_STEPS_N 8      // (round 6, 25e6-slot pool: 3: 201.0, 4: 198.8 - 201.2, 6: 197.7, 8: 196.0 - 197.3 ms; 24 service lanes 197.9, with 6 steps 198.7)
#endif
constexpr int HYP_TILE_SERVICE = HYP_TILE_SERVICE_N;      // lanes that must wait (visit finished / idle) before a wave runs its service phase
constexpr int HYP_TILE_STEPS = HYP_TILE_STEPS_N;         // cell steps between two scheduling decisions of a wave
#define TILE_DEPOSIT(p, v) do { if (!T.imaging) unsafeAtomicAdd(p, v); } while (0)      // (T: the walk kernel's TileGeom)
constexpr int HYP_TILE_MAX_POOLS = 4;
struct TileCtl {
    unsigned long long next_id, end_id, n_finished;
    unsigned long long first_id;                  // first packet id of the launch (imaging: index of a packet's EmitRec)
    unsigned int n_tasks[HYP_TILE_MAX_POOLS];     // per slot pool
    // split schedule: slots that need an interaction but sit in no task's list (a packet that drew a zero optical
    // depth); two lists per pool, filled and emptied in alternate generations
    unsigned int n_extra[HYP_TILE_MAX_POOLS][2];
    // slots that wait for an interaction / are free, listed by the walk of the previous generation (and by tile_interact for
    // the packets it ends): counters by generation parity, see tile_walk_publish_lists
    unsigned int n_gil[HYP_TILE_MAX_POOLS][2], n_gdl[HYP_TILE_MAX_POOLS][2];
    unsigned int n_live, live_cursor;             // the drain: slots that still hold a packet (TileGeom::drain_list), and how many of them have been handed out
    unsigned long long dbg[40];                   // debug builds only
};

struct TileGeom {
    int bx, by, bz;              // brick size in cells
    int nbx, nby, nbz, n_bricks;
    int n_slots, task_size;
    uint32_t iter_tag;
    int pool;                    // which slot pool (and stream) this launch belongs to
    int park;                    // tile_walk: park the last packets of a wave once this few lanes still walk
    int gen;                     // generation number (split schedule: which of the two extra lists is read)
    int imaging;                 // 1: the imaging iteration on this schedule -- walks deposit nothing (grid_integrate_noenergy)
    const int *drain_list;       // tile_drain: the slots that still hold a packet (tile_live_kernel), TileCtl::n_live of them
    int presort;                 // 1: the walk writes slot | kind << 30 into the interaction lists (HotRec::pad; one species, Cartesian walk)
    int vsplit;                  // 3 (2): every brick is three (two) entries of the sort -- 3 b for packets that have not interacted yet, 3 b + 1 for flights that start outwards, 3 b + 2 inwards
};                               //    (spherical grids: waves of one kind, hyp_ptile.h); 0 / 1: one entry per brick

// per task of the current generation: how many of its packets ended the visit waiting for an interaction
// (or a re-emission by a source) and how many slots it left free
struct TileCount { int n_int, n_dead; };
constexpr int HYP_TILE_EXTRA = 1024;       // capacity of each per-pool list behind TileCtl::n_extra; layout [list 0][list 1][freed slots]

struct TileTask { int brick, start, len, pad; };

__device__ __forceinline__ int pack_ow(const int ow[3]) { return (ow[0] + 1) | ((ow[1] + 1) << 2) | ((ow[2] + 1) << 4); }
__device__ __forceinline__ void unpack_ow(int w, int ow[3]) { ow[0] = (w & 3) - 1; ow[1] = ((w >> 2) & 3) - 1; ow[2] = ((w >> 4) & 3) - 1; }

__device__ __forceinline__ int brick_of(const TileGeom &T, const int ic[3])
{
    return ((ic[2] / T.bz) * T.nby + (ic[1] / T.by)) * T.nbx + (ic[0] / T.bx);
}

// How a packet's cell is kept in its slot record, and which brick (Cartesian) or cluster of cells (Voronoi, hyp_vtile.h)
// it belongs to.  The interaction / emission / drain kernels below are shared by the geometries through this.
template <int GEOM> struct TileCellIO;
template <> struct TileCellIO<GEOM_CAR> {
    template <int ND> static __device__ __forceinline__ void load(const DProblem &P, const HotRec<ND> &H, Cell<GEOM_CAR> &c)
    {
#pragma unroll
        for (int a = 0; a < 3; a++) c.ic[a] = H.ic[a];
        unpack_ow(H.ow, c.ow);
    }
    template <int ND> static __device__ __forceinline__ void store(const DProblem &P, HotRec<ND> &H, const Cell<GEOM_CAR> &c)
    {
#pragma unroll
        for (int a = 0; a < 3; a++) H.ic[a] = c.ic[a];
        H.ow = pack_ow(c.ow);
    }
    static __device__ __forceinline__ int brick(const DProblem &P, const TileGeom &T, const Cell<GEOM_CAR> &c) { return brick_of(T, c.ic); }
};
// Voronoi: ic = (cell, the cell the packet came from or -1, cluster << 16 | index of the cell in its cluster); ow = position of the
// wall the packet came through in the cell's wall list (VT_NO_BACK: none; VT_FIND_BACK: to be looked up from ic[1])
#define VT_FIND_BACK 254
template <> struct TileCellIO<GEOM_VOR> {
    template <int ND> static __device__ __forceinline__ void load(const DProblem &P, const HotRec<ND> &H, Cell<GEOM_VOR> &c)
    {
        c.id = H.ic[0]; c.ow[0] = 0; c.ow[1] = -(H.ic[1] + 1); c.ow[2] = 0;
    }
    template <int ND> static __device__ __forceinline__ void store(const DProblem &P, HotRec<ND> &H, const Cell<GEOM_VOR> &c)
    {
        const int prev = -c.ow[1] - 1;
        H.ic[0] = c.id; H.ic[1] = prev; H.ic[2] = P.vt_cluster[c.id]; H.ow = prev < 0 ? VT_NO_BACK : VT_FIND_BACK;
    }
    // A packet that an external source emits ON a face of the box (emit_from_extern_box, source_type.f90:822-907: the coordinate IS
    // the face's) moves away from it, and the reference's wall search never looks at that face (`ahead`, grid_geometry_voronoi.f90:
    // 362-371: the sign of one component of v).  The walk's FP32 filter leaves out ONE wall per step, the one the packet came
    // through: for such a packet that wall is the face -- its position in the cell's list goes into the record (round 6; before, the
    // filter carried the reference's rule for faces on every wall of every step).  Anything else keeps VT_NO_BACK; a packet on a
    // face without this (never seen: interaction points are interior) costs one pass of the reference's loop, not a wrong wall.
    template <int ND> static __device__ __forceinline__ void mark_face_behind(const DProblem &P, HotRec<ND> &H, const double r[3], const double v[3])
    {
        if (H.ow != VT_NO_BACK) return;
        int iw = -1;
#pragma unroll
        for (int a = 2; a >= 0; a--) {
            if (r[a] == P.vor_box[2 * a] && !(v[a] < 0.0)) iw = 2 * a;
            if (r[a] == P.vor_box[2 * a + 1] && !(v[a] > 0.0)) iw = 2 * a + 1;
        }
        if (iw < 0) return;
        const int k0 = P.vor_idx[H.ic[0]], k1 = P.vor_idx[H.ic[0] + 1];
        for (int k = k0; k < k1; k++)
            if (P.vor_neigh[k] == -(iw + 1) && k - k0 < VT_FIND_BACK) { H.ow = k - k0; break; }
    }
    static __device__ __forceinline__ int brick(const DProblem &P, const TileGeom &T, const Cell<GEOM_VOR> &c) { return P.vt_cluster[c.id] >> 16; }
};
// Polar grids (hyp_ptile.h): cells are numbered (i1, i2, i3) like on Cartesian grids; the spherical grid's `radial` (the sign of
// r.v at the start of the integration, Cell<GEOM_SPH>) rides in bit 6 of ow
template <> struct TileCellIO<GEOM_SPH> {
    template <int ND> static __device__ __forceinline__ void load(const DProblem &P, const HotRec<ND> &H, Cell<GEOM_SPH> &c)
    {
#pragma unroll
        for (int a = 0; a < 3; a++) c.ic[a] = H.ic[a];
        unpack_ow(H.ow & 63, c.ow);
        c.radial = (H.ow >> 6) & 1;
    }
    template <int ND> static __device__ __forceinline__ void store(const DProblem &P, HotRec<ND> &H, const Cell<GEOM_SPH> &c)
    {
#pragma unroll
        for (int a = 0; a < 3; a++) H.ic[a] = c.ic[a];
        H.ow = pack_ow(c.ow) | ((c.radial ? 1 : 0) << 6);
    }
    static __device__ __forceinline__ int brick(const DProblem &P, const TileGeom &T, const Cell<GEOM_SPH> &c) { return brick_of(T, c.ic); }
};
template <> struct TileCellIO<GEOM_CYL> {
    template <int ND> static __device__ __forceinline__ void load(const DProblem &P, const HotRec<ND> &H, Cell<GEOM_CYL> &c)
    {
#pragma unroll
        for (int a = 0; a < 3; a++) c.ic[a] = H.ic[a];
        unpack_ow(H.ow, c.ow);
    }
    template <int ND> static __device__ __forceinline__ void store(const DProblem &P, HotRec<ND> &H, const Cell<GEOM_CYL> &c)
    {
#pragma unroll
        for (int a = 0; a < 3; a++) H.ic[a] = c.ic[a];
        H.ow = pack_ow(c.ow);
    }
    static __device__ __forceinline__ int brick(const DProblem &P, const TileGeom &T, const Cell<GEOM_CYL> &c) { return brick_of(T, c.ic); }
};
// AMR (hyp_atile.h): ic = (cell, grid, brick); the position in the grid follows from the cell id
__device__ __forceinline__ int amr_brick_of(const DProblem &P, int grid, const int i[3])
{
    return P.at_grid_c0[grid] + ((i[2] / P.at_b[2]) * P.at_grid_nb[2 * grid + 1] + i[1] / P.at_b[1]) * P.at_grid_nb[2 * grid] + i[0] / P.at_b[0];
}
template <> struct TileCellIO<GEOM_AMR> {
    template <int ND> static __device__ __forceinline__ void load(const DProblem &P, const HotRec<ND> &H, Cell<GEOM_AMR> &c)
    {
        c.id = H.ic[0]; c.grid = H.ic[1];
        const AmrGrid &g = P.amr_grids[c.grid];
        const int local = c.id - (int)g.start;
        c.i[0] = local % g.n[0]; c.i[1] = (local / g.n[0]) % g.n[1]; c.i[2] = local / (g.n[0] * g.n[1]);
        unpack_ow(H.ow, c.ow);
    }
    template <int ND> static __device__ __forceinline__ void store(const DProblem &P, HotRec<ND> &H, const Cell<GEOM_AMR> &c)
    {
        H.ic[0] = c.id; H.ic[1] = c.grid; H.ic[2] = amr_brick_of(P, c.grid, c.i); H.ow = pack_ow(c.ow);
    }
    static __device__ __forceinline__ int brick(const DProblem &P, const TileGeom &T, const Cell<GEOM_AMR> &c) { return amr_brick_of(P, c.grid, c.i); }
};
// Octree (hyp_otile.h): ic = (leaf cell, -, cluster of the leaf), ow as on Cartesian grids; the rest of the cell record
// (centre, level, parent, sub-cell) is read back from the cell table
template <> struct TileCellIO<GEOM_OCT> {
    template <int ND> static __device__ __forceinline__ void load(const DProblem &P, const HotRec<ND> &H, Cell<GEOM_OCT> &c)
    {
        oct_load(P, H.ic[0], c);
        unpack_ow(H.ow, c.ow);
    }
    template <int ND> static __device__ __forceinline__ void store(const DProblem &P, HotRec<ND> &H, const Cell<GEOM_OCT> &c)
    {
        H.ic[0] = c.id; H.ic[1] = 0; H.ic[2] = P.ot_cluster[c.id]; H.ow = pack_ow(c.ow);
    }
    static __device__ __forceinline__ int brick(const DProblem &P, const TileGeom &T, const Cell<GEOM_OCT> &c) { return P.ot_cluster[c.id]; }
};

constexpr int HYP_PREP_WAVES = 2;

// ---------------------------------------------------------------------------
// The lists between the kernels: every task of tile_walk collects the slots whose packets wait for an
// interaction (or for re-emission by a source that absorbed them) and the slots it freed, appends
// both to the pool's two lists with ONE reservation per task (tile_walk_publish_lists), and adds the
// packets that move on to the brick histogram itself.  tile_interact and tile_emit work through the
// lists in full chunks -- full waves of one kind of work, no scan over slot_brick[], no nearly
// empty per-task workgroups, and each kernel only carries the registers of its own phase -- and
// add the bricks of the packets they hand to the next walk to the histogram, so that the counting
// pass of the sort is gone.
//
// Lists: ilist / dlist hold n_slots staging entries (a task's own range [start, start + len)) followed
// by n_slots entries of the pool-wide list.  Counters TileCtl::n_gil / n_gdl by parity: the walk of
// generation g adds to [g & 1]; tile_interact / tile_emit of generation g + 1 read [g & 1] (and
// tile_interact appends the slots of the packets it ends to the free list it is about to hand to
// tile_emit); tile_scan of generation g, which runs between them and the walk, clears [g & 1].
// ---------------------------------------------------------------------------

// The tallies of a workgroup go to the global tail through LDS: one set of (same-address) global atomics per
// workgroup instead of one per wave.  red[] must be zero before the first call; every thread of the block calls.
#define TILE_RED_N 6
__device__ __forceinline__ void block_tally_flush(const DProblem &P, TileCtl *__restrict__ ctl, double *red, const Counters &cnt,
                                                  unsigned int finished)
{
    const double e = wave_sum(cnt.energy_current);
    const double c = wave_sum((double)cnt.crossings);
    const double kg = wave_sum((double)cnt.killed_geo);
    const double ki = wave_sum((double)cnt.killed_int);
    const double ni = wave_sum((double)cnt.interactions);
    const double nf = wave_sum((double)finished);
    if (__lane_id() == 0) {
        if (e != 0.0) atomicAdd(&red[0], e);
        if (c != 0.0) atomicAdd(&red[1], c);
        if (kg != 0.0) atomicAdd(&red[2], kg);
        if (ki != 0.0) atomicAdd(&red[3], ki);
        if (ni != 0.0) atomicAdd(&red[4], ni);
        if (nf != 0.0) atomicAdd(&red[5], nf);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (red[0] != 0.0) unsafeAtomicAdd(&P.tail[TAIL_ENERGY], red[0]);
        if (red[1] != 0.0) unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], red[1]);
        if (red[2] != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], red[2]);
        if (red[3] != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], red[3]);
        if (red[4] != 0.0) unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], red[4]);
        if (red[5] != 0.0) atomicAdd(&ctl->n_finished, (unsigned long long)red[5]);
    }
}

// Brick histogram contributions of a workgroup, collected in a small LDS table first: the packets that one workgroup of
// tile_interact / tile_emit hands to the next walk sit in one or two bricks (the task's own; the source's), and many
// workgroups would otherwise hammer the same counts[] entries.  cache_b[] = brick or -1, cache_n[] = count.
#define TILE_BCACHE 8
__device__ __forceinline__ void count_bricks(unsigned int *__restrict__ counts, int *cache_b, unsigned int *cache_n, bool on, int brick)
{
    unsigned long long m = __ballot(on);
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const int b = __shfl(brick, leader, 64);
        const unsigned long long same = __ballot(on && brick == b) & m;
        if ((int)__lane_id() == leader) {
            const unsigned int n = (unsigned int)__popcll(same);
            bool done = false;
            for (int i = 0; i < TILE_BCACHE && !done; i++) {
                const int old = atomicCAS(&cache_b[i], -1, b);
                if (old == -1 || old == b) { atomicAdd(&cache_n[i], n); done = true; }
            }
            if (!done) atomicAdd(&counts[b], n);
        }
        m &= ~same;
    }
}
__device__ __forceinline__ void flush_bricks(unsigned int *__restrict__ counts, const int *cache_b, const unsigned int *cache_n)
{
    if (threadIdx.x < TILE_BCACHE && cache_b[threadIdx.x] >= 0 && cache_n[threadIdx.x]) atomicAdd(&counts[cache_b[threadIdx.x]], cache_n[threadIdx.x]);
}

template <int ND, int GEOM>
__device__ __forceinline__ void store_records(const DProblem &P, HotRec<ND> &H, ColdRec<ND> &C, const Packet<ND, GEOM> &p, const Rng &g,
                                              unsigned long long id, int state)
{
#pragma unroll
    for (int a = 0; a < 3; a++) { H.r[a] = p.r[a]; H.v[a] = p.v[a]; }
    TileCellIO<GEOM>::store(P, H, p.cell);
    if constexpr (GEOM == GEOM_VOR) TileCellIO<GEOM_VOR>::mark_face_behind(P, H, p.r, p.v);
    H.tau_req = p.tau_req; H.tau_ach = p.tau_ach; H.energy = p.energy;
#pragma unroll
    for (int d = 0; d < ND; d++) { H.chi[d] = p.chi[d]; H.kappa[d] = p.kappa[d]; C.albedo[d] = p.albedo[d]; }
    H.id = id; H.countdown = g.countdown; H.blk_b = g.blk_b; H.state = state;
    if (ND == 1) {
        // the kind of the packet's NEXT interaction (absorption: the first number of its stream exceeds the albedo) is known already --
        // the walk draws from the other stream --: the walk passes it on with the slot (TileGeom::presort), and tile_interact orders its
        // chunk by kind without reading the records a first time
        Rng g2 = g;
        H.pad = rng_uniform(g2) > p.albedo[0] ? 1 : 0;
    }
    C.a = p.a; C.s[0] = p.s[0]; C.s[1] = p.s[1]; C.s[2] = p.s[2]; C.s[3] = p.s[3];
    C.nu = p.nu; C.buf_a = g.buf_a; C.blk_a = g.blk_a; C.have_a = g.have_a; C.inter = p.inter;
    if (P.any_intersect) { C.t_src = p.t_src; C.t_ach = p.t_ach; C.reabs_id = p.reabs_id; C.reabs = p.reabs; }
}

// generation 0 of the split schedule: every slot is free and listed for tile_emit
static __global__ __launch_bounds__(256) void tile_init_kernel(TileGeom T, TileCtl *__restrict__ ctl, TileTask *__restrict__ tasks,
                                                        TileCount *__restrict__ tcount, int *__restrict__ dlist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T.n_slots) dlist[T.n_slots + i] = i;
    if (i == 0) {
        ctl->n_tasks[T.pool] = 0; ctl->n_extra[T.pool][0] = ctl->n_extra[T.pool][1] = 0;
        ctl->n_gil[T.pool][0] = ctl->n_gil[T.pool][1] = 0;
        ctl->n_gdl[T.pool][0] = 0; ctl->n_gdl[T.pool][1] = (unsigned int)T.n_slots;      // read by the emission of generation 0
    }
}

#ifndef HYP_INTERACT_WAVES_N
#define HYP_INTERACT_WAVES_N 2
#endif
constexpr int HYP_INTERACT_WAVES = HYP_INTERACT_WAVES_N;
constexpr int HYP_EMIT_WAVES = 2;
constexpr int HYP_EMIT_WAVES_SIMPLE = 3;     // 167 VGPRs, nothing spilled (4: 128 + 74 spilled; configs[1] 254.0-254.2 -> 251.5-252.5 ms; 2: 254.0)
// end of a walk task: its two lists (staged in the task's own range) go to the pool-wide lists, one reservation each
__device__ __forceinline__ void tile_walk_publish_lists(const TileGeom &T, TileCtl *__restrict__ ctl, const TileTask &tk, int *__restrict__ ilist,
                                                        int *__restrict__ dlist, int n_int, int n_dead, int *base /* 2 ints of LDS */)
{
    if (threadIdx.x == 0) base[0] = n_int ? (int)atomicAdd(&ctl->n_gil[T.pool][T.gen & 1], (unsigned int)n_int) : 0;
    if (threadIdx.x == 64) base[1] = n_dead ? (int)atomicAdd(&ctl->n_gdl[T.pool][T.gen & 1], (unsigned int)n_dead) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_int; i += blockDim.x) ilist[T.n_slots + base[0] + i] = ilist[tk.start + i];
    for (int i = threadIdx.x; i < n_dead; i += blockDim.x) dlist[T.n_slots + base[1] + i] = dlist[tk.start + i];
}
constexpr int HYP_INTERACT_CHUNK = 1024;     // list entries that tile_interact orders by kind at a time

// REABS: the problem has sources that can absorb packets (P.any_intersect), so slots may wait for a re-emission;
// MRW: the modified random walk is on (P.mrw).  Without them that code stays out of this kernel.
// IMG: the imaging iteration (do_final) on this schedule -- the interaction deposits nothing and leaves one PeelEvent in the
// event buffer B (peeloff_photon is made later by peel_kernel, hyp_defer.h); a workgroup reserves one event slot per entry of
// its chunk with ONE atomic and marks the slots of entries that peel nothing as empty.
template <int ND, bool REABS, bool MRW, int GEOM, bool IMG = false>
__global__ __launch_bounds__(256, HYP_INTERACT_WAVES) void tile_interact_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
        void *__restrict__ hot_v, void *__restrict__ cold_v, int *__restrict__ slot_brick,
        const TileTask *__restrict__ tasks, const int *__restrict__ ilist, int *__restrict__ dlist, TileCount *__restrict__ tcount,
        unsigned int *__restrict__ counts, int *__restrict__ extra, DeferBuf B)
{
    extern __shared__ double lds[];
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    constexpr int CH = HYP_INTERACT_CHUNK;
    const DProblem &P = *Pp;
    // the last workgroup takes the slots that are in neither list (TileCtl::n_extra); the others one chunk of the list each
    const bool is_extra = blockIdx.x == gridDim.x - 1;
    const int par = T.gen & 1, rp = (T.gen + 1) & 1;
    int n_int = 0;
    if (is_extra) n_int = (int)min(ctl->n_extra[T.pool][par], (unsigned int)HYP_TILE_EXTRA);
    else n_int = (int)ctl->n_gil[T.pool][rp] - (int)blockIdx.x * CH;
    if (n_int <= 0) return;
    if (n_int > CH) n_int = CH;
    const int *list = is_extra ? extra + par * HYP_TILE_EXTRA : ilist + T.n_slots + (size_t)blockIdx.x * CH;
    int *extra_next = extra + (par ^ 1) * HYP_TILE_EXTRA;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    __shared__ int sorted[CH];
    __shared__ int n_dead_x, n_abs, n_oth, dead_base;
    __shared__ unsigned long long ev_base;
    if (IMG && threadIdx.x == 0) {
        ev_base = atomicAdd(&B.ctl->reserved, (unsigned long long)(n_int + 64));      // (+ 64: the two kinds start on wave boundaries)
        if (ev_base + (unsigned long long)(n_int + 64) > B.cap) { raise_error(P, ERR_INTERNAL, (double)ev_base, (double)B.cap, 1.0); ev_base = ~0ull; }
    }
    PeelEvent<ND, GEOM> *__restrict__ ev = (PeelEvent<ND, GEOM> *)B.events;
    __shared__ int dead_l[CH];
    __shared__ double red[TILE_RED_N];
    __shared__ int cache_b[TILE_BCACHE];
    __shared__ unsigned int cache_n[TILE_BCACHE];
    if (threadIdx.x < TILE_RED_N) red[threadIdx.x] = 0.0;
    if (threadIdx.x < TILE_BCACHE) { cache_b[threadIdx.x] = -1; cache_n[threadIdx.x] = 0; }
    if (threadIdx.x == 0) n_dead_x = 0;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    const int nd = ndust<ND>(P);
    {
        const int c0 = 0;
        // Absorption + re-emission and scattering are two long, different code paths (dust_interact.f90:49-70); which one
        // a packet takes is decided by the first one or two numbers of its random stream.  Draw them ahead on a copy of
        // the stream and order the chunk by the outcome, so that the waves below run one path each.
        __syncthreads();
        if (threadIdx.x == 0) { n_abs = 0; n_oth = 0; }
        __syncthreads();
        const int n_chunk = min(CH, n_int - c0);
        for (int k = threadIdx.x; k < n_chunk; k += (int)blockDim.x) {
            const int entry = list[c0 + k];
            const int slot = entry & 0x3fffffff;
            if (T.presort) {        // the kind came with the slot: no look at the records
                if ((entry >> 30) & 1) sorted[atomicAdd(&n_abs, 1)] = slot;
                else sorted[CH - 1 - atomicAdd(&n_oth, 1)] = slot;
                continue;
            }
            const HotRec<ND> &H = hot[slot];
            const ColdRec<ND> &C = cold[slot];
            bool absorb = false;
            // everything the peek needs in ONE batch of loads: behind the short circuit of the test below they came as four
            // dependent batches (state; inter; stream state; id), a memory round trip each
            int h_state = H.state, c_inter = C.inter, c_have_a = C.have_a;
            unsigned int c_blk_a = C.blk_a;
            unsigned long long id = H.id;
            double c_buf_a = C.buf_a, albedo = C.albedo[0];
            asm volatile("" : "+v"(h_state), "+v"(c_inter), "+v"(c_have_a), "+v"(c_blk_a), "+v"(id), "+v"(c_buf_a), "+v"(albedo));
            if ((h_state == TS_INTERACT) & ((long long)c_inter != P.n_inter_max + 1)) {
                Rng g2;
                g2.key0 = P.seed_key; g2.key1 = T.iter_tag; g2.id_lo = (uint32_t)id; g2.id_hi = (uint32_t)(id >> 32);
                g2.blk_a = c_blk_a; g2.buf_a = c_buf_a; g2.have_a = c_have_a; g2.blk_b = 0; g2.countdown = 0;
                if (ND > 1 && nd > 1) {        // select_dust_chi_rho, as in interact()
                    Cell<GEOM> hc; TileCellIO<GEOM>::load(P, H, hc);
                    const size_t base = geo_index(P, hc) * (size_t)nd;
                    double cdf[ND], c = 0.0;
#pragma unroll
                    for (int d = 0; d < ND; d++) { if (d < nd) c += H.chi[d] * P.density[base + d]; cdf[d] = c; }
                    const double xi = rng_uniform(g2);
                    int idd = nd - 1; bool found = false;
#pragma unroll
                    for (int d = 0; d < ND; d++) if (d < nd && !found && d < nd - 1 && xi < cdf[d] / c) { idd = d; found = true; }
                    albedo = 0.0;
#pragma unroll
                    for (int d = 0; d < ND; d++) if (d == idd) albedo = C.albedo[d];
                }
                absorb = rng_uniform(g2) > albedo;
            }
            if (absorb) sorted[atomicAdd(&n_abs, 1)] = slot;
            else sorted[CH - 1 - atomicAdd(&n_oth, 1)] = slot;
        }
        __syncthreads();
        const int na = n_abs, oth0 = (na + 63) & ~63, nl = oth0 + n_oth;      // the other kind starts on a wave boundary
    for (int k0 = 0; k0 < nl; k0 += (int)blockDim.x) {
        const int k = k0 + (int)threadIdx.x;
        const bool valid = k < na || (k >= oth0 && k < nl);
        const int slot = !valid ? 0 : k < na ? sorted[k] : sorted[CH - 1 - (k - oth0)];
        int state = TS_DONE;
        Packet<ND, GEOM> p;
        Rng g;
        unsigned long long id = 0;
        PeelFlags f; f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = 0;
        unsigned int peel_seq = 0;
        p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
        if (valid) {
            const HotRec<ND> &H = hot[slot];
            const ColdRec<ND> &C = cold[slot];
            state = H.state;
#pragma unroll
            for (int a = 0; a < 3; a++) { p.r[a] = H.r[a]; p.v[a] = H.v[a]; }
            TileCellIO<GEOM>::load(P, H, p.cell);
            p.a = C.a;
            p.s[0] = C.s[0]; p.s[1] = C.s[1]; p.s[2] = C.s[2]; p.s[3] = C.s[3];
            p.nu = C.nu; p.energy = H.energy; p.tau_req = H.tau_req; p.tau_ach = H.tau_ach;
#pragma unroll
            for (int d = 0; d < ND; d++) { p.chi[d] = H.chi[d]; p.kappa[d] = H.kappa[d]; p.albedo[d] = C.albedo[d]; }
            p.inter = C.inter;
            id = H.id;
            g.key0 = P.seed_key; g.key1 = T.iter_tag; g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
            g.blk_a = C.blk_a; g.blk_b = H.blk_b; g.buf_a = C.buf_a; g.have_a = C.have_a; g.countdown = H.countdown;
            if (REABS) { p.reabs = C.reabs; p.reabs_id = C.reabs_id; }
            if (IMG) cold_flags_load(C, f, peel_seq, P.any_intersect != 0);
        }
        const Angle a_prev = p.a;
        const double s_prev[4] = {p.s[0], p.s[1], p.s[2], p.s[3]};
        int last = LAST_SR; bool last_iso = true, do_peel = false;
        if (REABS && state == TS_REEMIT) {
            // iter_lucy.f90:155-185: re-emission from the source that absorbed the packet
            const int inter = p.inter, reabs = p.reabs, rid = p.reabs_id;
            const double e = p.energy;
            if ((long long)reabs == P.n_reabs_max) { cnt.killed_int++; state = TS_DEAD; finished++; }
            else {
                int source_id; Angle src_normal;
                bool ok = emit_packet<ND, GEOM>(P, W, p, g, cnt, source_id, src_normal, rid, e);
                p.inter = inter; p.reabs = reabs + 1;
                if (IMG && ok) {
                    // the re-emission is peeled off like a scattering, also with peel_scattered_only (iter_final.f90:213-243), weighed with the angle to the
                    // surface normal (a_prev of the event), before the packet may turn out to have left the grid: final_defer_kernel<.., GEN>, peel == 3
                    f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
                    if (ev_base != ~0ull) write_peel_event<ND, GEOM>(ev[ev_base + (unsigned long long)k], p, g, src_normal, s_prev, f, peel_seq, LAST_SR, false);
                    peel_seq++; do_peel = true;
                }
                if (!ok || geo_escaped(P, p.cell)) { state = TS_DEAD; finished++; }
                else {
                    p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                    begin_integrate(P, p);
                    state = (p.tau_req == 0.0) ? TS_INTERACT : TS_WALK;
                }
            }
        } else if (state == TS_INTERACT) {
            p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
            if ((long long)p.inter == P.n_inter_max + 1) { cnt.killed_int++; state = TS_DEAD; finished++; }
            else {
                int scattered, dust_id;
                bool ok = interact<ND, GEOM>(P, p, g, cnt, scattered, dust_id);
                bool killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered);
                if (IMG) {          // iter_final.f90:245-268: the origin flags of the peel-off
                    f.dust_id = dust_id;
                    if (scattered) { f.scattered = 1; f.n_scat++; last = LAST_DS; last_iso = false; }
                    else { f.scattered = 0; f.reprocessed = 1; last = LAST_DE; last_iso = true; }
                }
                if (killed) { state = TS_DEAD; finished++; }
                else if (MRW && mrw_loop_lucy<ND, GEOM>(P, W, p, g, P.sum, cnt)) { state = TS_DEAD; finished++; }
                else {
                    p.inter++;
                    if (IMG) do_peel = !P.peel_scattered_only || last == LAST_DS;
                    if (IMG && do_peel) { if (ev_base != ~0ull) write_peel_event<ND, GEOM>(ev[ev_base + (unsigned long long)k], p, g, a_prev, s_prev, f, peel_seq, last, last_iso); peel_seq++; }
                    p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                    begin_integrate(P, p);
                    state = (p.tau_req == 0.0) ? TS_INTERACT : TS_WALK;
                }
            }
        }
        int brick = 0;
        if (IMG && !do_peel && k < n_int + 64 && ev_base != ~0ull) ev[ev_base + (unsigned long long)k].code = 0;       // nothing peeled here: an empty slot
        if (valid) {
            if (state == TS_WALK || state == TS_INTERACT) {
                store_records<ND, GEOM>(P, hot[slot], cold[slot], p, g, id, state);
                if (IMG) cold_flags_store(cold[slot], f, peel_seq, P.any_intersect != 0);
                if (state == TS_WALK) { brick = TileCellIO<GEOM>::brick(P, T, p.cell); if (T.vsplit > 1) {
                        int kind = 1;      // spherical grids: 1 = the integration starts outwards (find_wall skips the inner sphere for the whole flight), 2 = inwards
                        if constexpr (GEOM == GEOM_SPH) kind = (T.vsplit > 2 && !p.cell.radial) ? 2 : 1;
                        brick = brick * T.vsplit + kind;
                    }
                    slot_brick[slot] = brick; }
                else {
                    // zero optical depth drawn (probability 2^-53): the next generation's extra workgroup interacts again
                    slot_brick[slot] = TILE_NEEDS_INTERACT;
                    const unsigned int j = atomicAdd(&ctl->n_extra[T.pool][par ^ 1], 1u);
                    if (j < HYP_TILE_EXTRA) extra_next[j] = slot; else raise_error(P, ERR_INTERNAL, (double)j, 0.0, 0.0);
                }
            } else {        // the packet ended here: the slot is free for tile_emit
                hot[slot].state = TS_DEAD; slot_brick[slot] = TILE_NEEDS_PREPARE;
                dead_l[atomicAdd(&n_dead_x, 1)] = slot;
            }
        }
        count_bricks(counts, cache_b, cache_n, valid && state == TS_WALK, brick);
    }
        if (IMG && ev_base != ~0ull)        // reserved slots beyond the last pass of the loop
            for (int k = ((nl + (int)blockDim.x - 1) / (int)blockDim.x) * (int)blockDim.x + (int)threadIdx.x; k < n_int + 64; k += (int)blockDim.x)
                ev[ev_base + (unsigned long long)k].code = 0;
    }
    __syncthreads();
    if (is_extra && threadIdx.x == 0) ctl->n_extra[T.pool][par] = 0;
    // the slots of the packets that ended here join the free list tile_emit is about to work through
    if (threadIdx.x == 0) dead_base = n_dead_x ? (int)atomicAdd(&ctl->n_gdl[T.pool][rp], (unsigned int)n_dead_x) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_dead_x; i += (int)blockDim.x) dlist[T.n_slots + dead_base + i] = dead_l[i];
    flush_bricks(counts, cache_b, cache_n);
    block_tally_flush(P, ctl, red, cnt, finished);
}

// SIMPLE: every source is a point source with a tabulated or blackbody spectrum (emit_packet<.., SIMPLE>): the other
// emitters stay out of the kernel and its register budget allows twice the waves
// IMG: the imaging iteration -- a packet's emission, the escape walk of the forced first interaction and its first optical depth
// come from the record ff_walk_kernel left (B.ff, hyp_defer.h) when forced first interaction is on, and the emission leaves a
// PeelEvent (one slot per free-list entry reserved by the workgroup; the rare second emission into a slot reserves its own).
template <int ND, int GEOM, int SIMPLE /* emit_packet's CLASS */, bool IMG = false>
__global__ __launch_bounds__(256, SIMPLE ? HYP_EMIT_WAVES_SIMPLE : HYP_EMIT_WAVES) void tile_emit_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
        void *__restrict__ hot_v, void *__restrict__ cold_v, int *__restrict__ slot_brick,
        const TileTask *__restrict__ tasks, const int *__restrict__ dlist, const TileCount *__restrict__ tcount,
        unsigned int *__restrict__ counts, int *__restrict__ extra, DeferBuf B)
{
    extern __shared__ double lds[];
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    const DProblem &P = *Pp;
    // one chunk of 256 entries of the pool's free list per workgroup
    const int par = T.gen & 1, rp = (T.gen + 1) & 1;
    const int part = 0;
    int n_dead = (int)ctl->n_gdl[T.pool][rp] - (int)blockIdx.x * 256;
    if (n_dead <= 0) return;
    if (n_dead > 256) n_dead = 256;
    const int *list = dlist + T.n_slots + (size_t)blockIdx.x * 256;
    int *extra_next = extra + (par ^ 1) * HYP_TILE_EXTRA;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    __shared__ double red[TILE_RED_N];
    __shared__ int cache_b[TILE_BCACHE];
    __shared__ unsigned int cache_n[TILE_BCACHE];
    __shared__ unsigned long long id_base, ev_base;
    PeelEvent<ND, GEOM> *__restrict__ ev = (PeelEvent<ND, GEOM> *)B.events;
    if (IMG && threadIdx.x == 64) {
        ev_base = atomicAdd(&B.ctl->reserved, (unsigned long long)n_dead);
        if (ev_base + (unsigned long long)n_dead > B.cap) { raise_error(P, ERR_INTERNAL, (double)ev_base, (double)B.cap, 2.0); ev_base = ~0ull; }
    }
    if (threadIdx.x < TILE_RED_N) red[threadIdx.x] = 0.0;
    if (threadIdx.x < TILE_BCACHE) { cache_b[threadIdx.x] = -1; cache_n[threadIdx.x] = 0; }
    // one trip to the packet-id dispenser for all the entries of this workgroup (it is one address for the whole
    // chip); only packets that end at once (emitted outside the grid) go back to it, a wave at a time
    if (threadIdx.x == 0) {
        int mine = 0;
        for (int c0 = part * 256; c0 < n_dead; c0 += 256) mine += min(256, n_dead - c0);
        id_base = atomicAdd(&ctl->next_id, (unsigned long long)mine);
    }
    __syncthreads();
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    const unsigned long long end_id = ctl->end_id;
    int done_before = 0;
    for (int c0 = part * 256; c0 < n_dead; c0 += 256) {
        const int k = c0 + (int)threadIdx.x;
        const bool valid = k < n_dead;
        const int slot = valid ? list[k] : 0;
        int state = valid ? TS_DEAD : TS_DONE;
        Packet<ND, GEOM> p;
        Rng g;
        unsigned long long id = 0;
        p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
        PeelFlags f; f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = 0;
        unsigned int peel_seq = 0;
        bool wrote = false;        // IMG: this entry's reserved event slot holds an event
        // a packet that is emitted outside the grid (or leaves it at once) frees its slot again
        for (int round = 0;; round++) {
            const bool want = state == TS_DEAD;
            const unsigned long long m = __ballot(want);
            if (!m) break;
            unsigned long long base = 0;
            if (round > 0) {
                const int leader = __ffsll((long long)m) - 1;
                if ((int)__lane_id() == leader) base = atomicAdd(&ctl->next_id, (unsigned long long)__popcll(m));
                base = __shfl(base, leader, 64);
            }
            if (want) {
                id = round == 0 ? id_base + (unsigned long long)(done_before + (int)threadIdx.x)
                                : base + (unsigned long long)__popcll(m & ((1ull << __lane_id()) - 1ull));
                if (id >= end_id) state = TS_DONE;
                else if constexpr (!IMG) {
                    rng_init(g, P.seed_key, T.iter_tag, id);
                    int source_id; Angle src_normal;
                    bool ok = emit_packet<ND, GEOM, SIMPLE>(P, W, p, g, cnt, source_id, src_normal);
                    if (!ok || geo_escaped(P, p.cell)) finished++;
                    else {
                        p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                        begin_integrate(P, p);
                        state = (p.tau_req == 0.0) ? TS_INTERACT : TS_WALK;
                    }
                } else {
                    // iter_final.f90:160-209: emission, its peel-off event, the first optical depth (forced first interaction or not)
                    rng_init(g, P.seed_key, T.iter_tag, id);
                    bool ok = true, emit_iso = true;
                    double tau_first = 0.0, energy_after = 0.0;
                    int source_id = 0;
                    Angle src_normal = p.a;
                    if (B.ff) {
                        const EmitRec<ND> &R = ((const EmitRec<ND> *)B.ff)[id - ctl->first_id];
                        ok = (R.code >> 1) != 0;          // 0: the emission failed ahead of the rounds (its error is raised: the launch stops)
                        if (ok) {
                            g.buf_a = R.buf_a; g.blk_a = R.blk_a; g.blk_b = R.blk_b; g.have_a = R.code & 1; g.countdown = R.countdown;
                            source_id = R.source_id;
                            const DSource &S = P.sources[source_id];
                            p.r[0] = S.pos[0]; p.r[1] = S.pos[1]; p.r[2] = S.pos[2];
                            p.a = R.a;
                            angle_to_vector(p.a, p.v[0], p.v[1], p.v[2]);
                            p.s[0] = 1.0; p.s[1] = 0.0; p.s[2] = 0.0; p.s[3] = 0.0;
                            p.nu = R.nu; p.energy = R.energy0;
                            cnt.energy_current += R.energy0;
                            tau_first = R.tau_req; energy_after = R.energy;
#pragma unroll
                            for (int d = 0; d < ND; d++) { p.chi[d] = R.chi[d]; p.albedo[d] = R.albedo[d]; p.kappa[d] = R.kappa[d]; }
                            p.emiss_dust = -1;
                            geo_clear_wall(p.cell);
                            bool placed = false;
                            if constexpr (GEOM == GEOM_VOR) {
                                if (S.type == 1 && S.vor_cell1 > 0) { p.cell.id = S.vor_cell1 - 1; placed = true; }
                            }
                            if (!placed) (void)geo_place(P, W, p.r, p.v, p.cell);       // it did succeed ahead of the rounds
                            p.inter = 1; p.n_visited = 0;
                        }
                    } else {
                        // SIMPLE == 0: the general emitters (sources with a surface -- the emission's peel-off weighs with the angle to the
                        // normal, which the event carries in a_prev: final_defer_kernel<.., GEN>, hyp_defer.h)
                        ok = emit_packet<ND, GEOM, SIMPLE>(P, W, p, g, cnt, source_id, src_normal);
                        if (SIMPLE == 0 && ok) emit_iso = P.sources[source_id].type == 1 || P.sources[source_id].type == 8 || P.sources[source_id].type == 4;
                    }
                    f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
                    peel_seq = 0; p.reabs = 0;
                    if (!ok) finished++;
                    else {
                        if (!P.peel_scattered_only) {
                            // the emission's event: this entry's reserved slot, or one more for a second emission into the same slot
                            unsigned long long e_idx = ev_base == ~0ull ? ~0ull : ev_base + (unsigned long long)k;
                            if (wrote) { e_idx = atomicAdd(&B.ctl->reserved, 1ull); if (e_idx >= B.cap) { raise_error(P, ERR_INTERNAL, (double)e_idx, (double)B.cap, 3.0); e_idx = ~0ull; } }
                            const double s0[4] = {p.s[0], p.s[1], p.s[2], p.s[3]};
                            if (e_idx != ~0ull) write_peel_event<ND, GEOM>(ev[e_idx], p, g, emit_iso ? p.a : src_normal, s0, f, peel_seq, LAST_SR, emit_iso);
                            peel_seq++; wrote = true;
                        }
                        if (geo_escaped(P, p.cell)) finished++;
                        else {
                            if (B.ff) { p.tau_req = tau_first; p.energy = energy_after; }
                            else if (SIMPLE == 0 && P.forced_first) {
                                // forced first interaction of a packet that starts on its source's surface: the escape walk here, as final_defer_kernel<.., GEN>
                                // makes it (iter_final.f90:191-209)
                                bool killed = false, sampled = false;
                                const double tau_escape = escape_tau<ND, GEOM>(P, W, p.r, p.v, p.cell, p.chi, g, cnt, killed);
                                if (tau_escape > 1e-10 && !killed) {
                                    double weight, tau;
                                    forced_interaction(P, tau_escape, rng_uniform(g), tau, weight);
                                    p.tau_req = tau; p.energy *= weight; sampled = true;
                                }
                                if (!sampled) p.tau_req = rng_exp(g);
                            }
                            else p.tau_req = rng_exp(g);
                            p.tau_ach = 0.0;
                            begin_integrate(P, p);
                            state = (p.tau_req == 0.0) ? TS_INTERACT : TS_WALK;
                        }
                    }
                }
            }
        }
        if (IMG && valid && !wrote && ev_base != ~0ull) ev[ev_base + (unsigned long long)k].code = 0;       // nothing emitted into this entry: an empty slot
        done_before += min(256, n_dead - c0);
        int brick = 0;
        if (valid) {
            if (state == TS_WALK || state == TS_INTERACT) {
                store_records<ND, GEOM>(P, hot[slot], cold[slot], p, g, id, state);
                if (IMG) cold_flags_store(cold[slot], f, peel_seq, P.any_intersect != 0);
                if (state == TS_WALK) { brick = TileCellIO<GEOM>::brick(P, T, p.cell); if (T.vsplit > 1) brick = brick * T.vsplit; slot_brick[slot] = brick; }
                else {
                    // zero optical depth drawn (probability 2^-53): the next generation's extra workgroup interacts
                    slot_brick[slot] = TILE_NEEDS_INTERACT;
                    const unsigned int j = atomicAdd(&ctl->n_extra[T.pool][par ^ 1], 1u);
                    if (j < HYP_TILE_EXTRA) extra_next[j] = slot; else raise_error(P, ERR_INTERNAL, (double)j, 0.0, 0.0);
                }
            } else { hot[slot].state = TS_DONE; slot_brick[slot] = TILE_IDLE; }      // no packet ids left: the slot retires
        }
        count_bricks(counts, cache_b, cache_n, valid && state == TS_WALK, brick);
    }
    __syncthreads();
    flush_bricks(counts, cache_b, cache_n);
    block_tally_flush(P, ctl, red, cnt, finished);
}

// ---------------------------------------------------------------------------
// tile_drain: once the packet ids are used up and only a few packets are still in
// flight, generations stop paying (every one of them rescans all slots for a handful of
// steps).  This kernel takes every remaining packet to its end in one launch, one lane
// per packet, with the persistent kernel's walk_step (global atomics).
// ---------------------------------------------------------------------------
template <int ND, bool REABS, bool MRW, int GEOM>
__global__ __launch_bounds__(256, HYP_PREP_WAVES) void tile_drain_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                                       void *__restrict__ hot_v, void *__restrict__ cold_v,
                                                                       int *__restrict__ slot_brick)
{
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    double *sum = P.sum;
    if (P.n_copies > 1) {
        unsigned c = xcc_id();
        if (P.n_copies > 8) c += 8u * ((blockIdx.x >> 3) % (unsigned)(P.n_copies >> 3));
        sum += (size_t)(c % (unsigned)P.n_copies) * P.copy_stride;
    }
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    // The slots that still hold a packet come as one list (tile_live_kernel); a lane takes the next entry as soon as its packet has
    // ended (a wave refills when half of it is idle).  Round 3 gave every workgroup chunks of 2048 SLOTS to look through and ran the
    // packets it found 256 at a time to the end of the longest one: 18 passes at 40 % of the lanes, 10.6 ms for the last 1e6 packets
    // of configs[1] (profiles/r04_tiled_log.md).
    const int *__restrict__ live = T.drain_list;
    const unsigned int nl = ctl->n_live;
    const unsigned long long lt = (1ull << __lane_id()) - 1ull;
    int st = ST_DONE, slot = -1;
    bool exhausted = false;
    Packet<ND, GEOM> p;
    Rng g;
    rng_init(g, P.seed_key, T.iter_tag, 0);
    p.inter = 1; p.tau_req = 0.0; p.tau_ach = 0.0;
    p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
    {
        {
            for (;;) {
                unsigned long long m_walk = __ballot(st == ST_WALK);
                unsigned long long m_int = __ballot(st == ST_NEED_INTERACT);
                unsigned long long m_re = REABS ? __ballot(st == ST_NEED_REEMIT) : 0ull;
                const unsigned long long m_done = __ballot(st == ST_DONE);
                if (!exhausted && m_done && (__popcll(m_done) >= 32 || !(m_walk | m_int | m_re))) {
                    if (st == ST_DONE && slot >= 0) { hot[slot].state = TS_DONE; slot_brick[slot] = TILE_IDLE; slot = -1; }
                    const int want = __popcll(m_done);
                    unsigned int base = 0;
                    if (__lane_id() == 0) base = atomicAdd(&ctl->live_cursor, (unsigned int)want);
                    base = __shfl(base, 0, 64);
                    if (base + (unsigned int)want >= nl) exhausted = true;
                    const unsigned int kq = base + (unsigned int)__popcll(m_done & lt);
                    if (st == ST_DONE && kq < nl) {
                        slot = live[kq];
                        const HotRec<ND> &H = hot[slot];
                        const ColdRec<ND> &C = cold[slot];
#pragma unroll
                        for (int a = 0; a < 3; a++) { p.r[a] = H.r[a]; p.v[a] = H.v[a]; }
                        TileCellIO<GEOM>::load(P, H, p.cell);
                        p.a = C.a;
                        p.s[0] = C.s[0]; p.s[1] = C.s[1]; p.s[2] = C.s[2]; p.s[3] = C.s[3];
                        p.nu = C.nu; p.energy = H.energy; p.tau_req = H.tau_req; p.tau_ach = H.tau_ach;
#pragma unroll
                        for (int d = 0; d < ND; d++) { p.chi[d] = H.chi[d]; p.kappa[d] = H.kappa[d]; p.albedo[d] = C.albedo[d]; }
                        p.inter = C.inter;
                        const unsigned long long id = H.id;
                        g.key0 = P.seed_key; g.key1 = T.iter_tag; g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
                        g.blk_a = C.blk_a; g.blk_b = H.blk_b; g.buf_a = C.buf_a; g.have_a = C.have_a; g.countdown = H.countdown;
                        p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
                        if (REABS) { p.t_src = C.t_src; p.t_ach = C.t_ach; p.reabs_id = C.reabs_id; p.reabs = C.reabs; }
                        st = H.state == TS_INTERACT ? ST_NEED_INTERACT : (REABS && H.state == TS_REEMIT) ? ST_NEED_REEMIT : ST_WALK;
                    }
                    m_walk = __ballot(st == ST_WALK);
                    m_int = __ballot(st == ST_NEED_INTERACT);
                    m_re = REABS ? __ballot(st == ST_NEED_REEMIT) : 0ull;
                }
                if (!(m_walk | m_int | m_re)) { if (exhausted || nl == 0) break; else continue; }
                if (REABS && m_re && (__popcll(m_re) >= 16 || !m_walk)) {      // iter_lucy.f90:155-185
                    if (st == ST_NEED_REEMIT) {
                        if ((long long)p.reabs == P.n_reabs_max) { cnt.killed_int++; st = ST_DONE; finished++; }
                        else {
                            const int inter = p.inter, reabs = p.reabs + 1, rid = p.reabs_id;
                            const double e = p.energy;
                            int source_id; Angle src_normal;
                            bool ok = emit_packet<ND, GEOM>(P, W, p, g, cnt, source_id, src_normal, rid, e);
                            p.inter = inter; p.reabs = reabs;
                            if (!ok || geo_escaped(P, p.cell)) { st = ST_DONE; finished++; }
                            else {
                                p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                                begin_integrate(P, p);
                                st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                            }
                        }
                    }
                    m_walk = __ballot(st == ST_WALK);
                    m_int = __ballot(st == ST_NEED_INTERACT);
                }
                if (m_int && (__popcll(m_int) >= 16 || !m_walk)) {
                    if (st == ST_NEED_INTERACT) {
                        p.reabs = 0;
                        if ((long long)p.inter == P.n_inter_max + 1) { cnt.killed_int++; st = ST_DONE; finished++; }
                        else {
                            int scattered, dust_id;
                            bool ok = interact<ND, GEOM>(P, p, g, cnt, scattered, dust_id);
                            bool killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered);
                            if (killed) { st = ST_DONE; finished++; }
                            else if (MRW && mrw_loop_lucy<ND, GEOM>(P, W, p, g, sum, cnt)) { st = ST_DONE; finished++; }
                            else {
                                p.inter++;
                                p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                                begin_integrate(P, p);
                                st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                            }
                        }
                    }
                }
#pragma unroll 1
                for (int q = 0; q < 4; q++) {
                    if (st == ST_WALK) {
                        st = walk_step<ND, GEOM, true>(P, W, p, g, sum, cnt);
                        if (st == ST_NEED_EMIT) { st = ST_DONE; finished++; }
                    }
                }
            }
            if (st == ST_DONE && slot >= 0) { hot[slot].state = TS_DONE; slot_brick[slot] = TILE_IDLE; }
        }
    }
    double c = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    double ki = wave_sum((double)cnt.killed_int);
    double ni = wave_sum((double)cnt.interactions);
    double nf = wave_sum((double)finished);
    if (__lane_id() == 0) {
        if (c != 0.0) unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], c);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
        if (ki != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], ki);
        if (ni != 0.0) unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], ni);
        if (nf != 0.0) atomicAdd(&ctl->n_finished, (unsigned long long)nf);
    }
}

// End-game of the imaging iteration on this schedule (round 6; iter_final.f90:255-259 bounds a packet's interactions, not its
// generations): when no packet id is left and few packets are in flight, a generation is four launches per pool for a handful of
// packets.  The live slots (tile_live_kernel's list) become SuspRec of the deferred schedule -- "about to interact", or -3 "on its
// way" -- and final_defer_kernel resumes them as it resumes what a round set aside (hyp_defer.h): the packets end in a few
// persistent launches with their peel-off events sorted and peeled as always.  Entry k of the list goes to lane k of the next round.
template <int ND, int GEOM>
__global__ __launch_bounds__(256) void tile_to_susp_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
                                                          void *__restrict__ hot_v, void *__restrict__ cold_v, int *__restrict__ slot_brick, DeferBuf B)
{
    const DProblem &P = *Pp;
    HotRec<ND> *__restrict__ hot = (HotRec<ND> *)hot_v;
    ColdRec<ND> *__restrict__ cold = (ColdRec<ND> *)cold_v;
    const unsigned int nl = ctl->n_live;
    const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) B.ctl->n_susp[B.cur ^ 1] = nl;
    if (k >= nl) return;
    const int slot = T.drain_list[k];
    const HotRec<ND> &H = hot[slot];
    const ColdRec<ND> &C = cold[slot];
    SuspRec<ND, GEOM> &R = ((SuspRec<ND, GEOM> *)B.susp[B.cur ^ 1])[k];
    Packet<ND, GEOM> p;
#pragma unroll
    for (int a = 0; a < 3; a++) { p.r[a] = H.r[a]; p.v[a] = H.v[a]; }
    TileCellIO<GEOM>::load(P, H, p.cell);
    p.a = C.a;
    p.s[0] = C.s[0]; p.s[1] = C.s[1]; p.s[2] = C.s[2]; p.s[3] = C.s[3];
    p.nu = C.nu; p.energy = H.energy; p.tau_req = H.tau_req; p.tau_ach = H.tau_ach;
#pragma unroll
    for (int d = 0; d < ND; d++) { p.chi[d] = H.chi[d]; p.kappa[d] = H.kappa[d]; p.albedo[d] = C.albedo[d]; }
    p.inter = C.inter; p.emiss_dust = -1;
    p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
    if (P.any_intersect) { p.t_src = C.t_src; p.t_ach = C.t_ach; p.reabs_id = C.reabs_id; p.reabs = C.reabs; }      // (sources with a surface: final_defer_kernel<.., GEN> resumes)
    p.n_visited = 0; p.e_init = 0.0;
    PeelFlags f; unsigned int peel_seq;
    cold_flags_load(C, f, peel_seq, P.any_intersect != 0);
    p.peel_seq = peel_seq;
    p.spec_idx = H.state == TS_INTERACT ? 0 : H.state == TS_REEMIT ? 1 : -3;
    Rng g;
    const unsigned long long id = H.id;
    g.key0 = P.seed_key; g.key1 = T.iter_tag; g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
    g.blk_a = C.blk_a; g.blk_b = H.blk_b; g.buf_a = C.buf_a; g.have_a = C.have_a; g.countdown = H.countdown;
    R.p = p; R.g = g; R.f = f;
    hot[slot].state = TS_DONE; slot_brick[slot] = TILE_IDLE;
}

// the slots that still hold a packet, for tile_drain_kernel: 2048 slots per workgroup, one reservation in the list per workgroup
static __global__ __launch_bounds__(256) void tile_live_kernel(TileGeom T, const int *__restrict__ slot_brick, int *__restrict__ live, TileCtl *__restrict__ ctl)
{
    __shared__ int list[HYP_PREP_CHUNK];
    __shared__ int n_list, base;
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < HYP_PREP_CHUNK; k += blockDim.x) {
        const int s = blockIdx.x * HYP_PREP_CHUNK + k;
        const int sb = s < T.n_slots ? slot_brick[s] : TILE_IDLE;
        if (sb >= 0 || sb == TILE_NEEDS_INTERACT || sb == TILE_NEEDS_REEMIT) list[atomicAdd(&n_list, 1)] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) base = n_list ? (int)atomicAdd(&ctl->n_live, (unsigned int)n_list) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_list; i += blockDim.x) live[base + i] = list[i];
}

// ---------------------------------------------------------------------------
// counting sort of the walking slots by brick
// ---------------------------------------------------------------------------
// slots per thread of tile_sort.  Every workgroup makes one returning atomic per brick on cursor[], and
// atomics on one address are served one after the other by the memory side: the fewer workgroups, the shorter that queue.
#ifndef HYP_SORT_PER_THREAD_N
#define HYP_SORT_PER_THREAD_N 32
#endif
constexpr int HYP_SORT_PER_THREAD = HYP_SORT_PER_THREAD_N;

// Exclusive scan of the brick counts, task list and scatter of the slots in ONE launch.  Every workgroup
// scans the brick counts itself (a few hundred values; the offsets stay in LDS), workgroup 0 also writes the task list and resets
// what the next generation accumulates into.  The counts and the scatter cursors are kept twice, by generation parity: this
// generation's (`counts`, read by every workgroup for as long as the launch runs) and the next one's (`counts_next`: what the walk
// of this generation and the interaction / emission kernels of the next add to; it held generation g - 1's counts and is cleared
// here, after that generation's sort and before this generation's walk).
// dynamic LDS: 2 x n_bricks unsigned (histogram | offsets) + 512 unsigned
static __global__ __launch_bounds__(256) void tile_sort_kernel(TileGeom T, const int *__restrict__ slot_brick, const unsigned int *__restrict__ counts,
                                                       unsigned int *__restrict__ counts_next, unsigned int *__restrict__ cursor,
                                                       unsigned int *__restrict__ cursor_next, int *__restrict__ order,
                                                       TileTask *__restrict__ tasks, TileCtl *__restrict__ ctl)
{
    extern __shared__ unsigned int sort_lds[];
    unsigned int *hist = sort_lds, *offs = sort_lds + T.n_bricks, *part_c = offs + T.n_bricks, *part_t = part_c + 256;
    const int per = (T.n_bricks + 255) / 256;
    const int b0 = min((int)threadIdx.x * per, T.n_bricks), b1 = min(b0 + per, T.n_bricks);
    unsigned int sc = 0, stt = 0;
    for (int b = b0; b < b1; b++) { const unsigned int c = counts[b]; hist[b] = 0; sc += c; stt += (c + T.task_size - 1) / T.task_size; }
    part_c[threadIdx.x] = sc; part_t[threadIdx.x] = stt;
    __syncthreads();
    if (threadIdx.x < 64) {        // exclusive scan of the 256 partial sums by one wave (four per lane)
        unsigned int c4[4], t4[4], lc = 0, lt = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { c4[k] = part_c[4 * threadIdx.x + k]; t4[k] = part_t[4 * threadIdx.x + k]; lc += c4[k]; lt += t4[k]; }
        unsigned int ic = lc, it = lt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int uc = __shfl_up(ic, d, 64), ut = __shfl_up(it, d, 64);
            if ((int)threadIdx.x >= d) { ic += uc; it += ut; }
        }
        unsigned int ec = ic - lc, et = it - lt;
#pragma unroll
        for (int k = 0; k < 4; k++) { part_c[4 * threadIdx.x + k] = ec; part_t[4 * threadIdx.x + k] = et; ec += c4[k]; et += t4[k]; }
        if (blockIdx.x == 0 && threadIdx.x == 63) {
            ctl->n_tasks[T.pool] = it;
            ctl->n_gil[T.pool][T.gen & 1] = 0; ctl->n_gdl[T.pool][T.gen & 1] = 0;      // the walk of this generation fills them
        }
    }
    __syncthreads();
    {
        unsigned int oc = part_c[threadIdx.x], ot = part_t[threadIdx.x];
        for (int b = b0; b < b1; b++) {
            const unsigned int c = counts[b];
            offs[b] = oc;
            if (blockIdx.x == 0) {
                for (unsigned int q = 0; q < c; q += T.task_size) {
                    TileTask tk; tk.brick = b; tk.start = (int)(oc + q); tk.len = (int)min((unsigned int)T.task_size, c - q); tk.pad = 0;
                    tasks[ot++] = tk;
                }
                counts_next[b] = 0; cursor_next[b] = 0;
            }
            oc += c;
        }
    }
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
        if (br[k] >= 0) order[offs[br[k]] + hist[br[k]] + rank[k]] = slot;
    }
}


// brick shape per number of species: density + accumulators (16 B per cell and
// species) must leave room for two workgroups per CU in the 160 KB LDS
// Round 3: 32 x 16 x 16 bricks walked by ONE 1024-thread workgroup per CU instead of two 512-thread workgroups on 16^3 bricks:
// the mean chord grows from 10.7 to 12.8 cells, a fifth fewer visits (record traffic, sort, service phases); with the
// interaction / emission work in kernels of their own a CU that drains a task's tail alone costs less than it did in
// round 1 (311 against 298 ms then; now 255-259 against 268-277 at tasks of 8192 packets, profiles/r03_tiled_log.md).
constexpr int HYP_TILE_BX = 32;
constexpr int HYP_TILE_BY = 16;
constexpr int HYP_TILE_BZ = 16;
template <int ND> struct TileShape { static constexpr int X = HYP_TILE_BX, Y = HYP_TILE_BY, Z = HYP_TILE_BZ; };      // 128 KB
template <> struct TileShape<2> { static constexpr int X = 32, Y = 16, Z = 8; };         // 128 KB
template <> struct TileShape<3> { static constexpr int X = 32, Y = 8, Z = 8; };          // 96 KB
template <> struct TileShape<4> { static constexpr int X = 32, Y = 8, Z = 8; };          // 128 KB

// ---------------------------------------------------------------------------
// tile_walk: one workgroup per task; density and accumulators of the brick in LDS
//
// The kernel is VALU-issue bound (profiles/r01c_summary.md), so the per-step code is
// kept lean: everything that happens once per visit -- finishing the partial step of an
// interaction, writing the record back, taking the next packet -- is deferred to a
// "service" phase that runs when at least 16 lanes of the wave wait for it.
// ---------------------------------------------------------------------------

// (find_wall_ahead, the branch-free wall search of the step, lives in hyp_kernels.h: the deferred imaging kernels use it too)

// lane states of tile_walk_kernel
// LS_CHECK: the propagation check is due; LS_SLOW: geo_find_wall is needed for this step
// LS_REABS: the step would run into a source (grid_propagate_3d.f90:139-143)
enum { LS_IDLE = 0, LS_WALK = 1, LS_LEFT = 2, LS_DEAD = 3, LS_HIT = 4, LS_CHECK = 5, LS_SLOW = 6, LS_REABS = 7 };

template <int ND, int BX, int BY, int BZ>
__global__ __launch_bounds__(HYP_TILE_WG, HYP_TILE_OCC) void tile_walk_kernel(const DProblem *__restrict__ Pp, TileGeom T, TileCtl *__restrict__ ctl,
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
    const bool any_intersect = P.any_intersect != 0;      // read once: inside the loops it would be a scalar load and a wait per cell step
    constexpr int NC = BX * BY * BZ;
    Walls W;
    stage_walls<GEOM_CAR>(P, lds, W);
    double *dens = lds + 2 * ((size_t)P.n1 + P.n2 + P.n3 + 3);
    double *accum = dens + (size_t)NC * ND;
    __shared__ int next_pkt;
    // this task's lists of waiting / free slots and the bricks its packets move to
    // (index (dz+1)*9 + (dy+1)*3 + dx+1; 13 = packets parked in this brick)
    __shared__ int n_int_l, n_dead_l, pub_base[2];
    __shared__ unsigned int nb_cnt[27];
    __shared__ double red[TILE_RED_N];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + TILE_RED_N) red[threadIdx.x - 64] = 0.0;
    if (threadIdx.x < 27) nb_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 32) { n_int_l = 0; n_dead_l = 0; }
    const int bi = tk.brick % T.nbx, bj = (tk.brick / T.nbx) % T.nby, bk = tk.brick / (T.nbx * T.nby);
    const int x0 = bi * BX, y0 = bj * BY, z0 = bk * BZ;
    // (the grid's size in registers: read through P inside the step loop it was three dependent scalar loads, each with its wait, in
    // every wave-step in which some lane left the brick -- most of them)
    const int gn1 = P.n1, gn2 = P.n2, gn3 = P.n3;
    const int x1 = min(x0 + BX, gn1), y1 = min(y0 + BY, gn2), z1 = min(z0 + BZ, gn3);
    // Round 6: a lane keeps its cell as indices RELATIVE to the brick (cell.ic = global index - origin while it walks): the LDS index
    // of the cell is two shift-adds instead of three subtractions and two, "left the brick" three unsigned comparisons instead of
    // six signed ones, and the grid's edge is looked at only by packets that leave.  The wall tables are addressed through pointers
    // moved by the origin; records and the general functions of the service phase see global indices (to_global).
    const int ex = x1 - x0, ey = y1 - y0, ez = z1 - z0;          // the brick's extent (clipped to the grid)
    Walls Wl = W;
    Wl.w[0] += x0; Wl.w[1] += y0; Wl.w[2] += z0; Wl.ew[0] += x0; Wl.ew[1] += y0; Wl.ew[2] += z0;
    auto to_global = [&](const Cell<GEOM_CAR> &c) { Cell<GEOM_CAR> gcell = c; gcell.ic[0] += x0; gcell.ic[1] += y0; gcell.ic[2] += z0; return gcell; };
    for (int c = threadIdx.x; c < NC; c += blockDim.x) {
        int lx = c % BX, ly = (c / BX) % BY, lz = c / (BX * BY);
        int gx = x0 + lx, gy = y0 + ly, gz = z0 + lz;
        bool in = gx < P.n1 && gy < P.n2 && gz < P.n3;
        size_t gidx = ((size_t)gz * P.n2 + gy) * P.n1 + gx;
        for (int d = 0; d < ND; d++) {
            dens[c * ND + d] = in ? P.density[gidx * ND + d] : 0.0;
            accum[c * ND + d] = 0.0;
        }
    }
    if (threadIdx.x == 0) next_pkt = 0;
    __syncthreads();

    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    unsigned int finished = 0;
    // lane state: the walking part of a packet (the rest stays in its records).  ke = kappa x energy: what a deposit multiplies the
    // path length with (the records keep the two apart; the product is formed once per visit instead of once per crossing)
    double r[3], v[3], tau_req = 0.0, tau_ach = 0.0, chi[ND], ke[ND];
    double inv[3];                            // RN(1 / v) per axis, see find_wall_ahead
    int iu[3], smask[3];                      // iu = 1 where v > 0; smask = the sign bit where v <= 0: fixed during a visit
    bool v_ok = true;                         // no direction component is so small that 1 / v or d / v could overflow
    double hit_t = 0.0, hit_tau = 0.0;       // LS_HIT: step length to the wall and optical depth of the cell
    double t_src = HYP_INF, t_ach = 0.0;     // re-absorption by sources (P.any_intersect): see Packet
    int hit_lc = 0;
    Cell<GEOM_CAR> cell;
    Rng g; g.key0 = P.seed_key; g.key1 = T.iter_tag; g.blk_a = 0; g.have_a = 0; g.buf_a = 0.0;
    g.id_lo = g.id_hi = 0; g.blk_b = 0; g.countdown = 0;
    int slot = -1;                            // (bit 30: HotRec::pad, the kind of the packet's next interaction -- see store_records; a register of its own
                                              //  took the kernel from 119 to 124 VGPRs and cost more than the ordering saves)
#define SLOT (slot & 0x3fffffff)
    int st = LS_IDLE;
    bool exhausted = false, pre = false;
#pragma unroll
    for (int a = 0; a < 3; a++) { r[a] = 0.0; v[a] = 1.0; inv[a] = 1.0; cell.ic[a] = 0; cell.ow[a] = 0; smask[a] = 0; iu[a] = 1; }
#pragma unroll
    for (int d = 0; d < ND; d++) { chi[d] = 0.0; ke[d] = 0.0; }

    bool queue_empty = false;        // wave-uniform: some lane found the task's queue empty
#ifdef HYP_TILE_STATS
    unsigned long long dbg_outer = 0, dbg_wsteps = 0, dbg_lsteps = 0, dbg_service = 0, dbg_nservice = 0, dbg_wb = 0, dbg_claim = 0, dbg_nclaim = 0, dbg_nwb = 0, dbg_ncheck = 0, dbg_chk = 0;
    const long long dbg_t0 = clock64();
#endif
    for (;;) {
        if (queue_empty && st == LS_IDLE) exhausted = true;
        const unsigned long long m_walk = __ballot(st == LS_WALK);
        const unsigned long long m_out = __ballot(st >= LS_LEFT);      // anything the service phase must look at
        const unsigned long long m_idle = __ballot(st == LS_IDLE && !exhausted);
        if (!(m_walk | m_out | m_idle)) break;
        // Tail of a task: the queue is empty and only a few lanes of this wave still walk.  Their
        // packets go back to their slots as they are (same brick) and continue in the next
        // generation in a full wave, instead of dragging a nearly empty wave along.
        const bool park = !m_idle && queue_empty && __popcll(m_walk) <= T.park;
        // ---- service phase: write finished visits back, take new packets ----
        if (park || ((m_out | m_idle) && (__popcll(m_out | m_idle) >= HYP_TILE_SERVICE || !m_walk))) {
#ifdef HYP_TILE_STATS
            const long long dbg_ts = clock64();
            dbg_nwb += __popcll(m_out); if (__ballot(st == LS_CHECK || st == LS_SLOW)) dbg_ncheck++;
#endif
            // propagation check (grid_propagate_3d.f90:112-120), then the step goes on as usual.  The gap to a lane's next check costs a
            // Philox block and a logarithm whether one lane asks or sixty-four, and with the reference's default frequency a quarter of
            // the service phases would find ONE lane asking: a lane whose check is due waits until four are, or nobody walks any more
            const unsigned long long m_chk = __ballot(st == LS_CHECK);
            if (st == LS_CHECK && (__popcll(m_chk) >= 4 || !m_walk || park)) {
                const int gap = rng_check_gap(g, P.check_p, P.check_log1mp);
                g.countdown = gap < 0x7fffffff ? gap + 1 : gap;      // the step below takes one off again
                if (geo_in_correct_cell(P, W, r, to_global(cell))) st = LS_WALK;
                else { cnt.killed_geo++; st = LS_DEAD; }
            }
            // packets that round-off left outside their cell: the general wall search, handed to the
            // next step through (hit_t, hit_lc)
            if (st == LS_SLOW) {
                double tmin; int im[3];
                if (geo_find_wall(P, W, r, v, to_global(cell), tmin, im)) {
                    hit_t = tmin; hit_lc = (im[0] + 1) | ((im[1] + 1) << 2) | ((im[2] + 1) << 4);
                    pre = true; st = LS_WALK;
                } else { cnt.killed_geo++; st = LS_DEAD; }
            }
#ifdef HYP_TILE_STATS
            __builtin_amdgcn_s_waitcnt(0);
            const long long dbg_tc = clock64();
            dbg_chk += (unsigned long long)(dbg_tc - dbg_ts);
#endif
            if (st == LS_HIT && any_intersect) {
                const double tact0 = hit_t * ((tau_req - tau_ach) / hit_tau);
                t_ach += tact0;
                if (t_ach > t_src) st = LS_REABS;       // grid_propagate_3d.f90:184-188
            }
            if (st == LS_HIT) {
                // the interaction happens inside this cell: grid_propagate_3d.f90:170-200
                const double tau_needed = tau_req - tau_ach;
                const double tact = hit_t * (tau_needed / hit_tau);
#pragma unroll
                for (int a = 0; a < 3; a++) r[a] = r[a] + tact * v[a];
                tau_ach += tau_needed;
                geo_clear_wall(cell);
#pragma unroll
                for (int d = 0; d < ND; d++)
                    if (dens[hit_lc * ND + d] > 0.0) TILE_DEPOSIT(&accum[hit_lc * ND + d], tact * ke[d]);
            }
            if (st == LS_DEAD) {
                hot[SLOT].state = TS_DEAD; slot_brick[SLOT] = TILE_NEEDS_PREPARE;
                dlist[tk.start + atomicAdd(&n_dead_l, 1)] = SLOT;
                finished++; st = LS_IDLE;
            } else if (st == LS_LEFT || st == LS_HIT || st == LS_REABS || (park && st == LS_WALK)) {
                HotRec<ND> &H = hot[SLOT];
#pragma unroll
                for (int a = 0; a < 3; a++) H.r[a] = r[a];
                H.ic[0] = cell.ic[0] + x0; H.ic[1] = cell.ic[1] + y0; H.ic[2] = cell.ic[2] + z0;
                H.ow = pack_ow(cell.ow);
                H.tau_ach = tau_ach; H.countdown = g.countdown; H.blk_b = g.blk_b;
                if (any_intersect) cold[SLOT].t_ach = t_ach;
                if (st == LS_REABS) { H.state = TS_REEMIT; slot_brick[SLOT] = TILE_NEEDS_REEMIT; }
                else if (st == LS_HIT) { H.state = TS_INTERACT; slot_brick[SLOT] = TILE_NEEDS_INTERACT; }
                else if (st == LS_LEFT) {                                             // H.state stays TS_WALK
                    // one cell step leaves the brick through a face, an edge or a corner
                    const int dx = cell.ic[0] < 0 ? -1 : (cell.ic[0] >= ex ? 1 : 0);
                    const int dy = cell.ic[1] < 0 ? -1 : (cell.ic[1] >= ey ? 1 : 0);
                    const int dz = cell.ic[2] < 0 ? -1 : (cell.ic[2] >= ez ? 1 : 0);
                    slot_brick[SLOT] = tk.brick + dx + T.nbx * (dy + T.nby * dz);
                    atomicAdd(&nb_cnt[(dz + 1) * 9 + (dy + 1) * 3 + dx + 1], 1u);
                } else atomicAdd(&nb_cnt[13], 1u);                                    // parked: same brick again
                if (st == LS_REABS || st == LS_HIT) ilist[tk.start + atomicAdd(&n_int_l, 1)] = (T.presort && st == LS_HIT) ? slot : SLOT;
                st = LS_IDLE;
            }
            if (park) break;
#ifdef HYP_TILE_STATS
            __builtin_amdgcn_s_waitcnt(0);
            const long long dbg_tw = clock64();
            dbg_wb += (unsigned long long)(dbg_tw - dbg_tc); dbg_nclaim += __popcll(__ballot(st == LS_IDLE && !exhausted));
#endif
            if (st == LS_IDLE && !exhausted) {
                const int j = atomicAdd(&next_pkt, 1);
                if (j >= tk.len) exhausted = true;
                else {
                    slot = order[tk.start + j];
                    const HotRec<ND> &H = hot[SLOT];
                    v_ok = true;
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        r[a] = H.r[a]; v[a] = H.v[a]; cell.ic[a] = H.ic[a] - (a == 0 ? x0 : a == 1 ? y0 : z0);
                        iu[a] = v[a] > 0.0 ? 1 : 0; smask[a] = v[a] > 0.0 ? 0 : (int)0x80000000;
                        inv[a] = 1.0 / v[a];
                        v_ok = v_ok & ((v[a] == 0.0) | (fabs(v[a]) >= 0x1p-400));
                    }
                    unpack_ow(H.ow, cell.ow);
                    tau_req = H.tau_req; tau_ach = H.tau_ach;
#pragma unroll
                    for (int d = 0; d < ND; d++) { chi[d] = H.chi[d]; ke[d] = H.kappa[d] * H.energy; }
                    unsigned long long id = H.id;
                    g.id_lo = (uint32_t)id; g.id_hi = (uint32_t)(id >> 32);
                    g.countdown = H.countdown; g.blk_b = H.blk_b;
                    slot |= (H.pad & 1) << 30;
                    if (any_intersect) { t_src = cold[SLOT].t_src; t_ach = cold[SLOT].t_ach; }
                    st = LS_WALK; pre = false;
                }
            }
            if (__ballot(exhausted)) queue_empty = true;
#ifdef HYP_TILE_STATS
            __builtin_amdgcn_s_waitcnt(0);      // charge the loads of the refill to the service phase
            dbg_service += (unsigned long long)(clock64() - dbg_ts); dbg_nservice++; dbg_claim += (unsigned long long)(clock64() - dbg_tw);
#endif
        }
        // ---- a few cell steps (the body of grid_integrate, grid_propagate_3d.f90:106-232) ----
#ifdef HYP_TILE_STATS
        dbg_outer++;
#endif
#pragma unroll 1
        for (int k = 0; k < HYP_TILE_STEPS; k++) {
#ifdef HYP_TILE_STATS
            { unsigned long long mw = __ballot(st == LS_WALK); if (mw) { dbg_wsteps++; dbg_lsteps += __popcll(mw); } }
#endif
            if (st == LS_WALK) {
                // rare events wait for the service phase: they would cost every step of the wave
                // their full code path for one or two lanes
                double tmin; int im[3]; bool found;
                bool simple = find_wall_ahead(Wl, r, v, inv, iu, smask, cell, tmin, im, found) && v_ok;
                if (pre) {      // wall found by geo_find_wall in the service phase (a wave-uniform test around this -- one lane in ~1e4 steps has such a wall -- measured 222.2 against 219.5 ms in round 6: not done)
                    tmin = hit_t; im[0] = (hit_lc & 3) - 1; im[1] = ((hit_lc >> 2) & 3) - 1; im[2] = ((hit_lc >> 4) & 3) - 1;
                    found = true; simple = true;
                }
                if (g.countdown == 0) st = LS_CHECK;
                else if (!simple) st = LS_SLOW;
                else if (!found) { cnt.killed_geo++; st = LS_DEAD; }
                else {
                    pre = false;
                    g.countdown--;
                    const int lc = (cell.ic[2] * BY + cell.ic[1]) * BX + cell.ic[0];
                    double rho[ND], chi_rho;
#pragma unroll
                    for (int d = 0; d < ND; d++) {
                        rho[d] = dens[lc * ND + d];
                        chi_rho = d == 0 ? chi[0] * rho[0] : chi_rho + chi[d] * rho[d];      // (the reference's sum starts at +0: the same value, a signed zero aside, which no comparison below tells apart)
                    }
                    const double tau_cell = chi_rho * tmin;
                    cnt.crossings++;
                    bool reabs = false;
                    if (any_intersect && tau_cell < tau_req - tau_ach) { t_ach += tmin; reabs = t_ach > t_src; }
                    if (reabs) st = LS_REABS;
                    else if (tau_cell < tau_req - tau_ach) {
#pragma unroll
                        for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
                        tau_ach += tau_cell;
#pragma unroll
                        for (int d = 0; d < ND; d++) if (rho[d] > 0.0) TILE_DEPOSIT(&accum[lc * ND + d], tmin * ke[d]);
                        geo_advance(P, r, cell, im);
                        // the extents are clipped to the grid, so leaving the grid is leaving the brick (one unsigned comparison per axis)
                        if ((unsigned)cell.ic[0] >= (unsigned)ex || (unsigned)cell.ic[1] >= (unsigned)ey || (unsigned)cell.ic[2] >= (unsigned)ez) {
                            const int gx = cell.ic[0] + x0, gy = cell.ic[1] + y0, gz = cell.ic[2] + z0;
                            st = (gx < 0 || gx >= gn1 || gy < 0 || gy >= gn2 || gz < 0 || gz >= gn3) ? LS_DEAD : LS_LEFT;      // geo_escaped, on copies of the grid's size
                        }
                    } else {
                        st = LS_HIT; hit_t = tmin; hit_tau = tau_cell; hit_lc = lc;      // finished in the service phase
                    }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 27 && nb_cnt[threadIdx.x]) {
        const int dx = (int)threadIdx.x % 3 - 1, dy = ((int)threadIdx.x / 3) % 3 - 1, dz = (int)threadIdx.x / 9 - 1;
        atomicAdd(&counts[tk.brick + dx + T.nbx * (dy + T.nby * dz)], nb_cnt[threadIdx.x]);
    }
    tile_walk_publish_lists(T, ctl, tk, ilist, dlist, n_int_l, n_dead_l, pub_base);
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
                if (val != 0.0) hyp_atomic_add_g(&sum[gidx * ND + d], val);
            }
        }
    }
#ifdef HYP_TILE_STATS
    if (__lane_id() == 0) {
        atomicAdd(&ctl->dbg[0], dbg_outer); atomicAdd(&ctl->dbg[1], dbg_wsteps); atomicAdd(&ctl->dbg[2], dbg_lsteps);
        atomicAdd(&ctl->dbg[3], 1ull);
        atomicAdd(&ctl->dbg[6], dbg_service); atomicAdd(&ctl->dbg[7], dbg_nservice); atomicAdd(&ctl->dbg[8], (unsigned long long)(clock64() - dbg_t0));
        atomicAdd(&ctl->dbg[10], dbg_wb); atomicAdd(&ctl->dbg[11], dbg_claim); atomicAdd(&ctl->dbg[12], dbg_nwb); atomicAdd(&ctl->dbg[13], dbg_nclaim);
        atomicAdd(&ctl->dbg[14], dbg_ncheck); atomicAdd(&ctl->dbg[15], dbg_chk);
        if (threadIdx.x == 0) { atomicAdd(&ctl->dbg[4], 1ull); atomicAdd(&ctl->dbg[5], (unsigned long long)tk.len); }
    }
#endif
#undef SLOT
    block_tally_flush(P, ctl, red, cnt, finished);
}
