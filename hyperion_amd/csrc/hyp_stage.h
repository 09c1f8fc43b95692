// hyp_stage.h -- the imaging iteration in STAGES (do_final / propagate, iter_final.f90:60-273), for the problems
// final_kernel<.., PLAIN> covers; the deferred schedule of hyp_defer.h taken one step further.
//
// final_defer_kernel carries emission, interaction (Mueller algebra, table searches) and the cell walk in one kernel: 256
// VGPRs + spills, two waves per SIMD, and its walks -- the packet's own and the forced-first-interaction walk to the edge,
// 82 crossings per packet on BASELINE configs[3] -- run at half the rate of the peel kernel's.  Here packets live in slot
// records and a ROUND is three launches:
//   stage_event_kernel  one lane per slot: what the packet in the slot needs next -- emission into a free slot
//                       (source.f90:100-179), an interaction (dust_interact.f90:22-79), or the return to the source after
//                       the forced-first-interaction walk with the first optical depth drawn (iter_final.f90:195-209) --
//                       writes ONE PeelEvent per emission / interaction into the slot's own place in the event buffer, and
//                       prepares the next integration;
//   peel_kernel         (hyp_defer.h, unchanged) walks every (event, view) pair of the round to the observer;
//   stage_walk_kernel   only walks: lanes take slots from a queue, cross cells (defer_step: grid_integrate_noenergy, or the
//                       optical-depth sum of the forced-first walk) until the packet interacts, leaves or dies, write the
//                       walk's part of the record back and take the next slot.  No physics beyond the step: a third of
//                       the registers, and a lane never waits for an emission or an interaction of its neighbours.
// Every packet in flight is at an event when a round starts, so no lists are needed; a packet's random numbers depend on
// (seed, iteration, packet id) only and a peel-off walk's on (packet, event number, view), so the images are the sums of
// the other schedules in another order.  Rounds go on until the packet ids are used up and no slot is live.
#pragma once
#include "hyp_defer.h"

#ifndef HYP_STAGE_WALK_OCC
#define HYP_STAGE_WALK_OCC 3      // workgroups of the walk kernel per CU the register budget is set for (4: 68 spilled VGPRs on the octree)
#endif
#ifndef HYP_STAGE_REFILL
#define HYP_STAGE_REFILL 16       // idle lanes that trigger a refill in the walk kernel
#endif
#ifndef HYP_STAGE_STEPS
#define HYP_STAGE_STEPS 16        // cell crossings between two refill checks
#endif
#define HYP_STAGE_CHUNK 256       // slots a wave of the walk kernel reserves at a time

// what is in a slot
enum { SS_FREE = 0, SS_WALK = 1, SS_FF = 2, SS_INTERACT = 3, SS_FFDONE = 4, SS_FFKILLED = 5, SS_RETIRED = 6 };

// the walk's part of a slot record; the rest of the packet (SuspRec: packet, random stream, origin flags) is touched by
// stage_event_kernel only
template <int NDT, int GEOM>
struct alignas(16) StageHot {
    double r[3], v[3], tau_req, tau_ach, chi[NDT];
    Cell<GEOM> cell;
    unsigned int id_lo, id_hi, blk_b;
    int countdown, state;
};

template <int NDT, int GEOM>
__global__ __launch_bounds__(256) void stage_init_kernel(StageBuf B)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if (i < B.n_slots) ((StageHot<NDT, GEOM> *)B.hot)[i].state = SS_FREE;
}

template <int NDT, int GEOM>
__global__ __launch_bounds__(256, HYP_FINAL_WAVES) void stage_event_kernel(const DProblem *__restrict__ Pp, LaunchParams L, StageBuf B)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    StageHot<NDT, GEOM> *__restrict__ hot = (StageHot<NDT, GEOM> *)B.hot;
    SuspRec<NDT, GEOM> *__restrict__ cold = (SuspRec<NDT, GEOM> *)B.cold;
    PeelEvent<NDT, GEOM> *__restrict__ ev = (PeelEvent<NDT, GEOM> *)B.events;
    const unsigned long long slot = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    const bool valid = slot < B.n_slots;
    int st = valid ? hot[slot].state : SS_RETIRED;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;

    // packet ids for the free slots of this workgroup: one trip to the dispenser
    __shared__ int n_free;
    __shared__ unsigned long long id_base;
    __shared__ unsigned int live_wg, events_wg;
    if (threadIdx.x == 0) { n_free = 0; live_wg = 0; events_wg = 0; }
    __syncthreads();
    int rank = 0;
    if (st == SS_FREE) rank = atomicAdd(&n_free, 1);
    __syncthreads();
    if (threadIdx.x == 0) id_base = n_free ? atomicAdd(P.counter, (unsigned long long)n_free) : 0ull;
    __syncthreads();

    Packet<NDT, GEOM> p;
    Rng g;
    PeelFlags f; f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = 0;
    rng_init(g, P.seed_key, L.iter_tag, 0);
    p.inter = 1; p.tau_req = 0.0; p.tau_ach = 0.0;
    p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0; p.peel_seq = 0;
    if (st == SS_INTERACT || st == SS_FFDONE || st == SS_FFKILLED) {
        const SuspRec<NDT, GEOM> &R = cold[slot];
        p = R.p; g = R.g; f = R.f;
        const StageHot<NDT, GEOM> &H = hot[slot];       // what the walk changed
#pragma unroll
        for (int a = 0; a < 3; a++) p.r[a] = H.r[a];
        p.cell = H.cell; p.tau_ach = H.tau_ach;
        g.countdown = H.countdown; g.blk_b = H.blk_b;
    }

    // peel: 0 none, 1 after emission, 2 after interaction
    int peel = 0;
    Angle a_prev = p.a;
    double s_prev[4] = {p.s[0], p.s[1], p.s[2], p.s[3]};
    int last = LAST_SR; bool last_iso = true;

    if (st == SS_FFDONE || st == SS_FFKILLED) {
        // the optical depth to the edge is known: back to the source, first optical depth (iter_final.f90:195-209)
        const double tau_escape = p.tau_ach;
        const bool killed = st == SS_FFKILLED;
        const DSource &S = P.sources[f.source_id];
        p.r[0] = S.pos[0]; p.r[1] = S.pos[1]; p.r[2] = S.pos[2];
        geo_clear_wall(p.cell);
        (void)geo_place(P, W, p.r, p.v, p.cell);        // it did succeed when the packet was emitted
        bool sampled = false;
        if (tau_escape > 1e-10 && !killed) {
            double weight, tau;
            forced_interaction(P, tau_escape, rng_uniform(g), tau, weight);
            p.tau_req = tau; p.energy *= weight; sampled = true;
        }
        if (!sampled) p.tau_req = rng_exp(g);
        p.tau_ach = 0.0;
        begin_integrate(P, p);
        st = (p.tau_req == 0.0) ? SS_INTERACT : SS_WALK;
    } else if (st == SS_INTERACT) {
        if ((long long)p.inter == P.n_inter_max + 1) { cnt.killed_int++; st = SS_FREE; }
        else {
            int scattered, dust_id;
            bool ok = interact<NDT, GEOM>(P, p, g, cnt, scattered, dust_id, false);
            f.dust_id = dust_id;
            if (scattered) { f.scattered = 1; f.n_scat++; last = LAST_DS; last_iso = false; }
            else { f.scattered = 0; f.reprocessed = 1; last = LAST_DE; last_iso = true; }
            bool killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered);
            if (killed) st = SS_FREE;
            else { p.inter++; peel = 2; }
        }
    } else if (st == SS_FREE) {
        const unsigned long long id = id_base + (unsigned long long)rank;
        if (id >= L.end_id || *((volatile int *)P.err) != 0) st = SS_RETIRED;
        else {
            rng_init(g, P.seed_key, L.iter_tag, id);
            int source_id = 0;
            Angle src_normal;
            bool ok = emit_packet<NDT, GEOM, true>(P, W, p, g, cnt, source_id, src_normal);
            f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
            p.reabs = 0; p.peel_seq = 0; p.inter = 1;
            if (ok) { peel = 1; last = LAST_SR; last_iso = true; a_prev = p.a; s_prev[0] = p.s[0]; s_prev[1] = p.s[1]; s_prev[2] = p.s[2]; s_prev[3] = p.s[3]; }
            // (a packet emitted outside the grid is gone: the slot stays free)
        }
    }

    const bool do_peel = peel != 0 && (!P.peel_scattered_only || (peel == 2 && last == LAST_DS));
    if (valid) {
        PeelEvent<NDT, GEOM> &E = ev[slot];
        if (do_peel) {
            E.r[0] = p.r[0]; E.r[1] = p.r[1]; E.r[2] = p.r[2]; E.nu = p.nu; E.energy = p.energy;
            E.a_prev = a_prev;
            E.s_prev[0] = s_prev[0]; E.s_prev[1] = s_prev[1]; E.s_prev[2] = s_prev[2]; E.s_prev[3] = s_prev[3];
#pragma unroll
            for (int d = 0; d < NDT; d++) E.chi[d] = p.chi[d];
            E.id = ((unsigned long long)g.id_hi << 32) | g.id_lo;
            E.peel_seq = p.peel_seq;
            E.code = 1 | (last << 1) | ((last_iso ? 1 : 0) << 3);
            E.f = f;
            E.cell = p.cell;
            p.peel_seq++;
        } else E.code = 0;
    }
    if (peel == 1) {
        // first propagation after emission: iter_final.f90:191-209
        if (geo_escaped(P, p.cell)) st = SS_FREE;
        else if (P.forced_first) {
            p.tau_ach = 0.0; p.tau_req = 0.0;
            geo_begin(p.r, p.v, p.cell);
            st = SS_FF;
        } else {
            p.tau_req = rng_exp(g); p.tau_ach = 0.0;
            begin_integrate(P, p);
            st = (p.tau_req == 0.0) ? SS_INTERACT : SS_WALK;
        }
    } else if (peel == 2) {
        p.tau_req = rng_exp(g); p.tau_ach = 0.0;
        begin_integrate(P, p);
        st = (p.tau_req == 0.0) ? SS_INTERACT : SS_WALK;
    }

    if (valid) {
        StageHot<NDT, GEOM> &H = hot[slot];
        if (st == SS_WALK || st == SS_FF || st == SS_INTERACT) {
            SuspRec<NDT, GEOM> &R = cold[slot];
            R.p = p; R.g = g; R.f = f;
#pragma unroll
            for (int a = 0; a < 3; a++) { H.r[a] = p.r[a]; H.v[a] = p.v[a]; }
            H.tau_req = p.tau_req; H.tau_ach = p.tau_ach;
#pragma unroll
            for (int d = 0; d < NDT; d++) H.chi[d] = p.chi[d];
            H.cell = p.cell;
            H.id_lo = g.id_lo; H.id_hi = g.id_hi; H.blk_b = g.blk_b; H.countdown = g.countdown;
        }
        H.state = st;
    }
    const unsigned long long m_live = __ballot(valid && (st == SS_WALK || st == SS_FF || st == SS_INTERACT));
    const unsigned long long m_ev = __ballot(valid && do_peel);
    if (__lane_id() == 0) { if (m_live) atomicAdd(&live_wg, (unsigned int)__popcll(m_live)); if (m_ev) atomicAdd(&events_wg, (unsigned int)__popcll(m_ev)); }
    double e = wave_sum(cnt.energy_current);
    double kg = wave_sum((double)cnt.killed_geo);
    double ki = wave_sum((double)cnt.killed_int);
    double ni = wave_sum((double)cnt.interactions);
    if (__lane_id() == 0) {
        if (e != 0.0) unsafeAtomicAdd(&P.tail[TAIL_ENERGY], e);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
        if (ki != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], ki);
        if (ni != 0.0) unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], ni);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (live_wg) atomicAdd(&B.ctl->n_live, (unsigned long long)live_wg);
        if (events_wg) atomicAdd(&B.ctl->n_events, (unsigned long long)events_wg);
    }
}

template <int NDT, int GEOM>
__global__ __launch_bounds__(256, HYP_STAGE_WALK_OCC) void stage_walk_kernel(const DProblem *__restrict__ Pp, StageBuf B, uint32_t iter_tag)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    StageHot<NDT, GEOM> *__restrict__ hot = (StageHot<NDT, GEOM> *)B.hot;
    const unsigned int lane = __lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    Packet<NDT, GEOM> p;          // r, v, cell, tau_req, tau_ach, chi are used
    Rng g; g.key0 = P.seed_key; g.key1 = iter_tag; g.id_lo = g.id_hi = 0; g.blk_a = 0; g.blk_b = 0; g.have_a = 0; g.buf_a = 0.0; g.countdown = 0;
    p.tau_req = 0.0; p.tau_ach = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++) { p.r[a] = 0.0; p.v[a] = 0.0; }
    p.v[2] = 1.0;
#pragma unroll
    for (int d = 0; d < NDT; d++) p.chi[d] = 0.0;
    double inv[3] = {1.0, 1.0, 1.0};
    bool v_ok = false;
    int st = ST_DONE;             // ST_DONE: idle; ST_WALK / ST_FF: walking
    unsigned long long slot = 0;
    unsigned long long q_next = 0, q_end = 0;       // the wave's reserved slots
    bool exhausted = B.n_slots == 0;

    for (;;) {
        const unsigned long long m_idle = __ballot(st == ST_DONE);
        const unsigned long long m_walk = __ballot(st == ST_WALK || st == ST_FF);
        if (!exhausted && (__popcll(m_idle) >= HYP_STAGE_REFILL || !m_walk)) {
            // hand slots to the idle lanes; a slot whose packet does not walk this round is skipped
            unsigned long long mask = m_idle;
            while (mask) {
                if (q_next >= q_end) {
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(&B.ctl->walk_cursor, (unsigned long long)HYP_STAGE_CHUNK);
                    b = __shfl(b, 0, 64);
                    if (b >= B.n_slots) { exhausted = true; break; }
                    q_next = b; q_end = b + HYP_STAGE_CHUNK < B.n_slots ? b + HYP_STAGE_CHUNK : B.n_slots;
                }
                const unsigned long long avail = q_end - q_next;
                const unsigned int rank = __popcll(mask & lt);
                const bool mine = ((mask >> lane) & 1ull) && rank < avail;
                bool walks = false;
                if (mine) {
                    slot = q_next + rank;
                    const StageHot<NDT, GEOM> &H = hot[slot];
                    const int s = H.state;
                    if (s == SS_WALK || s == SS_FF) {
#pragma unroll
                        for (int a = 0; a < 3; a++) { p.r[a] = H.r[a]; p.v[a] = H.v[a]; }
                        p.tau_req = H.tau_req; p.tau_ach = H.tau_ach;
#pragma unroll
                        for (int d = 0; d < NDT; d++) p.chi[d] = H.chi[d];
                        p.cell = H.cell;
                        g.id_lo = H.id_lo; g.id_hi = H.id_hi; g.blk_b = H.blk_b; g.countdown = H.countdown;
                        if (GEOM == GEOM_OCT) {
                            v_ok = true;
#pragma unroll
                            for (int a = 0; a < 3; a++) { inv[a] = 1.0 / p.v[a]; v_ok = v_ok && (p.v[a] == 0.0 || fabs(p.v[a]) >= 0x1p-400); }
                        }
                        st = s == SS_FF ? ST_FF : ST_WALK;
                        walks = true;
                    }
                }
                const unsigned long long taken = __ballot(mine);
                q_next += __popcll(taken);
                mask &= ~__ballot(walks);           // lanes that got a walking packet are served; the others look at the next slots
                if (!taken) break;
            }
        }
        if (!__ballot(st == ST_WALK || st == ST_FF)) { if (exhausted) break; else continue; }

#pragma unroll 1
        for (int k = 0; k < HYP_STAGE_STEPS; k++) {
            if (st == ST_WALK || st == ST_FF) {
                const int s2 = defer_step<NDT, GEOM>(P, W, p, g, cnt, st == ST_FF, inv, v_ok);
                if (s2 != ST_WALK && s2 != ST_FF) {
                    StageHot<NDT, GEOM> &H = hot[slot];
#pragma unroll
                    for (int a = 0; a < 3; a++) H.r[a] = p.r[a];
                    H.tau_ach = p.tau_ach; H.cell = p.cell; H.blk_b = g.blk_b; H.countdown = g.countdown;
                    H.state = s2 == ST_NEED_INTERACT ? SS_INTERACT : s2 == ST_FF_DONE ? SS_FFDONE : s2 == ST_FF_KILLED ? SS_FFKILLED : SS_FREE;
                    st = ST_DONE;
                } else st = s2;
            }
        }
    }
    double cr = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    if (lane == 0) {
        if (cr != 0.0) unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], cr);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
    }
}
