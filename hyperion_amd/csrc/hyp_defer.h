// hyp_defer.h -- the imaging iteration with DEFERRED peel-off (do_final / propagate, iter_final.f90:60-273, with
// peeloff_photon, images_peeled.f90:95-270), for the problems final_kernel<.., PLAIN> covers.
//
// In final_kernel a lane that has just emitted or scattered walks to the edge of the grid once per viewing angle while the
// other lanes of its wave wait: with ~3.5 such walks per packet, each ~40 cells long on BASELINE configs[3], the wave runs
// at 15-20 % lane occupancy.  Here the propagation kernel only WRITES an event (position, cell, frequency, energy, incoming
// direction and Stokes vector, origin flags: one PeelEvent) per emission / interaction, and a second kernel walks all
// (event, view) pairs to the observer, one lane each, lanes taking the next pair as soon as theirs has left the grid.
// The walk of a pair depends on the event alone -- its propagation checks draw from Philox stream 2 keyed by (packet,
// event number, view), see peel_rng -- so the images are the same sums as the inline kernel's, in a different order.
//
// Rounds.  The event buffer is finite and the number of events per packet is not bounded (a packet in an optically thick
// envelope is re-emitted thousands of times), so the host runs rounds of {propagate, peel}.  A wave of the propagation
// kernel reserves event slots HYP_PEEL_CHUNK at a time; when a reservation fails the wave stops emitting, returns the
// packet ids it had been handed but not used (DeferBuf::ret) and sets aside every packet that reaches its next
// interaction (DeferBuf::susp: packet, RNG state, origin flags).  The next round starts from those.  No event is ever
// dropped and no packet is emitted twice.
#pragma once
#include "hyp_kernels.h"

#ifndef HYP_PAIR_CHUNK_N
#define HYP_PAIR_CHUNK_N 256
#endif
constexpr int HYP_PAIR_CHUNK = HYP_PAIR_CHUNK_N;      // (event, view) pairs a wave of the peel kernel reserves at a time
#ifndef HYP_PEEL_REFILL_N
#define HYP_PEEL_REFILL_N 32
#endif
#ifndef HYP_PEEL_STEPS_N
#define HYP_PEEL_STEPS_N 16
#endif
constexpr int HYP_PEEL_REFILL = HYP_PEEL_REFILL_N;      // idle lanes that trigger a refill in the peel kernel (its set-up runs with those lanes only)
#ifndef HYP_PEEL_OCC_N
#define HYP_PEEL_OCC_N 3
#endif
constexpr int HYP_PEEL_OCC = HYP_PEEL_OCC_N;        // workgroups of the peel kernel per CU the register budget is set for
// cell crossings of the propagation kernel between two state checks (configs[3], 1e8 packets, emission and forced first interaction made
// ahead of the rounds: 8 / 12 / 16 crossings 331.8 / 325.4 / 326.0 ms)
template <int GEOM> __host__ __device__ constexpr int defer_steps() { return GEOM == GEOM_OCT ? 12 : final_walk_steps<GEOM>(); }
constexpr int HYP_PEEL_STEPS = HYP_PEEL_STEPS_N;       // cell crossings between two refill / deposit checks
// ... of the peel kernel by geometry (round 6: walks of hundreds of crossings on the 400 x 200 polar grids want longer runs of steps and an earlier refill --
// spherical 707 -> 664 ms, with a stellar sphere 802 -> 761, cylindrical 158.7 -> 154.7; the octree loses with them, 292 -> 307; Cartesian 213.5 -> 212)
#ifdef HYP_PEEL_SHAPE_ALL
template <int GEOM> __host__ __device__ constexpr int peel_steps() { return HYP_PEEL_STEPS_N; }
template <int GEOM> __host__ __device__ constexpr int peel_refill() { return HYP_PEEL_REFILL_N; }
#else
template <int GEOM> __host__ __device__ constexpr int peel_steps() { return (GEOM == GEOM_SPH || GEOM == GEOM_CYL) ? 32 : HYP_PEEL_STEPS_N; }
template <int GEOM> __host__ __device__ constexpr int peel_refill() { return (GEOM == GEOM_SPH || GEOM == GEOM_CYL) ? 16 : HYP_PEEL_REFILL_N; }
#endif

template <int NDT, int GEOM>
struct alignas(16) PeelEvent {
    double r[3], nu, energy;
    Angle a_prev;                       // direction before the scattering
    double s_prev[4];                   // Stokes vector before the scattering
    double chi[NDT];
    unsigned long long id;
    unsigned int peel_seq;
    int code;                           // 0: empty slot, else 1 | last << 1 | last_isotropic << 3
    PeelFlags f;
    Cell<GEOM> cell;
};

// What ff_walk_kernel leaves of a packet it emitted and walked to the edge of the grid ahead of the rounds
template <int NDT>
struct alignas(16) EmitRec {
    Angle a;                            // direction of emission
    double nu, energy0, energy;         // energy0: as emitted (the emission's peel-off event, energy_current); energy: with the weight of
    double tau_req;                     //   the forced first interaction; tau_req: the first optical depth (iter_final.f90:195-209)
    double chi[NDT], albedo[NDT], kappa[NDT];
    double buf_a;                       // Rng::buf_a, blk_a, blk_b, have_a, countdown after all of that
    uint32_t blk_a, blk_b;
    int code;                           // have_a | status << 1: 0 emission failed (error raised), 1 emitted outside the grid, 2 walked
    int countdown;
    int source_id, pad;
};

template <int NDT, int GEOM>
struct alignas(16) SuspRec {
    Packet<NDT, GEOM> p;
    Rng g;
    PeelFlags f;
};

template <int GEOM>
__global__ void defer_reset_kernel(PeelCtl *ctl, int cur, int first)
{
    ctl->reserved = 0; ctl->pair_cursor = 0; ctl->written = 0;
    ctl->n_susp[cur] = 0; ctl->n_ret[cur] = 0;
    if (first) { ctl->n_susp[cur ^ 1] = 0; ctl->n_ret[cur ^ 1] = 0; }
}

// Forced first interaction (iter_final.f90:191-209) needs the optical depth from the source to the edge of the grid along
// the packet's direction before the packet takes its first step: in final_kernel that is a grid_escape_tau walk of the
// lanes that have just emitted, with the rest of the wave waiting.  Here it is a state of the lane like any other walk:
// ST_FF lanes cross cells in the same loop as the ST_WALK lanes (defer_step), summing the optical depth the way
// grid_escape_tau does and never interacting; once out they go back to the source (a point: PLAIN) and sample tau.
enum { ST_FF = 8, ST_FF_DONE = 9, ST_FF_KILLED = 10 };

// walk_step<NDT, GEOM, false> without re-absorbing sources (PLAIN), plus the `ff` mode = one pass of the loop of
// escape_tau (same order of operations, same association of the optical-depth sum, same use of the check stream)
template <int NDT, int GEOM, bool REABS = false>
__device__ __forceinline__ int defer_step(const DProblem &P, const Walls &W, Packet<NDT, GEOM> &p, Rng &g, Counters &cnt, bool ff,
                                          const double inv[3], bool v_ok)
{
    const int nd = ndust<NDT>(P);
    if (g.countdown == 0) {
        g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp);
        if (!geo_check_cell(P, W, p.r, p.v, p.cell)) { cnt.killed_geo++; return ff ? ST_FF_KILLED : ST_NEED_EMIT; }
    } else g.countdown--;
    double tmin; int im[3];
    bool found;
    found = find_wall_fixed_dir<GEOM>(P, W, p.r, p.v, inv, v_ok, p.cell, tmin, im);
    if (!found) { cnt.killed_geo++; return ff ? ST_FF_KILLED : ST_NEED_EMIT; }
    const size_t base = geo_index(P, p.cell) * (size_t)nd;
    double rho[NDT];
    double chi_rho = 0.0;
#pragma unroll
    for (int d = 0; d < NDT; d++) {
        rho[d] = 0.0;
        if (d < nd) { rho[d] = hyp_ldg(P.density + base + d); chi_rho += p.chi[d] * rho[d]; }
    }
    const double tau_cell = chi_rho * tmin;
    const double tau_needed = p.tau_req - p.tau_ach;
    cnt.crossings++;
    if (ff || tau_cell < tau_needed) {
        if (REABS && P.any_intersect) { p.t_ach += tmin; if (p.t_ach > p.t_src) return ST_NEED_REEMIT; }     // re-absorbed by a source: grid_propagate_3d.f90:139-143
#pragma unroll
        for (int a = 0; a < 3; a++) p.r[a] = p.r[a] + tmin * p.v[a];
        if (ff) {
#pragma unroll
            for (int d = 0; d < NDT; d++) if (d < nd) p.tau_ach += p.chi[d] * rho[d] * tmin;
        } else p.tau_ach += tau_cell;
        geo_advance(P, p.r, p.cell, im);
        if (geo_invalid(P, p.cell)) { cnt.killed_geo++; return ff ? ST_FF_KILLED : ST_NEED_EMIT; }
        if (geo_escaped(P, p.cell)) return ff ? ST_FF_DONE : ST_ESCAPED;
        return ff ? ST_FF : ST_WALK;
    }
    const double tact = tmin * (tau_needed / tau_cell);
    if (REABS && P.any_intersect) { p.t_ach += tact; if (p.t_ach > p.t_src) return ST_NEED_REEMIT; }     // :184-188
#pragma unroll
    for (int a = 0; a < 3; a++) p.r[a] = p.r[a] + tact * p.v[a];
    p.tau_ach += tau_needed;
    geo_clear_wall(p.cell);
    return ST_NEED_INTERACT;
}

// The propagation half: final_kernel<NDT, GEOM, true> with the peel-off replaced by an event record.  FFIN = false: every escape
// walk of the forced first interaction was made ahead of the rounds (ff_walk_kernel, B.ff), the ST_FF state is compiled out.
// MONO: one launch of the monochromatic final iteration (iter_final_mono.f90:58-343; P.mono_which = 1 source packets, 2 thermal
// packets from the grid pdf) of a problem that is plain otherwise: every interaction is a scattering weighted by the albedo, the
// packet ends below mono_threshold of the energy it was emitted with (Packet::e_init, set aside with the packet between rounds),
// and the peel kernel bins every event into the launch's frequency plane (image_bin_keys reads P.mono_inu).
template <int NDT, int GEOM, bool FFIN, bool MONO = false, bool GEN = false, bool MRWF = false>
__global__ __launch_bounds__(256, HYP_FINAL_WAVES) void final_defer_kernel(const DProblem *__restrict__ Pp, LaunchParams L, DeferBuf B)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    static_assert(!(MONO || GEN) || FFIN, "the monochromatic launches and the ones with general sources emit in the kernel");
    static_assert(!MRWF || (GEN && !MONO), "the modified random walk: polychromatic launches, GEN instances");
    constexpr bool FFS = FFIN && !MONO && !GEN;       // the escape walk of the forced first interaction as a lane state (MONO / GEN: inline, see below)
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    Packet<NDT, GEOM> p;
    Rng g;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    Dispenser dsp; dsp.next = 0; dsp.end = 0;
    PeelFlags f; f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = 0;
    int st = ST_NEED_EMIT;
    int mrw_k = 1;                            // MRWF: steps of the modified random walk since the last interaction (state ST_MRW; iter_final.f90:165-183)
    double inv[3] = {1.0, 1.0, 1.0};          // octree: RN(1 / v) of the packet's direction since its last emission / interaction
    bool v_ok = false;                        //   (oct_find_wall_inv); false: geo_find_wall
    bool pool_empty = false, full = false;
    unsigned long long w_pos = 0, w_end = 0;        // the wave's reserved event slots
    unsigned int n_written = 0;
    PeelEvent<NDT, GEOM> *__restrict__ ev = (PeelEvent<NDT, GEOM> *)B.events;
    const unsigned int lane = __lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    rng_init(g, P.seed_key, L.iter_tag, 0);
    p.inter = 1; p.tau_req = 0.0; p.tau_ach = 0.0;
    p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0; p.peel_seq = 0;
    {
        // what the previous round set aside: lane i resumes packet i, wave w takes back id range w (the grid is the same in
        // every round and a lane sets aside at most one packet, a wave returns at most one range per round)
        const unsigned int gl = blockIdx.x * 256u + threadIdx.x;
        if (gl < B.ctl->n_susp[B.cur ^ 1]) {
            const SuspRec<NDT, GEOM> &R = ((const SuspRec<NDT, GEOM> *)B.susp[B.cur ^ 1])[gl];
            p = R.p; g = R.g; f = R.f;
            // (spec_idx, unused by the imaging iteration, says what the packet was set aside before: 0 an interaction, 1 its re-emission
            // by a source, 2 + k the k-th step of its modified random walk)
            // (-3: a packet on its way between two interactions, handed over by the tiled schedule's end-game, tile_to_susp_kernel)
            st = p.spec_idx == -3 ? ST_WALK : (GEN && p.spec_idx == 1) ? ST_NEED_REEMIT : (MRWF && p.spec_idx >= 2) ? ST_MRW : ST_NEED_INTERACT;
            if (MRWF && p.spec_idx >= 2) mrw_k = p.spec_idx - 2;
        }
        const unsigned int wv = gl >> 6;
        if (wv < B.ctl->n_ret[B.cur ^ 1]) { dsp.next = B.ret[B.cur ^ 1][2 * (size_t)wv]; dsp.end = B.ret[B.cur ^ 1][2 * (size_t)wv + 1]; }
    }

    for (;;) {
        if (st == ST_ESCAPED) st = ST_NEED_EMIT;
        unsigned long long m_walk = __ballot(st == ST_WALK || (FFS && st == ST_FF));
        const unsigned long long m_ffd = FFS ? __ballot(st == ST_FF_DONE || st == ST_FF_KILLED) : 0ull;
        if (FFS && m_ffd && (__popcll(m_ffd) >= L.interact_threshold || !m_walk)) {
            // the optical depth to the edge is known: back to the source, first optical depth (iter_final.f90:195-209)
            if (st == ST_FF_DONE || st == ST_FF_KILLED) {
                const double tau_escape = p.tau_ach;
                const bool killed = st == ST_FF_KILLED;
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
                st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
            }
            m_walk = __ballot(st == ST_WALK || (FFS && st == ST_FF));
        }
        unsigned long long m_int = __ballot(st == ST_NEED_INTERACT);
        unsigned long long m_emit = __ballot(st == ST_NEED_EMIT);
        unsigned long long m_re = GEN ? __ballot(st == ST_NEED_REEMIT) : 0ull;       // re-absorbed by a source with a radius: iter_final.f90:213-243
        unsigned long long m_mrw = MRWF ? __ballot(st == ST_MRW) : 0ull;
        if (!(m_walk | m_int | m_emit | m_re | m_mrw | (FFS ? __ballot(st == ST_FF_DONE || st == ST_FF_KILLED) : 0ull))) break;

        // One event slot per lane must be there before anything that peels off is started.  A wave that only wants to emit
        // does not ask while there is no packet id left to emit with: the slots then go to the waves that hold the packets
        // set aside by the round before (with a buffer of fewer chunks than waves they would otherwise never get one).
        if (!full && w_end - w_pos < 64ull &&
            (m_int || m_re || m_mrw || (m_emit && !pool_empty && (dsp.next < dsp.end || *((volatile unsigned long long *)P.counter) < L.end_id)))) {
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&B.ctl->reserved, (unsigned long long)HYP_PEEL_CHUNK);
            b = __shfl(b, 0, 64);
            if (b + HYP_PEEL_CHUNK <= B.cap) {
                if (w_pos + lane < w_end) ev[w_pos + lane].code = 0;       // the < 64 slots left of the old chunk stay empty
                w_pos = b; w_end = b + HYP_PEEL_CHUNK;
            } else full = true;
        }
        if (full) {
            // this round's buffer is full: set aside what would peel off next, return the unused ids, walk the rest out
            const unsigned long long m = m_int | m_re | m_mrw;
            if (m) {
                unsigned int base = 0;
                if (lane == (unsigned int)(__ffsll((long long)m) - 1)) base = atomicAdd(&B.ctl->n_susp[B.cur], (unsigned int)__popcll(m));
                base = __shfl(base, __ffsll((long long)m) - 1, 64);
                if (st == ST_NEED_INTERACT || (GEN && st == ST_NEED_REEMIT) || (MRWF && st == ST_MRW)) {
                    SuspRec<NDT, GEOM> &R = ((SuspRec<NDT, GEOM> *)B.susp[B.cur])[base + __popcll(m & lt)];
                    p.spec_idx = (GEN && st == ST_NEED_REEMIT) ? 1 : (MRWF && st == ST_MRW) ? 2 + mrw_k : 0;
                    R.p = p; R.g = g; R.f = f;
                    st = ST_DONE;
                }
            }
            if (st == ST_NEED_EMIT) st = ST_DONE;
            if (dsp.next < dsp.end) {
                if (lane == 0) {
                    const unsigned int i = atomicAdd(&B.ctl->n_ret[B.cur], 1u);
                    B.ret[B.cur][2 * (size_t)i] = dsp.next; B.ret[B.cur][2 * (size_t)i + 1] = dsp.end;
                }
                dsp.next = dsp.end;
            }
            pool_empty = true;
            m_int = 0; m_emit = 0; m_re = 0; m_mrw = 0;
            if (!(m_walk | (FFS ? __ballot(st == ST_FF_DONE || st == ST_FF_KILLED) : 0ull))) break;
        }

        // peel: 0 none, 1 after emission, 2 after interaction
        int peel = 0;
        int ff_status = 0; double ff_tau_req = 0.0, ff_energy = 0.0;       // FFIN = false: from the packet's EmitRec
        Angle a_prev = p.a;
        double s_prev[4] = {p.s[0], p.s[1], p.s[2], p.s[3]};
        int last = LAST_SR; bool last_iso = true;

        if (GEN && m_re && (__popcll(m_re) >= L.emit_threshold || !m_walk)) {
            // packets re-absorbed by a source are re-emitted from it, and that is peeled off like a scattering: iter_final.f90:213-243
            if (st == ST_NEED_REEMIT) {
                if ((long long)p.reabs == P.n_reabs_max) { cnt.killed_int++; st = ST_NEED_EMIT; }
                else {
                    const int inter = p.inter, reabs = p.reabs + 1, rid = p.reabs_id;
                    const unsigned int seq = p.peel_seq;
                    const double e = p.energy;
                    int source_id = 0; Angle src_normal;
                    bool ok = emit_packet<NDT, GEOM>(P, W, p, g, cnt, source_id, src_normal, rid, e);
                    p.inter = inter; p.reabs = reabs; p.peel_seq = seq;
                    f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
                    if (!ok) st = ST_NEED_EMIT;
                    else { peel = 3; last = LAST_SR; st = ST_PLACED; last_iso = false; a_prev = src_normal; }
                }
            }
            m_walk = __ballot(st == ST_WALK);
            m_int = __ballot(st == ST_NEED_INTERACT);
            m_emit = __ballot(st == ST_NEED_EMIT);
        }

        if (MRWF && m_mrw) {
            // modified random walk, one step per pass, each peeled off as isotropic emission: iter_final.f90:165-183
            if (st == ST_MRW) {
                if ((long long)mrw_k == P.n_inter_mrw_max + 1) { cnt.killed_int++; st = ST_NEED_EMIT; }
                else if (mrw_wanted(P, W, p)) {
                    a_prev = p.a; s_prev[0] = p.s[0]; s_prev[1] = p.s[1]; s_prev[2] = p.s[2]; s_prev[3] = p.s[3];
                    f.dust_id = mrw_step<NDT, GEOM, false>(P, W, p, g, nullptr);
                    mrw_k++;
                    peel = 4; last = LAST_DE; last_iso = true;
                } else {
                    p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                    begin_integrate(P, p);
                    st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                }
            }
            m_walk = __ballot(st == ST_WALK);
            m_int = __ballot(st == ST_NEED_INTERACT);
            m_emit = __ballot(st == ST_NEED_EMIT);
        }

        if (m_int && (__popcll(m_int) >= L.interact_threshold || !m_walk)) {
            if (st == ST_NEED_INTERACT) {
                if (GEN) { p.reabs = 0; a_prev = p.a; s_prev[0] = p.s[0]; s_prev[1] = p.s[1]; s_prev[2] = p.s[2]; s_prev[3] = p.s[3]; }
                if ((long long)p.inter == P.n_inter_max + 1) {
                    cnt.killed_int++; st = ST_NEED_EMIT;
                } else {
                    int scattered, dust_id;
                    // MONO: always scatter, the energy decreases by the albedo (iter_final_mono.f90:330-336)
                    bool ok = interact<NDT, GEOM>(P, p, g, cnt, scattered, dust_id, MONO);
                    f.dust_id = dust_id;
                    if (scattered) { f.scattered = 1; f.n_scat++; last = LAST_DS; last_iso = false; }
                    else { f.scattered = 0; f.reprocessed = 1; last = LAST_DE; last_iso = true; }
                    bool killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered && !MONO);
                    if (MONO && p.energy < p.e_init * P.mono_threshold) killed = true;
                    if (killed) st = ST_NEED_EMIT;
                    else { p.inter++; peel = 2; }
                }
            }
            m_walk = __ballot(st == ST_WALK || (FFS && st == ST_FF));
            m_emit = __ballot(st == ST_NEED_EMIT);
        }

        if (m_emit && !pool_empty && (__popcll(m_emit) >= L.emit_threshold || !m_walk)) {
            const bool need = st == ST_NEED_EMIT;
            unsigned long long id = 0;
            bool got = take_id(P, L, dsp, need, id);
            if (need) {
                if (!got) st = ST_DONE;
                else if constexpr (FFIN) {
                    rng_init(g, P.seed_key, L.iter_tag, id);
                    int source_id = 0;
                    Angle src_normal;
                    if (MONO && P.mono_which == 2) {
                        // thermal packets of the monochromatic iteration: iter_final_mono.f90:176-196
                        int dust_id = 0;
                        bool ok = emit_mono_dust<NDT, GEOM>(P, W, p, g, cnt, dust_id);
                        f.scattered = 0; f.reprocessed = 1; f.n_scat = 0; f.dust_id = dust_id; f.source_id = 0;
                        if (!ok) st = ST_NEED_EMIT;
                        else { peel = 1; last = LAST_DE; st = ST_PLACED; p.reabs = 0; last_iso = true; p.e_init = p.energy; }
                    } else {
                        // (MONO: the energy carries the source's emission probability at the launch's frequency; the sources are
                        // isotropic points with tabulated or blackbody spectra, the host checks)
                        bool ok = emit_packet<NDT, GEOM, GEN ? 0 : (MONO ? 3 : 1)>(P, W, p, g, cnt, source_id, src_normal);
                        f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
                        if (!ok) st = ST_NEED_EMIT;
                        else {
                            if (MONO) { p.energy = p.energy / P.mono_n_total; p.e_init = p.energy; }     // iter_final_mono.f90:113-116
                            peel = 1; last = LAST_SR; st = ST_PLACED; p.reabs = 0; last_iso = true;
                            if (GEN) {      // sources with a surface: the emission's peel-off weighs with the angle to the normal (a_prev carries it)
                                last_iso = P.sources[source_id].type == 1 || P.sources[source_id].type == 8 || P.sources[source_id].type == 4;
                                if (!last_iso) a_prev = src_normal;
                            }
                        }
                    }
                } else {
                    // the packet was emitted ahead of the rounds (ff_walk_kernel): what emit_packet<.., SIMPLE> leaves in a packet,
                    // from the record; a packet whose emission failed raised its error there (the launch stops below)
                    const EmitRec<NDT> &R = ((const EmitRec<NDT> *)B.ff)[id - L.first_id];
                    ff_status = R.code >> 1;
                    if (ff_status != 0) {
                        rng_init(g, P.seed_key, L.iter_tag, id);
                        g.buf_a = R.buf_a; g.blk_a = R.blk_a; g.blk_b = R.blk_b; g.have_a = R.code & 1; g.countdown = R.countdown;
                        const int source_id = R.source_id;
                        const DSource &S = P.sources[source_id];
                        p.r[0] = S.pos[0]; p.r[1] = S.pos[1]; p.r[2] = S.pos[2];
                        p.a = R.a;
                        angle_to_vector(p.a, p.v[0], p.v[1], p.v[2]);
                        p.s[0] = 1.0; p.s[1] = 0.0; p.s[2] = 0.0; p.s[3] = 0.0;
                        p.nu = R.nu; p.energy = R.energy0;
                        cnt.energy_current += R.energy0;
                        ff_tau_req = R.tau_req; ff_energy = R.energy;
#pragma unroll
                        for (int d = 0; d < NDT; d++) { p.chi[d] = R.chi[d]; p.albedo[d] = R.albedo[d]; p.kappa[d] = R.kappa[d]; }
                        p.emiss_dust = -1;
                        geo_clear_wall(p.cell);
                        bool placed = false;
                        if constexpr (GEOM == GEOM_VOR) {
                            if (S.type == 1 && S.vor_cell1 > 0) { p.cell.id = S.vor_cell1 - 1; placed = true; }
                        }
                        if (!placed) (void)geo_place(P, W, p.r, p.v, p.cell);       // it did succeed ahead of the rounds
                        p.inter = 1; p.peel_seq = 0; p.n_visited = 0;
                        f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
                        peel = 1; last = LAST_SR; st = ST_PLACED; p.reabs = 0; last_iso = true;
                    }
                }
            }
            if (__ballot(st == ST_DONE)) pool_empty = true;
            if (*((volatile int *)P.err) != 0) { if (st == ST_NEED_EMIT) st = ST_DONE; pool_empty = true; }
        } else if (m_emit && pool_empty) {
            if (st == ST_NEED_EMIT) st = ST_DONE;
        }

        if (__ballot(peel != 0)) {
            const bool do_peel = peel != 0 && (!P.peel_scattered_only || (peel == 2 && last == LAST_DS) || peel == 3);       // (a re-emission is peeled in any case: "a kind of scattering", iter_final.f90:226-227)
            const unsigned long long m = __ballot(do_peel);
            if (m) {
                if (do_peel) {
                    PeelEvent<NDT, GEOM> &E = ev[w_pos + __popcll(m & lt)];
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
                }
                w_pos += __popcll(m); n_written += __popcll(m);
            }
            if (peel != 0) {
                walk_reciprocals<GEOM>(P, p.v, inv, v_ok);      // the direction is new
                if (peel == 1) {
                    // first propagation after emission: iter_final.f90:191-209
                    if (geo_escaped(P, p.cell)) st = ST_ESCAPED;
                    else if (MONO || GEN) {
                        // (GEN: the packet starts on the surface of its source.)  MONO: a thermal packet starts anywhere in the grid: the escape walk is made here, from where the packet is, the
                        // way final_kernel makes it (once per packet, against ~30 forced scatterings)
                        bool sampled = false;
                        if (P.forced_first) {
                            bool killed = false;
                            const double tau_escape = escape_tau<NDT, GEOM>(P, W, p.r, p.v, p.cell, p.chi, g, cnt, killed);
                            if (tau_escape > 1e-10 && !killed) {
                                double weight, tau;
                                forced_interaction(P, tau_escape, rng_uniform(g), tau, weight);
                                p.tau_req = tau; p.energy *= weight; sampled = true;
                            }
                        }
                        if (!sampled) p.tau_req = rng_exp(g);
                        p.tau_ach = 0.0;
                        begin_integrate(P, p);
                        st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                    } else if (!FFIN) {
                        // escape walk and first optical depth (iter_final.f90:195-209) were made ahead of the rounds; the packet
                        // has not moved
                        p.tau_req = ff_tau_req; p.energy = ff_energy;
                        p.tau_ach = 0.0;
                        begin_integrate(P, p);
                        st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                    } else if (FFS && P.forced_first) {
                        p.tau_ach = 0.0; p.tau_req = 0.0;
                        geo_begin(p.r, p.v, p.cell);
                        st = ST_FF;
                    } else {
                        p.tau_req = rng_exp(g);
                        p.tau_ach = 0.0;
                        begin_integrate(P, p);
                        st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                    }
                } else if (GEN && peel == 3 && geo_escaped(P, p.cell)) {
                    st = ST_ESCAPED;
                } else if (MRWF && peel == 4) {
                    // stays in ST_MRW: the next pass decides on another step
                } else if (MRWF && peel == 2) {
                    st = ST_MRW; mrw_k = 1;
                } else {
                    p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                    begin_integrate(P, p);
                    st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                }
            }
        }

#pragma unroll 1
        for (int k = 0; k < defer_steps<GEOM>(); k++) {
            if (FFS) { if (st == ST_WALK || st == ST_FF) st = defer_step<NDT, GEOM>(P, W, p, g, cnt, st == ST_FF, inv, v_ok); }
            else if (st == ST_WALK) st = defer_step<NDT, GEOM, GEN>(P, W, p, g, cnt, false, inv, v_ok);
        }
    }

    for (unsigned long long q = w_pos + lane; q < w_end; q += 64ull) ev[q].code = 0;      // reserved, never written

    double e = wave_sum(cnt.energy_current);
    double c = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    double ki = wave_sum((double)cnt.killed_int);
    double ni = wave_sum((double)cnt.interactions);
    if (lane == 0) {
        if (n_written) atomicAdd(&B.ctl->written, (unsigned long long)n_written);
        if (e != 0.0) unsafeAtomicAdd(&P.tail[TAIL_ENERGY], e);
        if (c != 0.0) unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], c);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
        if (ki != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], ki);
        if (ni != 0.0) unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], ni);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Emission and forced first interaction ahead of the rounds.  What happens to a packet between its emission and its first
// optical depth (iter_final.f90:191-209: emit, grid_escape_tau from the source along the direction it was emitted in, the
// forced first interaction's optical depth and weight) depends on nothing but the packet's id: emission, the walk's
// propagation checks and the sampling draw from the packet's own stream.  In final_defer_kernel<.., true> those walks are
// half of the crossings, made by lanes that carry a whole packet (256 VGPRs + spills, two waves per SIMD), and emission runs
// once 48 lanes of a wave wait for it.  Here all of it is done first, by a kernel of the peel kernel's budget: a lane takes
// an id, emits, walks, samples, and leaves EmitRec[id] -- direction, frequency, opacities, energy before and after the
// weight, first optical depth, and the state of the packet's stream.  final_defer_kernel<.., false> has neither emission
// code nor the ST_FF state: a lane that takes an id loads the record, writes the emission's peel-off event and walks.
// Crossings and packets killed by the escape walk are counted here, once; energy_current by the propagation kernel.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef HYP_FF_OCC_N
#define HYP_FF_OCC_N 3
#endif
constexpr int HYP_FF_OCC = HYP_FF_OCC_N;
constexpr int HYP_FF_REFILL = 32;

template <int NDT, int GEOM>
__global__ __launch_bounds__(256, HYP_FF_OCC) void ff_walk_kernel(const DProblem *__restrict__ Pp, LaunchParams L, DeferBuf B)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    const unsigned long long n_ids = L.end_id - L.first_id;
    EmitRec<NDT> *__restrict__ out = (EmitRec<NDT> *)B.ff;
    const unsigned int lane = __lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    Counters cnt, cnt_emit;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    Packet<NDT, GEOM> p;
    Rng g;
    p.inter = 1; p.tau_req = 0.0; p.tau_ach = 0.0;
    p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0; p.peel_seq = 0;
    rng_init(g, P.seed_key, L.iter_tag, 0);
    int st = 0;                                 // 0 idle, ST_FF walking, ST_FF_DONE / ST_FF_KILLED out of the grid (record to be finished)
    double inv[3] = {1.0, 1.0, 1.0};
    bool v_ok = false;
    unsigned long long mine_id = 0, q_next = 0, q_end = 0;
    bool exhausted = n_ids == 0;

    for (;;) {
        const unsigned long long m_idle = __ballot(st != ST_FF);
        const unsigned long long m_walk = __ballot(st == ST_FF);
        const bool service = __popcll(m_idle) >= HYP_FF_REFILL || !m_walk;
        if (service && (st == ST_FF_DONE || st == ST_FF_KILLED)) {
            // the optical depth to the edge is known: first optical depth (iter_final.f90:195-209)
            const double tau_escape = p.tau_ach;
            double energy = p.energy, tau_req = 0.0;
            bool sampled = false;
            if (tau_escape > 1e-10 && st != ST_FF_KILLED) {
                double weight, tau;
                forced_interaction(P, tau_escape, rng_uniform(g), tau, weight);
                tau_req = tau; energy *= weight; sampled = true;
            }
            if (!sampled) tau_req = rng_exp(g);
            EmitRec<NDT> &R = out[mine_id];
            R.energy = energy; R.tau_req = tau_req;
            R.buf_a = g.buf_a; R.blk_a = g.blk_a; R.blk_b = g.blk_b; R.code = (g.have_a & 1) | (2 << 1); R.countdown = g.countdown;
            st = 0;
        }
        if (service && !exhausted) {
            unsigned long long mask = m_idle, q = 0;
            bool got = false;
            while (mask) {
                if (q_next >= q_end) {
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(&B.ctl->ff_cursor, (unsigned long long)HYP_PAIR_CHUNK);
                    b = __shfl(b, 0, 64);
                    if (b >= n_ids) { exhausted = true; break; }
                    q_next = b; q_end = b + HYP_PAIR_CHUNK < n_ids ? b + HYP_PAIR_CHUNK : n_ids;
                }
                const unsigned long long avail = q_end - q_next;
                const unsigned int rank = __popcll(mask & lt);
                const bool take = ((mask >> lane) & 1ull) && rank < avail;
                if (take) { q = q_next + rank; got = true; }
                const unsigned long long taken = __ballot(take);
                q_next += __popcll(taken);
                mask &= ~taken;
            }
            if (got) {
                mine_id = q;
                rng_init(g, P.seed_key, L.iter_tag, L.first_id + q);
                int source_id = 0;
                Angle src_normal;
                cnt_emit.energy_current = 0.0; cnt_emit.crossings = 0; cnt_emit.killed_geo = 0; cnt_emit.killed_int = 0; cnt_emit.interactions = 0;
                const bool ok = emit_packet<NDT, GEOM, true>(P, W, p, g, cnt_emit, source_id, src_normal);
                EmitRec<NDT> &R = out[q];
                if (!ok) R.code = 0;
                else {
                    R.a = p.a; R.nu = p.nu; R.energy0 = p.energy; R.source_id = source_id; R.pad = 0;
#pragma unroll
                    for (int d = 0; d < NDT; d++) { R.chi[d] = p.chi[d]; R.albedo[d] = p.albedo[d]; R.kappa[d] = p.kappa[d]; }
                    if (geo_escaped(P, p.cell)) {
                        R.energy = p.energy; R.tau_req = 0.0;
                        R.buf_a = g.buf_a; R.blk_a = g.blk_a; R.blk_b = g.blk_b; R.code = (g.have_a & 1) | (1 << 1); R.countdown = g.countdown;
                    } else {
                        walk_reciprocals<GEOM>(P, p.v, inv, v_ok);      // the direction is new
                        p.tau_ach = 0.0; p.tau_req = 0.0;
                        geo_begin(p.r, p.v, p.cell);
                        st = ST_FF;
                    }
                }
            }
        }
        if (!__ballot(st != 0)) { if (exhausted) break; else continue; }

#pragma unroll 1
        for (int k = 0; k < HYP_PEEL_STEPS; k++) {
            if (st == ST_FF) st = defer_step<NDT, GEOM>(P, W, p, g, cnt, true, inv, v_ok);
        }
    }

    double cr = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    if (lane == 0) {
        if (cr != 0.0) unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], cr);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Sorted peel-off.  The events of a round are written in the order the propagation waves reach them: neighbouring lanes of
// a peel wave then walk through unrelated parts of the grid, and every load of the walk (next cell, its record, its
// density) is 64 different cache lines -- the imaging kernels are bound by exactly that, the L1's look-ups per scattered
// load, not by the latency of the chain (profiles/r03_tiled_log.md).  Ordered by the cell they happened in, the events of
// a wave start in the same few cells and, for one view, walk the same way out: the lanes touch the same lines (the
// direct light of a point source, one event per packet, is 64 times the same walk) and finish together.  The order is a
// counting sort of the event slots by key = cell index scaled to <= 4096 bins (indices follow the tree / the grid's
// rows, so a bin is a compact region): histogram + keys, scan, scatter; the peel kernel takes pair p as view p / n,
// event order[p % n].  Which pair a lane walks changes nothing about the walk (peel_rng is keyed by packet, event, view).
// ---------------------------------------------------------------------------------------------------------------------

template <int NDT, int GEOM>
__global__ __launch_bounds__(256) void peel_sort_hist_kernel(const DProblem *__restrict__ Pp, DeferBuf B)
{
    __shared__ unsigned int hist[HYP_SORT_MAX_BINS];
    const DProblem &P = *Pp;
    const PeelEvent<NDT, GEOM> *__restrict__ ev = (const PeelEvent<NDT, GEOM> *)B.events;
    unsigned long long n_slots = B.ctl->reserved;
    if (n_slots > B.cap) n_slots = B.cap;
    const unsigned long long i0 = (unsigned long long)blockIdx.x * HYP_SORT_PER_WG;
    if (i0 >= n_slots) return;
    for (unsigned int b = threadIdx.x; b < B.n_bins; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const unsigned long long i1 = i0 + HYP_SORT_PER_WG < n_slots ? i0 + HYP_SORT_PER_WG : n_slots;
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        unsigned int key = HYP_SORT_EMPTY;
        if (ev[i].code != 0) {
            const unsigned long long idx = (unsigned long long)geo_index(P, ev[i].cell);
            key = idx >= P.n_cells ? B.n_bins - 1u : (unsigned int)((idx * (unsigned long long)B.n_bins) / P.n_cells);
            atomicAdd(&hist[key], 1u);
        }
        B.keys[i] = key;
    }
    __syncthreads();
    for (unsigned int b = threadIdx.x; b < B.n_bins; b += blockDim.x) if (hist[b]) atomicAdd(&B.bins[b], hist[b]);
}

// one workgroup: offsets = exclusive scan of the counts, the counts become the cursors of the scatter, n_sorted = their sum
static __global__ __launch_bounds__(1024) void peel_sort_scan_kernel(DeferBuf B)
{
    __shared__ unsigned int part[1024];
    const unsigned int per = (B.n_bins + 1023u) / 1024u, b0 = threadIdx.x * per;
    unsigned int sum = 0;
    for (unsigned int b = b0; b < b0 + per && b < B.n_bins; b++) sum += B.bins[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned int d = 1; d < 1024u; d <<= 1) {
        const unsigned int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned int run = part[threadIdx.x] - sum;
    for (unsigned int b = b0; b < b0 + per && b < B.n_bins; b++) {
        const unsigned int c = B.bins[b];
        B.bins[B.n_bins + b] = run; B.bins[b] = 0;
        run += c;
    }
    if (threadIdx.x == 1023) B.ctl->n_sorted = part[1023];
}

template <int NDT, int GEOM>
__global__ __launch_bounds__(256) void peel_sort_scatter_kernel(const DProblem *__restrict__ Pp, DeferBuf B)
{
    // per workgroup: count its slots per bin, reserve that many places of the bin with ONE atomic, hand them out locally
    __shared__ unsigned int hist[HYP_SORT_MAX_BINS];
    __shared__ unsigned int base[HYP_SORT_MAX_BINS];
    unsigned long long n_slots = B.ctl->reserved;
    if (n_slots > B.cap) n_slots = B.cap;
    const unsigned long long i0 = (unsigned long long)blockIdx.x * HYP_SORT_PER_WG;
    if (i0 >= n_slots) return;
    for (unsigned int b = threadIdx.x; b < B.n_bins; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const unsigned long long i1 = i0 + HYP_SORT_PER_WG < n_slots ? i0 + HYP_SORT_PER_WG : n_slots;
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const unsigned int key = B.keys[i];
        if (key != HYP_SORT_EMPTY) atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    for (unsigned int b = threadIdx.x; b < B.n_bins; b += blockDim.x) {
        const unsigned int c = hist[b];
        base[b] = c ? B.bins[B.n_bins + b] + atomicAdd(&B.bins[b], c) : 0u;
        hist[b] = 0;
    }
    __syncthreads();
    for (unsigned long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const unsigned int key = B.keys[i];
        if (key != HYP_SORT_EMPTY) B.order[base[key] + atomicAdd(&hist[key], 1u)] = (unsigned int)i;
    }
}

// Direct light of the point sources: one lane per (source, view) makes the walk every emission event of that source makes
// towards that view in peel_kernel -- same placement, same wall search, same order of the checks -- and leaves the column density
// per species, the crossings and how the walk ends (DirectCol).  No propagation check is made here: a real walk draws the step
// of its first check when it is set up (peel_rng), and the peel kernel takes the recorded walk only for the pairs whose first check
// would fall behind the walk's last crossing (about 96 % of them at the default frequency of 1e-3 and ~40 crossings); the others are
// walked as before, check included.
template <int NDT, int GEOM>
__global__ __launch_bounds__(64) void direct_column_kernel(const DProblem *__restrict__ Pp, DirectCol *__restrict__ out)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    const int nd = ndust<NDT>(P);
    const int n_views = P.n_views_total, n = P.n_sources * n_views;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int is = k / n_views, vg = k % n_views;
        DirectCol dc;
        dc.col[0] = dc.col[1] = dc.col[2] = dc.col[3] = 0.0; dc.crossings = 0; dc.status = 0;
        int g_i = 0;
        while (g_i + 1 < P.n_peeled && vg >= P.peeled[g_i + 1].view_base) g_i++;
        const DPeeled &G = P.peeled[g_i];
        const int iv = vg - G.view_base;
        const DSource &S = P.sources[is];
        if (S.type == 1 && !G.inside_observer && !G.ignore_optical_depth && NDT <= 4) {
            Angle a_req;
            a_req.cost = G.view[4 * iv + 0]; a_req.sint = G.view[4 * iv + 1];
            a_req.cosp = G.view[4 * iv + 2]; a_req.sinp = G.view[4 * iv + 3];
            double r[3] = {S.pos[0], S.pos[1], S.pos[2]}, v[3];
            angle_to_vector(a_req, v[0], v[1], v[2]);
            double inv[3] = {1.0, 1.0, 1.0};
            bool v_ok = true;
            walk_reciprocals<GEOM>(P, v, inv, v_ok);
            Cell<GEOM> c;
            memset(&c, 0, sizeof c);
            if (geo_place(P, W, r, v, c)) {        // (not placed: the peel kernel counts a killed packet per event and does not walk)
                geo_begin(r, v, c);
                int status = 1;
                unsigned int crossings = 0;
                double col[NDT];
#pragma unroll
                for (int d = 0; d < NDT; d++) col[d] = 0.0;
                if (!geo_escaped(P, c)) for (;;) {
                    double tmin = 0.0; int im[3];
                    bool found;
                    found = find_wall_fixed_dir<GEOM>(P, W, r, v, inv, v_ok, c, tmin, im);
                    if (!found) { status = 2; break; }
                    const size_t base = geo_index(P, c) * (size_t)nd;
#pragma unroll
                    for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
#pragma unroll
                    for (int d = 0; d < NDT; d++) if (d < nd) col[d] += hyp_ldg(P.density + base + d) * tmin;
                    crossings++;
                    geo_advance(P, r, c, im);
                    if (geo_invalid(P, c)) { status = 2; break; }
                    if (geo_escaped(P, c)) break;
                    if (crossings > (1u << 26)) { status = 0; break; }
                }
                dc.status = status; dc.crossings = crossings;
#pragma unroll
                for (int d = 0; d < NDT; d++) if (d < 4) dc.col[d] = col[d];
            }
        }
        out[k] = dc;
    }
}

// The peel-off half: one lane per (event, view), peeloff<.., PLAIN> up to the walk, grid_escape_tau
// (grid_propagate_3d.f90:377-480) a few cells at a time, image_bin at the end.
// INSIDE: some peeled group has an inside observer (the walk towards the observer's position, ended at the observer: 18 spilled VGPRs
// at this budget, which the problems without one do not pay)
// GEN: the problem has sources with a surface (final_defer_kernel<.., GEN>): their emission / re-emission events weigh with the angle
// between the view and the surface normal (source_emit_peeloff, source_type.f90:512-533, 692-707), and a walk that meets a source
// ends without a deposit (grid_propagate_3d.f90:414-420)
template <int NDT, int GEOM, bool INSIDE, bool GEN = false>
__global__ __launch_bounds__(256, HYP_PEEL_OCC) void peel_kernel(const DProblem *__restrict__ Pp, DeferBuf B, uint32_t iter_tag)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    __shared__ unsigned long long img_keys[HYP_IMG_CACHE];
    __shared__ double img_vals[HYP_IMG_CACHE];
    ImgCache ic;
    img_cache_init(ic, img_keys, img_vals);
    const int nd = ndust<NDT>(P);
    const PeelEvent<NDT, GEOM> *__restrict__ ev = (const PeelEvent<NDT, GEOM> *)B.events;
    unsigned long long n_slots = B.ctl->reserved;
    if (n_slots > B.cap) n_slots = B.cap;
    const unsigned long long n_views = (unsigned long long)P.n_views_total;
    const bool sorted = B.order != nullptr;
    if (sorted) n_slots = B.ctl->n_sorted;       // only written slots are listed
    const unsigned long long n_pairs = n_slots * n_views;
    const unsigned int lane = __lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;

    // lane state: 0 idle, 1 walking, 2 out of the grid (to be binned)
    int st = 0, ig = 0;
    double r[3] = {0.0, 0.0, 0.0}, v[3] = {0.0, 0.0, 1.0}, tau = 0.0, chi[NDT];
    double inv[3] = {1.0, 1.0, 1.0};          // octree: RN(1 / v) of the walk's direction, see oct_find_wall_inv
    bool v_ok = true;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, energy = 0.0, nu_l = 0.0;
    double t_max = HYP_DBL_MAX, t_ach = 0.0;         // inside observers (images_peeled.f90:158-205): the walk ends at the observer
    long long k_img = -1, k_sed = -1;
    Cell<GEOM> c;
    Rng gp;
#pragma unroll
    for (int d = 0; d < NDT; d++) chi[d] = 0.0;
    gp.countdown = 0; gp.blk_b = 0;
    unsigned long long q_next = 0, q_end = 0;       // the wave's reserved pairs
    bool exhausted = n_pairs == 0;

    for (;;) {
        const unsigned long long m_idle = __ballot(st == 0);
        const unsigned long long m_walk = __ballot(st == 1);
        if (!exhausted && (__popcll(m_idle) >= peel_refill<GEOM>() || !m_walk)) {
            // hand pairs to the idle lanes
            unsigned long long mask = m_idle, pair = 0;
            bool got = false;
            while (mask) {
                if (q_next >= q_end) {
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(&B.ctl->pair_cursor, (unsigned long long)HYP_PAIR_CHUNK);
                    b = __shfl(b, 0, 64);
                    if (b >= n_pairs) { exhausted = true; break; }
                    q_next = b; q_end = b + HYP_PAIR_CHUNK < n_pairs ? b + HYP_PAIR_CHUNK : n_pairs;
                }
                const unsigned long long avail = q_end - q_next;
                const unsigned int rank = __popcll(mask & lt);
                const bool mine = ((mask >> lane) & 1ull) && rank < avail;
                if (mine) { pair = q_next + rank; got = true; }
                const unsigned long long taken = __ballot(mine);
                q_next += __popcll(taken);
                mask &= ~taken;
            }
            if (got) {
                // sorted: view-major, so that neighbouring lanes hold neighbouring events of ONE view
                const PeelEvent<NDT, GEOM> &E = ev[sorted ? (unsigned long long)B.order[pair % n_slots] : pair / n_views];
                const int code = E.code;
                if (code != 0) {
                    int vg = sorted ? (int)(pair / n_slots) : (int)(pair % n_views), g_i = 0;
                    while (g_i + 1 < P.n_peeled && vg >= P.peeled[g_i + 1].view_base) g_i++;
                    const DPeeled &G = P.peeled[g_i];
                    const int iv = vg - G.view_base;
                    const int last = (code >> 1) & 3;
                    const bool last_iso = (code >> 3) & 1;
                    const PeelFlags f = E.f;
                    Angle a_req;
                    a_req.cost = G.view[4 * iv + 0]; a_req.sint = G.view[4 * iv + 1];
                    a_req.cosp = G.view[4 * iv + 2]; a_req.sinp = G.view[4 * iv + 3];
                    const double nu = E.nu;
                    r[0] = E.r[0]; r[1] = E.r[1]; r[2] = E.r[2];
                    double d_obs = 0.0;
                    if ((INSIDE && G.inside_observer)) inside_direction(G, r, a_req, d_obs);      // towards the observer's position, d_obs away
                    if (last_iso) {
                        s[0] = 1.0; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
                    } else if (GEN && last == LAST_SR) {
                        // (a_prev holds the surface normal at the emission point)
                        double mu = 0.0;
                        const DSource &S = P.sources[f.source_id];
                        if (S.peeloff) {
                            double n0, n1, n2, q0, q1, q2;
                            angle_to_vector(E.a_prev, n0, n1, n2);
                            angle_to_vector(a_req, q0, q1, q2);
                            mu = q0 * n0 + q1 * n1 + q2 * n2;
                            if (mu < 0.0) mu = 0.0;
                        }
                        s[0] = (S.type == 2 && S.limb_darkening) ? 2.0 * (1.5 * mu * mu + mu) : 4.0 * mu;
                        s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
                    } else {
                        // dust_scatter_peeloff: dust_type_4elem.f90:421-444 (PLAIN: the sources are isotropic points)
                        const DDust &D = P.dust[f.dust_id];
                        s[0] = E.s_prev[0]; s[1] = E.s_prev[1]; s[2] = E.s_prev[2]; s[3] = E.s_prev[3];
                        if (last == LAST_DS) {
                            const Angle a_prev = E.a_prev;
                            Angle a_scat;
                            difference_angle(a_prev, a_req, a_scat);
                            if (a_scat.cost < D.mu_min || a_scat.cost > D.mu_max) { s[0] = s[1] = s[2] = s[3] = 0.0; }
                            else {
                                double P1, P2, P3, P4;
                                interp_P(D, a_scat.cost, nu, P1, P2, P3, P4);
                                scatter_stokes(s, a_prev, a_scat, a_req, P1, P2, P3, P4);
                            }
                        }
                    }
                    angle_to_vector(a_req, v[0], v[1], v[2]);
                    walk_reciprocals<GEOM>(P, v, inv, v_ok);
                    c = E.cell;
                    bool ok = geo_place(P, W, r, v, c);
                    if (!ok) cnt.killed_geo++;
                    const double d = (INSIDE && G.inside_observer) ? d_obs : -(v[0] * r[0] + v[1] * r[1] + v[2] * r[2]);
                    ok = ok && !(d < G.d_min || d > G.d_max);
                    const double dr0 = r[0] - G.origin[0], dr1 = r[1] - G.origin[1], dr2 = r[2] - G.origin[2];
                    double x_image = dr1 * a_req.cosp - dr0 * a_req.sinp;
                    double y_image = dr2 * a_req.sint - dr1 * a_req.cost * a_req.sinp - dr0 * a_req.cost * a_req.cosp;
                    if ((INSIDE && G.inside_observer)) inside_sky_position(G, iv, a_req, x_image, y_image);
                    bool inside = false;
                    if (G.compute_image)
                        inside = ((x_image >= G.x_min && x_image <= G.x_max) || (x_image <= G.x_min && x_image >= G.x_max)) &&
                                 ((y_image >= G.y_min && y_image <= G.y_max) || (y_image <= G.y_min && y_image >= G.y_max));
                    if (!inside && G.compute_sed) inside = x_image * x_image + y_image * y_image <= G.ap_max * G.ap_max;
                    ok = ok && inside;
                    if (ok) {
                        energy = E.energy;
                        // the bins do not depend on the attenuation (image_bin: image_type.f90:408-476); a NaN Stokes I
                        // after it is checked again when the lane deposits
                        // (filters, image_type.f90:467-475: the packet goes into every filter with a positive transmission at nu --
                        // the keys of filter 0 here, the filter index is the fastest one of the cubes)
                        image_bin_keys(P, G, nu, energy, s[0], f, x_image, y_image, iv, k_img, k_sed, G.use_filters ? 0 : -1);
                        nu_l = nu;
                        ig = g_i; tau = 0.0;
                        t_max = (INSIDE && G.inside_observer) ? d_obs : HYP_DBL_MAX; t_ach = 0.0;
#pragma unroll
                        for (int dd = 0; dd < NDT; dd++) chi[dd] = E.chi[dd];
                        peel_rng(P, gp, P.seed_key, iter_tag, E.id, E.peel_seq, vg);
                        st = 1;
                        if (G.ignore_optical_depth) st = 2;
                        else {
                            geo_begin(r, v, c);
                            if (geo_escaped(P, c)) st = 2;
                            else if (GEN && P.any_intersect) {
                                double t_source; int sid;
                                find_nearest_source(P, r, v, t_source, sid);
                                if (t_source < t_max) st = 0;          // a source in the way: nothing reaches the observer
                            } else if (B.direct && last == LAST_SR && NDT <= 4) {
                                // the direct light of a point source: the walk every packet of that source makes towards this view
                                const DirectCol dc = B.direct[(size_t)f.source_id * (size_t)n_views + (size_t)vg];
                                // (its first propagation check falls on step gp.countdown: behind the walk's crossings, or on the step that
                                // ends a status-2 walk anyway)
                                if (dc.status && (unsigned int)gp.countdown >= dc.crossings) {
#pragma unroll
                                    for (int dd = 0; dd < NDT; dd++) if (dd < nd && dd < 4) tau += chi[dd] * dc.col[dd];
                                    cnt.crossings += dc.crossings;
                                    if (dc.status == 2) { cnt.killed_geo++; st = 0; } else st = 2;
                                }
                            }
                        }
                    }
                }
            }
        }
        if (!__ballot(st != 0)) { if (exhausted) break; else continue; }

#pragma unroll 1
        for (int k = 0; k < peel_steps<GEOM>(); k++) {
            if (st == 1) {
                bool check_ok = true;
                if (gp.countdown == 0) {
                    gp.countdown = rng_check_gap(gp, P.check_p, P.check_log1mp, 2u);
                    check_ok = geo_check_cell(P, W, r, v, c);
                } else gp.countdown--;
                double tmin = 0.0; int im[3];
                bool found;
                found = find_wall_fixed_dir<GEOM>(P, W, r, v, inv, v_ok, c, tmin, im);
                if (!check_ok || !found) { cnt.killed_geo++; st = 0; }
                else {
                    const size_t base = geo_index(P, c) * (size_t)nd;
                    bool finished = false;          // grid_propagate_3d.f90:446-452
                    if (INSIDE && t_max < HYP_DBL_MAX) {
                        if (t_ach + tmin > t_max) { tmin = t_max - t_ach; finished = true; }
                        t_ach += tmin;
                    }
#pragma unroll
                    for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
#pragma unroll
                    for (int dd = 0; dd < NDT; dd++) if (dd < nd) tau += chi[dd] * hyp_ldg(P.density + base + dd) * tmin;
                    cnt.crossings++;
                    if (finished) st = 2;
                    else {
                        geo_advance(P, r, c, im);
                        if (geo_invalid(P, c)) { cnt.killed_geo++; st = 0; }
                        else if (geo_escaped(P, c)) st = 2;
                    }
                }
            }
        }

        if (__ballot(st == 2)) {
            double sa[4] = {0.0, 0.0, 0.0, 0.0};
            bool live = false;
            if (st == 2) {
                if (INSIDE && t_max < HYP_DBL_MAX) {        // 1 / (4 pi d^2) flux dilution: images_peeled.f90:236
                    const double dil = 1.0 / (4.0 * HYP_PI * (t_max * t_max));
                    s[0] = s[0] * dil; s[1] = s[1] * dil; s[2] = s[2] * dil; s[3] = s[3] * dil;
                }
                const double att = exp(-tau);
                sa[0] = s[0] * att; sa[1] = s[1] * att; sa[2] = s[2] * att; sa[3] = s[3] * att;
                live = sa[0] == sa[0];
            }
            for (int g_i = 0; g_i < P.n_peeled; g_i++) {
                const bool mine = live && ig == g_i;
                if (!__ballot(mine)) continue;
                const DPeeled &G = P.peeled[g_i];
                const size_t stride_img = (size_t)G.n_orig * G.n_view * G.n_y * G.n_x * G.n_nu, stride_sed = (size_t)G.n_orig * G.n_view * G.n_ap * G.n_nu;
                const int n_pass = G.use_filters ? G.n_nu : 1;
                for (int pass = 0; pass < n_pass; pass++) {
                    double val[4] = {sa[0] * energy, sa[1] * energy, sa[2] * energy, sa[3] * energy};
                    bool on = mine;
                    if (G.use_filters) {        // deposit_images: transmission of filter `pass` at nu, linear in the curve, 0 outside
                        double tr = 0.0;
                        if (mine) {
                            const int o0 = (int)G.filt_off[pass], o1 = (int)G.filt_off[pass + 1];
                            const double *fx = G.filt_nu + o0, *ft = G.filt_tr + o0;
                            const int j = locate(fx, o1 - o0, nu_l);
                            tr = j < 0 ? 0.0 : ft[j] + (nu_l - fx[j]) / (fx[j + 1] - fx[j]) * (ft[j + 1] - ft[j]);
                        }
                        on = mine && tr > 0.0;
                        val[0] *= tr; val[1] *= tr; val[2] *= tr; val[3] *= tr;
                    }
                    if (G.compute_image) wave_accumulate(G.img, G.uncertainties ? G.img2 : nullptr, on && k_img >= 0 ? k_img + pass : -1, stride_img, G.n_stokes, val, &ic);
                    if (G.compute_sed) wave_accumulate(G.sed, G.uncertainties ? G.sed2 : nullptr, on && k_sed >= 0 ? k_sed + pass : -1, stride_sed, G.n_stokes, val, &ic);
                }
            }
            if (st == 2) st = 0;
        }
    }

    img_cache_flush(ic);
    double cr = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    if (lane == 0) {
        if (cr != 0.0) unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], cr);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
    }
}
