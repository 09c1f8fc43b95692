// hyp_imaging.hip -- the imaging iterations: final (inline, deferred or tiled peel-off), raytracing, monochromatic (see hyp_engine.h)
#include "hyp_engine.h"

// ---- imaging iteration -------------------------------------------------------

// Buffers of the deferred peel-off, sized for `lanes` lanes of the propagation grid.  Returns nonzero when they cannot be had
// (the caller then peels off inline).
static int defer_buffers(hyp_handle h, const DeferKernels &dk, size_t lanes, uint64_t n_local)
{
    const size_t waves = lanes / 64;
    // `peel_events` is the ceiling; a small iteration does not need it (8 events per packet in one round, more rounds
    // beyond that) and the buffer only grows
    size_t cap = (size_t)h->peel_events;
    // (at least 4 Mi slots, ~0.9 GB: a packet of an optically thick run leaves thousands of events, and every round costs a host
    // synchronisation and three sort launches -- 2e4 packets with 1e4 events each took 1 413 rounds with the 8-per-packet rule alone)
    const size_t want = n_local > (1ull << 40) ? cap : std::max<size_t>((size_t)n_local * 8, (size_t)1 << 22);
    if (want < cap && !h->peel_events_exact) cap = want;
    cap = (cap + HYP_PEEL_CHUNK - 1) / HYP_PEEL_CHUNK * HYP_PEEL_CHUNK;
    if (h->d_peel_events && h->peel_event_bytes == dk.event_bytes && h->peel_lanes >= lanes &&
        (h->peel_events_exact ? h->peel_cap == cap : h->peel_cap >= cap)) return 0;
    free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
    h->peel_cap = 0;
    bool ok = hipMalloc(&h->d_peel_events, cap * dk.event_bytes) == hipSuccess;
    while (!ok && !h->peel_events_exact && cap > ((size_t)1 << 20)) {        // a smaller buffer means more rounds, not another schedule
        (void)hipGetLastError();
        cap = (cap / 2 + HYP_PEEL_CHUNK - 1) / HYP_PEEL_CHUNK * HYP_PEEL_CHUNK;
        ok = hipMalloc(&h->d_peel_events, cap * dk.event_bytes) == hipSuccess;
    }
    for (int i = 0; i < 2 && ok; i++)
        ok = hipMalloc(&h->d_peel_susp[i], lanes * dk.susp_bytes) == hipSuccess &&
             hipMalloc((void **)&h->d_peel_ret[i], waves * 2 * sizeof(unsigned long long)) == hipSuccess;
    if (ok && !h->d_peel_ctl)
        ok = hipMalloc((void **)&h->d_peel_ctl, sizeof(PeelCtl)) == hipSuccess && hipHostMalloc((void **)&h->h_peel_ctl, sizeof(PeelCtl)) == hipSuccess &&
             hipHostMalloc((void **)&h->h_peel_counter, sizeof(unsigned long long)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
        return 1;
    }
    h->peel_cap = cap; h->peel_event_bytes = dk.event_bytes; h->peel_lanes = lanes;
    return 0;
}

// Rounds of {propagate, peel} until every packet id has been used and no packet is left set aside (hyp_defer.h).
// the event buffer and, where the memory is there, the tables of the sorted peel-off
static void defer_setup_buffers(hyp_handle h, const DeferKernels &dk, DeferBuf &B)
{
    B.events = h->d_peel_events; B.cap = h->peel_cap; B.ctl = h->d_peel_ctl;
    B.susp[0] = h->d_peel_susp[0]; B.susp[1] = h->d_peel_susp[1]; B.ret[0] = h->d_peel_ret[0]; B.ret[1] = h->d_peel_ret[1];
    B.order = nullptr; B.keys = nullptr; B.bins = nullptr; B.n_bins = 0; B.ff = nullptr; B.cur = 0; B.direct = nullptr;
    h->last_direct_memo = 0;
    if (h->direct_memo && dk.direct && h->hp.n_sources > 0 && h->hp.n_views_total > 0 && !h->hp.peel_scattered_only) {
        // direct light of the point sources: one walk per (source, view) instead of one per packet (hyp_defer.h: direct_column_kernel)
        const size_t n = (size_t)h->hp.n_sources * (size_t)h->hp.n_views_total;
        if (h->direct_cap < n) {
            free_dev(h->d_direct);
            h->direct_cap = 0;
            if (hipMalloc((void **)&h->d_direct, n * sizeof(DirectCol)) == hipSuccess) h->direct_cap = n;
            else { (void)hipGetLastError(); h->d_direct = nullptr; }
        }
        if (h->direct_cap >= n) {
            hipLaunchKernelGGL(dk.direct, dim3((unsigned)std::min<size_t>((n + 63) / 64, 1024)), dim3(64), lds_bytes(h->hp), h->stream, (const DProblem *)h->d_problem, h->d_direct);
            B.direct = h->d_direct;
            h->last_direct_memo = 1;
        }
    }
    if (h->peel_sort && dk.sort_hist && h->peel_cap < 0xffffffffull) {
        // sorted peel-off: order + keys per event slot, counts | offsets per bin; without the memory the events are taken as written
        if (h->peel_sort_cap < h->peel_cap) {
            free_dev(h->d_peel_order); free_dev(h->d_peel_keys);
            h->peel_sort_cap = 0;
            if (hipMalloc((void **)&h->d_peel_order, sizeof(unsigned int) * h->peel_cap) == hipSuccess &&
                hipMalloc((void **)&h->d_peel_keys, sizeof(unsigned int) * h->peel_cap) == hipSuccess) h->peel_sort_cap = h->peel_cap;
            else { (void)hipGetLastError(); free_dev(h->d_peel_order); free_dev(h->d_peel_keys); }
        }
        if (!h->d_peel_bins && hipMalloc((void **)&h->d_peel_bins, sizeof(unsigned int) * 2 * HYP_SORT_MAX_BINS) != hipSuccess) { (void)hipGetLastError(); h->d_peel_bins = nullptr; }
        if (h->peel_sort_cap >= h->peel_cap && h->d_peel_bins) {
            B.order = h->d_peel_order; B.keys = h->d_peel_keys; B.bins = h->d_peel_bins;
            B.n_bins = (unsigned int)std::max<unsigned long long>(1ull, std::min<unsigned long long>(HYP_SORT_MAX_BINS, h->hp.n_cells));
        }
    }
}

// forced first interaction: every packet's emission, escape walk and first optical depth ahead of the rounds, one record per id
// (128 bytes at one dust species; without the memory the propagation kernel does it all itself).  Sets B.ff where it ran.
static void defer_ff_prepass(hyp_handle h, const DeferKernels &dk, const LaunchParams &L, DeferBuf &B, size_t lds)
{
    B.ff = nullptr;
    h->last_ff_prepass = 0;
    const unsigned long long n_ids = L.end_id - L.first_id;
    if (!(h->ff_prepass && h->hp.forced_first && dk.ff_walk && n_ids > 0)) return;
    const size_t want = (size_t)n_ids * dk.ff_bytes;
    if (h->ff_cap < want) {
        free_dev(h->d_ff);
        h->ff_cap = 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want < free_b / 2 && hipMalloc(&h->d_ff, want) == hipSuccess) h->ff_cap = want;
        else { (void)hipGetLastError(); h->d_ff = nullptr; }
    }
    if (h->ff_cap < want) return;
    B.ff = h->d_ff;
    (void)hipMemsetAsync(&h->d_peel_ctl->ff_cursor, 0, sizeof(unsigned long long), h->stream);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)dk.ff_walk, 256, lds) != hipSuccess || occ <= 0) occ = 2;
    const unsigned long long need = (n_ids + 255) / 256;
    const unsigned ff_blocks = (unsigned)std::min<unsigned long long>((unsigned long long)h->n_cu * occ, need);
    hipLaunchKernelGGL(dk.ff_walk, dim3(ff_blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L, B);
    h->last_ff_prepass = 1;
}

// the events in the buffer: ordered by cell (where the tables are there), then every (event, view) pair walked to the observer
static void defer_peel_events(hyp_handle h, const DeferKernels &dk, const DeferBuf &B, unsigned peel_blocks, size_t lds, uint32_t iter_tag)
{
    if (B.order) {
        const unsigned sort_blocks = (unsigned)((h->peel_cap + HYP_SORT_PER_WG - 1) / HYP_SORT_PER_WG);
        (void)hipMemsetAsync(B.bins, 0, sizeof(unsigned int) * B.n_bins, h->stream);
        hipLaunchKernelGGL(dk.sort_hist, dim3(sort_blocks), dim3(256), 0, h->stream, (const DProblem *)h->d_problem, B);
        hipLaunchKernelGGL(dk.sort_scan, dim3(1), dim3(1024), 0, h->stream, B);
        hipLaunchKernelGGL(dk.sort_scatter, dim3(sort_blocks), dim3(256), 0, h->stream, (const DProblem *)h->d_problem, B);
    }
    hipLaunchKernelGGL(h->inside_observers ? dk.peel_inside : dk.peel, dim3(peel_blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, B, iter_tag);
}

static int run_deferred_rounds(hyp_handle h, const DeferKernels &dk, const LaunchParams &L, unsigned blocks, size_t lds, bool ff_ahead = true)
{
    DeferBuf B;
    defer_setup_buffers(h, dk, B);
    if (ff_ahead) defer_ff_prepass(h, dk, L, B, lds);
    else h->last_ff_prepass = 0;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)(h->inside_observers ? dk.peel_inside : dk.peel), 256, lds) != hipSuccess || occ <= 0) occ = 2;
    const unsigned peel_blocks = (unsigned)(h->n_cu * occ);
    int idle_rounds = 0;
    for (int round = 0;; round++) {
        B.cur = round & 1;
        hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, B.cur, round == 0 ? 1 : 0);
        hipLaunchKernelGGL(B.ff ? dk.propagate_pre : dk.propagate, dim3(blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L, B);
        defer_peel_events(h, dk, B, peel_blocks, lds, L.iter_tag);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return h->set_error(std::string("deferred imaging launch: ") + hipGetErrorString(e));
        (void)hipMemcpyAsync(h->h_peel_ctl, h->d_peel_ctl, sizeof(PeelCtl), hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(h->h_peel_counter, h->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream);
        e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("propagation failed: ") + hipGetErrorString(e));
        const PeelCtl &C = *h->h_peel_ctl;
        h->last_defer_rounds = round + 1;
        h->last_defer_events += C.written;
        if (C.n_susp[B.cur] == 0 && C.n_ret[B.cur] == 0 && *h->h_peel_counter >= L.end_id) break;
        // a round without a single event can happen (all packets in flight left the grid), a long run of them cannot
        idle_rounds = C.written == 0 ? idle_rounds + 1 : 0;
        if (idle_rounds > 64) return h->set_error("deferred peel-off makes no progress (event buffer too small?)");
        int err = 0;
        if (hipMemcpy(&err, h->d_err, sizeof err, hipMemcpyDeviceToHost) != hipSuccess || err != 0) break;     // reported by hyp_final_accumulators
    }
    return 0;
}

// The imaging iteration with its propagation half on the slot-pool schedule of the Lucy iteration (hyp_tiled.h: IMG kernels):
// emission and the forced first interaction ahead of everything (ff_walk_kernel), then generations of interact / emit / sort /
// WALK FROM LDS -- the packets' own walks start at interaction points in random directions, which is what made them slow in
// final_defer_kernel (scattered loads) --, events appended to the buffer and peeled (sorted) when it could overflow and at the
// end.  Returns 0 done, 1 error, 2 not applicable (no tiled schedule for the grid, tables too large, forced first interaction
// without room for its records): the caller runs the rounds of hyp_defer.h instead.
// `gen`: the problem has general sources (a surface that emits with limb darkening and re-absorbs packets; dk holds the GEN kernels): the IMG
// kernels' GEN instances emit with the general emitter and make the escape walk of the forced first interaction themselves (no pre-pass: a
// packet starts on its source's surface), re-emissions from a source leave their event in tile_interact, and the walks watch t_src as in the
// Lucy iteration (round 6; before, such problems ran final_defer_kernel<.., GEN>'s own walks from global memory: 1.25e10 crossings/s on the
// 400 x 200 spherical grid against the brick walk's 4.3e10).
static int run_tiled_imaging(hyp_handle h, const DeferKernels &dk, const LaunchParams &L, size_t lds, uint64_t n_local, bool gen)
{
    const DProblem &P = h->hp;
    const TileKernels K = pick_tile_kernels(h->n_dust, P.grid_type);
    if (!K.walk || !K.interact_img || !K.emit_img || K.event_bytes != dk.event_bytes) return 2;
    if (gen && (!K.interact_img_gen || !K.emit_img_gen)) return 2;
    if (P.grid_type == 1 && car_tile_bricks(P, h->n_dust) < 0) return 2;
    if ((P.grid_type == 5 || P.grid_type == 6) && polar_tile_bricks(P, h->n_dust, h->pt_lds_kb) < 0) return 2;
    if (P.grid_type == 2 && !h->oct_neighbours) return 2;
    if (P.grid_type == 2 || P.grid_type == 3 || P.grid_type == 4) {
        const int rc = P.grid_type == 4 ? build_amr_slabs(h) : P.grid_type == 3 ? build_vor_clusters(h) : build_oct_clusters(h);
        if (rc) { h->err.clear(); return 2; }
        if (sync_problem(h)) return 1;
    }
    DeferBuf B;
    defer_setup_buffers(h, dk, B);
    {
        // the event buffer must hold a few generations' worth of events (one per slot and generation at most); decided BEFORE the
        // pre-pass runs: it counts its crossings and kills, and the caller's fall-back runs it again
        const long long want_slots = h->tile_slots > 0 ? h->tile_slots : tiled_imaging_slots(P);
        const unsigned long long slots = (unsigned long long)std::min<long long>(want_slots, (long long)n_local) + 4096ull;
        if (B.cap < 3ull * (slots + slots / 8)) return 2;
    }
    if (gen) { B.ff = nullptr; h->last_ff_prepass = 0; }
    else {
        defer_ff_prepass(h, dk, L, B, lds);
        if (P.forced_first && !B.ff) return 2;          // (no room for the records: the pre-pass did not run)
    }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)(h->inside_observers ? dk.peel_inside : dk.peel), 256, lds) != hipSuccess || occ <= 0) occ = 2;
    const unsigned peel_blocks = (unsigned)(h->n_cu * occ);
    hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, 0, 1);
    const uint32_t iter_tag = L.iter_tag;
    auto flush = [&]() -> int {
        // (every pool's stream is idle here)
        defer_peel_events(h, dk, B, peel_blocks, lds, iter_tag);
        (void)hipMemcpyAsync(h->h_peel_ctl, h->d_peel_ctl, sizeof(PeelCtl), hipMemcpyDeviceToHost, h->stream);
        hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, 0, 0);
        const hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("tiled imaging: peel-off failed: ") + hipGetErrorString(e));
        h->last_defer_rounds++;
        h->last_defer_events += B.order ? h->h_peel_ctl->n_sorted : h->h_peel_ctl->reserved;
        return 0;
    };
    // End-game (round 6): with no packet id left and at most one packet per lane of the deferred schedule's grid in flight, the live
    // slots are resumed by final_defer_kernel like packets a round set aside, and run to their ends in its rounds of {propagate,
    // sort, peel} -- instead of generations of four launches per pool for a handful of packets (an optically thick model with a high
    // albedo: hundreds of interactions per packet after the last id was handed out).
    int occ_p = 0;
    const DeferKernel prop = B.ff ? dk.propagate_pre : dk.propagate;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_p, (const void *)prop, 256, lds) != hipSuccess || occ_p <= 0) occ_p = 2;
    const unsigned prop_blocks = (unsigned)std::min<size_t>((size_t)h->n_cu * occ_p, h->peel_lanes / 256);
    TiledEndGame eg;
    eg.max_packets = (uint64_t)prop_blocks * 256ull;
    eg.run = [&]() -> int {
        // the tiled schedule took its ids from its own dispenser: the deferred kernel's must read "none left"
        const unsigned long long none = L.end_id;
        if (hipMemcpyAsync(h->d_counter, &none, sizeof none, hipMemcpyHostToDevice, h->stream) != hipSuccess) return h->set_error("tiled imaging: end-game set-up failed");
        int idle_rounds = 0;
        for (int round = 0;; round++) {
            DeferBuf R = B; R.cur = round & 1;
            hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, R.cur, 0);       // (round 0 resumes what tile_to_susp_kernel left in the other parity)
            hipLaunchKernelGGL(prop, dim3(prop_blocks), dim3(256), lds, h->stream, (const DProblem *)h->d_problem, L, R);
            defer_peel_events(h, dk, R, peel_blocks, lds, iter_tag);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return h->set_error(std::string("tiled imaging end-game launch: ") + hipGetErrorString(e));
            (void)hipMemcpyAsync(h->h_peel_ctl, h->d_peel_ctl, sizeof(PeelCtl), hipMemcpyDeviceToHost, h->stream);
            e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) return h->set_error(std::string("tiled imaging end-game failed: ") + hipGetErrorString(e));
            const PeelCtl &C = *h->h_peel_ctl;
            h->last_defer_rounds++;
            h->last_defer_events += C.written;
            if (C.n_susp[R.cur] == 0 && C.n_ret[R.cur] == 0) break;
            idle_rounds = C.written == 0 ? idle_rounds + 1 : 0;
            if (idle_rounds > 64) return h->set_error("tiled imaging end-game makes no progress (event buffer too small?)");
            int err = 0;
            if (hipMemcpy(&err, h->d_err, sizeof err, hipMemcpyDeviceToHost) != hipSuccess || err != 0) break;
        }
        hipLaunchKernelGGL(dk.reset, dim3(1), dim3(1), 0, h->stream, h->d_peel_ctl, 0, 1);
        return 0;
    };
    h->last_end_game = 0;
    h->tiled_img_gen = gen;
    const int rc = launch_tiled(h, L.first_id, n_local, iter_tag, &B, flush, h->img_end_game && prop && prop_blocks > 0 ? &eg : nullptr);
    h->tiled_img_gen = false;
    h->last_tiled_imaging = rc == 0 ? 1 : 0;
    return rc ? 1 : 0;
}

int hyp_final_launch(hyp_handle h, uint64_t first_id, uint64_t n_local)
{
    if (!h) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    if (P.n_sources == 0 && n_local > 0) return h->set_error("no sources set up - need sources for last iteration");      // setup_rt.f90:236
    double *tail;
    if (h->d_img_accum) {
        hipError_t e = hipMemsetAsync(h->d_img_accum, 0, sizeof(double) * h->img_accum_n, h->stream);
        if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(images): ") + hipGetErrorString(e));
        tail = h->d_img_accum + (h->img_accum_n - TAIL_SIZE);
    } else {
        hipError_t e = hipMemsetAsync(h->d_accum + h->n_elem, 0, sizeof(double) * TAIL_SIZE, h->stream);
        if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(tail): ") + hipGetErrorString(e));
        tail = h->d_accum + h->n_elem;
    }
    P.tail = tail; P.sum = h->d_accum; P.n_copies = 1;
    if (mrw_prepare(h)) return 1;
    if (sync_problem(h)) return 1;
    unsigned long long first = first_id;
    hipError_t e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));
    LucyKernel k = pick_final_kernel(h->n_dust, h->hp.grid_type, h->plain_imaging && !h->inside_observers && !h->hp.mono_which ? 1 : h->lean_imaging && !h->hp.mono_which ? 2 : 0);
    // deferred peel-off where the plain kernel applies and there is something to peel into (hyp_defer.h)
    const bool gen = !h->plain_imaging && h->gen_defer && h->gen_defer_opt;
    bool deferred = (h->plain_imaging || gen) && !h->hp.mono_which && h->defer_peel && P.n_peeled > 0 && P.n_views_total > 0 && !h->reproducible;
    DeferKernels dk;
    std::memset(&dk, 0, sizeof dk);
    if (deferred) dk = pick_defer_kernels(h->n_dust, h->hp.grid_type);
    if (deferred && gen) { dk.propagate = h->cfg.mrw ? dk.propagate_gen_mrw : dk.propagate_gen; dk.peel = dk.peel_gen; dk.direct = nullptr; }      // (a source may stand in the way of another's direct light)
    if (deferred && (!dk.propagate || !dk.peel)) deferred = false;
    const size_t lds = lds_bytes(P);
    int bpc = h->blocks_per_cu;
    if (bpc <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, deferred ? (const void *)dk.propagate : (const void *)k, 256, lds) != hipSuccess || occ <= 0) occ = 2;
        bpc = occ;
    }
    long long blocks = (long long)h->n_cu * bpc;
    if (deferred && defer_buffers(h, dk, (size_t)blocks * 256, n_local)) deferred = false;      // no memory for the buffers: peel off inline
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    if (h->reproducible) blocks = 1;          // one wave, inline peel-off: see hyp_engine.h
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = 0x10000u;
    int chunk = h->chunk;
    if (chunk <= 0) {
        unsigned long long c = n_local / ((unsigned long long)blocks * 32ull);
        if (c < 64) c = 64;
        if (c > 4096) c = 4096;
        chunk = (int)c;
    }
    L.chunk = chunk;
    L.interact_threshold = h->final_interact_threshold >= 0 ? h->final_interact_threshold : (deferred ? 16 : 32);
    L.emit_threshold = h->final_emit_threshold >= 0 ? h->final_emit_threshold : (!deferred ? 32 : h->hp.grid_type == 1 ? 16 : 48);
    h->last_defer_rounds = 0; h->last_defer_events = 0;
    h->last_tiled_imaging = 0;
    if (deferred) {
        (void)hipEventRecord(h->ev0, h->stream);
        // large launches of problems whose grid has a tiled schedule: the propagation half on it (defer_peel = 2 forces, 3 forbids)
        int rc = 2;
        // ... where flights are long: the tiled schedule trades a generation (four launches per pool, a record written and read) per
        // flight for walks from LDS.  A thick scattering medium -- 5 crossings per flight on a 64^3 grid at tau = 6, albedo 0.9 --
        // runs 0.162 s on it against 0.122 s on the deferred rounds (4e6 packets, profiles/r06_tiled_log.md); configs[3] has 29.
        const bool long_flights = h->lucy_cross_per_flight <= 0.0 || h->lucy_cross_per_flight >= 12.0;
        if ((!gen || !h->cfg.mrw) && !h->inside_observers && h->defer_peel != 3 && (h->defer_peel == 2 || (n_local >= 4000000ull && long_flights))) rc = run_tiled_imaging(h, dk, L, lds, n_local, gen);
        if (rc == 1) return 1;
        if (rc == 0) {
            (void)hipEventRecord(h->ev1, h->stream);
            h->final_pending = true;
            h->pending_packets = n_local;
            return 0;
        }
        if (run_deferred_rounds(h, dk, L, (unsigned)blocks, lds, !gen)) return 1;
        (void)hipEventRecord(h->ev1, h->stream);
        h->final_pending = true;
        h->pending_packets = n_local;
        return 0;
    }
    (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(h->reproducible ? 64 : 256), lds, h->stream, (const DProblem *)h->d_problem, L);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("final_kernel launch: ") + hipGetErrorString(e));
    h->final_pending = true;
    h->pending_packets = n_local;
    return 0;
}

int hyp_final_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->final_pending) return h->set_error("hyp_final_accumulators called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("propagation failed: ") + hipGetErrorString(e));
    (void)hipEventElapsedTime(&h->last_propagate_ms, h->ev0, h->ev1);
    if (check_device_error(h)) { h->final_pending = false; return 1; }
    if (h->d_img_accum) {
        if (device_ptr) *device_ptr = h->d_img_accum;
        if (n_doubles) *n_doubles = h->img_accum_n;
    } else {
        if (device_ptr) *device_ptr = h->d_accum + h->n_elem;
        if (n_doubles) *n_doubles = TAIL_SIZE;
    }
    return 0;
}

int hyp_final_finish(hyp_handle h, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->final_pending) return h->set_error("hyp_final_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->final_pending = false;
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->hp.tail, sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st;
    std::memset(&st, 0, sizeof st);
    st.energy_current = tail[TAIL_ENERGY];
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS]; st.interactions = (uint64_t)tail[TAIL_INTERACTIONS];
    st.n_packets = h->pending_packets;
    // peeled_images_adjust_scale(energy_total/energy_current): iter_final.f90:142-143
    if (st.energy_current > 0.0) {
        double scale = h->energy_total / st.energy_current;
        for (size_t g = 0; g < h->h_peeled.size(); g++) {
            // binned_images_adjust_scale (images_binned.f90:34-38): x n_theta x n_phi
            const double sc = (int)g == h->hp.binned ? scale * (double)h->hp.n_bin_theta * (double)h->hp.n_bin_phi : scale;
            if (h->sed_n[g]) image_scale_kernel<<<256, 256, 0, h->stream>>>(h->d_img_accum + h->sed_off[g], h->sed_n[g], sc);
            if (h->img_n[g]) image_scale_kernel<<<1024, 256, 0, h->stream>>>(h->d_img_accum + h->img_off[g], h->img_n[g], sc);
        }
        e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("image scaling failed: ") + hipGetErrorString(e));
    }
    h->last_stats = st;
    if (stats) *stats = st;
    return 0;
}

int hyp_final_iteration(hyp_handle h, uint64_t n_packets, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (n_packets == 0) return 0;   // "Skipping": iter_final.f90:78-85
    if (hyp_final_launch(h, 0, n_packets)) return 1;
    if (hyp_final_accumulators(h, nullptr, nullptr)) return 1;
    return hyp_final_finish(h, stats);
}

// ---- raytracing iteration (iter_raytracing.f90) --------------------------------------------

int hyp_raytracing_launch(hyp_handle h, int which, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first)
{
    if (!h) return 1;
    if (!h->cfg.raytracing) return h->set_error("raytracing was not requested in the configuration");
    if (which < 0 || which > 1) return h->set_error("hyp_raytracing_launch: which must be 0 (sources) or 1 (dust)");
    if (!h->d_img_accum) return h->set_error("no peeled images set up");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    double *tail = h->d_img_accum + (h->img_accum_n - TAIL_SIZE);
    hipError_t e;
    if (zero_first) e = hipMemsetAsync(h->d_img_accum, 0, sizeof(double) * h->img_accum_n, h->stream);
    else if (!h->ray_pending) e = hipMemsetAsync(tail, 0, sizeof(double) * TAIL_SIZE, h->stream);
    else e = hipSuccess;
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(images): ") + hipGetErrorString(e));
    P.tail = tail; P.sum = h->d_accum; P.n_copies = 1;
    if (sync_problem(h)) return 1;
    if (!h->ray_pending) h->ray_ms = 0.f;        // hyp_last_kernel_ms after hyp_raytracing_finish: the launches of this iteration
    h->ray_pending = true;
    if (which == 0 && P.n_sources == 0) n_local = 0;       // n_raytracing_photons_sources = 0: setup_rt.f90:238
    if (n_local == 0 || n_total == 0) return 0;
    unsigned long long first = first_id;
    e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));
    RayKernel k = pick_ray_kernel(h->n_dust, h->hp.grid_type);
    const size_t lds = lds_bytes(P);
    long long blocks = (long long)h->n_cu * 2;
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    if (h->reproducible) blocks = 1;
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = which == 0 ? 0x20000u : 0x30000u;
    unsigned long long c = n_local / ((unsigned long long)blocks * 32ull);
    if (c < 64) c = 64;
    if (c > 4096) c = 4096;
    L.chunk = (int)c;
    L.interact_threshold = h->interact_threshold; L.emit_threshold = h->emit_threshold;
    (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(h->reproducible ? 64 : 256), lds, h->stream, (const DProblem *)h->d_problem, L, which, (double)n_total);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("ray_kernel launch: ") + hipGetErrorString(e));
    // the two parts share the id dispenser: finish this launch before the next one resets it
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("raytracing failed: ") + hipGetErrorString(e));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->ray_ms += ms; }
    if (check_device_error(h)) { h->ray_pending = false; return 1; }
    return 0;
}

int hyp_raytracing_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->ray_pending) return h->set_error("hyp_raytracing_accumulators called without a launched iteration");
    if (device_ptr) *device_ptr = h->d_img_accum;
    if (n_doubles) *n_doubles = h->img_accum_n;
    return 0;
}

int hyp_raytracing_finish(hyp_handle h, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->ray_pending) return h->set_error("hyp_raytracing_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->ray_pending = false;
    h->last_propagate_ms = h->ray_ms; h->last_finish_ms = 0.f;
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->d_img_accum + (h->img_accum_n - TAIL_SIZE), sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st;
    std::memset(&st, 0, sizeof st);
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS];
    if (stats) *stats = st;
    return 0;
}

int hyp_raytracing_iteration(hyp_handle h, uint64_t n_sources, uint64_t n_dust, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (hyp_raytracing_launch(h, 0, 0, n_sources, n_sources, 0)) return 1;
    if (hyp_raytracing_launch(h, 1, 0, n_dust, n_dust, 0)) return 1;
    hyp_iter_stats st;
    if (hyp_raytracing_finish(h, &st)) return 1;
    st.n_packets = n_sources + n_dust;
    if (stats) *stats = st;
    return 0;
}

// ---- monochromatic final iteration (iter_final_mono.f90) ---------------------------------------

int hyp_mono_launch(hyp_handle h, int which, int inu, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first)
{
    if (!h) return 1;
    if (!h->cfg.monochromatic) return h->set_error("monochromatic mode was not requested in the configuration");
    if (which < 0 || which > 1) return h->set_error("hyp_mono_launch: which must be 0 (sources) or 1 (dust)");
    if (inu < 0 || inu >= (int)h->frequencies.size()) return h->set_error("incorrect inu");
    if (!h->d_img_accum) return h->set_error("no peeled images set up");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    double *tail = h->d_img_accum + (h->img_accum_n - TAIL_SIZE);
    hipError_t e = hipSuccess;
    if (zero_first) e = hipMemsetAsync(h->d_img_accum, 0, sizeof(double) * h->img_accum_n, h->stream);
    else if (!h->mono_pending) e = hipMemsetAsync(tail, 0, sizeof(double) * TAIL_SIZE, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(images): ") + hipGetErrorString(e));
    if (!h->mono_pending) { std::memset(&h->mono_stats, 0, sizeof h->mono_stats); h->ray_ms = 0.f; }
    h->mono_pending = true;
    P.tail = tail; P.sum = h->d_accum; P.n_copies = 1;
    P.mono_which = 0; P.mono_inu = inu; P.mono_nu = h->frequencies[inu]; P.mono_n_total = (double)n_total;
    if (which == 0 && h->hp.n_sources == 0) n_local = 0;       // n_last_photons_sources = 0: setup_rt.f90:232
    if (n_local == 0 || n_total == 0) return sync_problem(h);
    if (which == 1) {
        // setup_monochromatic_grid_pdfs: precompute_jnu_var ran in the last finish step (jnu_id / jnu_frac are current)
        const size_t nc = h->n_cells;
        if (!h->d_mono_cdf) {
            if (hipMalloc(&h->d_mono_cdf, sizeof(double) * nc * h->n_dust) != hipSuccess ||
                hipMalloc(&h->d_mono_mean, sizeof(double) * 2 * HYP_MAXD) != hipSuccess) return h->set_error("hipMalloc(monochromatic pdfs) failed");
        }
        P.mono_which = 2;       // dust_emit_probability reads mono_inu
        if (sync_problem(h)) return 1;
        mono_weight_kernel<<<dim3(1024), dim3(256), 0, h->stream>>>((const DProblem *)h->d_problem, h->d_mono_cdf);
        mono_scan_kernel<<<dim3(h->n_dust), dim3(1024), 0, h->stream>>>(h->d_mono_cdf, nc, h->d_mono_mean);
        double mean[2 * HYP_MAXD];
        e = hipMemcpyAsync(mean, h->d_mono_mean, sizeof mean, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return h->set_error(std::string("monochromatic emission pdfs: ") + hipGetErrorString(e));
        double tot = 0.0;
        for (int d = 0; d < h->n_dust; d++) { P.mono_mean_prob[d] = mean[d]; tot += mean[d]; }
        P.mono_cdf = h->d_mono_cdf;
        if (tot == 0.0) { P.mono_which = 0; return sync_problem(h); }      // "No emission at this frequency"
    }
    P.mono_which = which + 1;
    if (sync_problem(h)) return 1;
    unsigned long long first = first_id;
    e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));
    LucyKernel k = pick_final_kernel(h->n_dust, h->hp.grid_type, 0);
    const size_t lds = lds_bytes(P);
    // problems that are plain apart from being monochromatic: the launch on the deferred schedule (hyp_defer.h: the propagation
    // kernel writes events, the peel kernel walks them sorted by cell into the launch's frequency plane); option mono_defer = 0: inline
    const bool mgen = h->mono_gen_defer && h->gen_defer_opt;
    bool deferred = (h->mono_defer || mgen) && h->mono_defer_opt && h->defer_peel && P.n_peeled > 0 && P.n_views_total > 0 && !h->reproducible;
    DeferKernels dk;
    std::memset(&dk, 0, sizeof dk);
    if (deferred) dk = pick_defer_kernels(h->n_dust, h->hp.grid_type);
    if (deferred && mgen) { dk.propagate_mono = dk.propagate_mono_gen; dk.peel = dk.peel_gen; dk.direct = nullptr; }
    if (deferred && (!dk.propagate_mono || !dk.peel)) deferred = false;
    long long blocks = (long long)h->n_cu * 2;
    if (deferred) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)dk.propagate_mono, 256, lds) != hipSuccess || occ <= 0) occ = 2;
        blocks = (long long)h->n_cu * occ;
        if (defer_buffers(h, dk, (size_t)blocks * 256, n_local * 4)) deferred = false;      // (a packet leaves tens of events: fewer rounds)
    }
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    if (h->reproducible) blocks = 1;
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = (which == 0 ? 0x40000u : 0x50000u) + (uint32_t)inu;
    unsigned long long c = n_local / ((unsigned long long)blocks * 32ull);
    if (c < 64) c = 64;
    if (c > 4096) c = 4096;
    L.chunk = (int)c;
    // the monochromatic iteration is final_kernel with inline peel-off: the imaging iteration's batch sizes
    L.interact_threshold = h->final_interact_threshold >= 0 ? h->final_interact_threshold : 32;
    L.emit_threshold = h->final_emit_threshold >= 0 ? h->final_emit_threshold : 48;
    h->last_mono_deferred = deferred ? 1 : 0;
    (void)hipEventRecord(h->ev0, h->stream);
    if (deferred) {
        L.interact_threshold = h->final_interact_threshold >= 0 ? h->final_interact_threshold : 16;
        L.emit_threshold = h->final_emit_threshold >= 0 ? h->final_emit_threshold : 48;
        dk.propagate = dk.propagate_mono;
        if (run_deferred_rounds(h, dk, L, (unsigned)blocks, lds, false)) { P.mono_which = 0; h->mono_pending = false; return 1; }
    } else
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(h->reproducible ? 64 : 256), lds, h->stream, (const DProblem *)h->d_problem, L);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("final_kernel (monochromatic) launch: ") + hipGetErrorString(e));
    // the launches share the id dispenser and the problem block: finish this one before the next changes them
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("monochromatic iteration failed: ") + hipGetErrorString(e));
    { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->ray_ms += ms; }
    P.mono_which = 0;
    if (sync_problem(h)) return 1;
    h->mono_stats.n_packets += n_local;
    if (check_device_error(h)) { h->mono_pending = false; return 1; }
    return 0;
}

int hyp_mono_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->mono_pending) return h->set_error("hyp_mono_accumulators called without a launched iteration");
    if (device_ptr) *device_ptr = h->d_img_accum;
    if (n_doubles) *n_doubles = h->img_accum_n;
    return 0;
}

int hyp_mono_finish(hyp_handle h, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->mono_pending) return h->set_error("hyp_mono_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->mono_pending = false;
    h->last_propagate_ms = h->ray_ms; h->last_finish_ms = 0.f;       // hyp_last_kernel_ms: the propagation kernels of all launches of this iteration
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->d_img_accum + (h->img_accum_n - TAIL_SIZE), sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st = h->mono_stats;
    st.energy_current = tail[TAIL_ENERGY];
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS]; st.interactions = (uint64_t)tail[TAIL_INTERACTIONS];
    if (stats) *stats = st;
    return 0;
}

int hyp_mono_iteration(hyp_handle h, uint64_t n_sources, uint64_t n_dust, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->cfg.monochromatic) return h->set_error("monochromatic mode was not requested in the configuration");
    bool first = true;
    for (int which = 0; which < 2; which++) {
        const uint64_t n = which == 0 ? n_sources : n_dust;
        for (int inu = 0; inu < (int)h->frequencies.size(); inu++) {
            if (hyp_mono_launch(h, which, inu, 0, n, n, first ? 1 : 0)) return 1;
            first = false;
        }
    }
    return hyp_mono_finish(h, stats);
}

