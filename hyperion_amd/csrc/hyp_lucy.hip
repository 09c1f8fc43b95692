// hyp_lucy.hip -- the Lucy iteration: persistent kernel or host-driven generations of the tiled schedule, and the epilogue of an
// iteration (see hyp_engine.h)
#include "hyp_engine.h"

// ---- brick- / cluster-tiled iteration: host-driven generations ----

TileKernels pick_tile_kernels(int nd, int grid_type)
{
#ifdef HYP_VARIANT_GEOM
    return pick_tile_kernels_g<HYP_VARIANT_GEOM>(nd);
#endif
    switch (grid_type) {
    case 1: return pick_tile_kernels_g<GEOM_CAR>(nd);
    case 2: return pick_tile_kernels_g<GEOM_OCT>(nd);
    case 4: return pick_tile_kernels_g<GEOM_AMR>(nd);
    case 3: return pick_tile_kernels_g<GEOM_VOR>(nd);
    case 5: return pick_tile_kernels_g<GEOM_SPH>(nd);
    case 6: return pick_tile_kernels_g<GEOM_CYL>(nd);
    default: { TileKernels k; memset(&k, 0, sizeof k); return k; }
    }
}

void tile_shape(int nd, int &x, int &y, int &z)
{
    switch (nd) {
    case 1: x = TileShape<1>::X; y = TileShape<1>::Y; z = TileShape<1>::Z; break;
    case 2: x = TileShape<2>::X; y = TileShape<2>::Y; z = TileShape<2>::Z; break;
    case 3: x = TileShape<3>::X; y = TileShape<3>::Y; z = TileShape<3>::Z; break;
    default: x = TileShape<4>::X; y = TileShape<4>::Y; z = TileShape<4>::Z; break;
    }
}

int tile_bricks(const DProblem &P, int nd)
{
    int x, y, z;
    tile_shape(nd, x, y, z);
    return ((P.n1 + x - 1) / x) * ((P.n2 + y - 1) / y) * ((P.n3 + z - 1) / z);
}

// Bricks of a Cartesian grid on the tiled schedule, or -1 when the grid has no such schedule: more bricks than the sort's tables
// hold, or wall arrays that do not fit the LDS next to a brick (16 (n1 + n2 + n3 + 3) bytes on top of 128 KB: grids with
// n1 + n2 + n3 beyond ~1800 run on the persistent kernel in auto mode instead of failing in hipFuncSetAttribute)
// slots of the imaging iteration's tiled half when the option tile_slots is 0 (run_tiled_imaging sizes its event buffer by it)
long long tiled_imaging_slots(const DProblem &P) { return (P.grid_type == 2 || P.grid_type == 4) ? 3ll << 23 : 3ll << 22; }

long long car_tile_bricks(const DProblem &P, int nd)
{
    int x, y, z;
    tile_shape(nd, x, y, z);
    const long long nb = tile_bricks(P, nd);
    const size_t lds = lds_bytes(P) + sizeof(double) * 2 * (size_t)x * y * z * nd;
    return (nb <= HYP_TILE_MAX_BRICKS && lds + 4096 <= 160u * 1024u) ? nb : -1;
}

// Bricks of a polar grid (hyp_ptile.h): boxes of (r, theta, phi) / (w, z, phi) indices whose densities and accumulators fit `cells`
// cells of LDS.  Packets move mostly along r, so the brick is long in the first index: at most 8 cells in phi, 32 in theta / z,
// and what is left of the budget in r; theta / z shrink before r falls below 16 cells.
void polar_tile_shape(const DProblem &P, int nd, int lds_kb, int &x, int &y, int &z)
{
    const long long cells = std::max<long long>(64, (long long)lds_kb * 1024 / (16ll * nd));
    z = (int)std::min<long long>(P.n3, 8);
    y = (int)std::min<long long>(P.n2, 32);
    while ((long long)y * z * 16 > cells && y > 1) y = (y + 1) / 2;
    while ((long long)y * z * 16 > cells && z > 1) z = (z + 1) / 2;
    x = (int)std::max<long long>(1, std::min<long long>(P.n1, cells / ((long long)y * z)));
}

// Number of bricks of a polar grid on the tiled schedule, or -1 when the grid has no such schedule: more bricks than the sort's
// tables hold (HYP_TILE_MAX_BRICKS), or a brick beyond the LDS of a CU (pt_lds_kb is an option; 160 KB per CU on gfx950)
long long polar_tile_bricks(const DProblem &P, int nd, int lds_kb)
{
    int bx, by, bz;
    polar_tile_shape(P, nd, lds_kb, bx, by, bz);
    const long long nb = (long long)((P.n1 + bx - 1) / bx) * ((P.n2 + by - 1) / by) * ((P.n3 + bz - 1) / bz);
    const size_t lds = sizeof(double) * 2 * (size_t)bx * by * bz * nd;
    return (nb <= HYP_TILE_MAX_BRICKS && lds + 4096 <= 160u * 1024u) ? nb : -1;      // (4 KB: the kernel's static LDS -- counters, brick histogram)
}

// LDS of one AMR brick (hyp_atile.h): n cells, g goto entries (16 bits), w walls
size_t amr_slab_lds(size_t n, size_t g, size_t w, int nd) { return sizeof(double) * (2 * n * nd + w) + sizeof(short) * g + 16; }

// LDS of one octree cluster (hyp_otile.h): n cells of which k are refined
size_t oct_cluster_lds(size_t n, size_t k, int nd) { return (sizeof(OctCell) + sizeof(double) * 2 * nd + sizeof(short) * 6) * n + sizeof(short) * 8 * k + 16; }

// LDS of one walk workgroup
size_t tile_walk_lds(hyp_handle h, const TileKernels &K, const TileGeom &T)
{
    if (h->hp.grid_type == 3)       // cluster: its tables (VtInfo) + densities + accumulators
        return h->vt_max_lds;
    if (h->hp.grid_type == 5 || h->hp.grid_type == 6)      // polar brick: densities + accumulators
        return sizeof(double) * 2 * (size_t)T.bx * T.by * T.bz * K.nd;
    if (h->hp.grid_type == 4)       // slab: densities + accumulators + walls + goto slice
        return amr_slab_lds((size_t)T.bx, (size_t)T.by, (size_t)T.bz, K.nd);
    if (h->hp.grid_type == 2)       // cluster: cell records + densities + accumulators + children of the refined cells + neighbour table
        return oct_cluster_lds((size_t)T.bx, (size_t)T.by, K.nd);
    return lds_bytes(h->hp) + sizeof(double) * 2 * (size_t)K.bx * K.by * K.bz * K.nd;      // walls + densities + accumulators of the brick
}

// `img`: the imaging iteration on the tiled schedule -- the event buffer the IMG kernels append to; `flush` empties it (sort +
// peel_kernel) and is called with every pool's stream idle, when the buffer could overflow before the next look and at the end
int run_tiled_generations(hyp_handle h, const TileKernels &K, const TileGeom &T0, uint64_t n_local, int n_pools, const DeferBuf *img = nullptr,
                          const std::function<int()> &flush = std::function<int()>(), const TiledEndGame *end_game = nullptr)
{
    DeferBuf no_events;
    std::memset(&no_events, 0, sizeof no_events);
    const size_t lds_w = lds_bytes(h->hp);
    const size_t lds_int = lds_w;
    const TileWalkK walk_k = K.walk;
    const size_t lds_walk = tile_walk_lds(h, K, T0);
    const int grid_s = (T0.n_slots + 256 * HYP_SORT_PER_THREAD - 1) / (256 * HYP_SORT_PER_THREAD);
    const int grid_w = T0.n_slots / T0.task_size + T0.n_bricks + 1;
    // tile_interact: one workgroup per HYP_INTERACT_CHUNK entries of the pool's list (+ one for the extra list); tile_emit:
    // one per 256 free slots.  Workgroups beyond the lists' lengths (known on the device only) leave at once.
    const int grid_i = (T0.n_slots + HYP_INTERACT_CHUNK - 1) / HYP_INTERACT_CHUNK + 1;
    const int grid_e = (T0.n_slots + 255) / 256;
    if (hipFuncSetAttribute((const void *)walk_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_walk) != hipSuccess)
        return h->set_error("cannot reserve LDS for the tiled walk kernel");
    const size_t hot_sz = K.hot_bytes, cold_sz = K.cold_bytes;
    const int ri = h->hp.any_intersect ? 1 : 0, mi = h->hp.mrw ? 1 : 0;
    // Each pool of slots runs its own prepare -> sort -> walk sequence on its own stream, so the
    // latency-bound prepare of one pool overlaps the walk of the other.  The pools share only the
    // packet-id dispenser, the finished counter and the (atomic) accumulators.
    const size_t tasks_cap = (size_t)T0.n_slots / 256 + HYP_TILE_MAX_BRICKS + 2;
    int gen = 0, next_check = h->tile_poll;
    // Every packet in a slot makes one interaction per generation and is killed at n_inter_max of them (iter_lucy.f90:186-190,
    // iter_final.f90:255-259), and a slot takes a new id when its packet has ended: the generations are bounded by the interactions
    // of the packets that pass through one slot.  A sanity bound, not a schedule: a run that needs 1e6 generations is slow here
    // (launch-bound generations for a handful of packets; the Lucy iteration drains them in one kernel, the imaging iteration
    // has no such kernel) but it ends with the reference's result, not with an error.
    const long long per_slot = (long long)(n_local / ((uint64_t)T0.n_slots * (uint64_t)n_pools)) + 2;
    const long long max_gen_ll = std::max<long long>(200000, ((long long)h->cfg.n_inter_max + 2) * per_slot + 16);
    const int max_gen = (int)std::min<long long>(max_gen_ll, 2000000000ll);
    size_t n_timed = 0;
    // imaging: every generation can add at most one event per slot (plus the padding of the interaction chunks)
    const unsigned long long ev_per_gen = (unsigned long long)n_pools * ((unsigned long long)T0.n_slots + 64ull * (unsigned long long)grid_i);
    if (img) next_check = (int)std::max<unsigned long long>(1, std::min<unsigned long long>((unsigned long long)h->tile_poll, img->cap / ev_per_gen));
    for (;; gen++) {
        for (int pool = 0; pool < n_pools; pool++) {
            TileGeom T = T0; T.pool = pool;
            hipStream_t st = pool == 0 ? h->stream : h->pool_stream[pool];
            void *hot = (char *)h->d_hot + hot_sz * (size_t)pool * T.n_slots;
            void *cold = (char *)h->d_cold + cold_sz * (size_t)pool * T.n_slots;
            int *slot_brick = h->d_slot_brick + (size_t)pool * T.n_slots;
            int *order = h->d_order + (size_t)pool * T.n_slots;
            TileTask *tasks = h->d_tasks + pool * tasks_cap;
            // counts and cursors by generation parity (tile_sort_kernel); `counts` = what this generation's sort reads
            const size_t par_off = (size_t)HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS;
            const int gp = gen & 1, gn = (gen + 1) & 1;
            unsigned *counts = h->d_counts + gp * par_off + pool * HYP_TILE_MAX_BRICKS, *cursor = h->d_cursor + gp * par_off + pool * HYP_TILE_MAX_BRICKS;
            unsigned *counts_next = h->d_counts + gn * par_off + pool * HYP_TILE_MAX_BRICKS, *cursor_next = h->d_cursor + gn * par_off + pool * HYP_TILE_MAX_BRICKS;
            int *ilist = h->d_ilist + (size_t)pool * 2 * T.n_slots, *dlist = h->d_dlist + (size_t)pool * 2 * T.n_slots;      // [staging | pool-wide list]
            int *extra = h->d_extra + (size_t)pool * 3 * HYP_TILE_EXTRA;
            T.gen = gen;
            TileCount *tcount = h->d_tcount + pool * tasks_cap;
            // walk (previous generation) left per-task lists: interactions, then emission into the freed slots
            if (gen == 0) tile_init_kernel<<<(T.n_slots + 255) / 256, 256, 0, st>>>(T, h->d_ctl, tasks, tcount, dlist);
            else
                (img ? (h->tiled_img_gen ? K.interact_img_gen : K.interact_img) : K.interact[ri][mi])<<<grid_i, 256, lds_int, st>>>(h->d_problem, T, h->d_ctl, hot, cold, slot_brick, tasks, ilist, dlist,
                                                                                     tcount, counts, extra, img ? *img : no_events);
            (img ? (h->tiled_img_gen ? K.emit_img_gen : K.emit_img) : h->simple_sources && K.emit_simple ? K.emit_simple : h->ext_sources && K.emit_ext ? K.emit_ext : K.emit)<<<grid_e, 256, lds_w, st>>>(h->d_problem, T, h->d_ctl, hot, cold, slot_brick, tasks, dlist, tcount, counts, extra, img ? *img : no_events);
            tile_sort_kernel<<<grid_s, 256, sizeof(unsigned) * (2 * (size_t)T.n_bricks + 512), st>>>(T, slot_brick, counts, counts_next, cursor, cursor_next, order, tasks, h->d_ctl);
            const bool timed = h->tile_time_walk && n_timed + 2 <= 16384;
            if (timed) {
                while (h->walk_events.size() < n_timed + 2) {
                    hipEvent_t e = nullptr;
                    if (hipEventCreate(&e) != hipSuccess) return h->set_error("cannot create a timing event");
                    h->walk_events.push_back(e);
                }
                (void)hipEventRecord(h->walk_events[n_timed], st);
            }
            walk_k<<<grid_w, K.walk_threads, lds_walk, st>>>(h->d_problem, T, h->d_ctl, hot, cold, order, tasks, slot_brick, ilist, dlist, tcount, counts_next);
            if (timed) { (void)hipEventRecord(h->walk_events[n_timed + 1], st); n_timed += 2; }
        }
        if (gen + 1 >= next_check || gen > max_gen) {
            next_check = gen + 1 + h->tile_poll;
            hipError_t e = hipMemcpyAsync(h->h_ctl, h->d_ctl, sizeof(TileCtl), hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) return h->set_error(std::string("tiled generation failed: ") + hipGetErrorString(e));
            if (img) {
                // how many more generations are sure to fit the event buffer decides when to look again; with fewer than two (or at
                // the end) the buffer is emptied: sort + peel, every pool's stream idle
                for (int pool = 1; pool < n_pools; pool++)
                    if (hipStreamSynchronize(h->pool_stream[pool]) != hipSuccess) return h->set_error("tiled imaging generation failed");
                unsigned long long reserved = 0;
                (void)hipMemcpy(&reserved, &img->ctl->reserved, sizeof reserved, hipMemcpyDeviceToHost);
                if (reserved > img->cap) return h->set_error("tiled imaging: the event buffer overflowed");
                unsigned long long room = (img->cap - reserved) / ev_per_gen;
                if (h->h_ctl->n_finished >= n_local || room < 2) {
                    if (flush()) return 1;
                    room = img->cap / ev_per_gen;
                }
                // (once the last id is out the end-game is near: look every other generation, so that it starts with as many packets
                // as it may take rather than with what eight more generations leave of them)
                const unsigned long long poll = h->h_ctl->next_id >= h->h_ctl->end_id ? std::min<unsigned long long>(2, (unsigned long long)h->tile_poll) : (unsigned long long)h->tile_poll;
                next_check = gen + 1 + (int)std::max<unsigned long long>(1, std::min<unsigned long long>(poll, room));
            }
            if (h->h_ctl->n_finished >= n_local) break;
            if (gen > max_gen) return h->set_error(img ? "tiled imaging iteration did not terminate" : "tiled Lucy iteration did not terminate");
            // few packets left and no ids to hand out: finish them in one launch
            const uint64_t in_flight = n_local - h->h_ctl->n_finished;
            const uint64_t drain_at = h->tile_drain >= 0 ? (uint64_t)h->tile_drain : 1000000ull;      // (flat between 4e5 and 1.5e6 since the drain takes its packets from one list, profiles/r04_tiled_log.md)
            if (!img && h->h_ctl->next_id >= h->h_ctl->end_id && in_flight <= drain_at) {      // (the drain kernel deposits: Lucy only)
                for (int pool = 1; pool < n_pools; pool++) {
                    (void)hipEventRecord(h->ev_pool, h->pool_stream[pool]);
                    (void)hipStreamWaitEvent(h->stream, h->ev_pool, 0);
                }
                TileGeom T = T0; T.n_slots = T0.n_slots * n_pools;
                // the slots that still hold a packet as one list (in the sort's order[] array: nobody sorts any more), then the drain
                (void)hipMemsetAsync(&h->d_ctl->n_live, 0, 2 * sizeof(unsigned int), h->stream);
                tile_live_kernel<<<(T.n_slots + HYP_PREP_CHUNK - 1) / HYP_PREP_CHUNK, 256, 0, h->stream>>>(T, h->d_slot_brick, h->d_order, h->d_ctl);
                T.drain_list = h->d_order;
                const int grid_d = (int)std::min<uint64_t>((in_flight + 255) / 256 + 1, (uint64_t)h->n_cu * 8);
                K.drain[ri][mi]<<<grid_d, 256, lds_w, h->stream>>>(h->d_problem, T, h->d_ctl, h->d_hot, h->d_cold, h->d_slot_brick);
                e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) return h->set_error(std::string("tiled drain failed: ") + hipGetErrorString(e));
                break;
            }
            // imaging: the same moment -- no id left, few packets in flight -- hands the live slots to the deferred schedule's rounds
            // (tile_to_susp_kernel; every pool's stream is idle here, see above)
            if (img && end_game && K.to_susp && h->h_ctl->next_id >= h->h_ctl->end_id && in_flight <= std::min<uint64_t>(drain_at, end_game->max_packets)) {
                if (flush()) return 1;                       // what the generations left in the event buffer
                TileGeom T = T0; T.n_slots = T0.n_slots * n_pools;
                (void)hipMemsetAsync(&h->d_ctl->n_live, 0, 2 * sizeof(unsigned int), h->stream);
                tile_live_kernel<<<(T.n_slots + HYP_PREP_CHUNK - 1) / HYP_PREP_CHUNK, 256, 0, h->stream>>>(T, h->d_slot_brick, h->d_order, h->d_ctl);
                T.drain_list = h->d_order;
                DeferBuf B = *img; B.cur = 0;
                K.to_susp<<<(unsigned)((in_flight + 255) / 256 + 1), 256, 0, h->stream>>>(h->d_problem, T, h->d_ctl, h->d_hot, h->d_cold, h->d_slot_brick, B);
                if (hipGetLastError() != hipSuccess) return h->set_error("tiled imaging: the end-game's hand-over failed to launch");
                h->last_end_game = (long long)in_flight;
                if (end_game->run()) return 1;
                break;
            }
            int err = 0;
            (void)hipMemcpy(&err, h->d_err, sizeof(int), hipMemcpyDeviceToHost);
            if (err) break;
        }
    }
    h->last_generations = gen + 1;
    h->last_walk_ms = 0.0; h->last_walk_launches = 0;
    if (n_timed) {
        (void)hipDeviceSynchronize();
        for (size_t i = 0; i + 1 < n_timed; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->walk_events[i], h->walk_events[i + 1]) == hipSuccess) { h->last_walk_ms += ms; h->last_walk_launches++; }
        }
    }
#ifdef HYP_TILE_STATS
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h->h_ctl, h->d_ctl, sizeof(TileCtl), hipMemcpyDeviceToHost);
    {
        const unsigned long long *d = h->h_ctl->dbg;
        fprintf(stderr, "tile stats: generations %d, outer loops %llu, wave-steps %llu, lane-steps %llu (lane utilisation %.3f), waves %llu, "
                        "tasks %llu (mean %.0f packets), steps per outer loop %.2f\n", gen + 1, d[0], d[1], d[2], (double)d[2] / (64.0 * d[1]),
                d[3], d[4], (double)d[5] / d[4], (double)d[1] / d[0]);
        fprintf(stderr, "tile stats: service phases %llu (%.2f per outer loop), wave clocks in the service phase %.3f of the loop's (%.0f clocks per service phase, %.0f per outer loop)\n",
                d[7], (double)d[7] / d[0], (double)d[6] / d[8], (double)d[6] / d[7], (double)d[8] / d[0]);
        if (d[10]) fprintf(stderr, "tile stats: service phase = check + write-back %.3f (%.1f lanes), claim %.3f (%.1f lanes) of its clocks\n", (double)d[10] / d[6], (double)d[12] / d[7],
                           (double)d[11] / d[6], (double)d[13] / d[7]);
        if (d[15]) fprintf(stderr, "tile stats: propagation check / general wall search ran in %.3f of the service phases and took %.3f of their clocks\n", (double)d[14] / d[7], (double)d[15] / d[6]);
        if (d[17]) fprintf(stderr, "tile stats: wall search (busiest lane of each wave): %.0f clocks per search, %.3f of the loop's clocks; %.2f walls per lane-search\n",
                           (double)d[16] / d[17], (double)d[16] / d[8], (double)d[18] / std::max(1ull, d[19]));
        if (d[17] && d[21]) fprintf(stderr, "tile stats: of a search's clocks: site + q %.0f, filter loop %.0f, the winner's FP64 evaluation and the rest %.0f\n",
                                    (double)d[20] / d[17], (double)d[21] / d[17], (double)(d[16] - d[20] - d[21]) / d[17]);
        if (d[32]) fprintf(stderr, "tile stats (polar walk): %.0f clocks per wave-step (%.3f of the loop's)\n", (double)d[32] / d[1], (double)d[32] / d[8]);
        if (d[31]) fprintf(stderr, "tile stats (polar walk): wall search %.0f clocks per wave-step (%.3f of the loop's); per wave-step: inner sphere solved in %.3f (%.1f lanes), "
                                   "cones solved %.3f of 2 (%.1f lanes each), a lane on a cone wall in %.3f (%.1f lanes), lanes on a sphere wall %.1f\n",
                           (double)d[31] / d[1], (double)d[31] / d[8], (double)d[24] / d[1], (double)d[25] / std::max(1ull, d[24]), (double)d[26] / d[1], (double)d[27] / std::max(1ull, d[26]),
                           (double)d[28] / d[1], (double)d[29] / std::max(1ull, d[28]), (double)d[30] / d[1]);
        fprintf(stderr, "tile stats: wave clocks waiting at the end of the task for the workgroup's last wave %.3f of the loop's\n", (double)d[9] / d[8]);
    }
#endif
    return 0;
}

// One iteration on the slot-pool schedule: the Lucy iteration (img == nullptr; `iter_tag` = the iteration number), or the imaging
// iteration's propagation half with its events appended to *img (run_tiled_imaging below)
int launch_tiled(hyp_handle h, uint64_t first_id, uint64_t n_local, uint32_t iter_tag, const DeferBuf *img, const std::function<int()> &flush, const TiledEndGame *end_game)
{
    const DProblem &P = h->hp;
    const int nd = h->n_dust;
    const TileKernels K = pick_tile_kernels(nd, P.grid_type);
    if (!K.walk) return h->set_error("no tiled schedule for this grid geometry");
    TileGeom T;
    memset(&T, 0, sizeof T);
    if (P.grid_type == 3) {
        T.bx = h->vt_max_cells; T.by = 1; T.bz = 1;
        T.nbx = T.n_bricks = h->vt_clusters; T.nby = T.nbz = 1;
    } else if (P.grid_type == 4) {
        T.bx = h->at_max_cells; T.by = h->at_max_go; T.bz = h->at_max_walls;
        T.nbx = T.n_bricks = h->at_slabs_n; T.nby = T.nbz = 1;
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    } else if (P.grid_type == 2) {
        T.bx = h->ot_max_cells; T.by = h->ot_max_kids; T.bz = 1;
        T.nbx = T.n_bricks = h->ot_clusters; T.nby = T.nbz = 1;
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    } else if (P.grid_type == 5 || P.grid_type == 6) {
        polar_tile_shape(P, nd, h->pt_lds_kb, T.bx, T.by, T.bz);
        T.nbx = (P.n1 + T.bx - 1) / T.bx; T.nby = (P.n2 + T.by - 1) / T.by; T.nbz = (P.n3 + T.bz - 1) / T.bz;
        T.n_bricks = T.nbx * T.nby * T.nbz;
        // spherical grids: packets that have not interacted yet (radial for a central source: no cone wall is ever in reach, hyp_polar.h:
        // sph_cone_out_of_reach) sorted apart from the others, so that their waves skip the cone quadratics
        if (P.grid_type == 5 && h->pt_vsplit) {
            // (and the flights that start outwards apart from those that start inwards: the reference's find_wall leaves the inner sphere out
            // for the former, `radial`, which a wave can only skip when all its lanes do)
            const int vs = 3 * T.n_bricks <= HYP_TILE_MAX_BRICKS ? 3 : 2 * T.n_bricks <= HYP_TILE_MAX_BRICKS ? 2 : 1;
            if (vs > 1) { T.vsplit = vs; T.n_bricks *= vs; }
        }
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    } else {
        tile_shape(nd, T.bx, T.by, T.bz);
        T.nbx = (P.n1 + T.bx - 1) / T.bx; T.nby = (P.n2 + T.by - 1) / T.by; T.nbz = (P.n3 + T.bz - 1) / T.bz;
        T.n_bricks = T.nbx * T.nby * T.nbz;
        T.presort = nd == 1 && h->tile_presort ? 1 : 0;
    }
    // the sort's tables (d_counts / d_cursor, tile_sort_kernel's LDS) hold HYP_TILE_MAX_BRICKS entries per pool
    if (T.n_bricks < 1 || T.n_bricks > HYP_TILE_MAX_BRICKS) return h->set_error("grid has too many bricks for the tiled schedule");
    int n_pools = std::max(1, std::min(h->tile_pools, HYP_TILE_MAX_POOLS));
    // Pool size.  A walk workgroup loads its brick's densities into LDS and flushes its accumulators for however many packets its task
    // holds, and every generation costs four launches per pool with their tails: what counts is packets in flight per brick and per launch.
    // Rounds 2-3 settled on 6.3e6 slots (12.6e6 on trees) when the schedule was younger; round 6 swept again at 1e8 packets
    // (profiles/r06_tiled_log.md): 128^3 215.2 -> 200.1 ms at 25e6 slots, tessellation 346 -> 331, AMR 163 -> 159, spherical 1 658 -> 1 517,
    // octree flat; Cartesian grids with thousands of bricks want 50e6 (256^3 1 476 -> 994 ms, 400^3 4 357 -> 1 971, 512^3 10 236 -> 3 738;
    // 1e8 slots: no further gain).  280 B per slot at one species: 7 / 14 GB of the 288 -- within a third of the free memory.
    // The imaging iteration keeps a smaller pool: its event buffer holds three generations of events (run_tiled_imaging).
    long long def_slots = tiled_imaging_slots(P);       // (imaging: 354.5 -> 348.0 ms on 128^3 / 5e7 packets and 296.4 -> 290.5 ms on configs[3] for the doubled pool)
    if (!img) {
        def_slots = (P.grid_type == 1 && T.n_bricks >= 1024) ? 3ll << 24 : 3ll << 23;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const long long per_slot = (long long)(K.hot_bytes + K.cold_bytes + 6 * sizeof(int));
            const long long have = (long long)h->tile_slots_alloc * per_slot;       // (a pool of an earlier iteration is given back first)
            def_slots = std::min(def_slots, std::max(3ll << 21, ((long long)free_b + have) / 3 / per_slot));
        }
    }
    const long long want_slots = h->tile_slots > 0 ? h->tile_slots : def_slots;
    long long slots = std::min<long long>(want_slots, (long long)n_local);
    if (slots < 65536) n_pools = 1;
    slots = (((slots + n_pools - 1) / n_pools + 255) / 256) * 256;       // per pool
    T.n_slots = (int)slots;
    size_t all_slots = (size_t)slots * n_pools;
    // packets per walk task: 8 192; Voronoi clusters (156 KB of wall records loaded per task) 16 384 in the Lucy iteration -- round 6, 25e6 slots: 2 048 / 4 096 /
    // 8 192 / 16 384 / 32 768 / 65 536: 405 / 352 / 316-325 / 308-309 / 305-314 / 330-332 ms on configs[4]; the other geometries are best at 8 192
    T.task_size = h->tile_task <= 0 ? (P.grid_type == 3 && !img ? 16384 : 8192) : h->tile_task < 256 ? 256 : h->tile_task;
    T.iter_tag = iter_tag; T.pool = 0; T.park = h->tile_park;
    T.imaging = img ? 1 : 0;
    const size_t hot_sz = K.hot_bytes, cold_sz = K.cold_bytes;
    if (all_slots > (size_t)h->tile_slots_alloc || nd != h->tile_nd_alloc) {
        // (a pool of the default size that the device cannot give -- other handles or processes hold its memory -- is halved until it fits:
        // fewer packets in flight are slower, not wrong; a size asked for by the option tile_slots is an error when it cannot be had)
        for (;;) {
            free_dev(h->d_hot); free_dev(h->d_cold); free_dev(h->d_slot_brick); free_dev(h->d_order); free_dev(h->d_tasks);
            free_dev(h->d_ilist); free_dev(h->d_dlist); free_dev(h->d_tcount); free_dev(h->d_extra);
            h->tile_slots_alloc = 0;
            const size_t n_tasks_max = HYP_TILE_MAX_POOLS * ((size_t)all_slots / 256 + HYP_TILE_MAX_BRICKS + 2);
            if (hipMalloc(&h->d_hot, hot_sz * all_slots) == hipSuccess && hipMalloc(&h->d_cold, cold_sz * all_slots) == hipSuccess &&
                hipMalloc(&h->d_slot_brick, sizeof(int) * all_slots) == hipSuccess && hipMalloc(&h->d_order, sizeof(int) * all_slots) == hipSuccess &&
                hipMalloc(&h->d_tasks, sizeof(TileTask) * n_tasks_max) == hipSuccess &&
                hipMalloc(&h->d_ilist, sizeof(int) * 2 * all_slots) == hipSuccess && hipMalloc(&h->d_dlist, sizeof(int) * 2 * all_slots) == hipSuccess &&
                hipMalloc(&h->d_tcount, sizeof(TileCount) * n_tasks_max) == hipSuccess &&
                hipMalloc(&h->d_extra, sizeof(int) * 3 * HYP_TILE_EXTRA * HYP_TILE_MAX_POOLS) == hipSuccess) break;
            (void)hipGetLastError();
            if (h->tile_slots > 0 || slots * n_pools <= (3ll << 18)) {
                free_dev(h->d_hot); free_dev(h->d_cold); free_dev(h->d_slot_brick); free_dev(h->d_order); free_dev(h->d_tasks);
                free_dev(h->d_ilist); free_dev(h->d_dlist); free_dev(h->d_tcount); free_dev(h->d_extra);
                return h->set_error("cannot allocate the packet pool of the tiled Lucy iteration");
            }
            slots = ((slots / 2 + 255) / 256) * 256;
            T.n_slots = (int)slots;
            all_slots = (size_t)slots * n_pools;
        }
        h->tile_slots_alloc = all_slots; h->tile_nd_alloc = nd;
    }
    h->last_tile_slots = (long long)all_slots;
    if (!h->d_counts) {
        const size_t nb = sizeof(unsigned) * HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS;
        if (hipMalloc(&h->d_counts, 2 * nb) != hipSuccess || hipMalloc(&h->d_cursor, 2 * nb) != hipSuccess ||
            hipMalloc(&h->d_ctl, sizeof(TileCtl)) != hipSuccess || hipHostMalloc(&h->h_ctl, sizeof(TileCtl)) != hipSuccess)
            return h->set_error("cannot allocate the control blocks of the tiled Lucy iteration");
        if (hipEventCreateWithFlags(&h->ev_pool, hipEventDisableTiming) != hipSuccess)
            return h->set_error("cannot create the pool event of the tiled Lucy iteration");
    }
    for (int pool = 1; pool < n_pools; pool++)
        if (!h->pool_stream[pool] && hipStreamCreateWithFlags(&h->pool_stream[pool], hipStreamNonBlocking) != hipSuccess)
            return h->set_error("cannot create a stream for the tiled Lucy iteration");
    {
        static bool sort_attr = false;       // (n_bricks near HYP_TILE_MAX_BRICKS: more than the default 64 KB of dynamic LDS)
        if (!sort_attr) { (void)hipFuncSetAttribute((const void *)tile_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned) * (2 * HYP_TILE_MAX_BRICKS + 512))); sort_attr = true; }
    }
    TileCtl c0; memset(&c0, 0, sizeof(c0));
    c0.next_id = first_id; c0.end_id = first_id + n_local; c0.first_id = first_id;
    if (!img) (void)hipEventRecord(h->ev0, h->stream);        // (the imaging iteration's clock starts before its pre-pass)
    (void)hipMemsetAsync(h->d_hot, 0, hot_sz * all_slots, h->stream);          // state 0 = TS_DEAD
    (void)hipMemsetD32Async((hipDeviceptr_t)h->d_slot_brick, TILE_NEEDS_PREPARE, all_slots, h->stream);   // every slot is free
    (void)hipMemsetAsync(h->d_counts, 0, 2 * sizeof(unsigned) * HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS, h->stream);
    (void)hipMemsetAsync(h->d_cursor, 0, 2 * sizeof(unsigned) * HYP_TILE_MAX_BRICKS * HYP_TILE_MAX_POOLS, h->stream);
    (void)hipMemcpyAsync(h->d_ctl, &c0, sizeof(c0), hipMemcpyHostToDevice, h->stream);
    (void)hipStreamSynchronize(h->stream);      // c0 lives on this stack frame; the other pools start after the resets
    const int rc = run_tiled_generations(h, K, T, n_local, n_pools, img, flush, end_game);
    for (int pool = 1; pool < n_pools; pool++) {      // join the other pools into the engine's stream
        (void)hipEventRecord(h->ev_pool, h->pool_stream[pool]);
        (void)hipStreamWaitEvent(h->stream, h->ev_pool, 0);
    }
    (void)hipEventRecord(h->ev1, h->stream);
    return rc;
}

int sync_problem(hyp_handle h)
{
    hipError_t e = hipMemcpyAsync(h->d_problem, &h->hp, sizeof(DProblem), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(problem): ") + hipGetErrorString(e));
    return 0;
}

int run_finish_kernel(hyp_handle h, int mode, double scale, double *d_out_ref)
{
    FinishParams F;
    F.scale = scale; F.enforce_energy_range = h->cfg.enforce_energy_range;
    F.additional = (h->d_additional != nullptr); F.write_out = d_out_ref != nullptr; F.pad = 0;
    hipError_t e = hipMemsetAsync(h->d_energy_abs_tot, 0, sizeof(double) * HYP_MAXD, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    int blocks = h->n_cu * 8;
    size_t need = (h->n_elem + 255) / 256;
    if ((size_t)blocks > need) blocks = (int)need;
    if (blocks < 1) blocks = 1;
    // option "reproducible": update_energy_abs_tot (grid_physics_3d.f90:601-611) is a sum over all cells, and the dust packets of the
    // raytracing iteration multiply and divide by it: one wave forms it in a fixed order
    const int threads = h->reproducible ? 64 : 256;
    if (h->reproducible) blocks = 1;
    finish_kernel<<<blocks, threads, 0, h->stream>>>(h->d_problem, F, mode, h->d_specific_energy, h->d_density,
                                                 h->d_additional, h->d_jnu_id, h->d_jnu_frac, h->d_energy_abs_tot, d_out_ref,
                                                 h->d_spec, h->n_bins);
    e = hipGetLastError();
    if (e != hipSuccess) return h->set_error(std::string("finish_kernel launch: ") + hipGetErrorString(e));
    return 0;
}

// prepare_mrw + update_alpha_inv_planck at the start of an iteration (iter_lucy.f90:109-112,
// iter_final.f90:93-96); must run before sync_problem (it sets table pointers of the problem)
int mrw_prepare(hyp_handle h)
{
    DProblem &P = h->hp;
    if (!P.mrw) return 0;
    if (P.grid_type == 3) return h->set_error("distance_to_closest_wall: not implemented for Voronoi grid");
    if (!h->d_mrw_alpha) {
        if (hipMalloc(&h->d_mrw_alpha, sizeof(double) * h->n_cells) != hipSuccess ||
            hipMalloc(&h->d_mrw_diff, sizeof(double) * h->n_cells) != hipSuccess ||
            hipMalloc(&h->d_mrw_kp, sizeof(double) * h->n_elem) != hipSuccess)
            return h->set_error("cannot allocate the MRW tables");
    }
    P.mrw_alpha = h->d_mrw_alpha; P.mrw_diff = h->d_mrw_diff; P.mrw_kp = h->d_mrw_kp;     // reach the device with the caller's sync_problem
    unsigned blocks = (unsigned)std::min<size_t>((h->n_cells + 255) / 256, 65535);
    mrw_prepare_kernel<<<blocks, 256, 0, h->stream>>>(h->d_problem, h->d_specific_energy, h->d_density,
                                                      h->d_mrw_alpha, h->d_mrw_diff, h->d_mrw_kp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return h->set_error(std::string("mrw_prepare_kernel launch: ") + hipGetErrorString(e));
    return 0;
}

int check_device_error(hyp_handle h)
{
    int code = 0;
    double data[3] = {0, 0, 0};
    if (hipMemcpy(&code, h->d_err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return h->set_error("cannot read device error flag");
    if (code == ERR_NONE) return 0;
    (void)hipMemcpy(data, h->d_err_data, sizeof(data), hipMemcpyDeviceToHost);
    (void)hipMemset(h->d_err, 0, sizeof(int));
    char buf[512];
    if (code == ERR_NU_RANGE) {
        // message of src/dust/dust.f90:71
        std::snprintf(buf, sizeof buf,
                      "photon frequency (%10.4E Hz) is outside the range defined for the dust optical properties (%10.4E to %10.4E Hz)",
                      data[0], data[1], data[2]);
    } else if (code == ERR_NOT_IN_CELL) {
        // message of src/sources/source.f90:177
        std::snprintf(buf, sizeof buf,
                      "photon was not emitted inside a cell - this usually indicates that a source is not inside the grid");
    } else if (code == ERR_NEGATIVE_T) {
        // error("find_wall","negative t"), src/grid/grid_geometry_amr.f90:829
        std::snprintf(buf, sizeof buf, "negative t");
    } else if (code == ERR_RAY_GRID) {
        std::snprintf(buf, sizeof buf, "raytracing of dust emission is not available for this grid type");
    } else if (code == ERR_INTERNAL) {
        std::snprintf(buf, sizeof buf, "internal error: a work list of the tiled Lucy iteration overflowed (%g entries)", data[0]);
    } else std::snprintf(buf, sizeof buf, "device error %d", code);
    return h->set_error(buf);
}

// solve_pda (src/grid/grid_pda_3d.f90:84-172) on the device, after update_energy_abs.  The reference solves the
// diffusion equation for the mean intensity in the cells that saw fewer than max(30, 0.5 % of the mean) packets:
// with fewer than 10 000 such cells by Gaussian elimination, otherwise by Gauss-Seidel sweeps in cell order down to
// a relative change of 1e-4 per sweep, and repeats with the updated Rosseland means until the specific energy moves
// by less than 1e-5 / 1e-4.  Here: the Gauss pivot branch is a dense elimination on the device (pda_dense_* kernels;
// rows are diagonally dominant, no pivoting, zero rows skipped), the iterative branch Gauss-Seidel sweeps ordered by
// hyperplanes (pda_gs_kernel), which reproduce the reference's sequential sweeps exactly.
int solve_pda(hyp_handle h)
{
    h->pda_last_cells = 0; h->pda_last_outer = 0; h->pda_last_sweeps = 0;
    const DProblem &P = h->hp;
    if (!(P.grid_type == 1 || P.grid_type == 5 || P.grid_type == 6)) return 0;      // grid_pda_disabled.f90
    const size_t nc = h->n_cells;
    const int n_hp = P.n1 + P.n2 + P.n3 - 2;       // i1 + i2 + i3 = 0 .. n1 + n2 + n3 - 3
    if (!h->d_pda_ctl) {
        if (hipMalloc(&h->d_pda_ctl, sizeof(PdaCtl)) != hipSuccess || hipMalloc(&h->d_pda_mask, nc) != hipSuccess ||
            hipMalloc(&h->d_pda_cells, sizeof(unsigned int) * nc) != hipSuccess ||
            hipMalloc(&h->d_pda_hp, sizeof(unsigned int) * 3 * (n_hp + 1)) != hipSuccess ||
            hipMalloc(&h->d_pda_emean, sizeof(double) * nc) != hipSuccess)
            return h->set_error("cannot allocate the PDA work arrays");
    }
    unsigned int *hp_count = h->d_pda_hp, *hp_off = h->d_pda_hp + (n_hp + 1), *hp_cursor = h->d_pda_hp + 2 * (n_hp + 1);
    const double *nphot = h->d_accum + h->ext_nphot;
    const int blocks = h->n_cu * 4;
    PdaCtl ctl;
    (void)hipMemsetAsync(h->d_pda_ctl, 0, sizeof(PdaCtl), h->stream);
    (void)hipMemsetAsync(h->d_pda_hp, 0, sizeof(unsigned int) * 3 * (n_hp + 1), h->stream);
    pda_total_kernel<<<blocks, 256, 0, h->stream>>>(nphot, nc, h->d_pda_ctl);
    if (hipMemcpyAsync(&ctl, h->d_pda_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("PDA: cannot read the packet total");
    // mean_n_photons = sum(n_photons) / size(n_photons) is an INTEGER division (:99); threshold max(30, ceiling(0.005 mean))
    const double mean_n = (double)((long long)ctl.total_photons / (long long)nc);
    const double threshold = std::max(30.0, std::ceil(0.005 * mean_n));
    pda_mask_kernel<<<blocks, 256, 0, h->stream>>>(h->d_problem, nphot, threshold, h->d_specific_energy, h->d_density, h->d_pda_mask,
                                                   h->d_pda_emean, hp_count, h->d_pda_ctl);
    if (hipMemcpyAsync(&ctl, h->d_pda_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("PDA: cannot read the cell count");
    const unsigned int n_pda = ctl.n_pda;
    h->pda_last_cells = (int)n_pda;
    if (n_pda == 0) return 0;        // " [pda] not necessary for this iteration"
    pda_scan_kernel<<<1, 64, 0, h->stream>>>(hp_count, hp_off, hp_cursor, n_hp);
    pda_list_kernel<<<blocks, 256, 0, h->stream>>>(h->d_problem, h->d_pda_mask, hp_off, hp_cursor, h->d_pda_cells);
    if ((size_t)n_pda * 6 > h->pda_coef_alloc) {
        free_dev(h->d_pda_coef);
        if (hipMalloc(&h->d_pda_coef, sizeof(double) * 6 * n_pda) != hipSuccess) return h->set_error("cannot allocate the PDA coefficients");
        h->pda_coef_alloc = (size_t)n_pda * 6;
    }
    const bool exact = n_pda < 10000;
    const double tolerance = exact ? 1.e-5 : 1.e-4, gs_tol = 1.e-4;
    const int cb = (int)std::min<size_t>((n_pda + 255) / 256, (size_t)h->n_cu * 4);
    if (exact) {
        if (!h->d_pda_id && hipMalloc(&h->d_pda_id, sizeof(unsigned int) * nc) != hipSuccess) return h->set_error("cannot allocate the PDA index");
        if ((size_t)n_pda > h->pda_dense_alloc) {
            free_dev(h->d_pda_a); free_dev(h->d_pda_b); free_dev(h->d_pda_f);
            if (hipMalloc(&h->d_pda_a, sizeof(double) * (size_t)n_pda * n_pda) != hipSuccess || hipMalloc(&h->d_pda_b, sizeof(double) * n_pda) != hipSuccess ||
                hipMalloc(&h->d_pda_f, sizeof(double) * n_pda) != hipSuccess) return h->set_error("cannot allocate the dense PDA system");
            h->pda_dense_alloc = n_pda;
        }
        (void)hipMemsetAsync(h->d_pda_id, 0xff, sizeof(unsigned int) * nc, h->stream);
        pda_id_kernel<<<cb, 256, 0, h->stream>>>(h->d_pda_cells, n_pda, h->d_pda_id);
    }
    for (int outer = 1; outer <= 10000; outer++) {
        h->pda_last_outer = outer;
        pda_coef_kernel<<<cb, 256, 0, h->stream>>>(h->d_problem, h->d_pda_cells, n_pda, h->d_specific_energy, h->d_density, h->d_pda_emean,
                                                  h->d_pda_coef, exact ? 1 : 0);
        if (exact) {
            (void)hipMemsetAsync(h->d_pda_a, 0, sizeof(double) * (size_t)n_pda * n_pda, h->stream);
            pda_dense_build_kernel<<<cb, 256, 0, h->stream>>>(h->d_problem, h->d_pda_cells, n_pda, h->d_pda_id, h->d_pda_coef, h->d_pda_emean,
                                                             h->d_pda_a, h->d_pda_b);
            for (unsigned int k = 0; k + 1 < n_pda; k++) {
                const unsigned int rows = n_pda - k - 1;
                pda_pivot_kernel<<<1, 1024, 0, h->stream>>>(h->d_pda_a, n_pda, k, (unsigned int *)h->d_pda_f);       // f[0 .. k] is free: the pivot row's index lives in f[0]
                pda_swap_kernel<<<std::min(64u, (rows + 256) / 256), 256, 0, h->stream>>>(h->d_pda_a, h->d_pda_b, n_pda, k, (const unsigned int *)h->d_pda_f);
                pda_elim_factor_kernel<<<(rows + 255) / 256, 256, 0, h->stream>>>(h->d_pda_a, n_pda, k, h->d_pda_f);
                pda_elim_update_kernel<<<dim3(std::min(8u, (rows + 255) / 256), rows), 256, 0, h->stream>>>(h->d_pda_a, h->d_pda_b, n_pda, k, h->d_pda_f);
            }
            pda_backsub_kernel<<<1, 1024, 0, h->stream>>>(h->d_pda_a, h->d_pda_b, n_pda);
            pda_scatter_solution_kernel<<<cb, 256, 0, h->stream>>>(h->d_pda_cells, n_pda, h->d_pda_b, h->d_pda_emean);
        } else
            pda_gs_kernel<<<1, 1024, 0, h->stream>>>(h->d_problem, h->d_pda_cells, hp_off, n_hp, h->d_pda_coef, h->d_pda_emean, gs_tol,
                                                     20000000, h->d_pda_ctl);
        (void)hipMemsetAsync(&h->d_pda_ctl->maxdiff_bits, 0, sizeof(unsigned long long), h->stream);
        pda_update_kernel<<<cb, 256, 0, h->stream>>>(h->d_problem, h->d_pda_cells, n_pda, h->d_pda_emean, h->d_specific_energy, h->d_spec,
                                                    h->n_bins, h->d_pda_ctl);
        if (hipMemcpyAsync(&ctl, h->d_pda_ctl, sizeof ctl, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error(std::string("PDA solve failed: ") + hipGetErrorString(hipGetLastError()));
        h->pda_last_sweeps += ctl.sweeps;
        double maxdiff;
        std::memcpy(&maxdiff, &ctl.maxdiff_bits, sizeof maxdiff);
        if (maxdiff < tolerance) return 0;      // " [pda] converged"
    }
    return h->set_error("PDA did not converge");
}

int hyp_get_n_photons(hyp_handle h, double *out)
{
    if (!h || !out) return 1;
    if (!h->count_photons) return h->set_error("n_photons array is not allocated");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    // after hyp_lucy_accumulators (and the all-reduce) the block holds the whole-job counts
    hipError_t e = hipMemcpy(out, h->d_accum + h->ext_nphot, sizeof(double) * h->n_cells, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(n_photons): ") + hipGetErrorString(e));
    return 0;
}

int hyp_get_specific_energy_spectrum(hyp_handle h, double *out, double *bin_edges_out)
{
    if (!h) return 1;
    if (!h->n_bins) return h->set_error("specific_energy_spectrum array is not allocated");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    if (bin_edges_out) std::memcpy(bin_edges_out, h->spectrum_edges.data(), sizeof(double) * (h->n_bins + 1));
    if (!out) return 0;
    double *tmp = nullptr;
    const size_t n = (size_t)h->n_bins * h->n_elem;
    if (hipMalloc(&tmp, sizeof(double) * n) != hipSuccess) return h->set_error("cannot allocate the spectrum staging buffer");
    spectrum_to_ref_kernel<<<h->n_cu * 8, 256, 0, h->stream>>>(h->d_spec, tmp, h->n_cells, h->n_dust, h->n_bins);
    hipError_t e = hipMemcpyAsync(out, tmp, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return h->set_error(std::string("copy out failed: ") + hipGetErrorString(e));
    return 0;
}

// specific_energy_converged (grid_physics_3d.f90:637-689): the `percentile` quantile of max(a/b, b/a) between the
// specific energy at the previous call and now.  status 0: value computed; 1: nothing changed (value 0); 2: could not
// check (only cells that were or became zero changed); 3: first call (no previous state).  fortranlib's quantile
// (source absent) is restated as the element of rank nint(percentile / 100 * (n - 1)) of the sorted sample; it is
// found by a search over the bit patterns of the (positive) ratios: 63 counting passes, no sort.
int hyp_convergence_value(hyp_handle h, double percentile, double *value, int *status)
{
    if (!h || !value || !status) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    const size_t n = h->n_elem;
    if (!h->d_prev_se) {
        if (hipMalloc(&h->d_prev_se, sizeof(double) * n) != hipSuccess || hipMalloc(&h->d_ratio, sizeof(double) * n) != hipSuccess ||
            hipMalloc(&h->d_conv_ctl, sizeof(ConvCtl)) != hipSuccess) return h->set_error("cannot allocate the convergence work arrays");
    }
    *value = 0.0;
    if (!h->have_prev) {
        (void)hipMemcpyAsync(h->d_prev_se, h->d_specific_energy, sizeof(double) * n, hipMemcpyDeviceToDevice, h->stream);
        (void)hipStreamSynchronize(h->stream);
        h->have_prev = true; *status = 3;
        return 0;
    }
    const int blocks = h->n_cu * 8;
    ConvCtl c;
    (void)hipMemsetAsync(h->d_conv_ctl, 0, sizeof(ConvCtl), h->stream);
    conv_ratio_kernel<<<blocks, 256, 0, h->stream>>>(h->d_prev_se, h->d_specific_energy, n, h->d_ratio, h->d_conv_ctl);
    (void)hipMemcpyAsync(h->d_prev_se, h->d_specific_energy, sizeof(double) * n, hipMemcpyDeviceToDevice, h->stream);
    if (hipMemcpyAsync(&c, h->d_conv_ctl, sizeof c, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("convergence check failed");
    if (c.n_changed == 0) { *status = 1; return 0; }
    if (c.n_changed_nonzero == 0 || c.n_valid == 0) { *status = 2; return 0; }
    long long rank = (long long)std::floor(percentile / 100.0 * (double)(c.n_valid - 1) + 0.5);
    if (rank < 0) rank = 0;
    if ((unsigned long long)rank > c.n_valid - 1) rank = (long long)(c.n_valid - 1);
    // largest bit pattern v with #(ratios < v) <= rank is the ratio of that rank
    unsigned long long prefix = 0;
    for (int bit = 62; bit >= 0; bit--) {
        const unsigned long long cand = prefix | (1ull << bit);
        (void)hipMemsetAsync(&h->d_conv_ctl->count, 0, sizeof(unsigned long long), h->stream);
        conv_count_kernel<<<blocks, 256, 0, h->stream>>>(h->d_ratio, n, cand, h->d_conv_ctl);
        if (hipMemcpyAsync(&c.count, &h->d_conv_ctl->count, sizeof c.count, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) return h->set_error("convergence check failed");
        if (c.count <= (unsigned long long)rank) prefix = cand;
    }
    std::memcpy(value, &prefix, sizeof(double));
    *status = 0;
    return 0;
}

int hyp_lucy_launch(hyp_handle h, uint64_t first_id, uint64_t n_local, int iteration)
{
    if (!h) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    DProblem &P = h->hp;
    if (P.n_sources == 0) return h->set_error("no sources set up - need sources for initial iteration(s)");      // setup_rt.f90:230
    int copies = h->reproducible ? 1 : h->accum_copies;
    if (copies < 1) copies = 1;
    if (copies > 256) copies = 256;
    if (copies > h->accum_copies_alloc) {   // grow the replica pool on demand
        double *nb = nullptr;
        if (hipMalloc(&nb, sizeof(double) * h->accum_stride * copies) != hipSuccess)
            return h->set_error("cannot allocate accumulator replicas");
        (void)hipStreamSynchronize(h->stream);
        (void)hipFree(h->d_accum);
        h->d_accum = nb; h->accum_copies_alloc = copies;
    }
    P.sum = h->d_accum; P.tail = h->d_accum + h->n_elem; P.n_copies = copies; P.copy_stride = h->accum_stride;
    P.sum_spec = h->n_bins ? h->d_accum + h->ext_spec : nullptr;
    if (h->count_photons) {      // grid_reset_energy: grid_generic.f90:21-27
        (void)hipMemsetAsync(h->d_nphot, 0, sizeof(unsigned int) * h->n_cells, h->stream);
        (void)hipMemsetAsync(h->d_nphot_inexact, 0, sizeof(int), h->stream);
        // the visited sets are sized by hyp_lucy_launch below, once the grid of the persistent kernel is known
    }
    if (mrw_prepare(h)) return 1;
    // The brick-tiled iteration pays off once the grid has many bricks and the
    // iteration is long enough to amortise its generations (measured: profiles/r01c_*).
    // (the per-cell packet counter and the spectrum planes live in global memory: those runs use the persistent kernel)
    bool tile_ok = false, tile_auto = false;
    if (P.grid_type == 1) {
        tile_ok = h->n_dust <= 4 && car_tile_bricks(P, h->n_dust) > 0 && !h->count_photons && !h->n_bins;
        // (128^3: 14.2 against 18.0 ms at 2e6 packets, even at 1e6; tools/small_probe.py.  Large grids: a brick's load and flush must be shared by
        // enough packets -- break-even against the persistent kernel at ~1 200 packets in flight per brick, 400^3 and 512^3, tools/big_grid_probe.py)
        const unsigned long long nbk = (unsigned long long)tile_bricks(P, h->n_dust);
        tile_auto = tile_ok && nbk >= 32 && n_local >= std::max(1500000ull, 1500ull * nbk);
    } else if (P.grid_type == 3) {
        // Voronoi: clusters of cells in LDS (hyp_vtile.h); the modified random walk does not exist on these grids
        tile_ok = h->n_dust <= 4 && !h->count_photons && !h->n_bins && !P.mrw;
        tile_auto = tile_ok && h->n_cells >= 8192 && n_local >= 2000000ull;
    }
    else if (P.grid_type == 2) {
        // octree: clusters of sibling subtrees in LDS (hyp_otile.h)
        tile_ok = h->n_dust <= 4 && !h->count_photons && !h->n_bins && h->oct_neighbours;
        tile_auto = tile_ok && h->n_cells >= 4096 && n_local >= 2000000ull;
    }
    else if (P.grid_type == 5 || P.grid_type == 6) {
        // spherical / cylindrical polar grids: index bricks in LDS (hyp_ptile.h)
        tile_ok = h->n_dust <= 4 && polar_tile_bricks(P, h->n_dust, h->pt_lds_kb) > 0 && !h->count_photons && !h->n_bins;
        tile_auto = tile_ok && h->n_cells >= 4096 && n_local >= 3000000ull;      // (400 x 200: 91 against 81 ms at 2e6 packets, 140 against 150 at 4e6)
    }
    else if (P.grid_type == 4) {
        // AMR: bricks of the grids in LDS (hyp_atile.h)
        tile_ok = h->n_dust <= 4 && !h->count_photons && !h->n_bins;
        tile_auto = tile_ok && h->n_cells >= 32768 && n_local >= 2000000ull;
    }
    bool tiled = tile_ok && !h->reproducible && (h->lucy_mode == 1 || (h->lucy_mode < 0 && tile_auto && !h->tile_unbuildable));
    if (tiled && (P.grid_type == 2 || P.grid_type == 3 || P.grid_type == 4)) {
        // the builders have limits of their own (HYP_TILE_MAX_BRICKS clusters / bricks, the LDS budget, 16-bit grid numbers):
        // a grid beyond them runs on the persistent kernel as before; only a FORCED tiled iteration (lucy_mode = 1) reports the limit
        const int rc = P.grid_type == 4 ? build_amr_slabs(h) : P.grid_type == 3 ? build_vor_clusters(h) : build_oct_clusters(h);
        if (rc) {
            if (h->lucy_mode == 1) return 1;
            h->tile_unbuildable = true;
            h->err.clear();
            tiled = false;
        }
    }
    if (sync_problem(h)) return 1;
    hipError_t e = hipMemsetAsync(h->d_accum, 0, sizeof(double) * h->accum_stride * copies, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemsetAsync(accum): ") + hipGetErrorString(e));
    unsigned long long first = first_id;
    e = hipMemcpyAsync(h->d_counter, &first, sizeof(first), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync(counter): ") + hipGetErrorString(e));

    if (tiled) {
        if (launch_tiled(h, first_id, n_local, (uint32_t)iteration)) return 1;
        h->last_lucy_mode = 1;
        h->lucy_pending = true;
        h->pending_packets = n_local;
        return 0;
    }
    h->last_lucy_mode = 0;
    LucyKernel k = pick_lucy_kernel(h->n_dust, h->hp.grid_type);
    const size_t lds = lds_bytes(P);
    int bpc = h->blocks_per_cu;
    if (bpc <= 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, lds) != hipSuccess || occ <= 0) occ = 2;
        bpc = occ;
    }
    long long blocks = (long long)h->n_cu * bpc;
    long long need_blocks = (long long)((n_local + 255) / 256);
    if (need_blocks < 1) need_blocks = 1;
    if (blocks > need_blocks) blocks = need_blocks;
    const unsigned threads = h->reproducible ? 64u : 256u;       // "reproducible": one wave takes the ids in order and makes every deposit in program order
    if (h->reproducible) blocks = 1;
    if (h->count_photons) {
        // one visited set per lane of THIS launch (HYP_VISIT_SLOTS words each); with less memory than that, fewer workgroups
        for (;;) {
            const size_t lanes = (size_t)blocks * 256;
            if (h->visit_lanes >= lanes) break;
            free_dev(h->d_visit);
            h->visit_lanes = 0;
            if (hipMalloc((void **)&h->d_visit, lanes * HYP_VISIT_SLOTS * sizeof(unsigned long long)) == hipSuccess) { h->visit_lanes = lanes; break; }
            (void)hipGetLastError();
            h->d_visit = nullptr;
            if (blocks <= 1) return h->set_error("no memory for the per-lane visited sets of the n_photons counter");
            blocks = (blocks + 1) / 2;
        }
        (void)hipMemsetAsync(h->d_visit, 0, (size_t)blocks * 256 * HYP_VISIT_SLOTS * sizeof(unsigned long long), h->stream);
        if (P.visit_tab != h->d_visit) { P.visit_tab = h->d_visit; if (sync_problem(h)) return 1; }
    }
    LaunchParams L;
    L.first_id = first_id; L.end_id = first_id + n_local; L.iter_tag = (uint32_t)iteration;
    int chunk = h->chunk;
    if (chunk <= 0) {
        unsigned long long waves = (unsigned long long)blocks * 4ull;
        unsigned long long c = n_local / (waves * 8ull);
        if (c < 64) c = 64;
        if (c > 4096) c = 4096;
        chunk = (int)c;
    }
    L.chunk = chunk;
    L.interact_threshold = h->interact_threshold; L.emit_threshold = h->emit_threshold;
    (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(threads), lds, h->stream, (const DProblem *)h->d_problem, L);
    e = hipGetLastError();
    (void)hipEventRecord(h->ev1, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("lucy_kernel launch: ") + hipGetErrorString(e));
    h->lucy_pending = true;
    h->pending_packets = n_local;
    return 0;
}

int hyp_lucy_accumulators(hyp_handle h, void **device_ptr, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (!h->lucy_pending) return h->set_error("hyp_lucy_accumulators called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    if (h->hp.n_copies > 1) {
        int blocks = h->n_cu * 8;
        reduce_copies_kernel<<<blocks, 256, 0, h->stream>>>(h->d_accum, h->n_elem + TAIL_SIZE, h->accum_stride, h->hp.n_copies);
    }
    if (h->count_photons) nphot_to_block_kernel<<<h->n_cu * 4, 256, 0, h->stream>>>(h->d_nphot, h->d_accum + h->ext_nphot, h->n_cells);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("propagation failed: ") + hipGetErrorString(e));
    (void)hipEventElapsedTime(&h->last_propagate_ms, h->ev0, h->ev1);
    if (h->count_photons) (void)hipMemcpy(&h->nphot_inexact, h->d_nphot_inexact, sizeof(int), hipMemcpyDeviceToHost);
    if (check_device_error(h)) { h->lucy_pending = false; return 1; }
    if (device_ptr) *device_ptr = h->d_accum;
    if (n_doubles) *n_doubles = h->block_doubles;
    return 0;
}

int hyp_lucy_finish(hyp_handle h, double *specific_energy_out, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (!h->lucy_pending) return h->set_error("hyp_lucy_finish called without a launched iteration");
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    h->lucy_pending = false;
    double tail[TAIL_SIZE];
    hipError_t e = hipMemcpy(tail, h->d_accum + h->n_elem, sizeof(tail), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(tail): ") + hipGetErrorString(e));
    if (tail[TAIL_RANK_ERROR] != 0.0) return h->set_error("another rank reported an engine error");
    hyp_iter_stats st;
    std::memset(&st, 0, sizeof st);
    st.energy_current = tail[TAIL_ENERGY];
    st.killed_geo = (uint64_t)tail[TAIL_KILLED_GEO]; st.killed_int = (uint64_t)tail[TAIL_KILLED_INT];
    st.crossings = (uint64_t)tail[TAIL_CROSSINGS]; st.interactions = (uint64_t)tail[TAIL_INTERACTIONS];
    st.n_packets = h->pending_packets;
    if (!(st.energy_current > 0.0)) return h->set_error("no energy emitted");
    // update_energy_abs(energy_total/energy_current): iter_lucy.f90:224
    (void)hipEventRecord(h->ev2, h->stream);
    double *d_out = (specific_energy_out && h->n_dust > 1) ? h->d_scratch : nullptr;
    const double scale = h->energy_total / st.energy_current;
    if (h->n_bins) {
        spectrum_update_kernel<<<h->n_cu * 8, 256, 0, h->stream>>>(h->d_problem, h->d_accum + h->ext_spec, h->d_spec, scale, h->n_bins);
        if (hipGetLastError() != hipSuccess) return h->set_error("spectrum_update_kernel launch failed");
    }
    if (h->pda) {
        // update_energy_abs, then solve_pda, then sublimate_dust: iter_lucy.f90:224-235
        if (run_finish_kernel(h, 2, scale, nullptr)) return 1;
        if (solve_pda(h)) return 1;
        if (run_finish_kernel(h, 3, scale, d_out)) return 1;
    } else if (run_finish_kernel(h, 0, scale, d_out)) return 1;
    (void)hipEventRecord(h->ev3, h->stream);
    double tot[HYP_MAXD];
    e = hipMemcpyAsync(tot, h->d_energy_abs_tot, sizeof(tot), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && specific_energy_out)
        e = hipMemcpyAsync(specific_energy_out, h->n_dust > 1 ? h->d_scratch : h->d_specific_energy,
                           sizeof(double) * h->n_elem, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("finish failed: ") + hipGetErrorString(e));
    (void)hipEventElapsedTime(&h->last_finish_ms, h->ev2, h->ev3);
    for (int d = 0; d < h->n_dust; d++) st.energy_abs_tot[d] = tot[d];
    h->last_stats = st;
    // cell crossings per flight (emission or interaction -> next interaction or escape) of this iteration: what the imaging iteration's
    // choice of schedule looks at (hyp_final_launch)
    if (st.n_packets + st.interactions > 0) h->lucy_cross_per_flight = (double)st.crossings / (double)(st.n_packets + st.interactions);
    if (stats) *stats = st;
    return 0;
}

int hyp_lucy_iteration(hyp_handle h, uint64_t n_packets, int iteration, double *specific_energy_out, hyp_iter_stats *stats)
{
    if (!h) return 1;
    if (n_packets == 0) return 0;   // "Skipping": iter_lucy.f90:87-94
    if (hyp_lucy_launch(h, 0, n_packets, iteration)) return 1;
    if (hyp_lucy_accumulators(h, nullptr, nullptr)) return 1;
    return hyp_lucy_finish(h, specific_energy_out, stats);
}

