// hyp_engine.hip -- handle life cycle, problem digest, getters and setters, options (see hyp_engine.h for the other units)
#include "hyp_engine.h"

std::string g_error;

extern "C" {


int hyp_abi_version(void) { return HYP_ABI_VERSION; }

namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void bytes(const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } }
    void i64(int64_t v) { bytes(&v, sizeof v); }
    void f64(double v) { if (v == 0.0) v = 0.0; bytes(&v, sizeof v); }      // -0.0 and 0.0 alike
    void arr(const double *p, size_t n) { i64(p ? (int64_t)n : -1); if (p) for (size_t i = 0; i < n; i++) f64(p[i]); }
    void arr32(const int32_t *p, size_t n) { i64(p ? (int64_t)n : -1); if (p) bytes(p, n * sizeof(int32_t)); }
};
void digest_peeled(Fnv &f, const hyp_peeled_desc &d, bool binned)
{
    for (int64_t v : {(int64_t)(binned ? 0 : d.n_view), (int64_t)d.inside_observer, (int64_t)d.ignore_optical_depth, (int64_t)d.compute_image, (int64_t)d.compute_sed,
                      (int64_t)d.n_x, (int64_t)d.n_y, (int64_t)d.n_ap, (int64_t)d.n_nu, (int64_t)d.track_origin, (int64_t)d.track_n_scat, (int64_t)d.uncertainties,
                      (int64_t)d.compute_stokes, (int64_t)d.use_filters}) f.i64(v);
    if (d.compute_image) for (double v : {d.x_min, d.x_max, d.y_min, d.y_max}) f.f64(v);
    if (d.compute_sed) for (double v : {d.ap_min, d.ap_max}) f.f64(v);
    for (double v : {d.nu_min, d.nu_max, d.d_min, d.d_max}) f.f64(v);
    if (d.inside_observer) for (double v : d.peeloff_origin) f.f64(v);
    if (!binned) { f.arr(d.theta, (size_t)d.n_view); f.arr(d.phi, (size_t)d.n_view); }
    if (d.use_filters) {
        f.arr32(d.filt_n, (size_t)d.n_nu);
        size_t tot = 0;
        for (int i = 0; i < d.n_nu && d.filt_n; i++) tot += (size_t)d.filt_n[i];
        f.arr(d.filt_nu, tot); f.arr(d.filt_tr, tot);
    }
}
}  // namespace

int hyp_problem_digest(const hyp_problem *pr, uint64_t out[4])
{
    if (!pr || !out) return 1;
    const hyp_grid_desc &g = pr->grid;
    if (pr->n_dust < 0 || pr->n_sources < 0 || pr->n_peeled < 0 || g.n_cells < 0) return 1;
    size_t nc = (size_t)g.n_cells;
    Fnv a;
    a.i64(g.type);
    if (g.type == 1 || g.type == 5 || g.type == 6) {
        nc = (size_t)g.n1 * g.n2 * g.n3;
        a.i64(g.n1); a.i64(g.n2); a.i64(g.n3);
        a.arr(g.w1, (size_t)g.n1 + 1); a.arr(g.w2, (size_t)g.n2 + 1); a.arr(g.w3, (size_t)g.n3 + 1);
    } else if (g.type == 2) {
        a.arr32(g.refined, nc);
        for (double v : g.oct_center) a.f64(v);
        for (double v : g.oct_half) a.f64(v);
    } else if (g.type == 3) {
        a.arr(g.vor_sites, 3 * nc); a.arr(g.vor_volume, nc); a.arr32(g.vor_idx, nc + 1);
        a.arr32(g.vor_neighs, g.vor_idx ? (size_t)g.vor_idx[nc] : 0);
        for (double v : g.vor_box) a.f64(v);
        a.arr(g.vor_bb, 6 * nc);
    } else if (g.type == 4) {
        a.i64(g.n_amr_levels); a.i64(g.n_amr_grids);
        a.arr32(g.amr_level, (size_t)g.n_amr_grids); a.arr32(g.amr_n, 3 * (size_t)g.n_amr_grids); a.arr(g.amr_bounds, 6 * (size_t)g.n_amr_grids);
        nc = 0;
        for (int k = 0; k < g.n_amr_grids && g.amr_n; k++) nc += (size_t)g.amr_n[3 * k] * g.amr_n[3 * k + 1] * g.amr_n[3 * k + 2];
    } else return 1;
    a.arr(pr->density, nc * (size_t)pr->n_dust);
    a.arr(pr->specific_energy, nc * (size_t)pr->n_dust);
    out[0] = a.h;
    Fnv d;
    d.i64(pr->n_dust);
    for (int i = 0; i < pr->n_dust; i++) {
        const hyp_dust_desc &D = pr->dust[i];
        for (int64_t v : {(int64_t)D.n_nu, (int64_t)D.n_mu, (int64_t)D.n_jnu, (int64_t)D.n_enu, (int64_t)D.n_e, (int64_t)D.sublimation_mode, (int64_t)D.version, (int64_t)D.is_lte}) d.i64(v);
        d.f64(D.sublimation_mode ? D.sublimation_specific_energy : 0.0); d.f64(D.minimum_specific_energy);
        d.arr(D.nu, (size_t)D.n_nu); d.arr(D.albedo, (size_t)D.n_nu); d.arr(D.chi, (size_t)D.n_nu); d.arr(D.mu, (size_t)D.n_mu);
        const size_t np = (size_t)D.n_nu * D.n_mu;
        d.arr(D.P1, np); d.arr(D.P2, np); d.arr(D.P3, np); d.arr(D.P4, np);
        d.arr(D.emiss_nu, (size_t)D.n_enu); d.arr(D.emiss_jnu, (size_t)D.n_enu * D.n_jnu); d.arr(D.emiss_var, (size_t)D.n_jnu);
        d.arr(D.mo_specific_energy, (size_t)D.n_e); d.arr(D.mo_chi_rosseland, (size_t)D.n_e); d.arr(D.mo_kappa_planck, (size_t)D.n_e);
        d.arr(D.mo_chi_inv_planck, (size_t)D.n_e);
    }
    out[1] = d.h;
    Fnv s;
    s.i64(pr->n_sources);
    for (int i = 0; i < pr->n_sources; i++) {
        const hyp_source_desc &S = pr->sources[i];
        for (int64_t v : {(int64_t)S.type, (int64_t)S.spectrum_type, (int64_t)S.peeloff}) s.i64(v);
        s.f64(S.luminosity);
        if (S.spectrum_type == 1) { s.arr(S.spec_nu, (size_t)S.n_spec); s.arr(S.spec_fnu, (size_t)S.n_spec); }
        else if (S.spectrum_type == 2) s.f64(S.temperature);
        if (S.type == 1 || S.type == 2 || S.type == 5 || S.type == 7) for (double v : S.position) s.f64(v);
        if (S.type == 2 || S.type == 5 || S.type == 7) s.f64(S.radius);
        if (S.type == 2) s.i64(S.limb_darkening);
        if (S.type == 6) for (double v : S.box) s.f64(v);
        if (S.type == 7) for (double v : S.direction) s.f64(v);
        if (S.type == 8) { s.arr(S.points, 3 * (size_t)S.n_points); s.arr(S.point_lum, (size_t)S.n_points); }
        if (S.type == 4) s.arr(S.map, nc);
        s.i64(S.type == 2 ? S.n_spots : 0);
        for (int k = 0; k < S.n_spots && S.type == 2 && S.spots; k++) {
            const hyp_spot_desc &T = S.spots[k];
            for (double v : {T.longitude, T.latitude, T.radius, T.luminosity}) s.f64(v);
            s.i64(T.spectrum_type);
            if (T.spectrum_type == 1) { s.arr(T.spec_nu, (size_t)T.n_spec); s.arr(T.spec_fnu, (size_t)T.n_spec); } else s.f64(T.temperature);
        }
    }
    out[2] = s.h;
    Fnv c;
    const hyp_config &K = pr->config;
    for (int64_t v : {K.seed, K.n_inter_max, K.n_reabs_max, (int64_t)K.kill_on_absorb, (int64_t)K.kill_on_scatter, (int64_t)K.sample_sources_evenly,
                      (int64_t)K.enforce_energy_range, (int64_t)K.forced_first_interaction, (int64_t)(K.forced_first_interaction ? K.forced_first_interaction_algorithm : 0),
                      (int64_t)K.specific_energy_type, (int64_t)K.raytracing, (int64_t)K.mrw, (int64_t)K.monochromatic, (int64_t)K.pda, (int64_t)K.count_photons,
                      (int64_t)K.n_spectrum_bins}) c.i64(v);
    if (K.forced_first_interaction && K.forced_first_interaction_algorithm == 2) c.f64(K.baes16_xi);
    c.f64(K.propagation_check_frequency);
    if (K.mrw) { c.i64(K.n_inter_mrw_max); c.f64(K.mrw_gamma); }
    if (K.monochromatic) { c.f64(K.monochromatic_energy_threshold); c.arr(K.frequencies, (size_t)K.n_frequencies); }
    if (K.n_spectrum_bins) c.arr(K.spectrum_bin_edges, (size_t)K.n_spectrum_bins + 1);
    c.i64(pr->n_peeled);
    for (int i = 0; i < pr->n_peeled; i++) {
        digest_peeled(c, pr->peeled[i], false);
        if (K.monochromatic) { c.i64(pr->peeled[i].inu_min); c.i64(pr->peeled[i].inu_max); }
    }
    c.i64(pr->binned ? 1 : 0);
    if (pr->binned) { digest_peeled(c, *pr->binned, true); c.i64(pr->n_binned_theta); c.i64(pr->n_binned_phi); }
    out[3] = c.h;
    return 0;
}

const char *hyp_last_error(hyp_handle h) { return h ? h->err.c_str() : g_error.c_str(); }

void hyp_destroy(hyp_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    free_dev(h->d_problem); free_dev(h->d_blob); free_dev(h->d_sources); free_dev(h->d_peeled);
    free_dev(h->d_oct_cells); free_dev(h->d_oct_children); free_dev(h->d_oct_neigh);
    free_dev(h->d_at_slabs); free_dev(h->d_at_go); free_dev(h->d_at_grid_c0); free_dev(h->d_at_grid_nz);
    free_dev(h->d_ot_cluster); free_dev(h->d_ot_c0); free_dev(h->d_ot_nc); free_dev(h->d_ot_kid_off); free_dev(h->d_ot_rec); free_dev(h->d_ot_kid); free_dev(h->d_ot_nb);
    free_dev(h->d_mono_cdf); free_dev(h->d_mono_mean); free_dev(h->d_direct);
    free_dev(h->d_vor_bb);
    free_dev(h->d_vor_sites); free_dev(h->d_vor_volume); free_dev(h->d_vor_idx); free_dev(h->d_vor_neigh); free_dev(h->d_vor_seed); free_dev(h->d_vor_walls);
    free_dev(h->d_mask_map);
    free_dev(h->d_vt_cluster); free_dev(h->d_vt_info); free_dev(h->d_vt_blob); free_dev(h->d_vt_members); free_dev(h->d_vt_adj); free_dev(h->d_vt_ghost);
    free_dev(h->d_amr_grids); free_dev(h->d_amr_go); free_dev(h->d_amr_walls); free_dev(h->d_amr_cell_grid);
    free_dev(h->d_density); free_dev(h->d_specific_energy); free_dev(h->d_additional);
    free_dev(h->d_accum); free_dev(h->d_jnu_id); free_dev(h->d_jnu_frac); free_dev(h->d_energy_abs_tot);
    free_dev(h->d_mrw_alpha); free_dev(h->d_mrw_diff); free_dev(h->d_mrw_kp);
    free_dev(h->d_scratch); free_dev(h->d_counter); free_dev(h->d_err); free_dev(h->d_err_data);
    free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
    free_dev(h->d_peel_order); free_dev(h->d_peel_keys); free_dev(h->d_peel_bins); free_dev(h->d_ff);
    free_dev(h->d_peel_ctl);
    if (h->h_peel_ctl) (void)hipHostFree(h->h_peel_ctl);
    if (h->h_peel_counter) (void)hipHostFree(h->h_peel_counter);
    free_dev(h->d_img_accum);
    free_dev(h->d_hot); free_dev(h->d_cold); free_dev(h->d_slot_brick); free_dev(h->d_order);
    free_dev(h->d_counts); free_dev(h->d_cursor); free_dev(h->d_tasks); free_dev(h->d_ctl);
    free_dev(h->d_ilist); free_dev(h->d_dlist); free_dev(h->d_tcount); free_dev(h->d_extra);
    for (hipEvent_t e : h->walk_events) (void)hipEventDestroy(e);
    free_dev(h->d_nphot); free_dev(h->d_visit); free_dev(h->d_nphot_inexact); free_dev(h->d_log_edges); free_dev(h->d_bin_frac); free_dev(h->d_spec);
    free_dev(h->d_pda_mask); free_dev(h->d_pda_cells); free_dev(h->d_pda_hp); free_dev(h->d_pda_emean); free_dev(h->d_pda_coef);
    free_dev(h->d_pda_id); free_dev(h->d_pda_a); free_dev(h->d_pda_b); free_dev(h->d_pda_f);
    free_dev(h->d_pda_ctl); free_dev(h->d_prev_se); free_dev(h->d_ratio); free_dev(h->d_conv_ctl);
    if (h->h_ctl) (void)hipHostFree(h->h_ctl);
    for (int i = 1; i < 4; i++) if (h->pool_stream[i]) (void)hipStreamDestroy(h->pool_stream[i]);
    if (h->ev_pool) (void)hipEventDestroy(h->ev_pool);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->ev3) (void)hipEventDestroy(h->ev3);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int copy_out_ref_layout(hyp_handle h, const double *d_src, double *out)
{
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    const double *src = d_src;
    if (h->n_dust > 1) {
        to_ref_layout_kernel<<<1024, 256, 0, h->stream>>>(d_src, h->d_scratch, h->n_cells, h->n_dust);
        src = h->d_scratch;
    }
    hipError_t e = hipMemcpyAsync(out, src, sizeof(double) * h->n_elem, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("copy out failed: ") + hipGetErrorString(e));
    return 0;
}

int hyp_get_specific_energy(hyp_handle h, double *out)
{
    if (!h || !out) return 1;
    return copy_out_ref_layout(h, h->d_specific_energy, out);
}

int hyp_get_density(hyp_handle h, double *out)
{
    if (!h || !out) return 1;
    return copy_out_ref_layout(h, h->d_density, out);
}

int hyp_set_specific_energy(hyp_handle h, const double *in)
{
    if (!h || !in) return 1;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    hipError_t e = hipMemcpyAsync(h->d_scratch, in, sizeof(double) * h->n_elem, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_specific_energy, h->n_cells, h->n_dust);
    if (run_finish_kernel(h, 1, 1.0, nullptr)) return 1;
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return h->set_error(std::string("set_specific_energy failed: ") + hipGetErrorString(e));
    return 0;
}

int hyp_last_kernel_ms(hyp_handle h, float *propagate_ms, float *finish_ms)
{
    if (!h) return 1;
    if (propagate_ms) *propagate_ms = h->last_propagate_ms;
    if (finish_ms) *finish_ms = h->last_finish_ms;
    return 0;
}

int hyp_set_option(hyp_handle h, const char *name, int64_t value)
{
    if (!h || !name) return 1;
    std::string n(name);
    if (n == "interact_threshold") h->interact_threshold = (int)value;
    else if (n == "emit_threshold") h->emit_threshold = (int)value;
    else if (n == "accum_copies") h->accum_copies = (int)value;
    else if (n == "blocks_per_cu") h->blocks_per_cu = (int)value;
    else if (n == "chunk") h->chunk = (int)value;
    else if (n == "lucy_mode") h->lucy_mode = (int)value;       // -1 auto, 0 persistent atomics kernel, 1 brick-tiled
    else if (n == "tile_slots") h->tile_slots = (int)value;
    else if (n == "tile_task") h->tile_task = (int)value;
    else if (n == "tile_pools") h->tile_pools = (int)value;
    else if (n == "tile_time_walk") h->tile_time_walk = (int)value;
    else if (n == "plain_imaging") h->plain_imaging = value != 0 && h->plain_imaging;      // can only be switched off
    else if (n == "defer_peel") h->defer_peel = value < 0 ? 0 : value > 3 ? 3 : (int)value;
    else if (n == "mono_defer") h->mono_defer_opt = value ? 1 : 0;
    else if (n == "gen_defer") h->gen_defer_opt = value ? 1 : 0;
    else if (n == "direct_memo") h->direct_memo = value ? 1 : 0;
    else if (n == "peel_sort") h->peel_sort = value != 0;
    else if (n == "ff_prepass") h->ff_prepass = value != 0;
    else if (n == "oct_neighbours") { h->oct_neighbours = value != 0; h->hp.oct_neigh = h->oct_neighbours ? h->d_oct_neigh : nullptr; }
    else if (n == "peel_events") {
        if (value < 1) return h->set_error("peel_events must be positive");
        if (h->peel_events != value) {      // the buffers are allocated by the next imaging iteration
            free_dev(h->d_peel_events); free_dev(h->d_peel_susp[0]); free_dev(h->d_peel_susp[1]); free_dev(h->d_peel_ret[0]); free_dev(h->d_peel_ret[1]);
            h->peel_cap = 0;
        }
        h->peel_events = value; h->peel_events_exact = true;
    }
    else if (n == "vt_cells") { h->vt_cells = (int)value; h->vt_built_for = -1; h->tile_unbuildable = false; }
    // (the tests' handles on the tiled schedules: small clusters / bricks, early drain, a look at the device after every generation)
    else if (n == "at_cells") { h->at_cells = (int)value; h->at_built_for = -1; h->tile_unbuildable = false; }
    else if (n == "ot_cells") { h->ot_cells = (int)value; h->ot_built_for = -1; h->tile_unbuildable = false; }
    else if (n == "pt_lds_kb") h->pt_lds_kb = (int)std::max<int64_t>(1, std::min<int64_t>(150, value));
    else if (n == "tile_drain") h->tile_drain = (int)value;
    else if (n == "tile_park") h->tile_park = value < 0 ? 0 : value > 64 ? 64 : (int)value;
    else if (n == "img_end_game") h->img_end_game = value ? 1 : 0;
    else if (n == "reproducible") {
        h->reproducible = value ? 1 : 0;
        for (auto &G : h->h_peeled) G.serial = h->reproducible;
        if (h->d_peeled && !h->h_peeled.empty() &&
            hipMemcpy(h->d_peeled, h->h_peeled.data(), sizeof(DPeeled) * h->h_peeled.size(), hipMemcpyHostToDevice) != hipSuccess)
            return h->set_error("cannot update the image groups on the device");
    }
    else if (n == "tile_poll") h->tile_poll = value < 1 ? 1 : (int)value;
    else return h->set_error("unknown option: " + n);
    return 0;
}

int hyp_get_option(hyp_handle h, const char *name, int64_t *value)
{
    if (!h || !name || !value) return 1;
    std::string n(name);
    if (n == "interact_threshold") *value = h->interact_threshold;
    else if (n == "emit_threshold") *value = h->emit_threshold;
    else if (n == "accum_copies") *value = h->accum_copies;
    else if (n == "blocks_per_cu") *value = h->blocks_per_cu;
    else if (n == "chunk") *value = h->chunk;
    else if (n == "lucy_mode") *value = h->lucy_mode;
    else if (n == "tile_slots") *value = h->tile_slots;
    else if (n == "tile_task") *value = h->tile_task;
    else if (n == "tile_pools") *value = h->tile_pools;
    else if (n == "lucy_block_doubles") *value = (int64_t)h->block_doubles;
    else if (n == "lucy_flag_index") *value = (int64_t)(h->n_elem + TAIL_RANK_ERROR);
    else if (n == "image_block_doubles") *value = (int64_t)(h->d_img_accum ? h->img_accum_n : (size_t)TAIL_SIZE);
    else if (n == "image_flag_index") *value = (int64_t)((h->d_img_accum ? h->img_accum_n - TAIL_SIZE : (size_t)0) + TAIL_RANK_ERROR);
    else if (n == "last_walk_us") *value = (int64_t)(h->last_walk_ms * 1000.0);
    else if (n == "last_walk_launches") *value = h->last_walk_launches;
    else if (n == "pda_last_cells") *value = h->pda_last_cells;
    else if (n == "defer_peel") *value = h->defer_peel;
    else if (n == "mono_defer") *value = h->mono_defer_opt;
    else if (n == "gen_defer") *value = h->gen_defer_opt && h->gen_defer ? 1 : 0;
    else if (n == "direct_memo") *value = h->direct_memo;
    else if (n == "last_direct_memo") *value = h->last_direct_memo;
    else if (n == "last_mono_deferred") *value = h->last_mono_deferred;
    else if (n == "last_tiled_imaging") *value = h->last_tiled_imaging;
    else if (n == "last_end_game") *value = h->last_end_game;
    else if (n == "peel_sort") *value = h->peel_sort;
    else if (n == "ff_prepass") *value = h->ff_prepass;
    else if (n == "last_ff_prepass") *value = h->last_ff_prepass;
    else if (n == "n_photons_inexact") *value = h->nphot_inexact;
    else if (n == "peel_events") *value = h->peel_events;
    else if (n == "plain_imaging") *value = h->plain_imaging ? 1 : 0;
    else if (n == "last_defer_rounds") *value = h->last_defer_rounds;
    else if (n == "last_defer_events") *value = (int64_t)h->last_defer_events;
    else if (n == "at_slabs") *value = h->at_slabs_n;
    else if (n == "ot_clusters") *value = h->ot_clusters;
    else if (n == "vt_cells") *value = h->vt_cells;
    else if (n == "vt_clusters") *value = h->vt_clusters;
    else if (n == "vt_max_cells") *value = h->vt_max_cells;
    else if (n == "vt_max_lds") *value = (int64_t)h->vt_max_lds;
    else if (n == "ot_max_cells") *value = h->ot_max_cells;
    else if (n == "at_max_cells") *value = h->at_max_cells;
    else if (n == "ot_cells") *value = h->ot_cells;
    else if (n == "at_cells") *value = h->at_cells;
    else if (n == "pt_lds_kb") *value = h->pt_lds_kb;
    else if (n == "tile_drain") *value = h->tile_drain;
    else if (n == "tile_park") *value = h->tile_park;
    else if (n == "last_tile_slots") *value = h->last_tile_slots;
    else if (n == "reproducible") *value = h->reproducible;
    else if (n == "img_end_game") *value = h->img_end_game;
    else if (n == "tile_poll") *value = h->tile_poll;
    else if (n == "tile_time_walk") *value = h->tile_time_walk;
    else if (n == "oct_neighbours") *value = h->oct_neighbours ? 1 : 0;
    else if (n == "last_vt_exact_steps") *value = h->h_ctl ? (int64_t)h->h_ctl->dbg[38] : 0;      // steps of the Voronoi walk that ran the reference's loop
    else if (n == "last_vt_mismatch") *value = h->h_ctl ? (int64_t)h->h_ctl->dbg[39] : 0;         // -DHYP_VTILE_VERIFY builds: steps on which the filter and the reference's loop disagreed
    else if (n.rfind("last_walk_why", 0) == 0 && n.size() == 14 && n[13] >= '0' && n[13] <= '7') *value = h->h_ctl ? (int64_t)h->h_ctl->dbg[30 + (n[13] - '0')] : 0;
    else if (n == "last_lucy_mode") *value = h->last_lucy_mode;         // schedule the last Lucy iteration ran with
    else if (n == "last_generations") *value = h->last_generations;   // generations of the last tiled iteration
    else return h->set_error("unknown option: " + n);
    return 0;
}

int hyp_peeled_n_orig(hyp_handle h, int g) { return (h && g >= 0 && g < (int)h->h_peeled.size()) ? h->h_peeled[g].n_orig : -1; }

int hyp_peeled_get(hyp_handle h, int g, int which, double *out, uint64_t *n_doubles)
{
    if (!h) return 1;
    if (g < 0 || g >= (int)h->h_peeled.size() || which < 0 || which > 3) return h->set_error("hyp_peeled_get: bad group or selector");
    size_t n = (which < 2) ? h->sed_n[g] : h->img_n[g];
    size_t off = ((which < 2) ? h->sed_off[g] : h->img_off[g]) + ((which & 1) ? n : 0);
    if (n_doubles) *n_doubles = n;
    if (!out || n == 0) return 0;
    if (hipSetDevice(h->device) != hipSuccess) return h->set_error("hipSetDevice failed");
    hipError_t e = hipMemcpy(out, h->d_img_accum + off, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return h->set_error(std::string("hipMemcpy(images): ") + hipGetErrorString(e));
    return 0;
}

}  // extern "C"
