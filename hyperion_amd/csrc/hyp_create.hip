// hyp_create.hip -- hyp_create: the problem's tables built on the host and made resident on the device; the builders of the tiled
// schedules' clusters and bricks (see hyp_engine.h)
#include "hyp_engine.h"

namespace {

// A host-side pool of doubles that becomes one device allocation; tables are
// addressed by offset until upload, then by pointer.
struct Blob {
    std::vector<double> h;
    size_t put(const double *a, size_t n) { size_t o = h.size(); h.insert(h.end(), a, a + n); return o; }
    size_t put(const std::vector<double> &v) { return put(v.data(), v.size()); }
};

double seg_loglog(double x1, double x2, double y1, double y2)
{
    if (!(y1 > 0.0 && y2 > 0.0)) return 0.0;
    double b = std::log10(y1 / y2) / std::log10(x1 / x2);
    if (std::fabs(b + 1.0) < 1e-10) return x1 * y1 * std::log(x2 / x1);
    return y1 * (x2 * std::pow(x2 / x1, b) - x1) / (b + 1.0);
}

// interpolate_pdf(pdf, xv, bounds_error=.false., fill_value=0) of a log pdf set from (x, y[stride]): the normalised pdf
// interpolated in log-log (linear where an ordinate is not positive), 0 outside the table
double interp_log_pdf(const double *x, const double *y, size_t stride, int n, double xv)
{
    if (!(xv >= x[0]) || !(xv <= x[n - 1])) return 0.0;
    double norm = 0.0;
    for (int i = 0; i + 1 < n; i++) norm += seg_loglog(x[i], x[i + 1], y[(size_t)i * stride], y[(size_t)(i + 1) * stride]);
    if (!(norm > 0.0)) return 0.0;
    int j;
    if (xv == x[n - 1]) j = n - 2;
    else { int jl = 0, ju = n - 1; while (ju - jl > 1) { int jm = (ju + jl) >> 1; if (xv >= x[jm]) jl = jm; else ju = jm; } j = jl; }
    const double y1 = y[(size_t)j * stride] / norm, y2 = y[(size_t)(j + 1) * stride] / norm;
    if (y1 > 0.0 && y2 > 0.0) {
        const double f = (std::log10(xv) - std::log10(x[j])) / (std::log10(x[j + 1]) - std::log10(x[j]));
        return std::pow(10.0, std::log10(y1) + f * (std::log10(y2) - std::log10(y1)));
    }
    return y1 + (xv - x[j]) / (x[j + 1] - x[j]) * (y2 - y1);
}

// normalized_B_nu: source_type.f90:1088-1096
double normalized_B_nu(double nu, double T)
{
    const double a = 2.0 * HYP_H_CGS / HYP_C_CGS / HYP_C_CGS / HYP_STEF_BOLTZ * HYP_PI, b = HYP_H_CGS / HYP_K_CGS;
    const double T4 = T * T * T * T;
    return a * nu * nu * nu / (std::exp(b * nu / T) - 1.0) / T4;
}

// type_pdf set_pdf(x, y, log=.true.): normalised pdf, cdf and per-bin power-law
// index (+1) used by the device-side inversion.  Returns false if the integral
// vanishes.
bool build_log_pdf(const double *x, const double *y, int n, size_t stride,
                   std::vector<double> &cdf, std::vector<double> &bp1)
{
    std::vector<double> pdf(n);
    for (int i = 0; i < n; i++) pdf[i] = y[(size_t)i * stride];
    double norm = 0.0;
    for (int i = 0; i + 1 < n; i++) norm += seg_loglog(x[i], x[i + 1], pdf[i], pdf[i + 1]);
    if (!(norm > 0.0)) return false;
    for (int i = 0; i < n; i++) pdf[i] /= norm;
    cdf.assign(n, 0.0); bp1.assign(n, std::nan(""));
    for (int i = 1; i < n; i++) cdf[i] = cdf[i - 1] + seg_loglog(x[i - 1], x[i], pdf[i - 1], pdf[i]);
    double last = cdf[n - 1];
    for (int i = 0; i < n; i++) cdf[i] /= last;
    for (int i = 0; i + 1 < n; i++)
        if (pdf[i] > 0.0 && pdf[i + 1] > 0.0)
            bp1[i] = std::log10(pdf[i + 1] / pdf[i]) / std::log10(x[i + 1] / x[i]) + 1.0;
    return true;
}

// integral_loglog(x, y[, xmin, xmax]) of fortranlib (reference equivalent: hyperion/util/integrate.py
// integrate_loglog_subset): piecewise power laws, end points interpolated in log-log, limits
// clipped to the table.  `stride` lets y be a column of a row-major table.
double interp_seg_loglog(double x1, double x2, double y1, double y2, double x)
{
    if (y1 > 0.0 && y2 > 0.0) return y1 * std::pow(x / x1, std::log10(y2 / y1) / std::log10(x2 / x1));
    return y1 + (x - x1) / (x2 - x1) * (y2 - y1);
}

// interp1d_loglog of fortranlib at one abscissa inside [x[0], x[n-1]] (NaN outside)
double interp1d_loglog_host(const double *x, const double *y, int n, double xv)
{
    if (!(xv >= x[0] && xv <= x[n - 1])) return std::nan("");
    int j = (int)(std::upper_bound(x, x + n, xv) - x) - 1;
    if (j > n - 2) j = n - 2;
    const double y1 = y[j], y2 = y[j + 1];
    if (y1 > 0.0 && y2 > 0.0) {
        const double f = (std::log10(xv) - std::log10(x[j])) / (std::log10(x[j + 1]) - std::log10(x[j]));
        return std::pow(10.0, std::log10(y1) + f * (std::log10(y2) - std::log10(y1)));
    }
    return y1 + (xv - x[j]) / (x[j + 1] - x[j]) * (y2 - y1);
}

double integral_loglog_range(const double *x, const double *y, size_t stride, int n, double xmin, double xmax)
{
    if (xmin < x[0]) xmin = x[0];
    if (xmax > x[n - 1]) xmax = x[n - 1];
    if (!(xmax > xmin)) return 0.0;
    double s = 0.0;
    for (int i = 0; i + 1 < n; i++) {
        const double a = x[i], b = x[i + 1], ya0 = y[(size_t)i * stride], yb0 = y[(size_t)(i + 1) * stride];
        if (b <= xmin || a >= xmax) continue;
        const double xa = a < xmin ? xmin : a, xb = b > xmax ? xmax : b;
        const double ya = xa == a ? ya0 : interp_seg_loglog(a, b, ya0, yb0, xa);
        const double yb = xb == b ? yb0 : interp_seg_loglog(a, b, ya0, yb0, xb);
        s += seg_loglog(xa, xb, ya, yb);
    }
    return s;
}

double integral_loglog_all(const double *x, const double *y, size_t stride, int n)
{
    double s = 0.0;
    for (int i = 0; i + 1 < n; i++) s += seg_loglog(x[i], x[i + 1], y[(size_t)i * stride], y[(size_t)(i + 1) * stride]);
    return s;
}

double integral_linlog(const double *x, const double *y, int n)
{
    double s = 0.0;
    for (int i = 0; i + 1 < n; i++) {
        double y1 = y[i], y2 = y[i + 1], dx = x[i + 1] - x[i];
        if (y1 == y2) s += y1 * dx;
        else if (y1 > 0.0 && y2 > 0.0) s += (y2 - y1) * dx / std::log(y2 / y1);
    }
    return s;
}

double spacing(double x)
{
    x = std::fabs(x);
    if (x == 0.0) return DBL_MIN;
    return std::nextafter(x, INFINITY) - x;
}

struct DustOffsets {
    size_t nu, log10_nu, chi, albedo, log10_chi, log10_albedo, mu, P1, P2, P3, P4, P1_cdf, P2_cdf;
    size_t emiss_x, emiss_cdf, emiss_bp1, emiss_coarse, jnu_var, log10_jnu_var, mo_e, mo_chi_ross;
    size_t mo_kappa_planck, mo_chi_inv_planck, bnu_cdf, bnu_bp1, bnu_coarse, mono_prob;
    bool have_mo_e, have_mo_chi, have_mrw, have_pda;
};

struct SourceOffsets { size_t x, cdf, bp1; bool have; size_t points, point_cdf; bool have_points; size_t map_cdf; bool have_map; size_t spot_tab; bool have_spots; };
struct PeeledOffsets { size_t view, src_spec, dust_em, dust_chi, filt_off, filt_nu, filt_tr; };

}  // namespace

extern "C" {

int hyp_create(const hyp_problem *pr, int device, hyp_handle *out)
{
    g_error.clear();
    if (out) *out = nullptr;
    if (!pr || !out) return set_error("null argument");
    if (pr->grid.type < 1 || pr->grid.type > 6) return set_error("Unexpected coordinate type (grid types: 1 cartesian, 2 octree, 3 voronoi, 4 amr, 5 spherical polar, 6 cylindrical polar)");
    if (pr->n_dust < 1 || pr->n_dust > HYP_MAX_DUST) return set_error("n_dust must be between 1 and 8");
    // no sources is a valid set-up for dust-only raytracing / monochromatic runs (setup_rt.f90:228-239): the iterations that
    // need sources refuse to start instead (hyp_lucy_launch, hyp_final_launch)
    if (pr->n_sources < 0 || (pr->n_sources > 0 && !pr->sources)) return set_error("invalid source list");
    if (pr->config.monochromatic && (pr->config.n_frequencies < 1 || !pr->config.frequencies)) return set_error("monochromatic mode needs a frequency table");
    const bool is_oct = pr->grid.type == 2, is_vor = pr->grid.type == 3, is_amr = pr->grid.type == 4;
    const bool is_sph = pr->grid.type == 5, is_cyl = pr->grid.type == 6, is_polar = is_sph || is_cyl;
    const bool is_xyz = pr->grid.type == 1;                 // Cartesian proper (walls staged in LDS, brick-tiled schedule)
    const bool is_car = is_xyz || is_polar;                 // three wall arrays, cells (i1, i2, i3)
    std::vector<AmrGrid> amr_grids;
    std::vector<int> amr_go, amr_cell_grid;
    std::vector<double> amr_walls;
    double amr_eps = 0.0;
    int amr_level1 = 0;
    int64_t amr_cells = 0;
    const int n[3] = {is_car ? pr->grid.n1 : 0, is_car ? pr->grid.n2 : 0, is_car ? pr->grid.n3 : 0};
    std::vector<int> vor_seed;
    int vor_g = 1;
    const double *win[3] = {pr->grid.w1, pr->grid.w2, pr->grid.w3};
    std::vector<OctCell> oct_cells;
    std::vector<int> oct_children, oct_neigh;
    if (is_vor) {
        // setup_grid_geometry: grid_geometry_voronoi.f90:96-188
        const int64_t nc = pr->grid.n_cells;
        if (nc < 1 || nc > 2000000000ll || !pr->grid.vor_sites || !pr->grid.vor_idx || !pr->grid.vor_neighs || !pr->grid.vor_volume)
            return set_error("voronoi grid needs sites, volumes and neighbour lists");
        const int32_t *idx = pr->grid.vor_idx, *nei = pr->grid.vor_neighs;
        for (int64_t i = 0; i < nc; i++) if (idx[i + 1] < idx[i]) return set_error("sparse_idx should be non-decreasing");
        for (int64_t k = 0; k < idx[nc]; k++) if (nei[k] < -6 || nei[k] >= nc) return set_error("neighbour index out of range");
        for (int a = 0; a < 3; a++) if (!(pr->grid.vor_box[2 * a + 1] > pr->grid.vor_box[2 * a])) return set_error("voronoi domain is empty");
        // seed grid of the nearest-site walk: nearest site of every seed-cell centre
        const double *S = pr->grid.vor_sites, *B = pr->grid.vor_box;
        auto d2 = [&](int i, const double r[3]) {
            double dx = S[3 * (size_t)i] - r[0], dy = S[3 * (size_t)i + 1] - r[1], dz = S[3 * (size_t)i + 2] - r[2];
            return dx * dx + dy * dy + dz * dz;
        };
        auto nearest_from = [&](const double r[3], int seed) {
            int cur = seed; double dcur = d2(cur, r);
            for (;;) {
                int best = cur; double dbest = dcur;
                for (int k = idx[cur]; k < idx[cur + 1]; k++) {
                    int nb = nei[k];
                    if (nb < 0) continue;
                    double d = d2(nb, r);
                    if (d < dbest) { dbest = d; best = nb; }
                }
                if (best == cur) return cur;
                cur = best; dcur = dbest;
            }
        };
        // about two seed cells per site: the walk from the seed to the nearest site is 0-1 hops for most positions (emission
        // from extended sources places every packet this way); the result does not depend on the seed
        vor_g = (int)std::ceil(std::cbrt((double)nc * 2.0));
        if (vor_g < 1) vor_g = 1;
        if (vor_g > 256) vor_g = 256;
        vor_seed.resize((size_t)vor_g * vor_g * vor_g);
        int last = 0;
        for (int k = 0; k < vor_g; k++) for (int j = 0; j < vor_g; j++) for (int i = 0; i < vor_g; i++) {
            double c[3] = {B[0] + (i + 0.5) / vor_g * (B[1] - B[0]), B[2] + (j + 0.5) / vor_g * (B[3] - B[2]),
                           B[4] + (k + 0.5) / vor_g * (B[5] - B[4])};
            last = nearest_from(c, last);
            vor_seed[((size_t)k * vor_g + j) * vor_g + i] = last;
        }
    } else if (is_polar) {
        // setup_grid_geometry: grid_geometry_spherical_3d.f90:90-203, grid_geometry_cylindrical_3d.f90:90-175
        const double pi = 3.14159265358979323846;
        for (int a = 0; a < 3; a++) if (n[a] < 1 || !win[a]) return set_error("grid walls missing");
        for (int i = 0; i <= n[0]; i++) if (win[0][i] < 0.0) return set_error(is_sph ? "r walls should be positive" : "w walls should be positive");
        for (int i = 0; i <= n[1] && is_sph; i++) if (win[1][i] < 0.0 || win[1][i] > pi) return set_error("theta walls should be between 0 and pi");
        for (int i = 0; i <= n[2]; i++) if (win[2][i] < 0.0 || win[2][i] > 2.0 * pi) return set_error("phi walls should be between 0 and 2*pi");
        static const char *names_s[3] = {"dr", "dt", "dphi"}, *names_c[3] = {"dw", "dz", "dphi"};
        for (int a = 0; a < 3; a++) for (int i = 0; i < n[a]; i++)
            if (win[a][i + 1] - win[a][i] == 0.0)
                return set_error(std::string("all ") + (is_sph ? names_s[a] : names_c[a]) + " values should be greater than zero");
        for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
            const double a0 = win[0][i], b0 = win[0][i + 1], dphi = win[2][k + 1] - win[2][k];
            const double vol = is_sph ? (b0 * b0 * b0 - a0 * a0 * a0) * (std::cos(win[1][j]) - std::cos(win[1][j + 1])) * dphi / 3.0
                                      : (b0 * b0 - a0 * a0) * (win[1][j + 1] - win[1][j]) * dphi / 2.0;
            if (vol == 0.0) return set_error("all volumes should be greater than zero");
        }
    } else if (is_car) {
        for (int a = 0; a < 3; a++) {
            if (n[a] < 1 || !win[a]) return set_error("grid walls missing");
            for (int i = 0; i < n[a]; i++)
                if (!(win[a][i + 1] - win[a][i] > 0.0))
                    return set_error(std::string("all d") + "xyz"[a] + " values should be greater than zero");
        }
        if (is_xyz && (size_t)n[0] + n[1] + n[2] + 3 > 9000) return set_error("grid has too many walls for LDS staging");
    } else if (is_amr) {
        // read_grid/read_level + setup_grid_geometry: grid_geometry_amr.f90:111-508
        const int ng = pr->grid.n_amr_grids, nl = pr->grid.n_amr_levels;
        if (ng < 1 || nl < 1 || !pr->grid.amr_level || !pr->grid.amr_n || !pr->grid.amr_bounds) return set_error("amr grid needs levels and grids");
        amr_grids.resize(ng);
        std::vector<int> level(ng);
        std::vector<std::array<double, 3>> width(ng);
        double min_width = DBL_MAX;
        for (int k = 0; k < ng; k++) {
            AmrGrid &g = amr_grids[k];
            level[k] = pr->grid.amr_level[k];
            if (level[k] < 1 || level[k] > nl || (k > 0 && level[k] < level[k - 1])) return set_error("amr grids must be listed level by level");
            if (level[k] == 1) amr_level1 = k + 1;
            for (int a = 0; a < 3; a++) {
                g.n[a] = pr->grid.amr_n[3 * k + a];
                g.lo[a] = pr->grid.amr_bounds[6 * k + 2 * a]; g.hi[a] = pr->grid.amr_bounds[6 * k + 2 * a + 1];
                if (g.n[a] < 1 || !(g.hi[a] > g.lo[a])) return set_error("all volumes should be greater than zero");
                g.w_off[a] = (int)amr_walls.size();
                // fortranlib linspace: x(i) = (xmax - xmin) * (i - 1) / (n - 1) + xmin
                for (int i = 0; i <= g.n[a]; i++) amr_walls.push_back((g.hi[a] - g.lo[a]) * (double)i / (double)g.n[a] + g.lo[a]);
                width[k][a] = (g.hi[a] - g.lo[a]) / (double)g.n[a];
                if (width[k][a] < min_width) min_width = width[k][a];
            }
            if (amr_cells > 2000000000ll) return set_error("amr grid has too many cells");
            g.start = (unsigned)amr_cells; amr_cells += (int64_t)g.n[0] * g.n[1] * g.n[2];
            g.go_off = (int)amr_go.size();
            amr_go.resize(amr_go.size() + (size_t)(g.n[0] + 2) * (g.n[1] + 2) * (g.n[2] + 2), 0);
        }
        if (amr_cells > 2000000000ll) return set_error("amr grid has too many cells");
        amr_eps = min_width / 2.0;
        auto aligned = [](double x1, double x2, double dx) {
            double r = std::fmod(std::fabs(x1 - x2), dx);
            if (r > 0.5 * dx) r = dx - r;
            return std::fabs(r / dx) < 1.e-8;
        };
        auto first_of_level = [&](int l) { for (int q = 0; q < ng; q++) if (level[q] == l) return q; return -1; };
        char msg[256];
        for (int k = 0; k < ng; k++) {
            const int ref = first_of_level(level[k]), igrid = k - ref + 1;
            for (int a = 0; a < 3; a++) {
                if (std::fabs(width[k][a] - width[ref][a]) > 1.e-10 * width[k][a]) {
                    std::snprintf(msg, sizeof msg, "Grids 1 and %d in level %d have differing cell widths in the %c direction", igrid, level[k], "xyz"[a]);
                    return set_error(msg);
                }
                if (!aligned(amr_grids[k].lo[a], amr_grids[ref].lo[a], width[ref][a])) {
                    std::snprintf(msg, sizeof msg, "Grids 1 and %d in level %d have edges that are not separated by an integer number of cells in the %c direction", igrid, level[k], "xyz"[a]);
                    return set_error(msg);
                }
            }
            if (level[k] > 1) {
                const int pref = first_of_level(level[k] - 1);
                if (pref < 0) return set_error("amr level without grids");
                for (int a = 0; a < 3; a++) {
                    const double rf = width[pref][a] / width[ref][a];
                    if (std::fabs(rf - std::nearbyint(rf)) > 1.e-10) {
                        std::snprintf(msg, sizeof msg, "Refinement factor in the %c direction between level %d and level %d is not an integer (%.3f)", "xyz"[a], level[k] - 1, level[k], rf);
                        return set_error(msg);
                    }
                    if (!aligned(amr_grids[k].lo[a], amr_grids[pref].lo[a], width[pref][a])) {
                        std::snprintf(msg, sizeof msg, "Grid %d in level %d is not aligned with cells in level %d in the %c direction", igrid, level[k], level[k] - 1, "xyz"[a]);
                        return set_error(msg);
                    }
                }
            }
        }
        auto in_grid = [&](int k, const double r[3]) {
            const AmrGrid &g = amr_grids[k];
            for (int a = 0; a < 3; a++) { if (r[a] < g.lo[a]) return false; if (r[a] > g.hi[a]) return false; }
            return true;
        };
        auto go_at = [&](int k, int i1, int i2, int i3) -> int & {
            const AmrGrid &g = amr_grids[k];
            return amr_go[g.go_off + ((size_t)i3 * (g.n[1] + 2) + i2) * (g.n[0] + 2) + i1];
        };
        auto wall = [&](int k, int a, int i) { return amr_walls[amr_grids[k].w_off[a] + i]; };
        // cells overlapped by a grid of the next level (:357-382)
        for (int l1 = nl - 1; l1 >= 1; l1--)
            for (int k1 = 0; k1 < ng; k1++) {
                if (level[k1] != l1) continue;
                const AmrGrid &g1 = amr_grids[k1];
                for (int k2 = 0; k2 < ng; k2++) {
                    if (level[k2] != l1 + 1) continue;
                    const AmrGrid &g2 = amr_grids[k2];
                    bool hit = true;
                    for (int a = 0; a < 3; a++) if (g1.hi[a] < g2.lo[a] || g1.lo[a] > g2.hi[a]) hit = false;
                    if (!hit) continue;
                    for (int i1 = 1; i1 <= g1.n[0]; i1++) for (int i2 = 1; i2 <= g1.n[1]; i2++) for (int i3 = 1; i3 <= g1.n[2]; i3++) {
                        const double r[3] = {0.5 * (wall(k1, 0, i1 - 1) + wall(k1, 0, i1)), 0.5 * (wall(k1, 1, i2 - 1) + wall(k1, 1, i2)),
                                             0.5 * (wall(k1, 2, i3 - 1) + wall(k1, 2, i3))};
                        if (in_grid(k2, r)) go_at(k1, i1, i2, i3) = k2 + 1;
                    }
                }
            }
        // one step outside each grid: the grid of the same or a coarser level found there (:384-486)
        for (int k1 = 0; k1 < ng; k1++) {
            const AmrGrid &g1 = amr_grids[k1];
            for (int l2 = level[k1]; l2 >= 1; l2--)
                for (int k2 = 0; k2 < ng; k2++) {
                    if (level[k2] != l2 || k2 == k1) continue;
                    const AmrGrid &g2 = amr_grids[k2];
                    bool close = true;
                    for (int a = 0; a < 3; a++)
                        if (g1.hi[a] < g2.lo[a] - width[k2][a] * 0.5 || g1.lo[a] > g2.hi[a] + width[k2][a] * 0.5) close = false;
                    if (!close) continue;
                    for (int a = 0; a < 3; a++) {
                        const int b = (a + 1) % 3, c = (a + 2) % 3;
                        for (int side = 0; side < 2; side++) {
                            int idx[3]; double r[3];
                            idx[a] = side ? g1.n[a] + 1 : 0;
                            r[a] = side ? g1.hi[a] + width[k1][a] * 0.5 : g1.lo[a] - width[k1][a] * 0.5;
                            for (int ib = 1; ib <= g1.n[b]; ib++) for (int ic = 1; ic <= g1.n[c]; ic++) {
                                idx[b] = ib; idx[c] = ic;
                                r[b] = 0.5 * (wall(k1, b, ib - 1) + wall(k1, b, ib)); r[c] = 0.5 * (wall(k1, c, ic - 1) + wall(k1, c, ic));
                                int &q = go_at(k1, idx[0], idx[1], idx[2]);
                                if (in_grid(k2, r) && q == 0) q = k2 + 1;
                            }
                        }
                    }
                }
        }
        amr_cell_grid.resize((size_t)amr_cells);
        for (int k = 0; k < ng; k++) {
            const AmrGrid &g = amr_grids[k];
            const size_t nc = (size_t)g.n[0] * g.n[1] * g.n[2];
            for (size_t c = 0; c < nc; c++) amr_cell_grid[g.start + c] = k;
        }
    } else {
        // setup_grid_geometry + octree_setup_indiv: grid_geometry_octree.f90:147-246
        const int64_t nc = pr->grid.n_cells;
        if (nc < 1 || nc > 2000000000ll || !pr->grid.refined) return set_error("octree needs a refined list");
        oct_cells.resize((size_t)nc);
        oct_children.assign((size_t)nc * 8, -1);
        for (int a = 0; a < 3; a++) if (!(pr->grid.oct_half[a] > 0.0)) return set_error("all volumes should be greater than zero");
        std::vector<double> hx((size_t)nc);   // x half-widths only to detect underflow of the level encoding
        OctCell &root = oct_cells[0];
        root.x = pr->grid.oct_center[0]; root.y = pr->grid.oct_center[1]; root.z = pr->grid.oct_center[2];
        root.parent = -1; root.subcell = -1; root.level = 0; root.refined = pr->grid.refined[0] == 1; root.pad = 0;
        std::vector<std::pair<int, int>> stack;
        if (root.refined) stack.push_back({0, 0});
        int64_t filled = 1;
        while (!stack.empty()) {
            int par = stack.back().first, k = stack.back().second;
            if (k == 8) { stack.pop_back(); continue; }
            stack.back().second = k + 1;
            if (filled >= nc) return set_error("refined array is not self-consistent");
            int c = (int)filled++;
            oct_children[(size_t)par * 8 + k] = c;
            const OctCell &pc = oct_cells[par];
            const int lev = pc.level;
            OctCell &cc = oct_cells[c];
            const double hpx = std::ldexp(pr->grid.oct_half[0], -lev), hpy = std::ldexp(pr->grid.oct_half[1], -lev),
                         hpz = std::ldexp(pr->grid.oct_half[2], -lev);
            cc.x = pc.x + ((k & 1) ? 1 : -1) * hpx / 2.0;
            cc.y = pc.y + ((k & 2) ? 1 : -1) * hpy / 2.0;
            cc.z = pc.z + ((k & 4) ? 1 : -1) * hpz / 2.0;
            cc.parent = par; cc.subcell = (signed char)k; cc.pad = 0;
            if (lev + 1 > 200) return set_error("octree too deep");
            cc.level = (unsigned char)(lev + 1);
            cc.refined = pr->grid.refined[c] == 1;
            if (cc.refined) stack.push_back({c, 0});
        }
        if (filled != nc) return set_error("refined array is not self-consistent");
        // neighbour across each face, no finer than the cell itself: what next_cell_int (:328-347) finds when its descent is
        // stopped at the cell's own level (geo_advance goes on from there)
        oct_neigh.assign((size_t)nc * 6, (int)nc);
        std::vector<int> subs(256);
        for (int64_t id = 1; id < nc; id++)
            for (int axis = 0; axis < 3; axis++)
                for (int up = 0; up < 2; up++) {
                    int cur = (int)id, depth = 0, n = (int)nc;
                    while (cur != 0) {
                        const int sub = oct_cells[cur].subcell, par = oct_cells[cur].parent;
                        if (((sub >> axis) & 1) != up) {
                            int S = oct_children[(size_t)par * 8 + (up ? (sub | (1 << axis)) : (sub & ~(1 << axis)))];
                            while (oct_cells[S].refined && depth > 0) {
                                const int sc = subs[--depth];
                                S = oct_children[(size_t)S * 8 + ((sc & ~(1 << axis)) | ((up ? 0 : 1) << axis))];
                            }
                            n = S;
                            break;
                        }
                        subs[depth++] = sub; cur = par;
                    }
                    oct_neigh[(size_t)id * 6 + 2 * axis + up] = n;
                }
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error("no HIP device available: the photon-packet engine requires an AMD GPU (gfx950)");
    if (device < 0 || device >= ndev) return set_error("invalid device ordinal");
    HIP_TRY(hipSetDevice(device));

    hyp_engine *h = new hyp_engine();
    h->device = device;
    h->cfg = pr->config;
    h->n_dust = pr->n_dust;
    h->n_cells = is_car ? (size_t)n[0] * n[1] * n[2] : is_amr ? (size_t)amr_cells : (size_t)pr->grid.n_cells;
    h->n_elem = h->n_cells * h->n_dust;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->n_cu = prop.multiProcessorCount;
    if (h->n_cu <= 0) h->n_cu = 256;

#define FAIL(msg) do { g_error = (msg); hyp_destroy(h); return 1; } while (0)
#define HIPC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_error = std::string(#call) + ": " + hipGetErrorString(e_); hyp_destroy(h); return 1; } } while (0)

    Blob B;
    DProblem &P = h->hp;
    std::memset(&P, 0, sizeof(P));
    P.n1 = n[0]; P.n2 = n[1]; P.n3 = n[2]; P.n_dust = pr->n_dust;
    P.grid_type = pr->grid.type;
    if (is_oct) {
        double m = 0.0;
        for (int a = 0; a < 3; a++) {
            P.oct_half[a] = pr->grid.oct_half[a];
            P.oct_box[2 * a] = pr->grid.oct_center[a] - pr->grid.oct_half[a];
            P.oct_box[2 * a + 1] = pr->grid.oct_center[a] + pr->grid.oct_half[a];
            if (pr->grid.oct_half[a] > m) m = pr->grid.oct_half[a];
        }
        P.oct_eps = spacing(m) * 3.0;   // grid_geometry_octree.f90:243
    }
    P.n_sources = pr->n_sources; P.n_peeled = pr->n_peeled;
    P.sample_sources_evenly = pr->config.sample_sources_evenly;
    P.kill_on_absorb = pr->config.kill_on_absorb; P.kill_on_scatter = pr->config.kill_on_scatter;
    P.forced_first = pr->config.forced_first_interaction; P.forced_algo = pr->config.forced_first_interaction_algorithm;
    P.n_inter_max = pr->config.n_inter_max; P.n_reabs_max = pr->config.n_reabs_max; P.n_cells = h->n_cells; P.baes16_xi = pr->config.baes16_xi;
    {
        P.check_p = pr->config.propagation_check_frequency;
        P.check_log1mp = (P.check_p > 0.0 && P.check_p < 1.0) ? std::log1p(-P.check_p) : -1.0;
        int64_t sd = pr->config.seed;
        uint64_t s = (uint64_t)(sd < 0 ? -sd : sd);
        P.seed_key = (uint32_t)s ^ (uint32_t)(s >> 32);
    }

    // walls + 3*spacing(w): grid_geometry_cartesian_3d.f90:97-132
    // (Cartesian walls beyond 2^300: the reference's cell volumes dx dy dz overflow there; the wall search relies on path lengths
    // (w - r) / v staying finite, find_wall_ahead in hyp_kernels.h)
    for (int a = 0; a < 3 && is_car && !is_polar; a++)
        for (int i = 0; i <= n[a]; i++)
            if (!(std::fabs(win[a][i]) < 0x1p300)) return set_error("grid walls beyond 2^300 are not supported (cell volumes overflow)");
    size_t w_off[3] = {0, 0, 0}, ew_off[3] = {0, 0, 0};
    for (int a = 0; a < 3 && is_car; a++) {
        w_off[a] = B.put(win[a], n[a] + 1);
        std::vector<double> ew(n[a] + 1);
        for (int i = 0; i <= n[a]; i++) ew[i] = 3.0 * spacing(win[a][i]);
        // angles: ew = 3 * spacing(1) (spherical_3d.f90:199-201, cylindrical_3d.f90:171-173)
        if (is_polar && (a == 2 || (a == 1 && is_sph))) for (int i = 0; i <= n[a]; i++) ew[i] = 3.0 * spacing(1.0);
        ew_off[a] = B.put(ew);
    }
    size_t polar_off[5] = {0, 0, 0, 0, 0};
    int midplane = -2;
    if (is_polar) {
        std::vector<double> wr2(n[0] + 1), wtanp(n[2] + 1);
        for (int i = 0; i <= n[0]; i++) wr2[i] = win[0][i] * win[0][i];
        for (int i = 0; i <= n[2]; i++) wtanp[i] = std::tan(win[2][i]);
        polar_off[0] = B.put(wr2); polar_off[4] = B.put(wtanp);
        if (is_sph) {
            std::vector<double> wtant(n[1] + 1), wtant2(n[1] + 1), wcost(n[1] + 1);
            double m = DBL_MAX; int im = 0;
            for (int i = 0; i <= n[1]; i++) {
                wtant[i] = std::tan(win[1][i]); wtant2[i] = wtant[i] * wtant[i]; wcost[i] = std::cos(win[1][i]);
                const double d = std::fabs(win[1][i] - 3.14159265358979323846 / 2.0);
                if (d < m) { m = d; im = i; }
            }
            if (m < 1.e-6) midplane = im;       // :175: minloc(abs(w2 - pi/2)) if any is within 1e-6
            polar_off[1] = B.put(wtant); polar_off[2] = B.put(wtant2); polar_off[3] = B.put(wcost);
        }
    }

    // dust tables: dust_type_4elem.f90:78-293
    std::vector<DustOffsets> doff(pr->n_dust);
    for (int d = 0; d < pr->n_dust; d++) {
        const hyp_dust_desc &in = pr->dust[d];
        DDust &D = P.dust[d];
        DustOffsets &O = doff[d];
        const int nn = in.n_nu, nm = in.n_mu;
        if (nn < 2 || nm < 2 || in.n_jnu < 2 || in.n_enu < 2) FAIL("dust tables too short");
        D.n_nu = nn; D.n_mu = nm; D.n_jnu = in.n_jnu; D.n_enu = in.n_enu; D.n_e = in.n_e;
        D.sublimation_mode = in.sublimation_mode;
        D.sublimation_specific_energy = in.sublimation_specific_energy;
        D.minimum_specific_energy = in.minimum_specific_energy;
        D.nu_min = in.nu[0]; D.nu_max = in.nu[nn - 1];
        D.mu_min = in.mu[0]; D.mu_max = in.mu[nm - 1];
        for (int i = 0; i + 1 < nn; i++) if (!(in.nu[i + 1] > in.nu[i])) FAIL("dust frequencies should be monotonically increasing");
        std::vector<double> lnu(nn), lchi(nn), lalb(nn);
        for (int i = 0; i < nn; i++) {
            lnu[i] = std::log10(in.nu[i]);
            lchi[i] = in.chi[i] > 0.0 ? std::log10(in.chi[i]) : std::nan("");
            lalb[i] = in.albedo[i] > 0.0 ? std::log10(in.albedo[i]) : std::nan("");
        }
        O.nu = B.put(in.nu, nn); O.log10_nu = B.put(lnu);
        O.chi = B.put(in.chi, nn); O.albedo = B.put(in.albedo, nn);
        O.log10_chi = B.put(lchi); O.log10_albedo = B.put(lalb);
        O.mu = B.put(in.mu, nm);
        const size_t np = (size_t)nn * nm;
        std::vector<double> P1(in.P1, in.P1 + np), P2(in.P2, in.P2 + np), P3(in.P3, in.P3 + np), P4(in.P4, in.P4 + np);
        D.zero_p2 = 1;
        for (size_t i = 0; i < np; i++) if (P2[i] != 0.0) { D.zero_p2 = 0; break; }
        const double dmu = D.mu_max - D.mu_min;
        std::vector<double> C1(np, 0.0), C2(np, 0.0);
        for (int j = 0; j < nn; j++) {
            double *p1 = &P1[(size_t)j * nm], *p2 = &P2[(size_t)j * nm], *p3 = &P3[(size_t)j * nm], *p4 = &P4[(size_t)j * nm];
            double norm = integral_linlog(in.mu, p1, nm);
            if (norm == 0.0) FAIL("P1 matrix normalization is zero");
            for (int i = 0; i < nm; i++) {
                p1[i] = p1[i] / norm * dmu; p2[i] = p2[i] / norm * dmu;
                p3[i] = p3[i] / norm * dmu; p4[i] = p4[i] / norm * dmu;
            }
            double *c1 = &C1[(size_t)j * nm], *c2 = &C2[(size_t)j * nm];
            for (int i = 1; i < nm; i++) {
                double dx = in.mu[i] - in.mu[i - 1];
                c1[i] = c1[i - 1] + 0.5 * (p1[i] + p1[i - 1]) * dx;
                c2[i] = c2[i - 1] + 0.5 * (p2[i] + p2[i - 1]) * dx;
            }
            bool z1 = true, z2 = true;
            for (int i = 0; i < nm; i++) { if (c1[i] != 0.0) z1 = false; if (c2[i] != 0.0) z2 = false; }
            if (!z1) { double l = c1[nm - 1]; for (int i = 0; i < nm; i++) c1[i] /= l; }
            if (!z2) { double l = c2[nm - 1]; for (int i = 0; i < nm; i++) c2[i] /= l; }
        }
        O.P1 = B.put(P1); O.P2 = B.put(P2); O.P3 = B.put(P3); O.P4 = B.put(P4);
        O.P1_cdf = B.put(C1); O.P2_cdf = B.put(C2);
        // emissivities
        O.emiss_x = B.put(in.emiss_nu, in.n_enu);
        std::vector<double> cdf_all, bp1_all, cdf, bp1;
        for (int i = 0; i < in.n_jnu; i++) {
            if (!build_log_pdf(in.emiss_nu, in.emiss_jnu + i, in.n_enu, in.n_jnu, cdf, bp1)) FAIL("emissivity has zero integral");
            cdf_all.insert(cdf_all.end(), cdf.begin(), cdf.end());
            bp1_all.insert(bp1_all.end(), bp1.begin(), bp1.end());
        }
        O.emiss_cdf = B.put(cdf_all); O.emiss_bp1 = B.put(bp1_all);
        {   // every HYP_COARSE-th CDF entry of each row, for the two-level search of sample_log_pdf_pair
            const int nc = (in.n_enu + HYP_COARSE - 1) / HYP_COARSE;
            std::vector<double> coarse((size_t)in.n_jnu * nc);
            for (int i = 0; i < in.n_jnu; i++)
                for (int m = 0; m < nc; m++) coarse[(size_t)i * nc + m] = cdf_all[(size_t)i * in.n_enu + (size_t)m * HYP_COARSE];
            O.emiss_coarse = B.put(coarse);
        }
        std::vector<double> ljv(in.n_jnu);
        for (int i = 0; i < in.n_jnu; i++) ljv[i] = std::log10(in.emiss_var[i]);
        O.jnu_var = B.put(in.emiss_var, in.n_jnu); O.log10_jnu_var = B.put(ljv);
        O.have_mo_e = in.n_e > 0 && in.mo_specific_energy;
        O.have_mo_chi = O.have_mo_e && in.mo_chi_rosseland;
        if (O.have_mo_e) {
            for (int i = 1; i < in.n_e; i++)
                if (in.mo_specific_energy[i] < in.mo_specific_energy[i - 1]) FAIL("energy per unit mass is not monotonically increasing");
            O.mo_e = B.put(in.mo_specific_energy, in.n_e);
            D.e_min = in.mo_specific_energy[0]; D.e_max = in.mo_specific_energy[in.n_e - 1]; D.have_e_range = 1;
        }
        if (O.have_mo_chi) O.mo_chi_ross = B.put(in.mo_chi_rosseland, in.n_e);
        if (in.sublimation_mode == 2 && !O.have_mo_chi) FAIL("slow sublimation needs the Rosseland mean opacity table");
        O.have_pda = false;
        if (pr->config.pda) {       // setup_rt.f90:289-300; grid_pda_3d.f90 reads kappa_planck and chi_rosseland
            if (in.version == 1)
                FAIL("version 1 dust files can no longer be used when PDA is computed due to a bug - to fix this, re-generate the dust file using the latest version of Hyperion");
            if (!(O.have_mo_chi && in.mo_kappa_planck)) FAIL("PDA needs the kappa_planck and chi_rosseland mean opacities of every dust type");
            if (!pr->config.mrw) { O.mo_kappa_planck = B.put(in.mo_kappa_planck, in.n_e); O.have_pda = true; }
        }
        // modified random walk: Planck means + b_nu = j_nu / kappa_nu pdfs (dust_type_4elem.f90:289-291)
        O.have_mrw = false;
        if (pr->config.mrw) {
            if (!(O.have_mo_e && in.mo_kappa_planck && in.mo_chi_inv_planck))
                FAIL("MRW needs the kappa_planck and chi_inv_planck mean opacities of every dust type");
            O.mo_kappa_planck = B.put(in.mo_kappa_planck, in.n_e);
            O.mo_chi_inv_planck = B.put(in.mo_chi_inv_planck, in.n_e);
            std::vector<double> kap(nn), y(in.n_enu), bcdf_all, bbp1_all;
            for (int k = 0; k < nn; k++) kap[k] = in.chi[k] * (1.0 - in.albedo[k]);
            for (int i = 0; i < in.n_jnu; i++) {
                for (int k = 0; k < in.n_enu; k++)
                    y[k] = in.emiss_jnu[(size_t)k * in.n_jnu + i] / interp1d_loglog_host(in.nu, kap.data(), nn, in.emiss_nu[k]);
                if (!build_log_pdf(in.emiss_nu, y.data(), in.n_enu, 1, cdf, bp1)) FAIL("emissivity / kappa_nu has zero integral");
                bcdf_all.insert(bcdf_all.end(), cdf.begin(), cdf.end());
                bbp1_all.insert(bbp1_all.end(), bp1.begin(), bp1.end());
            }
            O.bnu_cdf = B.put(bcdf_all); O.bnu_bp1 = B.put(bbp1_all);
            const int nc = (in.n_enu + HYP_COARSE - 1) / HYP_COARSE;
            std::vector<double> coarse((size_t)in.n_jnu * nc);
            for (int i = 0; i < in.n_jnu; i++)
                for (int m = 0; m < nc; m++) coarse[(size_t)i * nc + m] = bcdf_all[(size_t)i * in.n_enu + (size_t)m * HYP_COARSE];
            O.bnu_coarse = B.put(coarse);
            O.have_mrw = true;
        }
    }
    // cumulative of Min et al. (2009) eq. 6 on 100 points: grid_mrw_3d.f90:157-195
    size_t mrw_x_off = 0, mrw_y_off = 0;
    if (pr->config.mrw) {
        std::vector<double> mx(100), my(100);
        for (int i = 0; i < 100; i++) {
            mx[i] = (double)i / 99.0;
            double y = 0.0;
            if (i == 99) y = 0.5;
            else for (long long j = 1;; j++) {
                const double term = std::pow(mx[i], (double)(j * j));
                if (term == 0.0) break;
                if (j % 2 == 0) y -= term; else y += term;
            }
            my[i] = y * 2.0;
        }
        mrw_x_off = B.put(mx); mrw_y_off = B.put(my);
    }

    // sources: source.f90:47-84, source_type.f90:102-322
    std::vector<DSource> hs(pr->n_sources);
    std::vector<SourceOffsets> soff(pr->n_sources);
    h->energy_total = 0.0;
    // luminosity of a point collection = sum of its members (source_type.f90:271)
    std::vector<double> src_lum(pr->n_sources);
    for (int i = 0; i < pr->n_sources; i++) {
        const hyp_source_desc &s = pr->sources[i];
        src_lum[i] = s.luminosity;
        if (s.type == 8 && s.point_lum && s.n_points > 0) { src_lum[i] = 0.0; for (int k = 0; k < s.n_points; k++) src_lum[i] += s.point_lum[k]; }
        h->energy_total += src_lum[i];
        soff[i].have_points = false; soff[i].have_map = false; soff[i].have_spots = false;
    }
    {
        double c = 0.0;
        for (int i = 0; i < pr->n_sources; i++) {
            const hyp_source_desc &s = pr->sources[i];
            DSource &S = hs[i];
            std::memset(&S, 0, sizeof(S));
            if (s.type != 1 && s.type != 2 && s.type != 4 && s.type != 5 && s.type != 6 && s.type != 7 && s.type != 8) FAIL("unknown type in source list: " + std::to_string(s.type));
            S.type = s.type; S.peeloff = s.peeloff; S.radius = s.radius; S.limb_darkening = s.limb_darkening;
            if (is_vor && s.type == 1) {
                // every packet of a point source starts in the same cell: find_cell (grid_geometry_voronoi.f90:196-229) once, here
                const double *Sx = pr->grid.vor_sites, *Bx = pr->grid.vor_box;
                const double r[3] = {s.position[0], s.position[1], s.position[2]};
                if (!(r[0] < Bx[0] || r[0] > Bx[1] || r[1] < Bx[2] || r[1] > Bx[3] || r[2] < Bx[4] || r[2] > Bx[5])) {
                    int id[3];
                    for (int a = 0; a < 3; a++) {
                        const double f = (r[a] - Bx[2 * a]) / (Bx[2 * a + 1] - Bx[2 * a]);
                        const int q = (int)(f * vor_g);
                        id[a] = q < 0 ? 0 : (q >= vor_g ? vor_g - 1 : q);
                    }
                    auto d2 = [&](int c) { const double dx = Sx[3 * (size_t)c] - r[0], dy = Sx[3 * (size_t)c + 1] - r[1], dz = Sx[3 * (size_t)c + 2] - r[2]; return dx * dx + dy * dy + dz * dz; };
                    int cur = vor_seed[((size_t)id[2] * vor_g + id[1]) * vor_g + id[0]];
                    double dcur = d2(cur);
                    for (;;) {
                        int best = cur; double dbest = dcur;
                        for (int k = pr->grid.vor_idx[cur]; k < pr->grid.vor_idx[cur + 1]; k++) {
                            const int nb = pr->grid.vor_neighs[k];
                            if (nb < 0) continue;
                            const double d = d2(nb);
                            if (d < dbest) { dbest = d; best = nb; }
                        }
                        if (best == cur) break;
                        cur = best; dcur = dbest;
                    }
                    S.vor_cell1 = cur + 1;
                }
            }
            if (s.type == 2) P.any_intersect = 1;      // s%intersect = .true.: source_type.f90:148
            if (s.type == 7) {      // plane_parallel: source_type.f90:239-256
                const double th = s.direction[0] * HYP_PI / 180.0, ph = s.direction[1] * HYP_PI / 180.0;
                S.dir_cost = std::cos(th); S.dir_sint = std::sin(th); S.dir_cosp = std::cos(ph); S.dir_sinp = std::sin(ph);
                if (s.peeloff) FAIL("plane parallel sources cannot be peeled off (source_emit_peeloff has no case for them)");
            }
            if (s.type == 8) {      // point_collection: source_type.f90:258-277
                if (s.n_points < 1 || !s.points || !s.point_lum) FAIL("point source collection needs positions and luminosities");
                std::vector<double> cdf(s.n_points);
                double tot = 0.0, c = 0.0;
                for (int k = 0; k < s.n_points; k++) tot += s.point_lum[k];
                for (int k = 0; k < s.n_points; k++) { c += s.point_lum[k] / tot; cdf[k] = c; }
                for (int k = 0; k < s.n_points; k++) cdf[k] /= c;
                S.n_points = s.n_points;
                soff[i].points = B.put(s.points, 3 * (size_t)s.n_points); soff[i].point_cdf = B.put(cdf);
                soff[i].have_points = true;
            }
            if (s.n_spots > 0) {    // spotted sphere: source_type.f90:150-188
                if (s.type != 2 || !s.spots) FAIL("only spherical sources can have spots");
                const int ns = s.n_spots;
                std::vector<double> tab((size_t)(ns + 1) + (size_t)ns * SPOT_STRIDE, 0.0);
                double tot = s.luminosity, cc = 0.0;
                for (int k = 0; k < ns; k++) tot += s.spots[k].luminosity;
                for (int k = 0; k <= ns; k++) { cc += (k < ns ? s.spots[k].luminosity : s.luminosity) / tot; tab[k] = cc; }
                for (int k = 0; k <= ns; k++) tab[k] /= cc;
                for (int k = 0; k < ns; k++) {
                    const hyp_spot_desc &q = s.spots[k];
                    double *t = tab.data() + (ns + 1) + (size_t)k * SPOT_STRIDE;
                    // angle3d_deg(lon, lat) as the reference passes them (theta = lon, phi = lat), then angle3d_to_vector3d
                    const double th = q.longitude * HYP_PI / 180.0, ph = q.latitude * HYP_PI / 180.0;
                    t[0] = std::sin(th) * std::cos(ph); t[1] = std::sin(th) * std::sin(ph); t[2] = std::cos(th);
                    t[3] = std::cos(q.radius * HYP_PI / 180.0);
                    t[4] = q.spectrum_type; t[5] = q.temperature; t[6] = q.n_spec;
                    if (q.spectrum_type == 1) {
                        std::vector<double> cdf, bp1;
                        if (!build_log_pdf(q.spec_nu, q.spec_fnu, q.n_spec, 1, cdf, bp1)) FAIL("source spectrum has zero integral");
                        t[7] = (double)B.put(q.spec_nu, q.n_spec); t[8] = (double)B.put(cdf); t[9] = (double)B.put(bp1);
                    } else if (q.spectrum_type != 2) FAIL("Spot cannot have LTE spectrum");
                    if (pr->config.monochromatic) {
                        std::vector<double> mp(pr->config.n_frequencies);
                        for (int f = 0; f < pr->config.n_frequencies; f++)
                            mp[f] = q.spectrum_type == 1 ? interp_log_pdf(q.spec_nu, q.spec_fnu, 1, q.n_spec, pr->config.frequencies[f])
                                                         : normalized_B_nu(pr->config.frequencies[f], q.temperature);
                        t[10] = (double)B.put(mp);
                    }
                }
                S.n_spots = ns;
                soff[i].spot_tab = B.put(tab); soff[i].have_spots = true;
            }
            if (s.type == 4) {      // map: source_type.f90:190-199, set_pdf(luminosity_map, map) over all cells
                if (!s.map) FAIL("map source needs a luminosity map");
                const size_t nc = h->n_cells;
                std::vector<double> cdf(nc);
                double tot = 0.0, cc = 0.0;
                for (size_t k = 0; k < nc; k++) tot += s.map[k];
                if (!(tot > 0.0)) FAIL("luminosity map is zero everywhere");
                for (size_t k = 0; k < nc; k++) { cc += s.map[k] / tot; cdf[k] = cc; }
                for (size_t k = 0; k < nc; k++) cdf[k] /= cc;
                soff[i].map_cdf = B.put(cdf); soff[i].have_map = true;
            }
            for (int k = 0; k < 6; k++) S.box[k] = s.box[k];
            if (s.type == 6) {   // face pdf ~ face areas: source_type.f90:233-237
                double dx = s.box[1] - s.box[0], dy = s.box[3] - s.box[2], dz = s.box[5] - s.box[4];
                double a[6] = {dy * dz, dy * dz, dz * dx, dz * dx, dx * dy, dx * dy}, cc = 0.0, tot = 0.0;
                for (int k = 0; k < 6; k++) tot += a[k];
                for (int k = 0; k < 6; k++) { cc += a[k] / tot; S.face_cdf[k] = cc; }
                for (int k = 0; k < 6; k++) S.face_cdf[k] /= cc;
            }
            S.pos[0] = s.position[0]; S.pos[1] = s.position[1]; S.pos[2] = s.position[2];
            S.temperature = s.temperature; S.spectrum_type = s.spectrum_type; S.n_spec = s.n_spec;
            S.lum_pdf = src_lum[i] / h->energy_total;
            c += S.lum_pdf; S.lum_cdf = c;
            soff[i].have = false;
            if (s.spectrum_type == 1) {
                for (int k = 0; k + 1 < s.n_spec; k++)
                    if (s.spec_nu[k + 1] < s.spec_nu[k]) FAIL("spectrum frequency should be monotonically increasing");
                std::vector<double> cdf, bp1;
                if (!build_log_pdf(s.spec_nu, s.spec_fnu, s.n_spec, 1, cdf, bp1)) FAIL("source spectrum has zero integral");
                soff[i].x = B.put(s.spec_nu, s.n_spec); soff[i].cdf = B.put(cdf); soff[i].bp1 = B.put(bp1);
                soff[i].have = true;
            } else if (s.spectrum_type == 3 && s.type == 4) {
                // 'lte': the emissivity of the dust in the emitting cell
            } else if (s.spectrum_type != 2)
                FAIL(std::string(s.type == 5 ? "External spherical source" : s.type == 6 ? "External box source" : s.type == 2 ? "Spherical source" : s.type == 7 ? "Plane parallel" : s.type == 8 ? "Point source collection" : "Point source") + " cannot have LTE spectrum");
        }
        for (int i = 0; i < pr->n_sources; i++) hs[i].lum_cdf /= c;
    }

    // peeled image groups: images_peeled.f90:272-380, image_type.f90:153-335
    // the binned image group (images_binned.f90:42-56), if any, is one more image group after the peeled ones:
    // n_theta x n_phi "views", never peeled into (P.n_peeled stays the number of peeled groups)
    const int n_groups = pr->n_peeled + (pr->binned ? 1 : 0);
    std::vector<hyp_peeled_desc> pdesc(pr->peeled, pr->peeled + pr->n_peeled);
    std::vector<double> binned_angles;
    if (pr->binned) {
        if (pr->config.monochromatic) FAIL("can't use binned images in exact wavelength mode");                       // setup_rt.f90:328
        if (pr->config.forced_first_interaction) FAIL("can't use binned images with forced first interaction");      // :329
        if (pr->n_binned_theta < 1 || pr->n_binned_phi < 1) FAIL("n_theta and n_phi should be positive");
        hyp_peeled_desc bd = *pr->binned;
        bd.n_view = pr->n_binned_theta * pr->n_binned_phi; bd.inside_observer = 0;
        binned_angles.assign((size_t)bd.n_view, 0.0);
        bd.theta = binned_angles.data(); bd.phi = binned_angles.data();
        pdesc.push_back(bd);
    }
    h->h_peeled.resize(n_groups);
    std::vector<PeeledOffsets> poff(n_groups);
    std::vector<int> ray_groups;
    h->sed_off.assign(n_groups, 0); h->img_off.assign(n_groups, 0);
    h->sed_n.assign(n_groups, 0); h->img_n.assign(n_groups, 0);
    size_t img_total = 0;
    int views_total = 0;
    for (int g = 0; g < n_groups; g++) {
        const hyp_peeled_desc &in = pdesc[g];
        DPeeled &G = h->h_peeled[g];
        std::memset(&G, 0, sizeof(G));
        G.view_base = views_total;
        if (g < pr->n_peeled && in.n_view > 0) views_total += in.n_view;
        if (in.inside_observer) {       // images_peeled.f90:312-315, 356-363
            if (in.compute_image && in.x_min < in.x_max) FAIL("longitudes should increase towards the left for inside observers");
            if (in.compute_sed) FAIL("computing SEDs for inside observers is not supported");
        }
        if (in.n_view < 1) FAIL("n_view should be a positive integer");
        G.n_view = in.n_view; G.ignore_optical_depth = in.ignore_optical_depth;
        G.compute_image = in.compute_image; G.compute_sed = in.compute_sed;
        G.n_x = in.n_x; G.n_y = in.n_y; G.n_ap = in.n_ap; G.n_nu = in.n_nu;
        if (pr->config.monochromatic) {     // image_type.f90:243-258
            if (in.inu_min < 1 || in.inu_min > pr->config.n_frequencies) FAIL("inu_min value is out of range");
            if (in.inu_max < 1 || in.inu_max > pr->config.n_frequencies) FAIL("inu_max value is out of range");
            G.n_nu = in.inu_max - in.inu_min + 1; G.inu_min = in.inu_min;
            if (G.n_nu != in.n_nu) FAIL("n_nu of a monochromatic image group should be inu_max - inu_min + 1");
        }
        G.track_origin = in.track_origin; G.track_n_scat = in.track_n_scat; G.uncertainties = in.uncertainties;
        G.n_stokes = in.compute_stokes ? 4 : 1;
        switch (in.track_origin) {
        case 0: G.n_orig = 1; break;
        case 1: G.n_orig = 4; break;
        case 2: G.n_orig = 2 * (pr->n_sources + pr->n_dust); break;
        case 3: G.n_orig = 4 + 2 * in.track_n_scat; break;
        default: FAIL("unknown track_origin flag");
        }
        G.x_min = in.x_min; G.x_max = in.x_max; G.y_min = in.y_min; G.y_max = in.y_max;
        G.ap_min = in.ap_min; G.ap_max = in.ap_max;
        G.log10_nu_min = std::log10(in.nu_min); G.log10_nu_max = std::log10(in.nu_max);
        if (in.compute_sed) { G.log10_ap_min = std::log10(in.ap_min); G.log10_ap_max = std::log10(in.ap_max); }
        G.d_min = in.d_min; G.d_max = in.d_max;
        G.inside_observer = in.inside_observer ? 1 : 0;
        if (in.inside_observer && G.d_min < 0.0) G.d_min = 0.0;
        for (int k = 0; k < 3; k++) G.origin[k] = in.peeloff_origin[k];
        std::vector<double> view((size_t)in.n_view * 4);
        for (int v = 0; v < in.n_view; v++) {
            double t = in.theta[v] * HYP_PI / 180.0, f = in.phi[v] * HYP_PI / 180.0;
            view[4 * v + 0] = std::cos(t); view[4 * v + 1] = std::sin(t);
            view[4 * v + 2] = std::cos(f); view[4 * v + 3] = std::sin(f);
        }
        poff[g].view = B.put(view);
        if (in.use_filters) {       // image_type.f90:173-181,285-291; images_peeled.f90:349-351
            if (pr->config.monochromatic) FAIL("cannot use filters in monochromatic mode");
            if (pr->config.raytracing && g < pr->n_peeled) FAIL("filter convolution cannot be used with raytracing");
            if (!in.filt_n || !in.filt_nu || !in.filt_tr) FAIL("filter tables are missing");
            std::vector<double> off(in.n_nu + 1, 0.0);
            for (int i = 0; i < in.n_nu; i++) {
                if (in.filt_n[i] < 2) FAIL("a filter needs at least two points");
                off[i + 1] = off[i] + in.filt_n[i];
            }
            poff[g].filt_off = B.put(off);
            poff[g].filt_nu = B.put(in.filt_nu, (int)off[in.n_nu]);
            poff[g].filt_tr = B.put(in.filt_tr, (int)off[in.n_nu]);
            G.use_filters = 1;
        }
        if (pr->config.raytracing && g < pr->n_peeled) ray_groups.push_back(g);
        if (in.compute_sed) {
            h->sed_n[g] = (size_t)G.n_stokes * G.n_orig * in.n_view * in.n_ap * in.n_nu;
            h->sed_off[g] = img_total; img_total += 2 * h->sed_n[g];
        }
        if (in.compute_image) {
            h->img_n[g] = (size_t)G.n_stokes * G.n_orig * in.n_view * in.n_y * in.n_x * in.n_nu;
            h->img_off[g] = img_total; img_total += 2 * h->img_n[g];
        }
    }

    // Raytracing caches (images_peeled.f90:422-538): source spectra, dust emissivities and opacities
    // binned on each group's frequency grid with get_spectrum_binned (source_type.f90:1118-1172),
    // get_j_nu_binned and get_chi_nu_binned (dust_type_4elem.f90:722-750, 793-818).
    int nj_stride = 1;
    for (int d = 0; d < pr->n_dust; d++) nj_stride = std::max(nj_stride, pr->dust[d].n_jnu);
    if (!ray_groups.empty() && pr->config.monochromatic) {
        // use_exact_nu: get_spectrum_interp (source_type.f90:1098-1116), get_j_nu_interp / get_chi_nu_interp
        // (dust_type_4elem.f90:708-720, 780-791) at the group's own frequencies
        for (int g : ray_groups) {
            const int nn = h->h_peeled[g].n_nu;
            const double *nu = pr->config.frequencies + (pdesc[g].inu_min - 1);
            std::vector<double> spec((size_t)pr->n_sources * nn, 0.0), em((size_t)pr->n_dust * nj_stride * nn, 0.0), chi((size_t)pr->n_dust * nn, 0.0);
            for (int is = 0; is < pr->n_sources; is++) {
                const hyp_source_desc &src = pr->sources[is];
                for (int i = 0; i < nn; i++)
                    spec[(size_t)is * nn + i] = src.spectrum_type == 1 ? interp_log_pdf(src.spec_nu, src.spec_fnu, 1, src.n_spec, nu[i])
                                              : src.spectrum_type == 2 ? normalized_B_nu(nu[i], src.temperature) : 0.0;     // lte: the packets carry the dust emissivity
            }
            for (int d = 0; d < pr->n_dust; d++) {
                const hyp_dust_desc &in = pr->dust[d];
                for (int j = 0; j < in.n_jnu; j++)
                    for (int i = 0; i < nn; i++)
                        em[((size_t)d * nj_stride + j) * nn + i] = std::log10(interp_log_pdf(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu, nu[i]));
                for (int i = 0; i < nn; i++) {
                    double c = 0.0;
                    if (nu[i] >= in.nu[0] && nu[i] <= in.nu[in.n_nu - 1]) {
                        int j = in.n_nu - 2;
                        if (nu[i] != in.nu[in.n_nu - 1]) { int jl = 0, ju = in.n_nu - 1; while (ju - jl > 1) { int jm = (ju + jl) >> 1; if (nu[i] >= in.nu[jm]) jl = jm; else ju = jm; } j = jl; }
                        const double y1 = in.chi[j], y2 = in.chi[j + 1];
                        if (y1 > 0.0 && y2 > 0.0) {
                            const double f = (std::log10(nu[i]) - std::log10(in.nu[j])) / (std::log10(in.nu[j + 1]) - std::log10(in.nu[j]));
                            c = std::pow(10.0, std::log10(y1) + f * (std::log10(y2) - std::log10(y1)));
                        } else c = y1 + (nu[i] - in.nu[j]) / (in.nu[j + 1] - in.nu[j]) * (y2 - y1);
                    }
                    chi[(size_t)d * nn + i] = c;
                }
            }
            poff[g].src_spec = B.put(spec); poff[g].dust_em = B.put(em); poff[g].dust_chi = B.put(chi);
        }
    } else if (!ray_groups.empty()) {
        const double l0 = std::log10(3.e9), l1 = std::log10(3.e16);
        const int nb = (int)std::ceil((l1 - l0) * 100000);
        std::vector<double> bnu, bfnu;
        std::vector<std::vector<double>> lo(n_groups), hi(n_groups), spec(n_groups), em(n_groups), chi(n_groups);
        for (int g : ray_groups) {
            const DPeeled &G = h->h_peeled[g];
            const int nn = G.n_nu;
            lo[g].resize(nn); hi[g].resize(nn);
            for (int i = 0; i < nn; i++) {
                lo[g][i] = std::pow(10.0, G.log10_nu_min + (G.log10_nu_max - G.log10_nu_min) * (double)i / (double)nn);
                hi[g][i] = std::pow(10.0, G.log10_nu_min + (G.log10_nu_max - G.log10_nu_min) * (double)(i + 1) / (double)nn);
            }
            spec[g].assign((size_t)pr->n_sources * nn, 0.0);
            em[g].assign((size_t)pr->n_dust * nj_stride * nn, 0.0);
            chi[g].assign((size_t)pr->n_dust * nn, 0.0);
        }
        for (int is = 0; is < pr->n_sources; is++) {
            const hyp_source_desc &src = pr->sources[is];
            const double *x, *y; int n;
            if (src.spectrum_type == 3) continue;       // lte: the packets carry the dust emissivity
            if (src.spectrum_type == 1) { x = src.spec_nu; y = src.spec_fnu; n = src.n_spec; }
            else {
                // blackbody on 100000 points per decade between 3e9 and 3e16 Hz, normalized_B_nu :1088-1096
                if (bnu.empty()) {
                    bnu.resize(nb); bfnu.resize(nb);
                    for (int k = 0; k < nb; k++) bnu[k] = std::pow(10.0, (double)k / (double)(nb - 1) * (l1 - l0) + l0);
                }
                const double a = 2.0 * HYP_H_CGS / HYP_C_CGS / HYP_C_CGS / HYP_STEF_BOLTZ * HYP_PI, b = HYP_H_CGS / HYP_K_CGS;
                const double T = src.temperature, T4 = T * T * T * T;
                for (int k = 0; k < nb; k++) bfnu[k] = a * bnu[k] * bnu[k] * bnu[k] / (std::exp(b * bnu[k] / T) - 1.0) / T4;
                x = bnu.data(); y = bfnu.data(); n = nb;
            }
            const double tot = integral_loglog_all(x, y, 1, n);
            for (int g : ray_groups) {
                const int nn = h->h_peeled[g].n_nu;
                for (int i = 0; i < nn; i++) spec[g][(size_t)is * nn + i] = integral_loglog_range(x, y, 1, n, lo[g][i], hi[g][i]) / tot;
            }
        }
        for (int d = 0; d < pr->n_dust; d++) {
            const hyp_dust_desc &in = pr->dust[d];
            for (int j = 0; j < in.n_jnu; j++) {
                const double tot = integral_loglog_all(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu);
                for (int g : ray_groups) {
                    const int nn = h->h_peeled[g].n_nu;
                    for (int i = 0; i < nn; i++)
                        em[g][((size_t)d * nj_stride + j) * nn + i] =
                            std::log10(integral_loglog_range(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu, lo[g][i], hi[g][i]) / tot);
                }
            }
            for (int g : ray_groups) {
                const int nn = h->h_peeled[g].n_nu;
                for (int i = 0; i < nn; i++)
                    chi[g][(size_t)d * nn + i] = integral_loglog_range(in.nu, in.chi, 1, in.n_nu, lo[g][i], hi[g][i]) / (hi[g][i] - lo[g][i]);
            }
        }
        for (int g : ray_groups) { poff[g].src_spec = B.put(spec[g]); poff[g].dust_em = B.put(em[g]); poff[g].dust_chi = B.put(chi[g]); }
    }

    // monochromatic mode: emission probability of every source and of every emissivity row at the run's frequencies
    size_t mono_src_off = 0;
    if (pr->config.monochromatic) {
        const int nf = pr->config.n_frequencies;
        const double *fr = pr->config.frequencies;
        h->frequencies.assign(fr, fr + nf);
        std::vector<double> sp((size_t)pr->n_sources * nf);
        for (int is = 0; is < pr->n_sources; is++) {
            const hyp_source_desc &src = pr->sources[is];
            for (int i = 0; i < nf; i++)
                sp[(size_t)is * nf + i] = src.spectrum_type == 1 ? interp_log_pdf(src.spec_nu, src.spec_fnu, 1, src.n_spec, fr[i])
                                        : src.spectrum_type == 2 ? normalized_B_nu(fr[i], src.temperature) : 0.0;
        }
        mono_src_off = B.put(sp);
        for (int d = 0; d < pr->n_dust; d++) {
            const hyp_dust_desc &in = pr->dust[d];
            std::vector<double> lp((size_t)in.n_jnu * nf);
            for (int j = 0; j < in.n_jnu; j++)
                for (int i = 0; i < nf; i++) {
                    const double pv = interp_log_pdf(in.emiss_nu, in.emiss_jnu + j, in.n_jnu, in.n_enu, fr[i]);
                    lp[(size_t)j * nf + i] = pv == 0.0 ? -INFINITY : std::log10(pv);
                }
            doff[d].mono_prob = B.put(lp);
        }
    }

    // ---- device allocations ----
    HIPC(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPC(hipEventCreate(&h->ev0)); HIPC(hipEventCreate(&h->ev1));
    HIPC(hipEventCreate(&h->ev2)); HIPC(hipEventCreate(&h->ev3));
    HIPC(hipMalloc(&h->d_blob, sizeof(double) * B.h.size()));
    HIPC(hipMemcpy(h->d_blob, B.h.data(), sizeof(double) * B.h.size(), hipMemcpyHostToDevice));
    const double *db = h->d_blob;
    for (int a = 0; a < 3 && is_car; a++) { P.w[a] = db + w_off[a]; P.ew[a] = db + ew_off[a]; }
    if (pr->config.monochromatic) {
        P.n_frequencies = pr->config.n_frequencies; P.mono_threshold = pr->config.monochromatic_energy_threshold;
        P.mono_src_prob = db + mono_src_off; P.mono_which = 0; P.mono_inu = 0;
    }
    if (is_polar) {
        P.wr2 = db + polar_off[0]; P.wtanp = db + polar_off[4];
        if (is_sph) { P.wtant = db + polar_off[1]; P.wtant2 = db + polar_off[2]; P.wcost = db + polar_off[3]; }
        P.midplane = midplane; P.n_dim = n[2] == 1 ? 2 : 3;
    }
    if (is_vor) {
        const size_t nc = h->n_cells, nn = (size_t)pr->grid.vor_idx[nc];
        std::vector<double> vol(nc);
        for (size_t i = 0; i < nc; i++) vol[i] = pr->grid.vor_volume[i] < 0.0 ? 0.0 : pr->grid.vor_volume[i];
        HIPC(hipMalloc(&h->d_vor_sites, sizeof(double) * 3 * nc));
        HIPC(hipMemcpy(h->d_vor_sites, pr->grid.vor_sites, sizeof(double) * 3 * nc, hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_vor_volume, sizeof(double) * nc));
        HIPC(hipMemcpy(h->d_vor_volume, vol.data(), sizeof(double) * nc, hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_vor_idx, sizeof(int) * (nc + 1)));
        HIPC(hipMemcpy(h->d_vor_idx, pr->grid.vor_idx, sizeof(int) * (nc + 1), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_vor_neigh, sizeof(int) * (nn ? nn : 1)));
        HIPC(hipMemcpy(h->d_vor_neigh, pr->grid.vor_neighs, sizeof(int) * nn, hipMemcpyHostToDevice));
        {
            std::vector<VorWall> walls(nn ? nn : 1);
            for (size_t k = 0; k < nn; k++) {
                const int nb = pr->grid.vor_neighs[k];
                VorWall &w = walls[k];
                w.nb = nb; w.loc = 0; w.x = w.y = w.z = 0.0;
                if (nb >= 0) { w.x = pr->grid.vor_sites[3 * (size_t)nb]; w.y = pr->grid.vor_sites[3 * (size_t)nb + 1]; w.z = pr->grid.vor_sites[3 * (size_t)nb + 2]; }
            }
            HIPC(hipMalloc(&h->d_vor_walls, sizeof(VorWall) * walls.size()));
            HIPC(hipMemcpy(h->d_vor_walls, walls.data(), sizeof(VorWall) * walls.size(), hipMemcpyHostToDevice));
        }
        HIPC(hipMalloc(&h->d_vor_seed, sizeof(int) * vor_seed.size()));
        HIPC(hipMemcpy(h->d_vor_seed, vor_seed.data(), sizeof(int) * vor_seed.size(), hipMemcpyHostToDevice));
        if (pr->grid.vor_bb) {
            HIPC(hipMalloc(&h->d_vor_bb, sizeof(double) * 6 * nc));
            HIPC(hipMemcpy(h->d_vor_bb, pr->grid.vor_bb, sizeof(double) * 6 * nc, hipMemcpyHostToDevice));
        }
        h->h_vor_sites.assign(pr->grid.vor_sites, pr->grid.vor_sites + 3 * nc);
        h->h_vor_idx.assign(pr->grid.vor_idx, pr->grid.vor_idx + nc + 1);
        h->h_vor_neigh.assign(pr->grid.vor_neighs, pr->grid.vor_neighs + nn);
        P.vor_bb = h->d_vor_bb;
        P.vor_sites = h->d_vor_sites; P.vor_volume = h->d_vor_volume; P.vor_idx = h->d_vor_idx;
        P.vor_neigh = h->d_vor_neigh; P.vor_seed = h->d_vor_seed; P.vor_g = vor_g; P.vor_walls = h->d_vor_walls;
        for (int k = 0; k < 6; k++) P.vor_box[k] = pr->grid.vor_box[k];
    }
    if (is_amr) {
        HIPC(hipMalloc(&h->d_amr_grids, sizeof(AmrGrid) * amr_grids.size()));
        HIPC(hipMemcpy(h->d_amr_grids, amr_grids.data(), sizeof(AmrGrid) * amr_grids.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_amr_go, sizeof(int) * amr_go.size()));
        HIPC(hipMemcpy(h->d_amr_go, amr_go.data(), sizeof(int) * amr_go.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_amr_walls, sizeof(double) * amr_walls.size()));
        HIPC(hipMemcpy(h->d_amr_walls, amr_walls.data(), sizeof(double) * amr_walls.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_amr_cell_grid, sizeof(int) * amr_cell_grid.size()));
        HIPC(hipMemcpy(h->d_amr_cell_grid, amr_cell_grid.data(), sizeof(int) * amr_cell_grid.size(), hipMemcpyHostToDevice));
        P.amr_grids = h->d_amr_grids; P.amr_go = h->d_amr_go; P.amr_walls = h->d_amr_walls; P.amr_cell_grid = h->d_amr_cell_grid;
        P.amr_eps = amr_eps; P.n_amr_grids = (int)amr_grids.size(); P.n_amr_level1 = amr_level1;
        h->h_amr_grids = amr_grids; h->h_amr_go = amr_go;
    }
    if (is_oct) {
        HIPC(hipMalloc(&h->d_oct_cells, sizeof(OctCell) * oct_cells.size()));
        HIPC(hipMemcpy(h->d_oct_cells, oct_cells.data(), sizeof(OctCell) * oct_cells.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_oct_children, sizeof(int) * oct_children.size()));
        HIPC(hipMemcpy(h->d_oct_children, oct_children.data(), sizeof(int) * oct_children.size(), hipMemcpyHostToDevice));
        HIPC(hipMalloc(&h->d_oct_neigh, sizeof(int) * oct_neigh.size()));
        HIPC(hipMemcpy(h->d_oct_neigh, oct_neigh.data(), sizeof(int) * oct_neigh.size(), hipMemcpyHostToDevice));
        P.oct_cells = h->d_oct_cells; P.oct_children = h->d_oct_children; P.oct_neigh = h->d_oct_neigh;
        h->h_oct_cells = oct_cells; h->h_oct_children = oct_children; h->h_oct_neigh = oct_neigh;
    }
    for (int d = 0; d < pr->n_dust; d++) {
        DDust &D = P.dust[d]; const DustOffsets &O = doff[d];
        D.nu = db + O.nu; D.log10_nu = db + O.log10_nu; D.chi = db + O.chi; D.albedo = db + O.albedo;
        D.log10_chi = db + O.log10_chi; D.log10_albedo = db + O.log10_albedo; D.mu = db + O.mu;
        D.P1 = db + O.P1; D.P2 = db + O.P2; D.P3 = db + O.P3; D.P4 = db + O.P4;
        D.P1_cdf = db + O.P1_cdf; D.P2_cdf = db + O.P2_cdf;
        D.emiss_x = db + O.emiss_x; D.emiss_cdf = db + O.emiss_cdf; D.emiss_bp1 = db + O.emiss_bp1;
        D.emiss_coarse = db + O.emiss_coarse; D.n_ecoarse = (D.n_enu + HYP_COARSE - 1) / HYP_COARSE; D.pad2 = 0;
        D.jnu_var = db + O.jnu_var; D.log10_jnu_var = db + O.log10_jnu_var;
        D.mo_e = O.have_mo_e ? db + O.mo_e : nullptr;
        D.mo_chi_ross = O.have_mo_chi ? db + O.mo_chi_ross : nullptr;
        D.mo_kappa_planck = (O.have_mrw || O.have_pda) ? db + O.mo_kappa_planck : nullptr;
        D.mo_chi_inv_planck = O.have_mrw ? db + O.mo_chi_inv_planck : nullptr;
        D.mono_log10_prob = pr->config.monochromatic ? db + O.mono_prob : nullptr;
        D.bnu_cdf = O.have_mrw ? db + O.bnu_cdf : nullptr; D.bnu_bp1 = O.have_mrw ? db + O.bnu_bp1 : nullptr;
        D.bnu_coarse = O.have_mrw ? db + O.bnu_coarse : nullptr;
    }
    P.mrw = pr->config.mrw ? 1 : 0; P.pad5 = 0;
    P.n_inter_mrw_max = pr->config.n_inter_mrw_max; P.mrw_gamma = pr->config.mrw_gamma;
    P.mrw_x = P.mrw ? db + mrw_x_off : nullptr; P.mrw_y = P.mrw ? db + mrw_y_off : nullptr;
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have) { hs[i].spec_x = db + soff[i].x; hs[i].spec_cdf = db + soff[i].cdf; hs[i].spec_bp1 = db + soff[i].bp1; }
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have_points) { hs[i].points = db + soff[i].points; hs[i].point_cdf = db + soff[i].point_cdf; }
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have_map) hs[i].map_cdf = db + soff[i].map_cdf;
    for (int i = 0; i < pr->n_sources; i++)
        if (soff[i].have_spots) { hs[i].spot_tab = db + soff[i].spot_tab; hs[i].spot_blob = db; }
    HIPC(hipMalloc(&h->d_sources, sizeof(DSource) * hs.size()));
    HIPC(hipMemcpy(h->d_sources, hs.data(), sizeof(DSource) * hs.size(), hipMemcpyHostToDevice));
    P.sources = h->d_sources;

    const size_t ne = h->n_elem;
    HIPC(hipMalloc(&h->d_density, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_specific_energy, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_scratch, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_jnu_id, sizeof(int) * ne));
    HIPC(hipMalloc(&h->d_jnu_frac, sizeof(double) * ne));
    HIPC(hipMalloc(&h->d_energy_abs_tot, sizeof(double) * HYP_MAXD));
    HIPC(hipMalloc(&h->d_counter, sizeof(unsigned long long)));
    HIPC(hipMalloc(&h->d_err, sizeof(int)));
    HIPC(hipMalloc(&h->d_err_data, sizeof(double) * 4));
    HIPC(hipMemset(h->d_err, 0, sizeof(int)));
    // The accumulator block that the ranks all-reduce: [sums | tail | n_photons as doubles | spectrum sums]
    h->count_photons = pr->config.count_photons || pr->config.pda;
    h->pda = pr->config.pda != 0;
    h->n_bins = pr->config.n_spectrum_bins > 0 ? pr->config.n_spectrum_bins : 0;
    if (h->pda && !is_car) { /* grid_pda_disabled.f90: nothing to solve, but the counters are kept */ }
    h->ext_nphot = ne + TAIL_SIZE;
    h->ext_spec = h->ext_nphot + (h->count_photons ? h->n_cells : 0);
    h->block_doubles = h->ext_spec + (size_t)h->n_bins * ne;
    h->accum_stride = ((h->block_doubles + 31) / 32) * 32;
    h->accum_copies_alloc = h->n_bins ? 1 : 8;
    if (h->n_bins) h->accum_copies = 1;        // the spectrum planes are not replicated; their atomics dominate anyway
    if (h->count_photons) {
        HIPC(hipMalloc(&h->d_nphot, sizeof(unsigned int) * h->n_cells));
        HIPC(hipMalloc(&h->d_nphot_inexact, sizeof(int)));
        HIPC(hipMemset(h->d_nphot, 0, sizeof(unsigned int) * h->n_cells));
        HIPC(hipMemset(h->d_nphot_inexact, 0, sizeof(int)));
        P.n_photons = h->d_nphot; P.visit_tab = nullptr; P.nphot_inexact = h->d_nphot_inexact; P.count_photons = 1;
    }
    if (h->n_bins) {     // grid_physics_3d.f90:124-143,269-282,326-348
        const int nb = h->n_bins;
        if (!pr->config.spectrum_bin_edges) FAIL("specific_energy_spectrum_bin_edges should be present in the input when output_specific_energy_spectrum is enabled");
        h->spectrum_edges.assign(pr->config.spectrum_bin_edges, pr->config.spectrum_bin_edges + nb + 1);
        std::vector<double> le(nb + 1);
        for (int b = 0; b <= nb; b++) {
            if (b && !(h->spectrum_edges[b] > h->spectrum_edges[b - 1])) FAIL("specific_energy_spectrum_bin_edges should be strictly increasing");
            le[b] = std::log10(h->spectrum_edges[b]);
        }
        HIPC(hipMalloc(&h->d_log_edges, sizeof(double) * (nb + 1)));
        HIPC(hipMemcpy(h->d_log_edges, le.data(), sizeof(double) * (nb + 1), hipMemcpyHostToDevice));
        h->nj_max = 1;
        for (int d = 0; d < pr->n_dust; d++) h->nj_max = std::max(h->nj_max, pr->dust[d].n_jnu);
        // get_j_nu_bin_fractions (dust_type_4elem.f90:752-778): share of each emissivity row in each bin
        std::vector<double> frac((size_t)pr->n_dust * h->nj_max * nb, 0.0);
        for (int d = 0; d < pr->n_dust; d++) {
            const hyp_dust_desc &in = pr->dust[d];
            for (int iv = 0; iv < in.n_jnu; iv++) {
                double *f = frac.data() + ((size_t)d * h->nj_max + iv) * nb;
                for (int b = 0; b < nb; b++)
                    f[b] = integral_loglog_range(in.emiss_nu, in.emiss_jnu + iv, in.n_jnu, in.n_enu, h->spectrum_edges[b], h->spectrum_edges[b + 1]);
                const double norm = integral_loglog_all(in.emiss_nu, in.emiss_jnu + iv, in.n_jnu, in.n_enu);
                if (norm > 0.0) for (int b = 0; b < nb; b++) f[b] /= norm;
            }
        }
        HIPC(hipMalloc(&h->d_bin_frac, sizeof(double) * frac.size()));
        HIPC(hipMemcpy(h->d_bin_frac, frac.data(), sizeof(double) * frac.size(), hipMemcpyHostToDevice));
        // specific_energy_spectrum starts at the minimum specific energy unless an initial specific energy was given
        // (then it starts at 0): grid_physics_3d.f90:143,215-253
        std::vector<double> sp((size_t)nb * ne, 0.0);
        if (!pr->specific_energy || pr->config.specific_energy_type == 1)
            for (int b = 0; b < nb; b++) for (size_t ic = 0; ic < h->n_cells; ic++) for (int d = 0; d < pr->n_dust; d++)
                sp[((size_t)b * h->n_cells + ic) * pr->n_dust + d] = pr->dust[d].minimum_specific_energy;
        HIPC(hipMalloc(&h->d_spec, sizeof(double) * sp.size()));
        HIPC(hipMemcpy(h->d_spec, sp.data(), sizeof(double) * sp.size(), hipMemcpyHostToDevice));
        P.n_bins = nb; P.nj_max = h->nj_max; P.log_nu_edges = h->d_log_edges; P.jnu_bin_frac = h->d_bin_frac;
    }
    HIPC(hipMalloc(&h->d_accum, sizeof(double) * h->accum_stride * h->accum_copies_alloc));
    HIPC(hipMemset(h->d_accum, 0, sizeof(double) * h->accum_stride * h->accum_copies_alloc));

    if (img_total > 0) {
        h->img_accum_n = img_total + TAIL_SIZE;
        HIPC(hipMalloc(&h->d_img_accum, sizeof(double) * h->img_accum_n));
        HIPC(hipMemset(h->d_img_accum, 0, sizeof(double) * h->img_accum_n));
    }
    for (int g = 0; g < n_groups; g++) {
        DPeeled &G = h->h_peeled[g];
        G.view = db + poff[g].view;
        if (G.use_filters) { G.filt_off = db + poff[g].filt_off; G.filt_nu = db + poff[g].filt_nu; G.filt_tr = db + poff[g].filt_tr; }
        if (pr->config.raytracing) {
            G.src_spec = db + poff[g].src_spec; G.dust_log10_em = db + poff[g].dust_em; G.dust_chi = db + poff[g].dust_chi;
            G.nj_stride = nj_stride;
        }
        if (h->sed_n[g]) { G.sed = h->d_img_accum + h->sed_off[g]; G.sed2 = G.sed + h->sed_n[g]; }
        if (h->img_n[g]) { G.img = h->d_img_accum + h->img_off[g]; G.img2 = G.img + h->img_n[g]; }
    }
    {
        bool plain = !pr->config.mrw && !pr->config.monochromatic && !pr->binned;
        for (int i = 0; i < pr->n_sources; i++) plain = plain && pr->sources[i].type == 1 && (pr->sources[i].spectrum_type == 1 || pr->sources[i].spectrum_type == 2);
        // (filters are the peel kernel's / deposit_images' business; inside observers are the peel kernel's, not the inline plain kernel's)
        h->plain_imaging = plain && h->n_dust <= 4;      // five to eight species: the general kernel only (hyp_geom.hip)
        {
            bool md = pr->config.monochromatic && !pr->binned && h->n_dust <= 4;      // (the modified random walk is not made in monochromatic launches: iter_final_mono.f90 has none)
            for (int i = 0; i < pr->n_sources; i++) md = md && pr->sources[i].type == 1 && (pr->sources[i].spectrum_type == 1 || pr->sources[i].spectrum_type == 2);
            h->mono_defer = md;
        }
        {
            // sources with a surface (spheres, limb darkening and re-absorption included; no spots) next to points: the imaging iteration
            // on the deferred schedule with the GEN kernels (hyp_defer.h) instead of the general kernel with inline peel-off
            // (any sources: the GEN kernels emit with the general emitter; what stays on final_kernel is MRW, binned images, inside
            // observers together with such sources, and more than four species)
            bool gd = !plain && !pr->config.monochromatic && !pr->binned && h->n_dust <= 4 && pr->n_sources > 0;      // (with the modified random walk: the MRWF instance)
            for (int g = 0; g < pr->n_peeled; g++) gd = gd && !pr->peeled[g].inside_observer;
            h->gen_defer = gd;
            // ... and the same sources in a monochromatic run (the Pascucci / Pinte benchmark models: a stellar sphere)
            bool mg = pr->config.monochromatic && !h->mono_defer && !pr->binned && h->n_dust <= 4 && pr->n_sources > 0;
            for (int g = 0; g < pr->n_peeled; g++) mg = mg && !pr->peeled[g].inside_observer;
            h->mono_gen_defer = mg;
        }
        h->inside_observers = false;
        for (int g = 0; g < pr->n_peeled; g++) h->inside_observers = h->inside_observers || pr->peeled[g].inside_observer;
        {
            bool lean = !pr->config.mrw && !pr->config.monochromatic && !pr->binned;
            for (int g = 0; g < pr->n_peeled; g++) lean = lean && !pr->peeled[g].inside_observer;
            (void)lean;
            h->lean_imaging = false;      // round 4: the lean kernel's problems image on the deferred schedule (GEN kernels); with gen_defer = 0 they run on the general kernel
        }
        bool simple = pr->n_sources > 0, ext = pr->n_sources > 0;
        for (int i = 0; i < pr->n_sources; i++) {
            const bool spec = pr->sources[i].spectrum_type == 1 || pr->sources[i].spectrum_type == 2;
            simple = simple && pr->sources[i].type == 1 && spec;
            ext = ext && (pr->sources[i].type == 1 || pr->sources[i].type == 5 || pr->sources[i].type == 6) && spec;
        }
        h->simple_sources = simple; h->ext_sources = ext;
    }
    P.n_views_total = views_total;
    P.binned = pr->binned ? pr->n_peeled : -1; P.n_bin_theta = pr->n_binned_theta; P.n_bin_phi = pr->n_binned_phi;
    if (n_groups > 0) {
        HIPC(hipMalloc(&h->d_peeled, sizeof(DPeeled) * n_groups));
        HIPC(hipMemcpy(h->d_peeled, h->h_peeled.data(), sizeof(DPeeled) * n_groups, hipMemcpyHostToDevice));
    }
    P.peeled = h->d_peeled;

    // density / specific energy: reference layout -> cell-major device layout
    {
        std::vector<double> dens(pr->density, pr->density + ne);
        if (is_oct)   // density is reset to zero in masked (refined) cells: grid_physics_3d.f90:152-160
            for (int d = 0; d < h->n_dust; d++)
                for (size_t ic = 0; ic < h->n_cells; ic++)
                    if (oct_cells[ic].refined) dens[(size_t)d * h->n_cells + ic] = 0.0;
        if (is_amr)   // mask = cells not covered by a finer grid: grid_geometry_amr.f90:489-496
            for (size_t ic = 0; ic < h->n_cells; ic++) {
                const AmrGrid &g = amr_grids[amr_cell_grid[ic]];
                const size_t l = ic - g.start;
                const int i1 = (int)(l % g.n[0]), i2 = (int)((l / g.n[0]) % g.n[1]), i3 = (int)(l / ((size_t)g.n[0] * g.n[1]));
                if (amr_go[g.go_off + ((size_t)(i3 + 1) * (g.n[1] + 2) + (i2 + 1)) * (g.n[0] + 2) + (i1 + 1)] != 0)
                    for (int d = 0; d < h->n_dust; d++) dens[(size_t)d * h->n_cells + ic] = 0.0;
            }
        if (is_vor)   // mask = volume > 0: grid_geometry_voronoi.f90:161-173
            for (int d = 0; d < h->n_dust; d++)
                for (size_t ic = 0; ic < h->n_cells; ic++)
                    if (!(pr->grid.vor_volume[ic] > 0.0)) dens[(size_t)d * h->n_cells + ic] = 0.0;
        HIPC(hipMemcpy(h->d_scratch, dens.data(), sizeof(double) * ne, hipMemcpyHostToDevice));
    }
    to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_density, h->n_cells, h->n_dust);
    HIPC(hipStreamSynchronize(h->stream));
    // grid_physics_3d.f90:176-253
    std::vector<double> se(ne);
    if (pr->specific_energy) {
        if (pr->config.specific_energy_type == 1) {
            HIPC(hipMalloc(&h->d_additional, sizeof(double) * ne));
            HIPC(hipMemcpy(h->d_scratch, pr->specific_energy, sizeof(double) * ne, hipMemcpyHostToDevice));
            to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_additional, h->n_cells, h->n_dust);
            HIPC(hipStreamSynchronize(h->stream));
            for (int d = 0; d < h->n_dust; d++)
                for (size_t ic = 0; ic < h->n_cells; ic++) se[(size_t)d * h->n_cells + ic] = pr->dust[d].minimum_specific_energy;
        } else {
            std::memcpy(se.data(), pr->specific_energy, sizeof(double) * ne);
        }
    } else {
        if (pr->config.specific_energy_type == 1) FAIL("cannot specify specific_energy_type since specific_energy was not given");
        for (int d = 0; d < h->n_dust; d++)
            for (size_t ic = 0; ic < h->n_cells; ic++) se[(size_t)d * h->n_cells + ic] = pr->dust[d].minimum_specific_energy;
    }
    HIPC(hipMemcpy(h->d_scratch, se.data(), sizeof(double) * ne, hipMemcpyHostToDevice));
    to_cell_major_kernel<<<1024, 256, 0, h->stream>>>(h->d_scratch, h->d_specific_energy, h->n_cells, h->n_dust);
    HIPC(hipStreamSynchronize(h->stream));

    P.density = h->d_density;
    P.sum = h->d_accum;
    P.copy_stride = h->accum_stride;
    P.n_copies = 1;
    P.tail = h->d_accum + ne;
    P.sum_spec = h->n_bins ? h->d_accum + h->ext_spec : nullptr;
    P.jnu_id = h->d_jnu_id; P.jnu_frac = h->d_jnu_frac;
    P.specific_energy = h->d_specific_energy; P.energy_abs_tot = h->d_energy_abs_tot;
    P.energy_total = h->energy_total; P.peel_scattered_only = pr->config.raytracing ? 1 : 0;
    {   // geo%mask_map: cartesian_3d.f90:101, octree.f90:214-225, amr.f90:489-505, voronoi.f90:161-173
        std::vector<unsigned int> mask;
        mask.reserve(h->n_cells);
        for (size_t ic = 0; ic < h->n_cells; ic++) {
            bool valid = true;
            if (is_oct) valid = !oct_cells[ic].refined;
            else if (is_vor) valid = pr->grid.vor_volume[ic] > 0.0;
            else if (is_amr) {
                const AmrGrid &g = amr_grids[amr_cell_grid[ic]];
                const size_t l = ic - g.start;
                const int i1 = (int)(l % g.n[0]), i2 = (int)((l / g.n[0]) % g.n[1]), i3 = (int)(l / ((size_t)g.n[0] * g.n[1]));
                valid = amr_go[g.go_off + ((size_t)(i3 + 1) * (g.n[1] + 2) + (i2 + 1)) * (g.n[0] + 2) + (i1 + 1)] == 0;
            }
            if (valid) mask.push_back((unsigned int)ic);
        }
        P.n_masked = mask.size();
        if (pr->config.raytracing) {
            HIPC(hipMalloc(&h->d_mask_map, sizeof(unsigned int) * (mask.size() ? mask.size() : 1)));
            HIPC(hipMemcpy(h->d_mask_map, mask.data(), sizeof(unsigned int) * mask.size(), hipMemcpyHostToDevice));
            P.mask_map = h->d_mask_map;
        }
    }
    P.counter = h->d_counter; P.err = h->d_err; P.err_data = h->d_err_data;
    HIPC(hipMalloc(&h->d_problem, sizeof(DProblem)));
    HIPC(hipMemcpy(h->d_problem, &P, sizeof(DProblem), hipMemcpyHostToDevice));

    // check_energy_abs at set-up (grid_physics_3d.f90:277) + first jnu_var
    if (run_finish_kernel(h, 1, 1.0, nullptr)) { g_error = h->err; hyp_destroy(h); return 1; }
    HIPC(hipStreamSynchronize(h->stream));
#undef FAIL
#undef HIPC
    *out = h;
    return 0;
}

}  // extern "C"

// Clusters of Voronoi cells for the tiled schedule (hyp_vtile.h): recursive coordinate bisection of the sites into groups
// of equal cell count whose tables (VtInfo in hyp_device.h: sites of the cluster's cells and of the cells across its
// boundary, one FP32 record and one link word per wall, one header word per cell), densities and accumulators fit the LDS
// budget of one walk workgroup.
static size_t vt_blob16(size_t n_own, size_t n_site, size_t n_wall)
{
    return 3 * ((n_site + 1) / 2) + n_wall + (n_wall + 3) / 4 + 3 * ((n_own + 3) / 4) + 3 * ((n_site - n_own + 3) / 4);
}

int build_vor_clusters(hyp_handle h)
{
    const int nd = h->n_dust;
    if (h->vt_built_for == nd && h->d_vt_cluster) return 0;
    const size_t nc = h->n_cells;
    const double *S = h->h_vor_sites.data();
    const int *idx = h->h_vor_idx.data(), *nei = h->h_vor_neigh.data();
    if (h->h_vor_sites.size() != 3 * nc) return h->set_error("voronoi tables missing for the cluster builder");
    for (size_t i = 0; i < nc; i++) if (idx[i + 1] - idx[i] > 255) return h->set_error("a voronoi cell has more than 255 walls: no cluster-tiled schedule");
    const size_t budget = (size_t)h->vt_lds_kb * 1024;
    std::vector<int> perm(nc), cl_of(nc), cell_off;
    std::vector<int> local(nc, -1);          // index of a cell in the site table of the cluster being laid out (-1: not in it)
    struct Layout { std::vector<int> ghosts; size_t n_wall = 0; };
    std::vector<Layout> lay;
    int n_cl = 0;
    size_t max_lds = 0;
    // sites + per wall 20 bytes + header, densities, accumulators, and about as many ghost sites as own cells at these sizes
    double target = h->vt_cells > 0 ? (double)h->vt_cells : std::max(8.0, (double)budget / (24.0 * 2 + 20.0 * 16.5 + 4 + 16.0 * nd));
    for (int attempt = 0;; attempt++) {
        n_cl = (int)std::max<double>(1.0, std::ceil((double)nc / target));
        if (n_cl > HYP_TILE_MAX_BRICKS) return h->set_error("voronoi grid has too many cells for the cluster-tiled schedule");
        for (size_t i = 0; i < nc; i++) perm[i] = (int)i;
        cell_off.assign(n_cl + 1, 0);
        // iterative bisection: (first cell, number of cells, first cluster, number of clusters)
        struct Part { size_t lo, n; int c0, k; };
        std::vector<Part> stack{{0, nc, 0, n_cl}};
        while (!stack.empty()) {
            const Part p = stack.back(); stack.pop_back();
            if (p.k == 1) { cell_off[p.c0 + 1] = (int)p.n; for (size_t i = p.lo; i < p.lo + p.n; i++) cl_of[perm[i]] = p.c0; continue; }
            double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
            for (size_t i = p.lo; i < p.lo + p.n; i++)
                for (int a = 0; a < 3; a++) { const double x = S[3 * (size_t)perm[i] + a]; lo[a] = std::min(lo[a], x); hi[a] = std::max(hi[a], x); }
            int ax = 0;
            for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
            const int k1 = p.k / 2;
            const size_t n1 = (size_t)((double)p.n * k1 / p.k + 0.5);
            std::nth_element(perm.begin() + p.lo, perm.begin() + p.lo + n1, perm.begin() + p.lo + p.n,
                             [&](int a, int b) { const double xa = S[3 * (size_t)a + ax], xb = S[3 * (size_t)b + ax]; return xa < xb || (xa == xb && a < b); });
            stack.push_back({p.lo, n1, p.c0, k1});
            stack.push_back({p.lo + n1, p.n - n1, p.c0 + k1, p.k - k1});
        }
        for (int c = 0; c < n_cl; c++) cell_off[c + 1] += cell_off[c];
        // ghosts (cells of other clusters across a wall, each once, in the order met) and the LDS each cluster needs
        lay.assign(n_cl, Layout());
        std::vector<int> seen(nc, -1);
        for (size_t i = 0; i < nc; i++) {
            Layout &Lc = lay[cl_of[i]];
            Lc.n_wall += (size_t)(idx[i + 1] - idx[i]);
            for (int k = idx[i]; k < idx[i + 1]; k++) {
                const int nb = nei[k];
                if (nb >= 0 && cl_of[nb] != cl_of[i] && seen[nb] != cl_of[i]) { seen[nb] = cl_of[i]; Lc.ghosts.push_back(nb); }
            }
        }
        max_lds = 0;
        bool fits = true;
        for (int c = 0; c < n_cl; c++) {
            const size_t n_own = (size_t)(cell_off[c + 1] - cell_off[c]), n_site = n_own + lay[c].ghosts.size();
            const size_t lds = 16 * vt_blob16(n_own, n_site, lay[c].n_wall) + sizeof(double) * 2 * n_own * nd;
            max_lds = std::max(max_lds, lds);
            if (n_site > 65535 || lay[c].n_wall >= (1u << 20)) fits = false;
        }
        if (fits && max_lds <= budget) break;
        if (fits && h->vt_cells > 0 && max_lds <= (size_t)156 * 1024) break;       // a forced size may take a whole CU's LDS
        if (h->vt_cells > 0 || attempt > 60) return h->set_error("voronoi clusters do not fit in LDS");
        target *= std::min(0.95, 0.98 * (double)budget / (double)max_lds);
        if (target < 1.0) target = 1.0;
    }
    // members of each cluster in ascending cell order
    std::vector<int> members(nc), cursor(cell_off.begin(), cell_off.end() - 1), packed(nc);
    for (size_t i = 0; i < nc; i++) {
        const int c = cl_of[i], l = cursor[c]++ - cell_off[c];
        members[cell_off[c] + l] = (int)i;
        packed[i] = (c << 16) | l;
    }
    std::vector<VtInfo> info(n_cl);
    std::vector<VtGhost> ghosts;
    std::vector<int> adj((size_t)n_cl * VT_MAX_ADJ, -1);
    size_t total16 = 0;
    for (int c = 0; c < n_cl; c++) {
        VtInfo &I = info[c];
        I.n_own = cell_off[c + 1] - cell_off[c]; I.n_site = I.n_own + (int)lay[c].ghosts.size(); I.n_wall = (int)lay[c].n_wall;
        I.cell0 = cell_off[c]; I.ghost0 = (int)ghosts.size();
        if (total16 > 0x7fffffffull) return h->set_error("voronoi cluster tables too large");
        I.blob16 = (int)total16;
        total16 += vt_blob16((size_t)I.n_own, (size_t)I.n_site, (size_t)I.n_wall);
        int *ad = adj.data() + (size_t)c * VT_MAX_ADJ;
        for (int nb : lay[c].ghosts) {
            const int cn = cl_of[nb];
            int s = 0;
            while (s < VT_MAX_ADJ && ad[s] != cn && ad[s] != -1) s++;
            if (s < VT_MAX_ADJ) ad[s] = cn;
            ghosts.push_back(VtGhost{nb, s});
        }
    }
    std::vector<float4> blob(total16 ? total16 : 1, make_float4(0.f, 0.f, 0.f, 0.f));
    const double *B = h->hp.vor_box;
    for (int c = 0; c < n_cl; c++) {
        VtInfo &I = info[c];
        for (int j = 0; j < I.n_own; j++) local[members[I.cell0 + j]] = j;
        for (int gI = 0; gI < I.n_site - I.n_own; gI++) local[lay[c].ghosts[gI]] = I.n_own + gI;
        const int ns = (I.n_site + 1) & ~1;
        double *sx = (double *)(blob.data() + I.blob16), *sy = sx + ns, *sz = sy + ns;
        float4 *wrec = (float4 *)(sz + ns);
        uint32_t *wlink = (uint32_t *)(wrec + I.n_wall), *hdr = wlink + ((I.n_wall + 3) & ~3);
        const int ng = I.n_site - I.n_own, ngp = (ng + 3) & ~3;
        float *lmax = (float *)(hdr + ((I.n_own + 3) & ~3));
        int *mem = (int *)(lmax + ((I.n_own + 3) & ~3)), *gcell = mem + ((I.n_own + 3) & ~3), *gpacked = gcell + ngp, *gadj = gpacked + ngp;
        for (int j = 0; j < I.n_own; j++) mem[j] = members[I.cell0 + j];
        for (int gI = 0; gI < ng; gI++) {
            const VtGhost &gh = ghosts[(size_t)I.ghost0 + gI];
            gcell[gI] = gh.cell; gpacked[gI] = packed[gh.cell]; gadj[gI] = gh.adj;
        }
        double rmax = 0.0, len_sum = 0.0; size_t len_n = 0;
        for (int j = 0; j < I.n_site; j++) {
            const int cell = j < I.n_own ? members[I.cell0 + j] : lay[c].ghosts[j - I.n_own];
            sx[j] = S[3 * (size_t)cell]; sy[j] = S[3 * (size_t)cell + 1]; sz[j] = S[3 * (size_t)cell + 2];
            for (int a = 0; a < 3; a++) rmax = std::max(rmax, std::fabs(S[3 * (size_t)cell + a]));
        }
        for (int a = 0; a < 6; a++) rmax = std::max(rmax, std::fabs(B[a]));
        // first pass: the scale (a power of two that brings the mean |n| to order one)
        for (int j = 0; j < I.n_own; j++) {
            const int cell = members[I.cell0 + j];
            for (int k = idx[cell]; k < idx[cell + 1]; k++) {
                const int nb = nei[k];
                double n[3];
                if (nb >= 0) for (int a = 0; a < 3; a++) n[a] = S[3 * (size_t)nb + a] - S[3 * (size_t)cell + a];
                else { const int iw = -nb - 1, ax = iw >> 1; n[0] = n[1] = n[2] = 0.0; n[ax] = 2.0 * (B[iw] - S[3 * (size_t)cell + ax]); }
                const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                if (len > 0.0 && std::isfinite(len)) { len_sum += len; len_n++; }
            }
        }
        int e2 = 0;
        if (len_n) (void)std::frexp(len_sum / (double)len_n, &e2);
        const double scale = std::ldexp(1.0, -e2);
        I.scale = (float)scale;
        I.abs_eps = (float)(std::ldexp(1.0, -49) * rmax * scale * (1.0 + 1e-6));
        int kw = 0;
        for (int j = 0; j < I.n_own; j++) {
            const int cell = members[I.cell0 + j];
            const int k0 = kw;
            bool exact = idx[cell + 1] - idx[cell] > 32;        // (the filter keeps a 32-bit history of its comparisons)
            double cell_lmax = 0.0;
            for (int k = idx[cell]; k < idx[cell + 1]; k++, kw++) {
                const int nb = nei[k];
                double n[3];
                uint32_t link;
                if (nb >= 0) {
                    for (int a = 0; a < 3; a++) n[a] = S[3 * (size_t)nb + a] - S[3 * (size_t)cell + a];
                    int back = VT_NO_BACK;
                    for (int q = idx[nb]; q < idx[nb + 1]; q++) if (nei[q] == cell) { if (q - idx[nb] < VT_FIND_BACK) back = q - idx[nb]; break; }
                    link = (uint32_t)local[nb] | ((uint32_t)back << 16);
                    for (int q = idx[cell]; q < k; q++) if (nei[q] == nb) exact = true;      // a neighbour listed twice
                } else {
                    // a face of the box: the bisector plane with the site's mirror image in it (FP32 filter only)
                    const int iw = -nb - 1, ax = iw >> 1;
                    n[0] = n[1] = n[2] = 0.0; n[ax] = 2.0 * (B[iw] - S[3 * (size_t)cell + ax]);
                    link = 0xffffu | ((uint32_t)VT_NO_BACK << 16) | ((uint32_t)(iw + 1) << 24);
                }
                const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) * scale * (1.0 + 4.0 * 5.9604645e-8);
                // the filter's bounds assume FP32 values far from overflow and underflow: a wall 2^30 times shorter or longer than the
                // cluster's mean (coincident sites, absurd aspect ratios) sends its cell through the reference's loop
                if (!(len > 9.3e-10 && len < 1.07e9)) exact = true;
                const double hl2 = 0.5 * (n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) * scale * scale;
                wrec[kw] = make_float4((float)(n[0] * scale), (float)(n[1] * scale), (float)(n[2] * scale), (float)hl2);
                if (std::isfinite(len)) cell_lmax = std::max(cell_lmax, len);
                wlink[kw] = link;
            }
            hdr[j] = (uint32_t)k0 | ((uint32_t)(kw - k0) << 20) | (exact ? VT_HDR_EXACT : 0u);
            lmax[j] = std::nextafter((float)cell_lmax, INFINITY);
        }
        for (int j = 0; j < I.n_own; j++) local[members[I.cell0 + j]] = -1;
        for (int gI = 0; gI < I.n_site - I.n_own; gI++) local[lay[c].ghosts[gI]] = -1;
    }
    // unused adjacency slots point at the cluster itself (the walk adds a zero count there)
    for (int c = 0; c < n_cl; c++) for (int s = 0; s < VT_MAX_ADJ; s++) if (adj[(size_t)c * VT_MAX_ADJ + s] < 0) adj[(size_t)c * VT_MAX_ADJ + s] = c;
    if (ghosts.empty()) ghosts.push_back(VtGhost{0, VT_MAX_ADJ});
    free_dev(h->d_vt_cluster); free_dev(h->d_vt_info); free_dev(h->d_vt_blob); free_dev(h->d_vt_members); free_dev(h->d_vt_adj); free_dev(h->d_vt_ghost);
    auto up = [&](auto *&dst, const auto &v) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (hipMalloc((void **)&dst, sizeof(T) * v.size()) != hipSuccess) return 1;
        return hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    };
    if (up(h->d_vt_cluster, packed) || up(h->d_vt_info, info) || up(h->d_vt_blob, blob) || up(h->d_vt_members, members) ||
        up(h->d_vt_adj, adj) || up(h->d_vt_ghost, ghosts))
        return h->set_error("cannot allocate the cluster tables of the tiled Voronoi schedule");
    DProblem &P = h->hp;
    P.vt_cluster = h->d_vt_cluster; P.vt_info = h->d_vt_info; P.vt_blob = h->d_vt_blob; P.vt_members = h->d_vt_members;
    P.vt_adj = h->d_vt_adj; P.vt_ghost = h->d_vt_ghost;
    int max_cells = 0;
    for (int c = 0; c < n_cl; c++) max_cells = std::max(max_cells, info[c].n_own);
    h->vt_clusters = n_cl; h->vt_max_cells = max_cells; h->vt_max_lds = max_lds; h->vt_built_for = nd;
    return 0;
}

// Bricks of AMR grids for the tiled schedule (hyp_atile.h): every grid is cut into bricks of at most b0 x b1 x b2 cells, the
// shape of the Cartesian schedule for the number of species (16^3 for one), shrunk along z until densities + accumulators,
// walls and the brick's slice of the goto table (ghost layer included, 16 bits per entry) fit the LDS budget.
int build_amr_slabs(hyp_handle h)
{
    const int nd = h->n_dust;
    if (h->at_built_for == nd && h->d_at_slabs) return 0;
    const std::vector<AmrGrid> &G = h->h_amr_grids;
    const std::vector<int> &GO = h->h_amr_go;
    if (G.empty()) return h->set_error("amr tables missing for the brick builder");
    if (G.size() >= 32767) return h->set_error("too many amr grids for the 16-bit goto slices of the tiled schedule");
    const size_t budget = (size_t)h->at_lds_kb * 1024;
    int b[3] = {h->at_lds_kb > 100 ? 32 : 16, nd <= 2 ? 16 : 8, nd == 1 ? 16 : 8};      // one 1024-thread workgroup per CU (option at_lds_kb <= 100: 16-cell bricks, two fit a CU)
    if (h->at_cells > 0)         // option: smaller bricks (tests)
        while ((long long)b[0] * b[1] * b[2] > h->at_cells && (b[0] > 1 || b[1] > 1 || b[2] > 1)) {
            int a = b[2] >= b[1] && b[2] >= b[0] ? 2 : (b[1] >= b[0] ? 1 : 0);
            b[a] = (b[a] + 1) / 2;
        }
    auto lds_of = [&](const int n[3]) { return amr_slab_lds((size_t)n[0] * n[1] * n[2], (size_t)(n[0] + 2) * (n[1] + 2) * (n[2] + 2), (size_t)n[0] + n[1] + n[2] + 3, nd); };
    while (lds_of(b) > budget && (b[0] > 1 || b[1] > 1 || b[2] > 1)) {
        int a = b[2] >= b[1] && b[2] >= b[0] ? 2 : (b[1] >= b[0] ? 1 : 0);
        b[a]--;
    }
    if (lds_of(b) > budget) return h->set_error("the LDS budget of the tiled amr schedule is too small");
    std::vector<AtSlab> bricks;
    std::vector<short> go;
    std::vector<int> c0(G.size()), gnb(2 * G.size());
    int max_cells = 0, max_go = 0, max_walls = 0;
    for (size_t k = 0; k < G.size(); k++) {
        const AmrGrid &g = G[k];
        const int nb[3] = {(g.n[0] + b[0] - 1) / b[0], (g.n[1] + b[1] - 1) / b[1], (g.n[2] + b[2] - 1) / b[2]};
        c0[k] = (int)bricks.size(); gnb[2 * k] = nb[0]; gnb[2 * k + 1] = nb[1];
        for (int kz = 0; kz < nb[2]; kz++) for (int ky = 0; ky < nb[1]; ky++) for (int kx = 0; kx < nb[0]; kx++) {
            AtSlab s; std::memset(&s, 0, sizeof s);
            s.grid = (int)k;
            s.o[0] = kx * b[0]; s.o[1] = ky * b[1]; s.o[2] = kz * b[2];
            for (int a = 0; a < 3; a++) s.n[a] = std::min(b[a], g.n[a] - s.o[a]);
            s.go_off = (int)go.size();
            // goto entries of the brick's cells and one layer around them: 1-based positions o .. o + n + 1 of the grid's table
            for (int z = 0; z < s.n[2] + 2; z++) for (int y = 0; y < s.n[1] + 2; y++) for (int x = 0; x < s.n[0] + 2; x++)
                go.push_back((short)GO[(size_t)g.go_off + ((size_t)(s.o[2] + z) * (g.n[1] + 2) + (s.o[1] + y)) * (g.n[0] + 2) + (s.o[0] + x)]);
            bricks.push_back(s);
            max_cells = std::max(max_cells, s.n[0] * s.n[1] * s.n[2]);
            max_go = std::max(max_go, (s.n[0] + 2) * (s.n[1] + 2) * (s.n[2] + 2));
            max_walls = std::max(max_walls, s.n[0] + s.n[1] + s.n[2] + 3);
        }
    }
    if (bricks.size() > HYP_TILE_MAX_BRICKS) return h->set_error("amr grid has too many cells for the brick-tiled schedule");
    if (go.size() > 2000000000ull) return h->set_error("amr goto slices too large");
    free_dev(h->d_at_slabs); free_dev(h->d_at_go); free_dev(h->d_at_grid_c0); free_dev(h->d_at_grid_nz);
    auto up = [&](auto *&dst, const auto &v) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (hipMalloc((void **)&dst, sizeof(T) * std::max<size_t>(v.size(), 1)) != hipSuccess) return 1;
        return hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    };
    if (up(h->d_at_slabs, bricks) || up(h->d_at_go, go) || up(h->d_at_grid_c0, c0) || up(h->d_at_grid_nz, gnb))
        return h->set_error("cannot allocate the brick tables of the tiled AMR schedule");
    DProblem &P = h->hp;
    P.at_slabs = h->d_at_slabs; P.at_go = h->d_at_go; P.at_grid_c0 = h->d_at_grid_c0; P.at_grid_nb = h->d_at_grid_nz;
    for (int a = 0; a < 3; a++) P.at_b[a] = b[a];
    h->at_slabs_n = (int)bricks.size(); h->at_max_cells = max_cells; h->at_max_go = max_go; h->at_max_walls = max_walls; h->at_built_for = nd;
    return 0;
}

// Clusters of octree cells for the tiled schedule (hyp_otile.h).  Cells are numbered depth first
// (grid_geometry_octree.f90:206-246), so a subtree is a contiguous range of ids and so is a run of consecutive siblings.
// Top down: a subtree that fits the LDS budget is a unit; the children of one that does not are grouped, in order, into
// runs that fit; the cells above the units belong to no cluster (they are refined, a packet is never in one of them).
int build_oct_clusters(hyp_handle h)
{
    const int nd = h->n_dust;
    if (h->ot_built_for == nd && h->d_ot_cluster) return 0;
    const size_t nc = h->n_cells;
    const std::vector<OctCell> &C = h->h_oct_cells;
    const std::vector<int> &CH = h->h_oct_children, &NB = h->h_oct_neigh;
    if (C.size() != nc || NB.size() != 6 * nc) return h->set_error("octree tables missing for the cluster builder");
    // subtree sizes (cells, refined cells): children have larger ids than their parent
    std::vector<int> size(nc, 1), nref(nc, 0);
    for (size_t i = nc; i-- > 0;) {
        if (C[i].refined) nref[i] += 1;
        if (i > 0) { size[C[i].parent] += size[i]; nref[C[i].parent] += nref[i]; }
    }
    const size_t budget = (size_t)h->ot_lds_kb * 1024;
    const int cap = h->ot_cells > 0 ? h->ot_cells : 32767;
    auto fits = [&](long long n, long long k) { return n <= cap && n <= 32767 && oct_cluster_lds((size_t)n, (size_t)k, nd) <= budget; };
    std::vector<int> cl_of(nc, -1), c0v, ncv, kid_off{0};
    auto emit = [&](int first, int n, int k) {
        const int c = (int)c0v.size();
        c0v.push_back(first); ncv.push_back(n); kid_off.push_back(kid_off.back() + k);
        for (int i = first; i < first + n; i++) cl_of[i] = c;
    };
    std::vector<int> stack{0};
    if (fits(size[0], nref[0])) { emit(0, size[0], nref[0]); stack.clear(); }
    while (!stack.empty()) {
        const int p = stack.back(); stack.pop_back();      // a refined cell whose subtree does not fit
        int first = -1, n = 0, k = 0;
        std::vector<int> deeper;
        for (int s = 0; s < 8; s++) {
            const int c = CH[(size_t)p * 8 + s];
            if (!fits(size[c], nref[c])) {
                if (!C[c].refined) return h->set_error("octree cluster budget too small for a single cell");
                if (n) emit(first, n, k);
                n = 0; k = 0; deeper.push_back(c);
                continue;
            }
            if (n && !fits(n + size[c], k + nref[c])) { emit(first, n, k); n = 0; k = 0; }
            if (!n) first = c;
            n += size[c]; k += nref[c];
        }
        if (n) emit(first, n, k);
        for (size_t i = deeper.size(); i-- > 0;) stack.push_back(deeper[i]);
    }
    const int n_cl = (int)c0v.size();
    if (n_cl > HYP_TILE_MAX_BRICKS) return h->set_error("octree has too many cells for the cluster-tiled schedule");
    int max_cells = 0, max_kids = 0;
    for (int c = 0; c < n_cl; c++) { max_cells = std::max(max_cells, ncv[c]); max_kids = std::max(max_kids, kid_off[c + 1] - kid_off[c]); }
    // per-cluster images: records with the row of a refined cell's children in `parent`, children and neighbours as local indices
    std::vector<OctCell> rec(C);
    // `pad` of the cluster's copy: bit b set where the cell passes the second half of geo_advance's edge test on axis b,
    // h 1e-6 > 1e-14 (|c| + h) -- a property of the cell, formed here with the walk's own operations (otile_walk_kernel looks it up
    // instead of evaluating it at every step)
    for (size_t i = 0; i < rec.size(); i++) {
        const double cxyz[3] = {rec[i].x, rec[i].y, rec[i].z};
        unsigned char bits = 0;
        for (int b = 0; b < 3; b++) {
            const double hb = std::ldexp(h->hp.oct_half[b], -(int)rec[i].level);
            if (hb * 1e-6 > 1e-14 * (std::fabs(cxyz[b]) + hb)) bits |= (unsigned char)(1u << b);
        }
        rec[i].pad = bits;
    }
    std::vector<short> kid((size_t)std::max(1, kid_off[n_cl]) * 8, (short)-1), nb(6 * nc, (short)-2);
    for (int c = 0; c < n_cl; c++) {
        int row = 0;
        for (int i = c0v[c]; i < c0v[c] + ncv[c]; i++) {
            if (C[i].refined) {
                rec[i].parent = row;
                for (int s = 0; s < 8; s++) kid[((size_t)kid_off[c] + row) * 8 + s] = (short)(CH[(size_t)i * 8 + s] - c0v[c]);
                row++;
            }
            for (int f = 0; f < 6; f++) {
                const int n = NB[(size_t)i * 6 + f];
                nb[(size_t)i * 6 + f] = (size_t)n == nc ? (short)-1 : (cl_of[n] == c ? (short)(n - c0v[c]) : (short)-2);
            }
        }
    }
    free_dev(h->d_ot_cluster); free_dev(h->d_ot_c0); free_dev(h->d_ot_nc); free_dev(h->d_ot_kid_off); free_dev(h->d_ot_rec); free_dev(h->d_ot_kid); free_dev(h->d_ot_nb);
    auto up = [&](auto *&dst, const auto &v) {
        using T = typename std::remove_reference<decltype(v)>::type::value_type;
        if (hipMalloc((void **)&dst, sizeof(T) * v.size()) != hipSuccess) return 1;
        return hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess ? 1 : 0;
    };
    if (up(h->d_ot_cluster, cl_of) || up(h->d_ot_c0, c0v) || up(h->d_ot_nc, ncv) || up(h->d_ot_kid_off, kid_off) || up(h->d_ot_rec, rec) ||
        up(h->d_ot_kid, kid) || up(h->d_ot_nb, nb))
        return h->set_error("cannot allocate the cluster tables of the tiled octree schedule");
    DProblem &P = h->hp;
    P.ot_cluster = h->d_ot_cluster; P.ot_c0 = h->d_ot_c0; P.ot_nc = h->d_ot_nc; P.ot_kid_off = h->d_ot_kid_off;
    P.ot_rec = h->d_ot_rec; P.ot_kid = h->d_ot_kid; P.ot_nb = h->d_ot_nb;
    h->ot_clusters = n_cl; h->ot_max_cells = max_cells; h->ot_max_kids = max_kids; h->ot_built_for = nd;
    return 0;
}

