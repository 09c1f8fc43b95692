/*
 * hyp_oracle.c -- TEST INFRASTRUCTURE ONLY (see hyp_oracle.h).
 *
 * Plain-C FP64 restatement of the reference's photon-packet path.  Every
 * function cites the reference file:line (relative to /root/reference) it
 * follows.  Random numbers: the reference uses fortranlib's generator (source
 * absent); here every packet owns two counter-based Philox4x32-10 streams keyed
 * by (seed, iteration) and indexed by the global packet id, so results do not
 * depend on thread count or on how packets are sharded:
 *   stream A: FP64 uniforms, consumed in the reference's order of `random`
 *             calls (emit -> random_exp -> interact ...);
 *   stream B: the per-crossing propagation check of grid_propagate_3d.f90:108
 *             (`random(xi); if(xi < frac_check)`), a Bernoulli(frac_check)
 *             trial per cell step, is realised through its gap distribution:
 *             the number of steps until the next check is drawn from the
 *             geometric law (1-p)^k p with one uniform of stream B per check
 *             (identical in distribution, one draw per ~1/p steps).
 */
#include "hyp_oracle.h"

#include <math.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846
#define TWOPI 6.28318530717958647692
/* cgs constants, hyperion/util/constants.py:1-20 (same values as fortranlib lib_constants) */
#define H_CGS 6.6260755e-27
#define K_CGS 1.380658e-16

static char g_error[512];

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al. 2011)                                   */
/* ------------------------------------------------------------------ */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    uint32_t key[2];
    uint64_t id;
    uint32_t blk_a, blk_b;
    uint32_t stream_b;      /* 1: the packet's own propagation checks; 2: those of a peel-off walk (own block range) */
    int have_a;
    int32_t countdown;      /* cell steps left until the next propagation check */
    double buf_a;
} rng_t;

static uint32_t seed_key(int64_t seed)
{
    uint64_t s = (uint64_t)(seed < 0 ? -seed : seed);
    return (uint32_t)s ^ (uint32_t)(s >> 32);
}

static void rng_init(rng_t *g, int64_t seed, uint32_t iter_tag, uint64_t id)
{
    g->key[0] = seed_key(seed); g->key[1] = iter_tag;
    g->id = id; g->blk_a = 0; g->blk_b = 0; g->stream_b = 1u; g->have_a = 0; g->countdown = 0;
}

static inline double u64_to_unit(uint32_t hi, uint32_t lo)
{
    uint64_t u = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)u * (1.0 / 9007199254740992.0);
}

/* uniform in [0,1): the reference's `random(xi)` */
static double rng_uniform(rng_t *g)
{
    if (g->have_a) { g->have_a = 0; return g->buf_a; }
    uint32_t ctr[4] = {(uint32_t)g->id, (uint32_t)(g->id >> 32), g->blk_a++, 0u}, o[4];
    orc_philox4x32_10(ctr, g->key, o);
    g->buf_a = u64_to_unit(o[2], o[3]); g->have_a = 1;
    return u64_to_unit(o[0], o[1]);
}

/* number of cell steps before the next propagation check: geometric gap of a
 * Bernoulli(p) process, gap = floor(log(1-u) / log(1-p)), clamped to int32 */
static int32_t rng_check_gap(rng_t *g, double p, double log1mp)
{
    if (p >= 1.0) return 0;
    if (!(p > 0.0)) return INT32_MAX;
    uint32_t ctr[4] = {(uint32_t)g->id, (uint32_t)(g->id >> 32), g->blk_b++, g->stream_b}, o[4];
    orc_philox4x32_10(ctr, g->key, o);
    double gap = floor(log(1.0 - u64_to_unit(o[0], o[1])) / log1mp);
    return gap >= 2147483647.0 ? INT32_MAX : (int32_t)gap;
}

/* fortranlib random_exp: tau = -log(1 - xi) */
static double rng_exp(rng_t *g) { return -log(1.0 - rng_uniform(g)); }

/* ------------------------------------------------------------------ */
/* fortranlib lib_array restated                                       */
/* ------------------------------------------------------------------ */

/* locate (0-based): j with x[j] <= xv < x[j+1]; xv==x[n-1] -> n-2; -1 if
 * outside.  Ascending arrays only (all call sites on this path). */
static int locate(const double *x, int n, double xv)
{
    if (!(xv >= x[0]) || !(xv <= x[n - 1])) return -1;
    if (xv == x[n - 1]) return n - 2;
    int jl = 0, ju = n - 1;
    while (ju - jl > 1) {
        int jm = (ju + jl) >> 1;
        if (xv >= x[jm]) jl = jm; else ju = jm;
    }
    return jl;
}

/* interp1d_loglog: log-log where both ordinates are positive, linear
 * otherwise (same rule as hyperion/util/_interpolate_core.c:277-300). */
static double interp1d_loglog(const double *x, const double *y, int n, double xv)
{
    int j = locate(x, n, xv);
    if (j < 0) return NAN;
    double y1 = y[j], y2 = y[j + 1];
    if (y1 > 0.0 && y2 > 0.0) {
        double f = (log10(xv) - log10(x[j])) / (log10(x[j + 1]) - log10(x[j]));
        return pow(10.0, log10(y1) + f * (log10(y2) - log10(y1)));
    }
    return y1 + (xv - x[j]) / (x[j + 1] - x[j]) * (y2 - y1);
}

/* interp2d: bilinear, array a[iy][ix] with ix fastest (Fortran a(ix,iy)) */
static double interp2d(const double *x, int nx, const double *y, int ny,
                       const double *a, double xv, double yv)
{
    int i = locate(x, nx, xv), j = locate(y, ny, yv);
    if (i < 0 || j < 0) return NAN;
    double fx = (xv - x[i]) / (x[i + 1] - x[i]);
    double fy = (yv - y[j]) / (y[j + 1] - y[j]);
    double a00 = a[(size_t)j * nx + i], a10 = a[(size_t)j * nx + i + 1];
    double a01 = a[(size_t)(j + 1) * nx + i], a11 = a[(size_t)(j + 1) * nx + i + 1];
    return a00 * (1 - fx) * (1 - fy) + a10 * fx * (1 - fy) + a01 * (1 - fx) * fy + a11 * fx * fy;
}

/* one log-log trapezium segment (hyperion/util/_integrate_core.c:317-326) */
static double seg_loglog(double x1, double x2, double y1, double y2)
{
    if (!(y1 > 0.0 && y2 > 0.0)) return 0.0;
    double b = log10(y1 / y2) / log10(x1 / x2);
    if (fabs(b + 1.0) < 1e-10) return x1 * y1 * log(x2 / x1);
    return y1 * (x2 * pow(x2 / x1, b) - x1) / (b + 1.0);
}

/* integral_linlog: x linear, y logarithmic (_integrate_core.c:236-246) */
static double integral_linlog(const double *x, const double *y, int n)
{
    double s = 0.0;
    for (int i = 0; i < n - 1; i++) {
        double y1 = y[i], y2 = y[i + 1], dx = x[i + 1] - x[i];
        if (y1 == y2) s += y1 * dx;
        else if (y1 > 0.0 && y2 > 0.0) s += (y2 - y1) * dx / log(y2 / y1);
    }
    return s;
}

/* integral_loglog(x, y[, xmin, xmax]) (fortranlib; the reference's own equivalent is
 * hyperion/util/integrate.py integrate_loglog_subset): piecewise power laws between the
 * tabulated points, end points interpolated in log-log, limits clipped to the table. */
static double interp_seg_loglog(double x1, double x2, double y1, double y2, double x)
{
    if (y1 > 0.0 && y2 > 0.0) return y1 * pow(x / x1, log10(y2 / y1) / log10(x2 / x1));
    return y1 + (x - x1) / (x2 - x1) * (y2 - y1);
}

static double integral_loglog_range(const double *x, const double *y, int n, double xmin, double xmax)
{
    if (xmin < x[0]) xmin = x[0];
    if (xmax > x[n - 1]) xmax = x[n - 1];
    if (!(xmax > xmin)) return 0.0;
    double s = 0.0;
    for (int i = 0; i < n - 1; i++) {
        double a = x[i], b = x[i + 1];
        if (b <= xmin || a >= xmax) continue;
        double xa = a < xmin ? xmin : a, xb = b > xmax ? xmax : b;
        double ya = xa == a ? y[i] : interp_seg_loglog(a, b, y[i], y[i + 1], xa);
        double yb = xb == b ? y[i + 1] : interp_seg_loglog(a, b, y[i], y[i + 1], xb);
        s += seg_loglog(xa, xb, ya, yb);
    }
    return s;
}

static double integral_loglog_all(const double *x, const double *y, int n)
{
    double s = 0.0;
    for (int i = 0; i < n - 1; i++) s += seg_loglog(x[i], x[i + 1], y[i], y[i + 1]);
    return s;
}

/* ------------------------------------------------------------------ */
/* fortranlib type_pdf restated                                        */
/* ------------------------------------------------------------------ */

typedef struct {
    int n;
    double *x, *pdf, *cdf, *bp1; /* bp1[j] = power-law index + 1 of bin j */
} pdf_t;

/* set_pdf(p, x, y, log=.true.) */
static int pdf_set_log(pdf_t *p, const double *x, const double *y, int n, int stride)
{
    p->n = n;
    p->x = malloc(sizeof(double) * n); p->pdf = malloc(sizeof(double) * n);
    p->cdf = malloc(sizeof(double) * n); p->bp1 = malloc(sizeof(double) * n);
    double norm = 0.0;
    for (int i = 0; i < n; i++) { p->x[i] = x[i]; p->pdf[i] = y[(size_t)i * stride]; }
    for (int i = 0; i < n - 1; i++) norm += seg_loglog(p->x[i], p->x[i + 1], p->pdf[i], p->pdf[i + 1]);
    if (!(norm > 0.0)) return -1;
    for (int i = 0; i < n; i++) p->pdf[i] /= norm;
    p->cdf[0] = 0.0;
    for (int i = 1; i < n; i++)
        p->cdf[i] = p->cdf[i - 1] + seg_loglog(p->x[i - 1], p->x[i], p->pdf[i - 1], p->pdf[i]);
    double last = p->cdf[n - 1];
    for (int i = 0; i < n; i++) p->cdf[i] /= last;
    for (int i = 0; i < n - 1; i++) {
        if (p->pdf[i] > 0.0 && p->pdf[i + 1] > 0.0)
            p->bp1[i] = log10(p->pdf[i + 1] / p->pdf[i]) / log10(p->x[i + 1] / p->x[i]) + 1.0;
        else p->bp1[i] = NAN;
    }
    p->bp1[n - 1] = NAN;
    return 0;
}

static void pdf_free(pdf_t *p) { free(p->x); free(p->pdf); free(p->cdf); free(p->bp1); }

/* sample_pdf(p, xi) for a log pdf: invert the piecewise power-law CDF */
static double pdf_sample_log(const pdf_t *p, double xi)
{
    int j = locate(p->cdf, p->n, xi);
    if (j < 0) j = 0;
    double c1 = p->cdf[j], c2 = p->cdf[j + 1];
    double f = (c2 > c1) ? (xi - c1) / (c2 - c1) : 0.0;
    double x1 = p->x[j], x2 = p->x[j + 1], bp1 = p->bp1[j];
    if (bp1 != bp1) return x1 + f * (x2 - x1);
    if (fabs(bp1) < 1e-10) return x1 * pow(x2 / x1, f);
    return x1 * pow(1.0 + f * (pow(x2 / x1, bp1) - 1.0), 1.0 / bp1);
}

/* discrete pdf (fortranlib pdf_discrete): smallest i with xi < cdf[i] */
static int sample_discrete(const double *cdf, int n, double xi)
{
    for (int i = 0; i < n - 1; i++) if (xi < cdf[i]) return i;
    return n - 1;
}

/* ------------------------------------------------------------------ */
/* Angles (fortranlib type_angle3d restated)                            */
/* ------------------------------------------------------------------ */

typedef struct { double cost, sint, cosp, sinp; } angle_t;

static inline void angle_to_vector(const angle_t *a, double v[3])
{
    v[0] = a->sint * a->cosp; v[1] = a->sint * a->sinp; v[2] = a->cost;
}

/* random_sphere_angle3d: mu uniform in [-1,1], phi uniform in [0,2pi) */
static void random_sphere_angle(rng_t *g, angle_t *a)
{
    double mu = 2.0 * rng_uniform(g) - 1.0;
    double phi = TWOPI * rng_uniform(g);
    a->cost = mu; a->sint = sqrt(1.0 - mu * mu);
    a->cosp = cos(phi); a->sinp = sin(phi);
}

/* rotate_angle3d(a_local, a_coord, a_final): add the local (scattering)
 * angle to the direction a_coord.  Spherical triangle pole / old / new with
 * sides a=old theta, b=local theta, c=new theta and angle C = local phi at
 * the old direction, B = |new phi - old phi| at the pole.  Orientation: local
 * phi in (0,pi) -> new phi = old phi + B.  fortranlib's source is absent; the
 * orientation is pinned by the sign of Stokes U in the reference's golden
 * test_peeloff outputs (tests/test_oracle_golden.py): the mirror choice gives
 * the opposite sign of U at the same Q. */
static void rotate_angle(const angle_t *loc, const angle_t *co, angle_t *fin)
{
    double cos_a = co->cost, sin_a = co->sint;
    double cos_b = loc->cost, sin_b = loc->sint;
    double cos_C = loc->cosp, sin_C = fabs(loc->sinp);
    double cos_c = cos_a * cos_b + sin_a * sin_b * cos_C;
    if (cos_c > 1.0) cos_c = 1.0;
    if (cos_c < -1.0) cos_c = -1.0;
    double sin_c = sqrt(1.0 - cos_c * cos_c);
    double cos_B, sin_B;
    if (fabs(sin_a) < 1e-12 || sin_c < 1e-12) {
        /* old or new direction along the pole: azimuth difference is the
         * local azimuth itself (old at pole) or arbitrary (new at pole) */
        if (fabs(sin_a) < 1e-12) { cos_B = (cos_a > 0 ? -cos_C : cos_C); sin_B = sin_C; }
        else { cos_B = 1.0; sin_B = 0.0; }
    } else {
        cos_B = (cos_b - cos_a * cos_c) / (sin_a * sin_c);
        if (cos_B > 1.0) cos_B = 1.0;
        if (cos_B < -1.0) cos_B = -1.0;
        sin_B = sqrt(1.0 - cos_B * cos_B);
    }
    fin->cost = cos_c; fin->sint = sin_c;
    if (loc->sinp < 0.0) { /* new phi = old phi - B */
        fin->cosp = co->cosp * cos_B + co->sinp * sin_B;
        fin->sinp = co->sinp * cos_B - co->cosp * sin_B;
    } else {               /* new phi = old phi + B */
        fin->cosp = co->cosp * cos_B - co->sinp * sin_B;
        fin->sinp = co->sinp * cos_B + co->cosp * sin_B;
    }
}

/* difference_angle3d(a_coord, a_final, a_local): inverse of rotate_angle */
static void difference_angle(const angle_t *co, const angle_t *fin, angle_t *loc)
{
    double cos_a = co->cost, sin_a = co->sint;
    double cos_c = fin->cost, sin_c = fin->sint;
    double cos_B = co->cosp * fin->cosp + co->sinp * fin->sinp;   /* cos(new-old) */
    double sin_Bs = co->cosp * fin->sinp - co->sinp * fin->cosp;  /* sin(new-old) */
    double cos_b = cos_a * cos_c + sin_a * sin_c * cos_B;
    if (cos_b > 1.0) cos_b = 1.0;
    if (cos_b < -1.0) cos_b = -1.0;
    double sin_b = sqrt(1.0 - cos_b * cos_b);
    loc->cost = cos_b; loc->sint = sin_b;
    if (sin_b < 1e-12 || sin_a < 1e-12) {
        if (sin_a < 1e-12 && sin_b >= 1e-12) {
            loc->cosp = (cos_a > 0 ? -cos_B : cos_B); loc->sinp = sin_Bs;
        } else { loc->cosp = 1.0; loc->sinp = 0.0; }
        return;
    }
    double cos_C = (cos_c - cos_a * cos_b) / (sin_a * sin_b);
    if (cos_C > 1.0) cos_C = 1.0;
    if (cos_C < -1.0) cos_C = -1.0;
    double sin_C = sqrt(1.0 - cos_C * cos_C);
    loc->cosp = cos_C;
    loc->sinp = (sin_Bs >= 0.0) ? sin_C : -sin_C;
}

/* ------------------------------------------------------------------ */
/* State                                                                */
/* ------------------------------------------------------------------ */

typedef struct {
    int n_nu, n_mu, n_jnu, n_enu, n_e;
    int sublimation_mode, version, zero_p2;
    double sublimation_specific_energy, minimum_specific_energy;
    double *nu, *albedo, *chi;
    double *mu, mu_min, mu_max;
    double *P1, *P2, *P3, *P4;         /* normalised, [n_nu][n_mu] */
    double *P1_cdf, *P2_cdf;            /* [n_nu][n_mu] */
    double *j_nu_var, *log10_j_nu_var;  /* [n_jnu] */
    pdf_t *j_nu;                        /* [n_jnu] */
    double e_min, e_max;                /* mean_opacities specific_energy range */
    int have_e_range;
    double *mo_e, *mo_chi_ross;
    double *mo_kappa_planck, *mo_chi_inv_planck;   /* MRW: dust.f90:88-100 */
    pdf_t *b_nu;                        /* [n_jnu] emissivity / kappa_nu: dust_type_4elem.f90:289-291 */
} dust_t;

typedef struct {
    int type, spectrum_type, peeloff, limb_darkening, n_points;
    double luminosity, temperature, position[3], radius, box[6], face_cdf[6];
    angle_t direction;          /* plane_parallel: angle3d_deg(theta, phi) */
    double *points, *point_cdf; /* point_collection: [n][3] positions, luminosity cdf */
    double *map_cdf;            /* map: [n_cells] cumulative of the luminosity map */
    /* spotted sphere (the reference's type 3): spot_cdf over [spots..., sphere] (source_type.f90:159-188) */
    int n_spots; double *spot_cdf; angle_t *spot_a; double *spot_cost; int *spot_stype; double *spot_T; pdf_t *spot_spectrum;
    pdf_t spectrum;
} source_t;

typedef struct {
    orc_peeled_desc d;
    double *theta, *phi;
    angle_t *view;          /* [n_view] */
    int n_orig, n_stokes;
    int *filt_off;          /* filters: offsets into d.filt_nu / d.filt_tr, [n_nu + 1] */
    double *filt_nu, *filt_tr;
    double log10_nu_min, log10_nu_max, log10_ap_min, log10_ap_max;
    double *sed, *sed2, *img, *img2;
    size_t sed_size, img_size;
    /* raytracing caches (images_peeled.f90:57-82): spectra binned on the image's frequency grid */
    double *src_spec;       /* [n_sources][n_nu]  get_source_spectrum */
    double *dust_log10_em;  /* [n_dust][n_jnu][n_nu]  log10 of get_j_nu_binned */
    double *dust_chi;       /* [n_dust][n_nu]  get_chi_nu_binned */
    int nj_stride;
} peeled_t;

enum { GRID_CAR = 1, GRID_OCT = 2, GRID_VOR = 3, GRID_AMR = 4, GRID_SPH = 5, GRID_CYL = 6 };
#define GRID_IS_3IDX(t) ((t) == GRID_CAR || (t) == GRID_SPH || (t) == GRID_CYL)

/* one grid of an AMR level: type_grid_amr.f90:12-21 */
typedef struct amr_grid {
    int level, n[3];
    double lo[3], hi[3], width[3], volume;
    double *w[3];               /* linspace(lo, hi, n+1) */
    size_t start;               /* unique id of the first cell */
    int32_t *go;                /* [(n3+2)][(n2+2)][(n1+2)] grid to continue in (global index + 1), 0 = stay */
} amr_grid;

struct orc_state {
    char err[512];
    int grid_type;
    int n1, n2, n3;
    size_t n_cells;
    /* octree: type_grid_octree.f90:14-22 (0-based ids; -1 = none) */
    double *ox, *oy, *oz, *odx, *ody, *odz;
    uint8_t *orefined; int8_t *osubcell;
    int32_t *oparent, *ochildren;   /* ochildren[8*id + k] */
    double oct_eps, obox[6];
    /* voronoi: type_grid_voronoi.f90 (0-based ids; walls -1..-6) */
    double *vsite;              /* [n][3] */
    int32_t *vidx, *vneigh;
    double vbox[6];
    double *vbb;                /* [n][6] bb_min, bb_max of the cells (table `cells`), or NULL */
    int vg;                     /* seed grid for the nearest-site search: vg^3 cells */
    int32_t *vseed;
    /* amr: grid_geometry_amr.f90 */
    int n_amr_grids, n_amr_levels;
    amr_grid *amr;
    int32_t *amr_cell_grid;     /* [n_cells] grid of each unique id */
    double amr_eps;
    double *w[3], *ew[3];
    int n[3];
    /* spherical / cylindrical polar grids: type_grid_spherical_3d.f90:11-20, type_grid_cylindrical_3d.f90 */
    double *wr2, *wtanp, *wtant, *wtant2, *wcost;
    int midplane;               /* 0-based index of the theta wall at pi/2, -2 if none (Fortran default -1) */
    int n_dim;
    double *volume;
    int any_intersect;          /* some source can re-absorb packets (spheres) */
    /* modified random walk: grid_mrw_3d.f90 */
    double *alpha_inv_planck, *diff_coeff;      /* [n_cells], refreshed by prepare_mrw every iteration */
    double mrw_x[100], mrw_y[100];              /* cumulative of Min et al. (2009) eq. 6 */
    size_t n_masked; uint32_t *mask_map;   /* valid cells (geo%mask_map of every geometry) */
    orc_config cfg;
    double *frequencies;        /* copy of cfg.frequencies */
    /* grid_monochromatic.f90: emission pdf over cells per dust at the current frequency */
    int mono_inu; double *mono_cdf; double mono_mean_prob[ORC_MAX_DUST];
    double check_p, check_log1mp;
    int n_dust, n_sources, n_peeled;
    int n_views_total, view_base[64];   /* peeled views numbered through all groups */
    int has_binned, n_theta, n_phi, n_groups;   /* binned images (images_binned.f90): group index n_peeled; n_groups = n_peeled + has_binned */
    dust_t *dust;
    source_t *src;
    double *lum_pdf, *lum_cdf;
    double energy_total;
    peeled_t *peeled;
    double *density;            /* [n_dust][n_cells] */
    double *specific_energy;    /* [n_dust][n_cells] */
    double *specific_energy_add;
    double *specific_energy_sum;
    int32_t *jnu_var_id;        /* [n_dust][n_cells] */
    double *jnu_var_frac;
    double energy_abs_tot[ORC_MAX_DUST];
    /* n_photons (grid_physics_3d.f90:307-318): packets that entered each cell in the current Lucy iteration */
    int64_t *n_photons;
    /* frequency-resolved specific energy (grid_physics_3d.f90:41-56,111-282): [n_bins][n_dust][n_cells] */
    int n_bins;
    double *nu_edges, *log_nu_edges;
    double *spec, *spec_sum, *spec_add;
    double *jnu_bin_frac;       /* [n_dust][nj_max][n_bins] setup_j_nu_bin_fractions :326-348 */
    int nj_max;
    int pda_last_cells, pda_last_outer;   /* diagnostics of the last solve_pda */
    /* pending accumulate totals */
    orc_iter_stats pending;
    int fatal;                  /* set by update_optconsts range error */
};

const char *orc_last_error(const orc_state *st) { return st ? st->err : g_error; }
const char *orc_global_error(void) { return g_error; }
const double *orc_specific_energy_sum(const orc_state *st) { return st->specific_energy_sum; }
const double *orc_specific_energy(const orc_state *st) { return st->specific_energy; }
const double *orc_density(const orc_state *st) { return st->density; }

/* ------------------------------------------------------------------ */
/* Dust set-up: dust_type_4elem.f90:78-293                              */
/* ------------------------------------------------------------------ */

static double *dup(const double *a, size_t n)
{
    double *r = malloc(sizeof(double) * (n ? n : 1));
    if (a) memcpy(r, a, sizeof(double) * n);
    return r;
}

static int dust_setup(dust_t *d, const orc_dust_desc *in, char *err)
{
    memset(d, 0, sizeof(*d));
    d->n_nu = in->n_nu; d->n_mu = in->n_mu; d->n_jnu = in->n_jnu; d->n_enu = in->n_enu; d->n_e = in->n_e;
    d->sublimation_mode = in->sublimation_mode; d->version = in->version;
    d->sublimation_specific_energy = in->sublimation_specific_energy;
    d->minimum_specific_energy = in->minimum_specific_energy;
    int nn = d->n_nu, nm = d->n_mu;
    d->nu = dup(in->nu, nn); d->albedo = dup(in->albedo, nn); d->chi = dup(in->chi, nn);
    d->mu = dup(in->mu, nm);
    size_t np = (size_t)nn * nm;
    d->P1 = dup(in->P1, np); d->P2 = dup(in->P2, np); d->P3 = dup(in->P3, np); d->P4 = dup(in->P4, np);
    d->zero_p2 = 1;
    for (size_t i = 0; i < np; i++) if (d->P2[i] != 0.0) { d->zero_p2 = 0; break; }
    d->mu_min = d->mu[0]; d->mu_max = d->mu[nm - 1];
    double dmu = d->mu_max - d->mu_min;
    /* :183-193 normalise so that the integral over mu is dmu */
    for (int j = 0; j < nn; j++) {
        double norm = integral_linlog(d->mu, d->P1 + (size_t)j * nm, nm);
        if (norm == 0.0) { snprintf(err, 512, "P1 matrix normalization is zero"); return -1; }
        for (int i = 0; i < nm; i++) {
            size_t k = (size_t)j * nm + i;
            d->P1[k] = d->P1[k] / norm * dmu; d->P2[k] = d->P2[k] / norm * dmu;
            d->P3[k] = d->P3[k] / norm * dmu; d->P4[k] = d->P4[k] / norm * dmu;
        }
    }
    /* :195-212 cumulative (trapezium) integrals, normalised to the last value */
    d->P1_cdf = calloc(np, sizeof(double)); d->P2_cdf = calloc(np, sizeof(double));
    for (int j = 0; j < nn; j++) {
        double *c1 = d->P1_cdf + (size_t)j * nm, *c2 = d->P2_cdf + (size_t)j * nm;
        const double *p1 = d->P1 + (size_t)j * nm, *p2 = d->P2 + (size_t)j * nm;
        c1[0] = 0.0; c2[0] = 0.0;
        for (int i = 1; i < nm; i++) {
            double dx = d->mu[i] - d->mu[i - 1];
            c1[i] = c1[i - 1] + 0.5 * (p1[i] + p1[i - 1]) * dx;
            c2[i] = c2[i - 1] + 0.5 * (p2[i] + p2[i - 1]) * dx;
        }
        int all0 = 1; for (int i = 0; i < nm; i++) if (c1[i] != 0.0) all0 = 0;
        if (!all0) { double l = c1[nm - 1]; for (int i = 0; i < nm; i++) c1[i] /= l; }
        all0 = 1; for (int i = 0; i < nm; i++) if (c2[i] != 0.0) all0 = 0;
        if (!all0) { double l = c2[nm - 1]; for (int i = 0; i < nm; i++) c2[i] /= l; }
    }
    /* :214-262 mean opacities: only the specific-energy range is used here */
    if (in->n_e > 0 && in->mo_specific_energy) {
        d->mo_e = dup(in->mo_specific_energy, in->n_e);
        d->e_min = d->mo_e[0]; d->e_max = d->mo_e[in->n_e - 1]; d->have_e_range = 1;
        if (in->mo_chi_rosseland) d->mo_chi_ross = dup(in->mo_chi_rosseland, in->n_e);
        if (in->mo_kappa_planck) d->mo_kappa_planck = dup(in->mo_kappa_planck, in->n_e);
        if (in->mo_chi_inv_planck) d->mo_chi_inv_planck = dup(in->mo_chi_inv_planck, in->n_e);
    }
    /* :264-291 emissivities */
    d->j_nu_var = dup(in->emiss_var, d->n_jnu);
    d->log10_j_nu_var = malloc(sizeof(double) * d->n_jnu);
    for (int i = 0; i < d->n_jnu; i++) d->log10_j_nu_var[i] = log10(d->j_nu_var[i]);
    d->j_nu = calloc(d->n_jnu, sizeof(pdf_t));
    for (int i = 0; i < d->n_jnu; i++) {
        if (pdf_set_log(&d->j_nu[i], in->emiss_nu, in->emiss_jnu + i, d->n_enu, d->n_jnu)) {
            snprintf(err, 512, "emissivity %d has zero integral", i); return -1;
        }
    }
    /* b_nu = j_nu / kappa_nu on the emissivity grid (log pdf): sampled by the MRW (:289-291, 400-419) */
    d->b_nu = calloc(d->n_jnu, sizeof(pdf_t));
    {
        double *kap = malloc(sizeof(double) * nn), *y = malloc(sizeof(double) * d->n_enu);
        for (int k = 0; k < nn; k++) kap[k] = d->chi[k] * (1.0 - d->albedo[k]);      /* d%kappa_nu */
        int ok = 1;
        for (int i = 0; i < d->n_jnu && ok; i++) {
            for (int k = 0; k < d->n_enu; k++) {
                double nu = in->emiss_nu[k];
                double kk = (nu < d->nu[0] || nu > d->nu[nn - 1]) ? NAN : interp1d_loglog(d->nu, kap, nn, nu);
                y[k] = in->emiss_jnu[(size_t)k * d->n_jnu + i] / kk;
            }
            if (pdf_set_log(&d->b_nu[i], in->emiss_nu, y, d->n_enu, 1)) ok = 0;
        }
        free(kap); free(y);
        if (!ok) { for (int i = 0; i < d->n_jnu; i++) if (d->b_nu[i].x) pdf_free(&d->b_nu[i]); free(d->b_nu); d->b_nu = NULL; }
    }
    return 0;
}

static void dust_free(dust_t *d)
{
    free(d->nu); free(d->albedo); free(d->chi); free(d->mu);
    free(d->P1); free(d->P2); free(d->P3); free(d->P4); free(d->P1_cdf); free(d->P2_cdf);
    free(d->j_nu_var); free(d->log10_j_nu_var); free(d->mo_e); free(d->mo_chi_ross);
    free(d->mo_kappa_planck); free(d->mo_chi_inv_planck);
    if (d->b_nu) { for (int i = 0; i < d->n_jnu; i++) pdf_free(&d->b_nu[i]); free(d->b_nu); }
    if (d->j_nu) { for (int i = 0; i < d->n_jnu; i++) pdf_free(&d->j_nu[i]); free(d->j_nu); }
}

/* dust_jnu_var_pos_frac: dust_type_4elem.f90:295-320 (0-based id) */
static void dust_jnu_var_pos_frac(const dust_t *d, double e, int32_t *id, double *frac)
{
    int n = d->n_jnu;
    if (e < d->j_nu_var[0]) { *id = 0; *frac = 0.0; }
    else if (e > d->j_nu_var[n - 1]) { *id = n - 2; *frac = 1.0; }
    else {
        int j = locate(d->j_nu_var, n, e);
        *id = j;
        *frac = (log10(e) - d->log10_j_nu_var[j]) / (d->log10_j_nu_var[j + 1] - d->log10_j_nu_var[j]);
    }
}

/* ------------------------------------------------------------------ */
/* grid_physics_3d.f90: check_energy_abs :555-603, update_energy_abs_tot
 * :605-611, precompute_jnu_var :613-629, update_energy_abs :500-553,
 * sublimate_dust :420-498                                             */
/* ------------------------------------------------------------------ */

static void update_energy_abs_tot(orc_state *st)
{
    for (int d = 0; d < st->n_dust; d++) {
        double s = 0.0;
        const double *e = st->specific_energy + (size_t)d * st->n_cells;
        const double *rho = st->density + (size_t)d * st->n_cells;
        for (size_t ic = 0; ic < st->n_cells; ic++) s += e[ic] * rho[ic] * st->volume[ic];
        st->energy_abs_tot[d] = s;
    }
}

static void check_energy_abs(orc_state *st)
{
    for (int d = 0; d < st->n_dust; d++) {
        const dust_t *du = &st->dust[d];
        double *e = st->specific_energy + (size_t)d * st->n_cells;
        for (size_t ic = 0; ic < st->n_cells; ic++)
            if (e[ic] < du->minimum_specific_energy) e[ic] = du->minimum_specific_energy;
        if (st->cfg.enforce_energy_range && du->have_e_range) {
            for (size_t ic = 0; ic < st->n_cells; ic++) {
                if (e[ic] < du->e_min) e[ic] = du->e_min;
                if (e[ic] > du->e_max) e[ic] = du->e_max;
            }
        }
    }
    update_energy_abs_tot(st);
}

static void precompute_jnu_var(orc_state *st)
{
    for (int d = 0; d < st->n_dust; d++)
        for (size_t ic = 0; ic < st->n_cells; ic++) {
            size_t k = (size_t)d * st->n_cells + ic;
            dust_jnu_var_pos_frac(&st->dust[d], st->specific_energy[k], &st->jnu_var_id[k], &st->jnu_var_frac[k]);
        }
}

static double chi_rosseland(const dust_t *d, double e)
{
    return interp1d_loglog(d->mo_e, d->mo_chi_ross, d->n_e, e);
}

/* scale_specific_energy_spectrum: grid_physics_3d.f90:350-365 */
static void scale_spectrum(orc_state *st, size_t ic, int d, double factor)
{
    if (!st->n_bins) return;
    for (int b = 0; b < st->n_bins; b++) st->spec[((size_t)b * st->n_dust + d) * st->n_cells + ic] *= factor;
}

static void sublimate_dust(orc_state *st)
{
    for (int d = 0; d < st->n_dust; d++) {
        const dust_t *du = &st->dust[d];
        double *e = st->specific_energy + (size_t)d * st->n_cells;
        double *rho = st->density + (size_t)d * st->n_cells;
        double es = du->sublimation_specific_energy;
        switch (du->sublimation_mode) {
        case 1:
            for (size_t ic = 0; ic < st->n_cells; ic++)
                if (e[ic] > es) {
                    rho[ic] = 0.0; e[ic] = du->minimum_specific_energy;
                    for (int b = 0; b < st->n_bins; b++) st->spec[((size_t)b * st->n_dust + d) * st->n_cells + ic] = du->minimum_specific_energy;   /* :441-447 */
                }
            break;
        case 2:
            for (size_t ic = 0; ic < st->n_cells; ic++)
                if (e[ic] > es) {
                    double r = chi_rosseland(du, e[ic]) / chi_rosseland(du, es);
                    rho[ic] = rho[ic] * es / e[ic] * r * r;
                    scale_spectrum(st, ic, d, es / e[ic]);      /* :463-464 */
                    e[ic] = es;
                }
            break;
        case 3:
            for (size_t ic = 0; ic < st->n_cells; ic++) if (e[ic] > es) { scale_spectrum(st, ic, d, es / e[ic]); e[ic] = es; }
            break;
        default: break;
        }
    }
    update_energy_abs_tot(st);
    check_energy_abs(st);
}

static void update_energy_abs(orc_state *st, double scale)
{
    for (int d = 0; d < st->n_dust; d++) {
        double *e = st->specific_energy + (size_t)d * st->n_cells;
        const double *s = st->specific_energy_sum + (size_t)d * st->n_cells;
        for (size_t ic = 0; ic < st->n_cells; ic++) {
            e[ic] = s[ic] * scale / st->volume[ic];
            if (st->volume[ic] == 0.0) e[ic] = 0.0;
        }
    }
    if (st->n_bins) {        /* :517-524 */
        for (int b = 0; b < st->n_bins; b++)
            for (int d = 0; d < st->n_dust; d++)
                for (size_t ic = 0; ic < st->n_cells; ic++) {
                    size_t k = ((size_t)b * st->n_dust + d) * st->n_cells + ic;
                    st->spec[k] = st->spec_sum[k] * scale / st->volume[ic];
                    if (st->volume[ic] == 0.0) st->spec[k] = 0.0;
                }
    }
    if (st->cfg.specific_energy_type == 1 && st->specific_energy_add) {
        size_t n = (size_t)st->n_dust * st->n_cells;
        for (size_t k = 0; k < n; k++) st->specific_energy[k] += st->specific_energy_add[k];
        if (st->n_bins && st->spec_add)      /* :543-545 */
            for (size_t k = 0; k < n * st->n_bins; k++) st->spec[k] += st->spec_add[k];
    }
    update_energy_abs_tot(st);
    check_energy_abs(st);
}

/* ------------------------------------------------------------------ */
/* Partial diffusion approximation: src/grid/grid_pda_3d.f90 with       */
/* grid_pda_{cartesian,spherical,cylindrical}_3d.f90                    */
/* ------------------------------------------------------------------ */

static double kappa_planck(const dust_t *d, double e) { return interp1d_loglog(d->mo_e, d->mo_kappa_planck, d->n_e, e); }
static inline size_t cell_index(const orc_state *st, const int ic[3]);

/* cell_width: grid_geometry_cartesian_3d.f90:49-61, _spherical_3d.f90:60-72, _cylindrical_3d.f90:60-72 */
static double pda_cell_width(const orc_state *st, const int i[3], int dir)
{
    const double *w1 = st->w[0], *w2 = st->w[1], *w3 = st->w[2];
    if (st->grid_type == GRID_CAR) return st->w[dir][i[dir] + 1] - st->w[dir][i[dir]];
    /* cell centre in the first coordinate: half the outer wall if the inner one is 0, geometric mean otherwise */
    const double rc = w1[i[0]] == 0.0 ? w1[i[0] + 1] / 2.0 : pow(10.0, (log10(w1[i[0]]) + log10(w1[i[0] + 1])) / 2.0);
    if (st->grid_type == GRID_SPH) {
        if (dir == 0) return w1[i[0] + 1] - w1[i[0]];
        if (dir == 1) return rc * (w2[i[1] + 1] - w2[i[1]]);
        return rc * sin((w2[i[1]] + w2[i[1] + 1]) / 2.0) * (w3[i[2] + 1] - w3[i[2]]);
    }
    if (dir == 0) return w1[i[0] + 1] - w1[i[0]];
    if (dir == 1) return w2[i[1] + 1] - w2[i[1]];
    return rc * (w3[i[2] + 1] - w3[i[2]]);
}

/* geometrical_factor: grid_pda_*_3d.f90 (wall = 0..5: lower / upper wall of directions 1, 2, 3) */
static double pda_geom_factor(const orc_state *st, int wall, const int i[3])
{
    const double *w1 = st->w[0], *w2 = st->w[1];
    if (st->grid_type == GRID_CYL) {
        if (wall == 0) return 2.0 * w1[i[0]] / (w1[i[0]] + w1[i[0] + 1]);
        if (wall == 1) return 2.0 * w1[i[0] + 1] / (w1[i[0]] + w1[i[0] + 1]);
    } else if (st->grid_type == GRID_SPH) {
        const double sw = w1[i[0]] + w1[i[0] + 1];
        if (wall == 0) return 4.0 * (w1[i[0]] * w1[i[0]]) / (sw * sw);
        if (wall == 1) return 4.0 * (w1[i[0] + 1] * w1[i[0] + 1]) / (sw * sw);
        if (wall == 2) return 2.0 * sin(w2[i[1]]) / (sin(w2[i[1]]) + sin(w2[i[1] + 1]));
        if (wall == 3) return 2.0 * sin(w2[i[1] + 1]) / (sin(w2[i[1]]) + sin(w2[i[1] + 1]));
    }
    return 1.0;
}

static void pda_neighbour(const orc_state *st, const int i[3], int wall, int j[3])      /* next_cell_int */
{
    j[0] = i[0]; j[1] = i[1]; j[2] = i[2];
    const int dir = wall >> 1;
    j[dir] += (wall & 1) ? 1 : -1;
    if (dir == 2 && st->grid_type != GRID_CAR) { if (j[2] < 0) j[2] = st->n3 - 1; if (j[2] >= st->n3) j[2] = 0; }   /* phi is periodic */
}

static double pda_dtau_rosseland(const orc_state *st, const int i[3], int dir)
{
    const size_t ic = cell_index(st, i);
    double t = 0.0;
    for (int d = 0; d < st->n_dust; d++) {
        const size_t k = (size_t)d * st->n_cells + ic;
        t += st->density[k] * chi_rosseland(&st->dust[d], st->specific_energy[k]) * pda_cell_width(st, i, dir);
    }
    return t;
}

static double pda_e_mean(const orc_state *st, size_t ic)       /* update_e_mean :72-82 */
{
    double sr = 0.0, e = 0.0;
    for (int d = 0; d < st->n_dust; d++) sr += st->density[(size_t)d * st->n_cells + ic];
    if (!(sr > 0.0)) return 0.0;
    for (int d = 0; d < st->n_dust; d++) {
        const size_t k = (size_t)d * st->n_cells + ic;
        e += st->density[k] * st->specific_energy[k] / kappa_planck(&st->dust[d], st->specific_energy[k]);
    }
    return e / sr;
}

static void pda_update_specific_energy(orc_state *st, size_t ic, double e_mean)    /* update_specific_energy :36-70 */
{
    for (int d = 0; d < st->n_dust; d++) {
        const dust_t *du = &st->dust[d];
        const size_t k = (size_t)d * st->n_cells + ic;
        double s = st->specific_energy[k];
        const double s_old = s, smin = du->mo_e[0], smax = du->mo_e[du->n_e - 1];
        if (e_mean < smin / kappa_planck(du, smin)) s = smin;
        else if (e_mean > smax / kappa_planck(du, smax)) s = smax;
        else {
            for (;;) {
                const double s_prev = s;
                s = e_mean * kappa_planck(du, s);
                const double a = s / s_prev, b = s_prev / s;
                if ((a > b ? a : b) - 1.0 < 1.e-5) break;
                if (s != s) break;          /* NaN guard: the reference would loop for ever */
            }
        }
        st->specific_energy[k] = s;
        if (s_old > 0.0) scale_spectrum(st, ic, d, s / s_old);
    }
}

/* solve_pda :84-172.  The Gauss pivot branch (< 10 000 cells, lineq_gausselim of fortranlib, source absent) solves the
 * linear system  sum_walls c_w (e_next - e_curr) = 0  of the PDA cells exactly; it is restated as Gaussian elimination
 * with partial pivoting on the matrix with one ROW per cell's equation (the reference stores a(id_next, id_curr), i.e.
 * the equation of id_curr in COLUMN id_curr of a column-major array: its solver works on that transposed storage). */
static int solve_pda(orc_state *st)
{
    st->pda_last_cells = 0; st->pda_last_outer = 0;
    if (!GRID_IS_3IDX(st->grid_type)) return 0;          /* grid_pda_disabled.f90: do_pda = .false. */
    if (!st->n_photons) return 0;
    const size_t nc = st->n_cells;
    int64_t tot = 0;
    for (size_t ic = 0; ic < nc; ic++) tot += st->n_photons[ic];
    const double mean_n = (double)(tot / (int64_t)nc);    /* integer division, :99 */
    double thr_d = ceil(0.005 * mean_n); if (thr_d < 30.0) thr_d = 30.0;
    const int64_t thr = (int64_t)thr_d;
    uint8_t *do_pda = calloc(nc, 1);
    for (size_t ic = 0; ic < nc; ic++) {
        double sr = 0.0;
        for (int d = 0; d < st->n_dust; d++) sr += st->density[(size_t)d * nc + ic];
        do_pda[ic] = st->n_photons[ic] < thr && sr > 0.0;
    }
    /* check_allowed_pda: no cells on the outer faces (directions 1, 2; also 3 for Cartesian grids) */
    size_t n_pda = 0;
    for (int i3 = 0; i3 < st->n3; i3++) for (int i2 = 0; i2 < st->n2; i2++) for (int i1 = 0; i1 < st->n1; i1++) {
        const size_t ic = ((size_t)i3 * st->n2 + i2) * st->n1 + i1;
        if (i1 == 0 || i1 == st->n1 - 1 || i2 == 0 || i2 == st->n2 - 1) do_pda[ic] = 0;
        if (st->grid_type == GRID_CAR && (i3 == 0 || i3 == st->n3 - 1)) do_pda[ic] = 0;
        n_pda += do_pda[ic];
    }
    if (!n_pda) { free(do_pda); return 0; }
    st->pda_last_cells = (int)n_pda;
    const int exact = n_pda < 10000;
    const double tolerance = exact ? 1.e-5 : 1.e-4;
    double *e_mean = malloc(sizeof(double) * nc);
    for (size_t ic = 0; ic < nc; ic++) e_mean[ic] = pda_e_mean(st, ic);
    int32_t *cells = malloc(sizeof(int32_t) * 3 * n_pda);
    int64_t *id_pda = malloc(sizeof(int64_t) * nc);
    size_t np = 0;
    for (int i3 = 0; i3 < st->n3; i3++) for (int i2 = 0; i2 < st->n2; i2++) for (int i1 = 0; i1 < st->n1; i1++) {
        const size_t ic = ((size_t)i3 * st->n2 + i2) * st->n1 + i1;
        id_pda[ic] = -1;
        if (do_pda[ic]) { cells[3 * np] = i1; cells[3 * np + 1] = i2; cells[3 * np + 2] = i3; id_pda[ic] = (int64_t)np; np++; }
    }
    const int n_walls = st->grid_type == GRID_CAR ? 6 : st->n_dim * 2;      /* geo%n_dim * 2 */
    const size_t ntot = (size_t)st->n_dust * nc;
    double *prev = malloc(sizeof(double) * ntot);
    double *coef = malloc(sizeof(double) * 6 * n_pda);
    for (int outer = 1; outer < 100000; outer++) {
        st->pda_last_outer = outer;
        memcpy(prev, st->specific_energy, sizeof(double) * ntot);
        for (size_t q = 0; q < n_pda; q++) { const int *i = &cells[3 * q]; e_mean[cell_index(st, i)] = pda_e_mean(st, cell_index(st, i)); }
        /* the coefficients depend on the specific energy, which is only updated after the solve */
        for (size_t q = 0; q < n_pda; q++) {
            const int *i = &cells[3 * q];
            for (int wall = 0; wall < n_walls; wall++) {
                const int dir = wall >> 1; int j[3];
                pda_neighbour(st, i, wall, j);
                double dsum = pda_dtau_rosseland(st, i, dir) + pda_dtau_rosseland(st, j, dir);
                double c;
                if (exact) { if (dsum < 1e-100) dsum = 1e-100; c = 1. / dsum / pda_cell_width(st, i, dir); }
                else c = 1. / dsum / pda_cell_width(st, i, dir);
                coef[6 * q + wall] = c * pda_geom_factor(st, wall, i);
            }
        }
        if (exact) {      /* solve_pda_indiv_exact :185-256 */
            const size_t n = n_pda;
            double *a = calloc(n * n, sizeof(double)), *b = calloc(n, sizeof(double));
            for (size_t q = 0; q < n; q++) {
                const int *i = &cells[3 * q];
                for (int wall = 0; wall < n_walls; wall++) {
                    int j[3]; pda_neighbour(st, i, wall, j);
                    const double c = coef[6 * q + wall];
                    a[q * n + q] -= c;
                    const int64_t qn = id_pda[cell_index(st, j)];
                    if (qn >= 0) a[q * n + (size_t)qn] += c;
                    else b[q] -= c * e_mean[cell_index(st, j)];
                }
            }
            for (size_t k = 0; k < n; k++) {        /* Gaussian elimination, partial pivoting */
                size_t piv = k; double big = fabs(a[k * n + k]);
                for (size_t r = k + 1; r < n; r++) if (fabs(a[r * n + k]) > big) { big = fabs(a[r * n + k]); piv = r; }
                if (piv != k) {
                    for (size_t c2 = k; c2 < n; c2++) { double t = a[k * n + c2]; a[k * n + c2] = a[piv * n + c2]; a[piv * n + c2] = t; }
                    double t = b[k]; b[k] = b[piv]; b[piv] = t;
                }
                for (size_t r = k + 1; r < n; r++) {
                    const double f = a[r * n + k] / a[k * n + k];
                    if (f == 0.0) continue;
                    for (size_t c2 = k; c2 < n; c2++) a[r * n + c2] -= f * a[k * n + c2];
                    b[r] -= f * b[k];
                }
            }
            for (size_t kk = n; kk-- > 0;) {
                double t = b[kk];
                for (size_t c2 = kk + 1; c2 < n; c2++) t -= a[kk * n + c2] * b[c2];
                b[kk] = t / a[kk * n + kk];
            }
            for (size_t q = 0; q < n; q++) { const size_t ic = cell_index(st, &cells[3 * q]); e_mean[ic] = b[q]; pda_update_specific_energy(st, ic, e_mean[ic]); }
            free(a); free(b);
        } else {          /* solve_pda_indiv_iterative :258-325: Gauss-Seidel in cell order */
            for (;;) {
                double max_diff = 0.0;
                for (size_t q = 0; q < n_pda; q++) {
                    const int *i = &cells[3 * q];
                    double a = 0.0, b = 0.0;
                    for (int wall = 0; wall < n_walls; wall++) {
                        int j[3]; pda_neighbour(st, i, wall, j);
                        const double c = coef[6 * q + wall];
                        a = a - c;
                        b = b - c * e_mean[cell_index(st, j)];
                    }
                    const size_t ic = cell_index(st, i);
                    const double e_new = b / a, diff = fabs(e_new - e_mean[ic]) / e_mean[ic];
                    if (diff > max_diff) max_diff = diff;
                    e_mean[ic] = e_new;
                }
                if (max_diff < 1.e-4) break;
            }
            for (size_t q = 0; q < n_pda; q++) { const size_t ic = cell_index(st, &cells[3 * q]); pda_update_specific_energy(st, ic, e_mean[ic]); }
        }
        double maxdiff = 0.0;
        for (size_t k = 0; k < ntot; k++) {
            const double dv = fabs(st->specific_energy[k] - prev[k]) / prev[k];
            if (dv > maxdiff) maxdiff = dv;          /* NaN (0/0) compares false and is skipped, like maxval */
        }
        if (maxdiff < tolerance) break;
    }
    free(do_pda); free(e_mean); free(cells); free(id_pda); free(prev); free(coef);
    update_energy_abs_tot(st);
    check_energy_abs(st);
    return 0;
}

/* ------------------------------------------------------------------ */
/* create / destroy                                                    */
/* ------------------------------------------------------------------ */

static double spacing(double x)
{
    x = fabs(x);
    if (x == 0.0) return DBL_MIN;
    return nextafter(x, INFINITY) - x;
}

static int peeled_setup(orc_state *st, peeled_t *p, const orc_peeled_desc *in);
static void raytracing_caches(const orc_state *st);
static void peeled_free(peeled_t *p);

/* ---- voronoi: grid_geometry_voronoi.f90 -------------------------------- */

static inline double vdist2(const orc_state *st, int32_t i, const double r[3])
{
    const double *s = st->vsite + 3 * (size_t)i;
    double dx = s[0] - r[0], dy = s[1] - r[1], dz = s[2] - r[2];
    return dx * dx + dy * dy + dz * dz;
}

/* nearest site by steepest descent over the neighbour (Delaunay) graph from
 * `seed`: the reference uses a kd-tree (kdtree2_n_nearest, :224); both return the
 * nearest site. */
static int32_t vor_nearest_from(const orc_state *st, const double r[3], int32_t seed)
{
    int32_t cur = seed;
    double dcur = vdist2(st, cur, r);
    for (;;) {
        int32_t best = cur; double dbest = dcur;
        for (int32_t k = st->vidx[cur]; k < st->vidx[cur + 1]; k++) {
            int32_t nb = st->vneigh[k];
            if (nb < 0) continue;
            double d = vdist2(st, nb, r);
            if (d < dbest) { dbest = d; best = nb; }
        }
        if (best == cur) return cur;
        cur = best; dcur = dbest;
    }
}

static inline int vor_seed_cell(const orc_state *st, const double r[3])
{
    int g = st->vg, id[3];
    for (int a = 0; a < 3; a++) {
        double f = (r[a] - st->vbox[2 * a]) / (st->vbox[2 * a + 1] - st->vbox[2 * a]);
        int i = (int)(f * g);
        id[a] = i < 0 ? 0 : (i >= g ? g - 1 : i);
    }
    return (id[2] * g + id[1]) * g + id[0];
}

static int32_t vor_nearest(const orc_state *st, const double r[3])
{
    return vor_nearest_from(st, r, st->vseed[vor_seed_cell(st, r)]);
}

static int voronoi_setup(orc_state *st, const orc_grid_desc *gd)
{
    size_t n = (size_t)gd->n_cells;
    if (n < 1 || !gd->vor_sites || !gd->vor_idx || !gd->vor_neighs || !gd->vor_volume) {
        snprintf(g_error, sizeof g_error, "voronoi grid needs sites, volumes and neighbour lists"); return 1;
    }
    st->n_cells = n;
    st->vsite = dup(gd->vor_sites, 3 * n);
    st->vbb = gd->vor_bb ? dup(gd->vor_bb, 6 * n) : NULL;
    st->vidx = malloc(sizeof(int32_t) * (n + 1)); memcpy(st->vidx, gd->vor_idx, sizeof(int32_t) * (n + 1));
    size_t nn = (size_t)st->vidx[n];
    st->vneigh = malloc(sizeof(int32_t) * (nn ? nn : 1)); memcpy(st->vneigh, gd->vor_neighs, sizeof(int32_t) * nn);
    for (size_t k = 0; k < nn; k++)
        if (st->vneigh[k] < -6 || st->vneigh[k] >= (int32_t)n) { snprintf(g_error, sizeof g_error, "neighbour index out of range"); return 1; }
    memcpy(st->vbox, gd->vor_box, sizeof st->vbox);
    st->volume = malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; i++) st->volume[i] = gd->vor_volume[i] < 0.0 ? 0.0 : gd->vor_volume[i];
    /* seed grid: nearest site of each grid-cell centre (walk from the previous answer) */
    int g = (int)ceil(cbrt((double)n / 4.0));
    if (g < 1) g = 1;
    if (g > 256) g = 256;
    st->vg = g;
    st->vseed = malloc(sizeof(int32_t) * (size_t)g * g * g);
    int32_t last = 0;
    for (int k = 0; k < g; k++) for (int j = 0; j < g; j++) for (int i = 0; i < g; i++) {
        double c[3] = {st->vbox[0] + (i + 0.5) / g * (st->vbox[1] - st->vbox[0]),
                       st->vbox[2] + (j + 0.5) / g * (st->vbox[3] - st->vbox[2]),
                       st->vbox[4] + (k + 0.5) / g * (st->vbox[5] - st->vbox[4])};
        last = vor_nearest_from(st, c, last);
        st->vseed[((size_t)k * g + j) * g + i] = last;
    }
    return 0;
}


/* ---- AMR: grid_geometry_amr.f90 ------------------------------------------- */

static inline int amr_in_grid(const amr_grid *g, const double r[3])   /* in_grid :82-96 */
{
    for (int a = 0; a < 3; a++) { if (r[a] < g->lo[a]) return 0; if (r[a] > g->hi[a]) return 0; }
    return 1;
}

static inline size_t amr_go_index(const amr_grid *g, int i1, int i2, int i3)   /* indices 0..n+1 */
{
    return ((size_t)i3 * (g->n[1] + 2) + i2) * (g->n[0] + 2) + i1;
}

static int amr_covered(const orc_state *st, size_t ic)
{
    const amr_grid *g = &st->amr[st->amr_cell_grid[ic]];
    size_t l = ic - g->start;
    int i1 = (int)(l % g->n[0]), i2 = (int)((l / g->n[0]) % g->n[1]), i3 = (int)(l / ((size_t)g->n[0] * g->n[1]));
    return g->go[amr_go_index(g, i1 + 1, i2 + 1, i3 + 1)] != 0;
}

/* aligned :184-191 */
static int amr_aligned(double x1, double x2, double dx)
{
    double r = fmod(fabs(x1 - x2), dx);
    if (r > 0.5 * dx) r = dx - r;
    return fabs(r / dx) < 1.e-8;
}

/* read_grid/read_level + setup_grid_geometry :111-508 */
static int amr_setup(orc_state *st, const orc_grid_desc *gd)
{
    int ng = gd->n_amr_grids, nl = gd->n_amr_levels;
    if (ng < 1 || nl < 1 || !gd->amr_level || !gd->amr_n || !gd->amr_bounds) { snprintf(g_error, sizeof g_error, "amr grid needs levels and grids"); return 1; }
    st->n_amr_grids = ng; st->n_amr_levels = nl;
    st->amr = calloc(ng, sizeof(amr_grid));
    size_t start = 0; double min_width = DBL_MAX;
    for (int k = 0; k < ng; k++) {
        amr_grid *g = &st->amr[k];
        g->level = gd->amr_level[k];
        if (g->level < 1 || g->level > nl || (k > 0 && g->level < st->amr[k - 1].level)) { snprintf(g_error, sizeof g_error, "amr grids must be listed level by level"); return 1; }
        for (int a = 0; a < 3; a++) {
            g->n[a] = gd->amr_n[3 * k + a];
            g->lo[a] = gd->amr_bounds[6 * k + 2 * a]; g->hi[a] = gd->amr_bounds[6 * k + 2 * a + 1];
            g->w[a] = malloc(sizeof(double) * (g->n[a] + 1));
            /* fortranlib linspace: x(i) = (xmax - xmin) * (i - 1) / (n - 1) + xmin */
            for (int i = 0; i <= g->n[a]; i++) g->w[a][i] = (g->hi[a] - g->lo[a]) * (double)i / (double)g->n[a] + g->lo[a];
            g->width[a] = (g->hi[a] - g->lo[a]) / (double)g->n[a];
            if (g->width[a] < min_width) min_width = g->width[a];
        }
        g->volume = g->width[0] * g->width[1] * g->width[2];
        g->start = start; start += (size_t)g->n[0] * g->n[1] * g->n[2];
        g->go = calloc((size_t)(g->n[0] + 2) * (g->n[1] + 2) * (g->n[2] + 2), sizeof(int32_t));
    }
    st->n_cells = start;
    st->amr_eps = min_width / 2.0;
    /* consistency checks of :226-305, with the reference's messages */
    for (int k = 0; k < ng; k++) {
        const amr_grid *g = &st->amr[k];
        int ref = -1;
        for (int q = 0; q < ng; q++) if (st->amr[q].level == g->level) { ref = q; break; }
        const amr_grid *gr = &st->amr[ref];
        int igrid = k - ref + 1;
        for (int a = 0; a < 3; a++) {
            if (fabs(g->width[a] - gr->width[a]) > 1.e-10 * g->width[a]) {
                snprintf(g_error, sizeof g_error, "Grids 1 and %d in level %d have differing cell widths in the %c direction", igrid, g->level, "xyz"[a]);
                return 1;
            }
            if (!amr_aligned(g->lo[a], gr->lo[a], gr->width[a])) {
                snprintf(g_error, sizeof g_error, "Grids 1 and %d in level %d have edges that are not separated by an integer number of cells in the %c direction", igrid, g->level, "xyz"[a]);
                return 1;
            }
        }
        if (g->level > 1) {
            int pref = -1;
            for (int q = 0; q < ng; q++) if (st->amr[q].level == g->level - 1) { pref = q; break; }
            if (pref < 0) { snprintf(g_error, sizeof g_error, "amr level %d has no grids", g->level - 1); return 1; }
            const amr_grid *gp = &st->amr[pref];
            for (int a = 0; a < 3; a++) {
                double rf = gp->width[a] / gr->width[a];
                if (fabs(rf - nearbyint(rf)) > 1.e-10) {
                    snprintf(g_error, sizeof g_error, "Refinement factor in the %c direction between level %d and level %d is not an integer (%.3f)", "xyz"[a], g->level - 1, g->level, rf);
                    return 1;
                }
                if (!amr_aligned(g->lo[a], gp->lo[a], gp->width[a])) {
                    snprintf(g_error, sizeof g_error, "Grid %d in level %d is not aligned with cells in level %d in the %c direction", igrid, g->level, g->level - 1, "xyz"[a]);
                    return 1;
                }
            }
        }
    }
    st->volume = malloc(sizeof(double) * st->n_cells);
    st->amr_cell_grid = malloc(sizeof(int32_t) * st->n_cells);
    for (int k = 0; k < ng; k++) {
        const amr_grid *g = &st->amr[k];
        size_t nc = (size_t)g->n[0] * g->n[1] * g->n[2];
        for (size_t c = 0; c < nc; c++) { st->volume[g->start + c] = g->volume; st->amr_cell_grid[g->start + c] = k; }
    }
    /* cells overlapped by a grid of the next level (:357-382); later grids overwrite earlier ones */
    for (int l1 = nl - 1; l1 >= 1; l1--)
        for (int k1 = 0; k1 < ng; k1++) {
            amr_grid *g1 = &st->amr[k1];
            if (g1->level != l1) continue;
            for (int k2 = 0; k2 < ng; k2++) {
                const amr_grid *g2 = &st->amr[k2];
                if (g2->level != l1 + 1) continue;
                int hit = 1;
                for (int a = 0; a < 3; a++) if (g1->hi[a] < g2->lo[a] || g1->lo[a] > g2->hi[a]) hit = 0;   /* grids_intersect */
                if (!hit) continue;
                for (int i1 = 1; i1 <= g1->n[0]; i1++) for (int i2 = 1; i2 <= g1->n[1]; i2++) for (int i3 = 1; i3 <= g1->n[2]; i3++) {
                    double r[3] = {0.5 * (g1->w[0][i1 - 1] + g1->w[0][i1]), 0.5 * (g1->w[1][i2 - 1] + g1->w[1][i2]), 0.5 * (g1->w[2][i3 - 1] + g1->w[2][i3])};
                    if (amr_in_grid(g2, r)) g1->go[amr_go_index(g1, i1, i2, i3)] = k2 + 1;
                }
            }
        }
    /* one step outside each grid: which grid of the same or a coarser level is there (:384-486) */
    for (int k1 = 0; k1 < ng; k1++) {
        amr_grid *g1 = &st->amr[k1];
        for (int l2 = g1->level; l2 >= 1; l2--)
            for (int k2 = 0; k2 < ng; k2++) {
                const amr_grid *g2 = &st->amr[k2];
                if (g2->level != l2 || k2 == k1) continue;
                int close = 1;   /* grids_close :69-80 */
                for (int a = 0; a < 3; a++)
                    if (g1->hi[a] < g2->lo[a] - g2->width[a] * 0.5 || g1->lo[a] > g2->hi[a] + g2->width[a] * 0.5) close = 0;
                if (!close) continue;
                for (int a = 0; a < 3; a++) {
                    const int b = (a + 1) % 3, c = (a + 2) % 3;
                    for (int side = 0; side < 2; side++) {
                        int idx[3]; double r[3];
                        idx[a] = side ? g1->n[a] + 1 : 0;
                        r[a] = side ? g1->hi[a] + g1->width[a] * 0.5 : g1->lo[a] - g1->width[a] * 0.5;
                        for (int ib = 1; ib <= g1->n[b]; ib++) for (int ic = 1; ic <= g1->n[c]; ic++) {
                            idx[b] = ib; idx[c] = ic;
                            r[b] = 0.5 * (g1->w[b][ib - 1] + g1->w[b][ib]); r[c] = 0.5 * (g1->w[c][ic - 1] + g1->w[c][ic]);
                            size_t q = amr_go_index(g1, idx[0], idx[1], idx[2]);
                            if (amr_in_grid(g2, r) && g1->go[q] == 0) g1->go[q] = k2 + 1;
                        }
                    }
                }
            }
    }
    return 0;
}

/* ipos (fortranlib): 1-based bin of x in n equal bins of [xmin, xmax]; 0 below, n+1 above */
static inline int amr_ipos(double xmin, double xmax, double x, int n)
{
    if (x < xmin) return 0;
    if (x > xmax) return n + 1;
    if (x < xmax) { int i = (int)((x - xmin) / (xmax - xmin) * (double)n) + 1; return i > n ? n : i; }
    return n;
}

/* ipos2 :510-519 */
static inline int amr_ipos2(double xmin, double xmax, double x, int n)
{
    double eps = (xmax - xmin) * 1.e-10;
    int i = amr_ipos(xmin, xmax, x, n);
    if (i == 0 && fabs(x - xmin) < eps) i = 1;
    if (i == n + 1 && fabs(x - xmax) < eps) i = n;
    return i;
}

/* find_position_in_grid :521-545: unique id, or -1 (invalid_cell) */
static int64_t amr_find_position(const orc_state *st, const double r[3], int k)
{
    for (;;) {
        const amr_grid *g = &st->amr[k];
        int i[3];
        for (int a = 0; a < 3; a++) i[a] = amr_ipos2(g->lo[a], g->hi[a], r[a], g->n[a]);
        int32_t go = g->go[amr_go_index(g, i[0], i[1], i[2])];
        if (go == 0) {
            for (int a = 0; a < 3; a++) if (i[a] < 1 || i[a] > g->n[a]) return -1;
            return (int64_t)(g->start + ((size_t)(i[2] - 1) * g->n[1] + (i[1] - 1)) * g->n[0] + (i[0] - 1));
        }
        k = go - 1;
    }
}

/* find_cell_position :560-572 */
static int64_t amr_find_cell(const orc_state *st, const double r[3])
{
    for (int k = 0; k < st->n_amr_grids && st->amr[k].level == 1; k++)
        if (amr_in_grid(&st->amr[k], r)) return amr_find_position(st, r, k);
    return -1;
}

static inline void amr_cell_coords(const orc_state *st, size_t ic, const amr_grid **gp, int i[3])
{
    const amr_grid *g = &st->amr[st->amr_cell_grid[ic]];
    size_t l = ic - g->start;
    i[0] = (int)(l % g->n[0]); i[1] = (int)((l / g->n[0]) % g->n[1]); i[2] = (int)(l / ((size_t)g->n[0] * g->n[1]));
    *gp = g;
}

/* setup_grid_geometry + octree_setup_indiv: grid_geometry_octree.f90:147-246.
 * Cells are numbered depth-first (pre-order) as the `refined` list is read. */
static int octree_setup(orc_state *st, const orc_grid_desc *gd)
{
    size_t n = (size_t)gd->n_cells;
    if (n < 1 || !gd->refined) { snprintf(g_error, sizeof g_error, "octree needs a refined list"); return 1; }
    st->n_cells = n;
    st->ox = malloc(sizeof(double) * n); st->oy = malloc(sizeof(double) * n); st->oz = malloc(sizeof(double) * n);
    st->odx = malloc(sizeof(double) * n); st->ody = malloc(sizeof(double) * n); st->odz = malloc(sizeof(double) * n);
    st->orefined = malloc(n); st->osubcell = malloc(n);
    st->oparent = malloc(sizeof(int32_t) * n); st->ochildren = malloc(sizeof(int32_t) * 8 * n);
    for (size_t i = 0; i < n; i++) { st->orefined[i] = gd->refined[i] == 1; st->oparent[i] = -1; st->osubcell[i] = -1; }
    for (size_t i = 0; i < 8 * n; i++) st->ochildren[i] = -1;
    st->ox[0] = gd->oct_center[0]; st->oy[0] = gd->oct_center[1]; st->oz[0] = gd->oct_center[2];
    st->odx[0] = gd->oct_half[0]; st->ody[0] = gd->oct_half[1]; st->odz[0] = gd->oct_half[2];
    /* explicit stack of (parent, next child slot) reproducing the recursion order */
    int32_t *stack_p = malloc(sizeof(int32_t) * 64); int *stack_k = malloc(sizeof(int) * 64);
    int depth = 0; size_t filled = 1;
    if (st->orefined[0]) { stack_p[0] = 0; stack_k[0] = 0; depth = 1; }
    while (depth > 0) {
        int32_t par = stack_p[depth - 1]; int k = stack_k[depth - 1];
        if (k == 8) { depth--; continue; }
        stack_k[depth - 1] = k + 1;
        if (filled >= n) { snprintf(g_error, sizeof g_error, "refined array is not self-consistent"); free(stack_p); free(stack_k); return 1; }
        int32_t c = (int32_t)filled++;
        st->ochildren[8 * (size_t)par + k] = c;
        int sx = (k & 1) ? 1 : -1, sy = (k & 2) ? 1 : -1, sz = (k & 4) ? 1 : -1;
        st->ox[c] = st->ox[par] + sx * st->odx[par] / 2.0;
        st->oy[c] = st->oy[par] + sy * st->ody[par] / 2.0;
        st->oz[c] = st->oz[par] + sz * st->odz[par] / 2.0;
        st->odx[c] = st->odx[par] / 2.0; st->ody[c] = st->ody[par] / 2.0; st->odz[c] = st->odz[par] / 2.0;
        st->oparent[c] = par; st->osubcell[c] = (int8_t)k;
        if (st->orefined[c]) {
            if (depth >= 63) { snprintf(g_error, sizeof g_error, "octree too deep"); free(stack_p); free(stack_k); return 1; }
            stack_p[depth] = c; stack_k[depth] = 0; depth++;
        }
    }
    free(stack_p); free(stack_k);
    if (filled != n) { snprintf(g_error, sizeof g_error, "refined array is not self-consistent"); return 1; }
    st->volume = malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; i++) {
        st->volume[i] = st->odx[i] * st->ody[i] * st->odz[i] * 8.0;
        if (st->volume[i] == 0.0) { snprintf(g_error, sizeof g_error, "all volumes should be greater than zero"); return 1; }
    }
    st->obox[0] = st->ox[0] - st->odx[0]; st->obox[1] = st->ox[0] + st->odx[0];
    st->obox[2] = st->oy[0] - st->ody[0]; st->obox[3] = st->oy[0] + st->ody[0];
    st->obox[4] = st->oz[0] - st->odz[0]; st->obox[5] = st->oz[0] + st->odz[0];
    double m = st->odx[0] > st->ody[0] ? st->odx[0] : st->ody[0];
    if (st->odz[0] > m) m = st->odz[0];
    st->oct_eps = spacing(m) * 3.0;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Spherical / cylindrical polar grids: set-up                          */
/* grid_geometry_spherical_3d.f90:90-203, grid_geometry_cylindrical_3d.f90:90-175 */
/* ------------------------------------------------------------------ */
static int polar_setup(orc_state *st, const orc_grid_desc *gd)
{
    const int sph = st->grid_type == GRID_SPH;
    st->n1 = gd->n1; st->n2 = gd->n2; st->n3 = gd->n3;
    st->n[0] = st->n1; st->n[1] = st->n2; st->n[2] = st->n3;
    st->n_cells = (size_t)st->n1 * st->n2 * st->n3;
    const double *win[3] = {gd->w1, gd->w2, gd->w3};
    for (int a = 0; a < 3; a++) st->w[a] = dup(win[a], st->n[a] + 1);
    for (int i = 0; i <= st->n1; i++)
        if (st->w[0][i] < 0.0) { snprintf(g_error, sizeof g_error, sph ? "r walls should be positive" : "w walls should be positive"); return 1; }
    if (sph) for (int i = 0; i <= st->n2; i++)
        if (st->w[1][i] < 0.0 || st->w[1][i] > PI) { snprintf(g_error, sizeof g_error, "theta walls should be between 0 and pi"); return 1; }
    for (int i = 0; i <= st->n3; i++)
        if (st->w[2][i] < 0.0 || st->w[2][i] > 2.0 * PI) { snprintf(g_error, sizeof g_error, "phi walls should be between 0 and 2*pi"); return 1; }
    st->volume = malloc(sizeof(double) * st->n_cells);
    for (int k = 0; k < st->n3; k++) for (int j = 0; j < st->n2; j++) for (int i = 0; i < st->n1; i++) {
        const double *w1 = st->w[0], *w2 = st->w[1], *w3 = st->w[2];
        double dphi = w3[k + 1] - w3[k], vol;
        if (sph) {
            double dr3 = w1[i + 1] * w1[i + 1] * w1[i + 1] - w1[i] * w1[i] * w1[i];
            double dcost = cos(w2[j]) - cos(w2[j + 1]);
            vol = dr3 * dcost * dphi / 3.0;
        } else {
            double dw2 = w1[i + 1] * w1[i + 1] - w1[i] * w1[i];
            vol = dw2 * (w2[j + 1] - w2[j]) * dphi / 2.0;
        }
        st->volume[((size_t)k * st->n2 + j) * st->n1 + i] = vol;
        if (vol == 0.0) { snprintf(g_error, sizeof g_error, "all volumes should be greater than zero"); return 1; }
    }
    static const char *names_s[3] = {"dr", "dt", "dphi"}, *names_c[3] = {"dw", "dz", "dphi"};
    for (int a = 0; a < 3; a++) for (int i = 0; i < st->n[a]; i++)
        if (st->w[a][i + 1] - st->w[a][i] == 0.0) {
            snprintf(g_error, sizeof g_error, "all %s values should be greater than zero", sph ? names_s[a] : names_c[a]); return 1;
        }
    st->wr2 = malloc(sizeof(double) * (st->n1 + 1));
    for (int i = 0; i <= st->n1; i++) st->wr2[i] = st->w[0][i] * st->w[0][i];
    st->wtanp = malloc(sizeof(double) * (st->n3 + 1));
    for (int i = 0; i <= st->n3; i++) st->wtanp[i] = tan(st->w[2][i]);
    st->midplane = -2;
    if (sph) {
        st->wtant = malloc(sizeof(double) * (st->n2 + 1));
        st->wtant2 = malloc(sizeof(double) * (st->n2 + 1));
        st->wcost = malloc(sizeof(double) * (st->n2 + 1));
        /* :175 if(any(abs(w2 - pi/2) < 1e-6)) midplane = minloc(abs(w2 - pi/2), 1) */
        double m = DBL_MAX; int im = 0;
        for (int i = 0; i <= st->n2; i++) {
            st->wtant[i] = tan(st->w[1][i]); st->wtant2[i] = st->wtant[i] * st->wtant[i]; st->wcost[i] = cos(st->w[1][i]);
            double d = fabs(st->w[1][i] - PI / 2.0);
            if (d < m) { m = d; im = i; }
        }
        if (m < 1.e-6) st->midplane = im;
    }
    st->n_dim = st->n3 == 1 ? 2 : 3;
    for (int a = 0; a < 3; a++) st->ew[a] = malloc(sizeof(double) * (st->n[a] + 1));
    for (int i = 0; i <= st->n1; i++) st->ew[0][i] = 3.0 * spacing(st->w[0][i]);
    for (int i = 0; i <= st->n2; i++) st->ew[1][i] = sph ? 3.0 * spacing(1.0) : 3.0 * spacing(st->w[1][i]);
    for (int i = 0; i <= st->n3; i++) st->ew[2][i] = 3.0 * spacing(1.0);
    return 0;
}

int orc_create(const orc_problem *pr, orc_state **out)
{
    g_error[0] = 0;
    if (!pr || !out) { snprintf(g_error, sizeof g_error, "null argument"); return 1; }
    if (pr->grid.type < GRID_CAR || pr->grid.type > GRID_CYL) { snprintf(g_error, sizeof g_error, "unknown grid type"); return 1; }
    if (pr->n_dust < 0 || pr->n_dust > ORC_MAX_DUST) { snprintf(g_error, sizeof g_error, "n_dust out of range"); return 1; }
    orc_state *st = calloc(1, sizeof(*st));
    st->cfg = pr->config;
    if (st->cfg.monochromatic) {
        if (st->cfg.n_frequencies < 1 || !st->cfg.frequencies) { snprintf(g_error, sizeof g_error, "monochromatic mode needs a frequency table"); free(st); return 1; }
        st->frequencies = dup(st->cfg.frequencies, st->cfg.n_frequencies);
        st->cfg.frequencies = st->frequencies;
    }
    st->mono_inu = -1;
    st->grid_type = pr->grid.type;
    st->check_p = pr->config.propagation_check_frequency;
    st->check_log1mp = (st->check_p > 0.0 && st->check_p < 1.0) ? log1p(-st->check_p) : -1.0;
    if (st->grid_type == GRID_CAR) {
        st->n1 = pr->grid.n1; st->n2 = pr->grid.n2; st->n3 = pr->grid.n3;
        st->n[0] = st->n1; st->n[1] = st->n2; st->n[2] = st->n3;
        st->n_cells = (size_t)st->n1 * st->n2 * st->n3;
        const double *win[3] = {pr->grid.w1, pr->grid.w2, pr->grid.w3};
        for (int a = 0; a < 3; a++) {
            st->w[a] = dup(win[a], st->n[a] + 1);
            st->ew[a] = malloc(sizeof(double) * (st->n[a] + 1));
            /* grid_geometry_cartesian_3d.f90:130-132: ew = 3*spacing(w) */
            for (int i = 0; i <= st->n[a]; i++) st->ew[a][i] = 3.0 * spacing(st->w[a][i]);
            for (int i = 0; i < st->n[a]; i++)
                if (!(st->w[a][i + 1] - st->w[a][i] > 0.0)) {
                    snprintf(g_error, sizeof g_error, "all d%c values should be greater than zero", "xyz"[a]);
                    orc_destroy(st); return 1;
                }
        }
        st->volume = malloc(sizeof(double) * st->n_cells);
        for (int k = 0; k < st->n3; k++) for (int j = 0; j < st->n2; j++) for (int i = 0; i < st->n1; i++)
            st->volume[((size_t)k * st->n2 + j) * st->n1 + i] =
                (st->w[0][i + 1] - st->w[0][i]) * (st->w[1][j + 1] - st->w[1][j]) * (st->w[2][k + 1] - st->w[2][k]);
    } else if (st->grid_type == GRID_SPH || st->grid_type == GRID_CYL) {
        if (polar_setup(st, &pr->grid)) { orc_destroy(st); return 1; }
    } else if (st->grid_type == GRID_OCT) {
        if (octree_setup(st, &pr->grid)) { orc_destroy(st); return 1; }
    } else if (st->grid_type == GRID_AMR) {
        if (amr_setup(st, &pr->grid)) { orc_destroy(st); return 1; }
    } else {
        if (voronoi_setup(st, &pr->grid)) { orc_destroy(st); return 1; }
    }

    st->n_dust = pr->n_dust;
    st->dust = calloc(st->n_dust ? st->n_dust : 1, sizeof(dust_t));
    for (int d = 0; d < st->n_dust; d++)
        if (dust_setup(&st->dust[d], &pr->dust[d], g_error)) { orc_destroy(st); return 1; }

    /* sources: source.f90:47-84, source_type.f90:102-322 */
    st->n_sources = pr->n_sources;
    st->src = calloc(st->n_sources ? st->n_sources : 1, sizeof(source_t));
    st->lum_pdf = calloc(st->n_sources ? st->n_sources : 1, sizeof(double));
    st->lum_cdf = calloc(st->n_sources ? st->n_sources : 1, sizeof(double));
    st->energy_total = 0.0;
    for (int i = 0; i < st->n_sources; i++) {
        const orc_source_desc *s = &pr->sources[i];
        source_t *t = &st->src[i];
        t->type = s->type; t->spectrum_type = s->spectrum_type; t->peeloff = s->peeloff; t->limb_darkening = s->limb_darkening;
        if (s->type == 2) st->any_intersect = 1;     /* s%intersect = .true., source_type.f90:148 */
        if (s->type == 7) {   /* plane_parallel :239-256 */
            double th = s->direction[0] * PI / 180.0, ph = s->direction[1] * PI / 180.0;
            t->direction.cost = cos(th); t->direction.sint = sin(th); t->direction.cosp = cos(ph); t->direction.sinp = sin(ph);
            if (s->peeloff) { snprintf(g_error, sizeof g_error, "plane parallel sources cannot be peeled off (source_emit_peeloff has no case for them)"); orc_destroy(st); return 1; }
        }
        if (s->type == 8) {   /* point_collection :258-277: luminosity = sum, set_pdf(collection_pdf, luminosities) */
            if (s->n_points < 1 || !s->points || !s->point_lum) { snprintf(g_error, sizeof g_error, "point source collection needs positions and luminosities"); orc_destroy(st); return 1; }
            t->n_points = s->n_points;
            t->points = dup(s->points, 3 * (size_t)s->n_points);
            t->point_cdf = malloc(sizeof(double) * s->n_points);
            double tot = 0.0, c = 0.0;
            for (int k = 0; k < s->n_points; k++) tot += s->point_lum[k];
            for (int k = 0; k < s->n_points; k++) { c += s->point_lum[k] / tot; t->point_cdf[k] = c; }
            for (int k = 0; k < s->n_points; k++) t->point_cdf[k] /= c;
            t->luminosity = tot;
        }
        if (s->type == 4) {   /* map :190-199: set_pdf(luminosity_map, map) over all cells */
            if (!s->map) { snprintf(g_error, sizeof g_error, "map source needs a luminosity map"); orc_destroy(st); return 1; }
            t->map_cdf = malloc(sizeof(double) * st->n_cells);
            double tot = 0.0, c = 0.0;
            for (size_t k = 0; k < st->n_cells; k++) tot += s->map[k];
            if (!(tot > 0.0)) { snprintf(g_error, sizeof g_error, "luminosity map is zero everywhere"); orc_destroy(st); return 1; }
            for (size_t k = 0; k < st->n_cells; k++) { c += s->map[k] / tot; t->map_cdf[k] = c; }
            for (size_t k = 0; k < st->n_cells; k++) t->map_cdf[k] /= c;
        }
        if (s->n_spots > 0) {   /* source_type.f90:150-188 */
            if (s->type != 2 || !s->spots) { snprintf(g_error, sizeof g_error, "only spherical sources can have spots"); orc_destroy(st); return 1; }
            const int ns = s->n_spots;
            t->n_spots = ns;
            t->spot_cdf = malloc(sizeof(double) * (ns + 1)); t->spot_a = malloc(sizeof(angle_t) * ns); t->spot_cost = malloc(sizeof(double) * ns);
            t->spot_stype = malloc(sizeof(int) * ns); t->spot_T = malloc(sizeof(double) * ns); t->spot_spectrum = calloc(ns, sizeof(pdf_t));
            double tot = s->luminosity, c = 0.0;
            for (int k = 0; k < ns; k++) tot += s->spots[k].luminosity;
            for (int k = 0; k <= ns; k++) { c += (k < ns ? s->spots[k].luminosity : s->luminosity) / tot; t->spot_cdf[k] = c; }
            for (int k = 0; k <= ns; k++) t->spot_cdf[k] /= c;
            for (int k = 0; k < ns; k++) {
                const orc_spot_desc *q = &s->spots[k];
                /* angle3d_deg(lon, lat): theta = lon, phi = lat as the reference passes them */
                double th = q->longitude * PI / 180.0, ph = q->latitude * PI / 180.0;
                t->spot_a[k].cost = cos(th); t->spot_a[k].sint = sin(th); t->spot_a[k].cosp = cos(ph); t->spot_a[k].sinp = sin(ph);
                t->spot_cost[k] = cos(q->radius * PI / 180.0);
                t->spot_stype[k] = q->spectrum_type; t->spot_T[k] = q->temperature;
                if (q->spectrum_type == 1) {
                    if (pdf_set_log(&t->spot_spectrum[k], q->spec_nu, q->spec_fnu, q->n_spec, 1)) { snprintf(g_error, sizeof g_error, "source spectrum has zero integral"); orc_destroy(st); return 1; }
                } else if (q->spectrum_type != 2) { snprintf(g_error, sizeof g_error, "Spot cannot have LTE spectrum"); orc_destroy(st); return 1; }
            }
        }
        t->luminosity = s->luminosity; t->temperature = s->temperature;
        memcpy(t->position, s->position, sizeof t->position);
        if (s->type != 1 && s->type != 2 && s->type != 4 && s->type != 5 && s->type != 6 && s->type != 7 && s->type != 8) { snprintf(g_error, sizeof g_error, "unknown type in source list: %d", s->type); orc_destroy(st); return 1; }
        t->radius = s->radius; memcpy(t->box, s->box, sizeof t->box);
        if (s->type == 6) {   /* source_type.f90:233-237: face pdf ~ face areas */
            double dx = s->box[1] - s->box[0], dy = s->box[3] - s->box[2], dz = s->box[5] - s->box[4];
            double a[6] = {dy * dz, dy * dz, dz * dx, dz * dx, dx * dy, dx * dy}, c = 0.0, tot = 0.0;
            for (int k = 0; k < 6; k++) tot += a[k];
            for (int k = 0; k < 6; k++) { c += a[k] / tot; t->face_cdf[k] = c; }
            for (int k = 0; k < 6; k++) t->face_cdf[k] /= c;
        }
        if (s->spectrum_type == 1) {
            for (int k = 0; k + 1 < s->n_spec; k++)
                if (s->spec_nu[k + 1] < s->spec_nu[k]) {
                    snprintf(g_error, sizeof g_error, "spectrum frequency should be monotonically increasing");
                    orc_destroy(st); return 1;
                }
            if (pdf_set_log(&t->spectrum, s->spec_nu, s->spec_fnu, s->n_spec, 1)) {
                snprintf(g_error, sizeof g_error, "source spectrum has zero integral"); orc_destroy(st); return 1;
            }
        } else if (s->spectrum_type == 3 && s->type == 4) {
            /* 'lte': the emissivity of the dust in the emitting cell */
        } else if (s->spectrum_type != 2) {
            snprintf(g_error, sizeof g_error, "%s cannot have LTE spectrum",
                     s->type == 5 ? "External spherical source" : s->type == 6 ? "External box source" : s->type == 2 ? "Spherical source" : s->type == 7 ? "Plane parallel" : s->type == 8 ? "Point source collection" : "Point source");
            orc_destroy(st); return 1;
        }
        st->energy_total += t->luminosity;
    }
    {
        double c = 0.0;
        for (int i = 0; i < st->n_sources; i++) {
            st->lum_pdf[i] = st->src[i].luminosity / st->energy_total;
            c += st->lum_pdf[i]; st->lum_cdf[i] = c;
        }
        if (st->n_sources) for (int i = 0; i < st->n_sources; i++) st->lum_cdf[i] /= c;
    }

    size_t ntot = (size_t)st->n_dust * st->n_cells;
    st->density = dup(pr->density, ntot);
    if (st->grid_type == GRID_OCT)   /* reset density to zero in masked (refined) cells: grid_physics_3d.f90:152-160 */
        for (int d = 0; d < st->n_dust; d++)
            for (size_t ic = 0; ic < st->n_cells; ic++)
                if (st->orefined[ic]) st->density[(size_t)d * st->n_cells + ic] = 0.0;
    if (st->grid_type == GRID_AMR)   /* mask = cells not covered by a finer grid: grid_geometry_amr.f90:489-496 */
        for (int d = 0; d < st->n_dust; d++)
            for (size_t ic = 0; ic < st->n_cells; ic++)
                if (amr_covered(st, ic)) st->density[(size_t)d * st->n_cells + ic] = 0.0;
    if (st->grid_type == GRID_VOR)   /* mask = volume > 0: grid_geometry_voronoi.f90:161-173 */
        for (int d = 0; d < st->n_dust; d++)
            for (size_t ic = 0; ic < st->n_cells; ic++)
                if (!(st->volume[ic] > 0.0)) st->density[(size_t)d * st->n_cells + ic] = 0.0;
    /* geo%mask / geo%mask_map: cartesian_3d.f90:101, octree.f90:214-225, amr.f90:489-505, voronoi.f90:161-173 */
    st->mask_map = malloc(sizeof(uint32_t) * (st->n_cells ? st->n_cells : 1));
    st->n_masked = 0;
    for (size_t ic = 0; ic < st->n_cells; ic++) {
        int valid = 1;
        if (st->grid_type == GRID_OCT) valid = !st->orefined[ic];
        else if (st->grid_type == GRID_AMR) valid = !amr_covered(st, ic);
        else if (st->grid_type == GRID_VOR) valid = st->volume[ic] > 0.0;
        if (valid) st->mask_map[st->n_masked++] = (uint32_t)ic;
    }
    st->specific_energy = malloc(sizeof(double) * (ntot ? ntot : 1));
    st->specific_energy_sum = calloc(ntot ? ntot : 1, sizeof(double));
    st->jnu_var_id = calloc(ntot ? ntot : 1, sizeof(int32_t));
    st->jnu_var_frac = calloc(ntot ? ntot : 1, sizeof(double));
    /* grid_physics_3d.f90:176-253 */
    if (pr->specific_energy) {
        memcpy(st->specific_energy, pr->specific_energy, sizeof(double) * ntot);
        if (st->cfg.specific_energy_type == 1) {
            st->specific_energy_add = dup(pr->specific_energy, ntot);
            for (int d = 0; d < st->n_dust; d++)
                for (size_t ic = 0; ic < st->n_cells; ic++)
                    st->specific_energy[(size_t)d * st->n_cells + ic] = st->dust[d].minimum_specific_energy;
        }
    } else {
        if (st->cfg.specific_energy_type == 1) {
            snprintf(g_error, sizeof g_error, "cannot specify specific_energy_type since specific_energy was not given");
            orc_destroy(st); return 1;
        }
        for (int d = 0; d < st->n_dust; d++)
            for (size_t ic = 0; ic < st->n_cells; ic++)
                st->specific_energy[(size_t)d * st->n_cells + ic] = st->dust[d].minimum_specific_energy;
    }
    check_energy_abs(st);
    if (st->cfg.count_photons || st->cfg.pda) st->n_photons = calloc(st->n_cells ? st->n_cells : 1, sizeof(int64_t));
    if (st->cfg.pda)      /* setup_rt.f90:289-300 */
        for (int d = 0; d < st->n_dust; d++)
            if (st->dust[d].version == 1) {
                snprintf(g_error, sizeof g_error, "version 1 dust files can no longer be used when PDA is computed due to a bug - to fix this, re-generate the dust file using the latest version of Hyperion");
                orc_destroy(st); return 1;
            }
    st->n_bins = st->cfg.n_spectrum_bins > 0 ? st->cfg.n_spectrum_bins : 0;
    if (st->n_bins) {     /* grid_physics_3d.f90:124-143,215-282 */
        const int nb = st->n_bins;
        st->nu_edges = dup(st->cfg.spectrum_bin_edges, nb + 1);
        st->cfg.spectrum_bin_edges = st->nu_edges;
        for (int b = 0; b < nb; b++)
            if (!(st->nu_edges[b + 1] > st->nu_edges[b])) { snprintf(g_error, sizeof g_error, "specific_energy_spectrum_bin_edges should be strictly increasing"); orc_destroy(st); return 1; }
        st->log_nu_edges = malloc(sizeof(double) * (nb + 1));
        for (int b = 0; b <= nb; b++) st->log_nu_edges[b] = log10(st->nu_edges[b]);
        const size_t ns = (size_t)nb * (ntot ? ntot : 1);
        st->spec = calloc(ns, sizeof(double)); st->spec_sum = calloc(ns, sizeof(double));
        if (pr->specific_energy && st->cfg.specific_energy_type == 1) st->spec_add = calloc(ns, sizeof(double));   /* copy of the zeros */
        if (!pr->specific_energy || st->cfg.specific_energy_type == 1)
            for (int b = 0; b < nb; b++) for (int d = 0; d < st->n_dust; d++) for (size_t ic = 0; ic < st->n_cells; ic++)
                st->spec[((size_t)b * st->n_dust + d) * st->n_cells + ic] = st->dust[d].minimum_specific_energy;
        /* setup_j_nu_bin_fractions :326-348 with get_j_nu_bin_fractions dust_type_4elem.f90:752-778 */
        st->nj_max = 1;
        for (int d = 0; d < st->n_dust; d++) if (st->dust[d].n_jnu > st->nj_max) st->nj_max = st->dust[d].n_jnu;
        st->jnu_bin_frac = calloc((size_t)st->n_dust * st->nj_max * nb, sizeof(double));
        for (int d = 0; d < st->n_dust; d++)
            for (int iv = 0; iv < st->dust[d].n_jnu; iv++) {
                const pdf_t *q = &st->dust[d].j_nu[iv];
                double *f = st->jnu_bin_frac + ((size_t)d * st->nj_max + iv) * nb;
                for (int b = 0; b < nb; b++) f[b] = integral_loglog_range(q->x, q->pdf, q->n, st->nu_edges[b], st->nu_edges[b + 1]);
                const double norm = integral_loglog_all(q->x, q->pdf, q->n);
                if (norm > 0.0) for (int b = 0; b < nb; b++) f[b] /= norm;
            }
    }

    st->n_peeled = pr->n_peeled;
    if (st->n_peeled > 64) { snprintf(g_error, sizeof g_error, "at most 64 peeled image groups"); orc_destroy(st); return 1; }
    st->n_views_total = 0;
    for (int ig = 0; ig < st->n_peeled; ig++) { st->view_base[ig] = st->n_views_total; st->n_views_total += pr->peeled[ig].n_view; }
    st->has_binned = pr->binned != NULL;
    st->n_groups = st->n_peeled + st->has_binned;
    st->peeled = calloc(st->n_groups ? st->n_groups : 1, sizeof(peeled_t));
    if (st->has_binned) {   /* setup_rt.f90:327-331, binned_images_setup :42-56 */
        if (st->cfg.monochromatic) { snprintf(g_error, sizeof g_error, "can't use binned images in exact wavelength mode"); orc_destroy(st); return 1; }
        if (st->cfg.forced_first_interaction) { snprintf(g_error, sizeof g_error, "can't use binned images with forced first interaction"); orc_destroy(st); return 1; }
        st->n_theta = pr->n_binned_theta; st->n_phi = pr->n_binned_phi;
        if (st->n_theta < 1 || st->n_phi < 1) { snprintf(g_error, sizeof g_error, "n_theta and n_phi should be positive"); orc_destroy(st); return 1; }
        orc_peeled_desc bd = *pr->binned;
        bd.n_view = st->n_theta * st->n_phi;
        double *zeros = calloc(bd.n_view, sizeof(double));
        bd.theta = zeros; bd.phi = zeros; bd.inside_observer = 0;
        int rc = peeled_setup(st, &st->peeled[st->n_peeled], &bd);
        free(zeros);
        if (rc) { orc_destroy(st); return 1; }
    }
    for (int g = 0; g < st->n_peeled; g++)
        if (peeled_setup(st, &st->peeled[g], &pr->peeled[g])) { orc_destroy(st); return 1; }
    if (st->cfg.raytracing) raytracing_caches(st);
    *out = st;
    return 0;
}

void orc_destroy(orc_state *st)
{
    if (!st) return;
    for (int a = 0; a < 3; a++) { free(st->w[a]); free(st->ew[a]); }
    free(st->frequencies); free(st->mono_cdf);
    free(st->volume); free(st->wr2); free(st->wtanp); free(st->wtant); free(st->wtant2); free(st->wcost);
    free(st->ox); free(st->oy); free(st->oz); free(st->odx); free(st->ody); free(st->odz);
    free(st->orefined); free(st->osubcell); free(st->oparent); free(st->ochildren);
    free(st->vsite); free(st->vidx); free(st->vneigh); free(st->vseed);
    if (st->amr) { for (int g = 0; g < st->n_amr_grids; g++) { free(st->amr[g].go); for (int a = 0; a < 3; a++) free(st->amr[g].w[a]); } free(st->amr); }
    free(st->amr_cell_grid); free(st->mask_map); free(st->alpha_inv_planck); free(st->diff_coeff);
    if (st->dust) { for (int d = 0; d < st->n_dust; d++) dust_free(&st->dust[d]); free(st->dust); }
    if (st->src) {
        for (int i = 0; i < st->n_sources; i++) {
            if (st->src[i].spectrum_type == 1 && st->src[i].spectrum.x) pdf_free(&st->src[i].spectrum);
            free(st->src[i].points); free(st->src[i].point_cdf); free(st->src[i].map_cdf);
            if (st->src[i].spot_spectrum) for (int k = 0; k < st->src[i].n_spots; k++) if (st->src[i].spot_stype[k] == 1) pdf_free(&st->src[i].spot_spectrum[k]);
            free(st->src[i].spot_cdf); free(st->src[i].spot_a); free(st->src[i].spot_cost); free(st->src[i].spot_stype); free(st->src[i].spot_T); free(st->src[i].spot_spectrum);
        }
        free(st->src);
    }
    if (st->peeled) { for (int g = 0; g < st->n_groups; g++) peeled_free(&st->peeled[g]); free(st->peeled); }
    free(st->lum_pdf); free(st->lum_cdf);
    free(st->density); free(st->specific_energy); free(st->specific_energy_add);
    free(st->vbb);
    free(st->n_photons); free(st->nu_edges); free(st->log_nu_edges); free(st->spec); free(st->spec_sum); free(st->spec_add); free(st->jnu_bin_frac);
    free(st->specific_energy_sum); free(st->jnu_var_id); free(st->jnu_var_frac);
    free(st);
}

/* ------------------------------------------------------------------ */
/* Photon packet: src/core/type_photon.f90:14-73                        */
/* ------------------------------------------------------------------ */

enum { LAST_SR = 0, LAST_DS = 1, LAST_DE = 2 };

typedef struct {
    double r[3], v[3];
    angle_t a;
    double s[4];
    double nu, energy;
    int ic[3];          /* 0-based cell indices */
    int on_wall[3];     /* -1 lower wall, +1 upper wall, 0 none */
    int in_cell, killed;
    int inu;            /* monochromatic: 0-based index into the frequency table */
    int radial;         /* (r.v) > 0 at the start of the current integration: grid_propagate_3d.f90:73 */
    double chi[ORC_MAX_DUST], albedo[ORC_MAX_DUST], kappa[ORC_MAX_DUST];
    int last, last_isotropic, scattered, reprocessed, n_scat, dust_id, source_id, face_id;
    angle_t a_prev; double s_prev[4], v_prev[3];
    angle_t source_a;   /* inward normal at the emission point of an external source */
    int reabsorbed, reabsorbed_id;
    int emiss_type, emiss_var_id; double emiss_var_frac;   /* raytracing: 1/2 source spectrum, 3 dust emissivity */
    uint32_t peel_seq;  /* peel-off events of this packet so far (keys the check stream of the peel-off walks) */
} photon_t;

typedef struct {
    double *sum;        /* thread-local specific_energy_sum or NULL (noenergy) */
    int64_t *nphot;     /* thread-local n_photons or NULL */
    uint64_t *last_id;  /* thread-local last_photon_id (packet id + 1; 0 = none) */
    uint64_t cur_id;    /* id + 1 of the packet being propagated */
    double *sum_spec;   /* thread-local specific_energy_sum_spectrum or NULL */
    uint64_t killed_geo, killed_int, crossings, interactions;
    double energy_current;
    int fatal; char err[256];
    /* thread-local image accumulators (final iteration) */
    double **sed, **sed2, **img, **img2;
} acc_t;

/* update_optconsts: dust.f90:64-79 */
static int update_optconsts(const orc_state *st, photon_t *p, acc_t *acc)
{
    for (int d = 0; d < st->n_dust; d++) {
        const dust_t *du = &st->dust[d];
        if (p->nu < du->nu[0] || p->nu > du->nu[du->n_nu - 1]) {
            if (!acc->fatal) {
                acc->fatal = 1;
                snprintf(acc->err, sizeof acc->err,
                         "photon frequency (%10.4E Hz) is outside the range defined for the dust optical properties (%10.4E to %10.4E Hz)",
                         p->nu, du->nu[0], du->nu[du->n_nu - 1]);
            }
            p->killed = 1;
            return -1;
        }
        p->chi[d] = interp1d_loglog(du->nu, du->chi, du->n_nu, p->nu);
        p->albedo[d] = interp1d_loglog(du->nu, du->albedo, du->n_nu, p->nu);
        p->kappa[d] = p->chi[d] * (1.0 - p->albedo[d]);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Cartesian geometry: grid_geometry_cartesian_3d.f90                   */
/* ------------------------------------------------------------------ */

static inline size_t cell_index(const orc_state *st, const int ic[3])
{
    if (!GRID_IS_3IDX(st->grid_type)) return (size_t)ic[0];
    return ((size_t)ic[2] * st->n2 + ic[1]) * st->n1 + ic[0];
}

/* escaped_cell :267-275 */
static inline int escaped(const orc_state *st, const int ic[3])
{
    /* octree: escaped_cell grid_geometry_octree.f90:320-326 (ic == n_cells+1) */
    if (st->grid_type == GRID_SPH) return ic[0] < 0 || ic[0] >= st->n1;      /* spherical_3d.f90:483-490 */
    if (st->grid_type == GRID_CYL) return ic[0] < 0 || ic[0] >= st->n1 || ic[1] < 0 || ic[1] >= st->n2;   /* cylindrical_3d.f90:381-390 */
    if (st->grid_type != GRID_CAR) return (size_t)ic[0] == st->n_cells;
    return ic[0] < 0 || ic[0] >= st->n1 || ic[1] < 0 || ic[1] >= st->n2 || ic[2] < 0 || ic[2] >= st->n3;
}

/* ---- octree: grid_geometry_octree.f90 ---------------------------------- */

/* subcell_id :101-133 (0-based: bit0 = x, bit1 = y, bit2 = z) */
static inline int oct_subcell(const orc_state *st, int32_t id, const double r[3])
{
    return (r[0] < st->ox[id] ? 0 : 1) | (r[1] < st->oy[id] ? 0 : 2) | (r[2] < st->oz[id] ? 0 : 4);
}

/* locate_cell :135-146 */
static int32_t oct_locate(const orc_state *st, const double r[3], int32_t id)
{
    while (st->orefined[id]) id = st->ochildren[8 * (size_t)id + oct_subcell(st, id, r)];
    return id;
}

/* next_cell_int :328-347: wall 0..5 = -x,+x,-y,+y,-z,+z; opposite_cell table :53-59 */
static int32_t oct_next_cell(const orc_state *st, int32_t id, int wall, const double r[3])
{
    for (;;) {
        if (id == 0) return (int32_t)st->n_cells;
        int sub = st->osubcell[id], axis = wall >> 1, up = wall & 1;
        int bit = (sub >> axis) & 1;
        int32_t par = st->oparent[id];
        if (bit != up) {   /* the sibling on that side exists inside the parent */
            int sib = up ? (sub | (1 << axis)) : (sub & ~(1 << axis));
            return oct_locate(st, r, st->ochildren[8 * (size_t)par + sib]);
        }
        id = par;
    }
}

/* find_cell :143-166 (car) / :260-283 (oct); returns 0 if outside */
static int find_cell_polar(const orc_state *st, const double r[3], const double v[3], int ic[3]);
static int find_cell(const orc_state *st, const double r[3], const double v[3], int ic[3])
{
    if (st->grid_type == GRID_SPH || st->grid_type == GRID_CYL) return find_cell_polar(st, r, v, ic);
    if (st->grid_type == GRID_VOR) {   /* grid_geometry_voronoi.f90:196-229 */
        if (r[0] < st->vbox[0] || r[0] > st->vbox[1]) return 0;
        if (r[1] < st->vbox[2] || r[1] > st->vbox[3]) return 0;
        if (r[2] < st->vbox[4] || r[2] > st->vbox[5]) return 0;
        ic[0] = vor_nearest(st, r); ic[1] = ic[2] = 0;
        return 1;
    }
    if (st->grid_type == GRID_AMR) {
        int64_t id = amr_find_cell(st, r);
        if (id < 0) return 0;
        ic[0] = (int)id; ic[1] = ic[2] = 0;
        return 1;
    }
    if (st->grid_type == GRID_OCT) {
        if (r[0] < st->obox[0] || r[0] > st->obox[1]) return 0;
        if (r[1] < st->obox[2] || r[1] > st->obox[3]) return 0;
        if (r[2] < st->obox[4] || r[2] > st->obox[5]) return 0;
        ic[0] = oct_locate(st, r, 0); ic[1] = ic[2] = 0;
        return 1;
    }
    for (int a = 0; a < 3; a++) {
        int i = locate(st->w[a], st->n[a] + 1, r[a]);
        if (i < 0 || i >= st->n[a]) return 0;
        ic[a] = i;
    }
    return 1;
}

/* adjust_wall :168-232 */
static void adjust_wall(const orc_state *st, photon_t *p)
{
    for (int a = 0; a < 3; a++) {
        p->on_wall[a] = 0;
        const double *w = st->w[a];
        int i = p->ic[a];
        if (p->v[a] > 0.0) {
            if (p->r[a] == w[i]) p->on_wall[a] = -1;
            else if (p->r[a] == w[i + 1]) { p->on_wall[a] = -1; p->ic[a] = i + 1; }
        } else if (p->v[a] < 0.0) {
            if (p->r[a] == w[i]) { p->on_wall[a] = +1; p->ic[a] = i - 1; }
            else if (p->r[a] == w[i + 1]) p->on_wall[a] = +1;
        }
    }
}


/* ------------------------------------------------------------------ */
/* Spherical / cylindrical polar walks: grid_geometry_spherical_3d.f90, */
/* grid_geometry_cylindrical_3d.f90 (0-based cell and wall indices)     */
/* ------------------------------------------------------------------ */

/* equal_nulp :49-59 */
static inline int equal_nulp(double x, double y, int n)
{
    if (x == y) return 1;
    return fabs(x - y) <= n * spacing(x > y ? x : y);
}

/* theta and phi of a position, from the direction where the position cannot tell (:251-268) */
static inline double polar_theta(const double r[3], const double v[3], double r_sq)
{
    if (r_sq == 0.0) return atan2(sqrt(v[0] * v[0] + v[1] * v[1]), v[2]);
    return atan2(sqrt(r[0] * r[0] + r[1] * r[1]), r[2]);
}
static inline double polar_phi(const double r[3], const double v[3], double w_sq)
{
    double phi = w_sq == 0.0 ? atan2(v[1], v[0]) : atan2(r[1], r[0]);
    if (phi < 0.0) phi = phi + 2.0 * PI;
    return phi;
}

/* find_cell: spherical :226-299, cylindrical :183-236 */
static int find_cell_polar(const orc_state *st, const double r[3], const double v[3], int ic[3])
{
    double w_sq = r[0] * r[0] + r[1] * r[1];
    double phi = polar_phi(r, v, w_sq);
    int i1, i2;
    if (st->grid_type == GRID_SPH) {
        double r_sq = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];      /* p%r .dot. p%r */
        i1 = locate(st->wr2, st->n1 + 1, r_sq);
        i2 = locate(st->w[1], st->n2 + 1, polar_theta(r, v, r_sq));
    } else {
        i1 = locate(st->wr2, st->n1 + 1, w_sq);
        i2 = locate(st->w[1], st->n2 + 1, r[2]);
    }
    int i3 = locate(st->w[2], st->n3 + 1, phi);
    if (i1 < 0 || i1 >= st->n1 || i2 < 0 || i2 >= st->n2 || i3 < 0 || i3 >= st->n3) return 0;
    ic[0] = i1; ic[1] = i2; ic[2] = i3;
    return 1;
}

/* the azimuthal part shared by both adjust_wall: spherical :425-462, cylindrical :312-349 */
static void adjust_wall_phi(const orc_state *st, photon_t *p, double phi)
{
    const double *w3 = st->w[2];
    if (p->r[0] == 0.0 && p->r[1] == 0.0 && p->v[0] == 0.0 && p->v[1] == 0.0) return;   /* on all phi walls at once */
    if (equal_nulp(phi, w3[p->ic[2]], 3)) {
        double dphi = atan2(p->v[1], p->v[0]) - w3[p->ic[2]];
        if (dphi < -PI) dphi = dphi + 2.0 * PI;
        if (dphi > 0.0) p->on_wall[2] = -1;
        else { p->on_wall[2] = +1; p->ic[2]--; if (p->ic[2] == -1) p->ic[2] = st->n3 - 1; }
    } else if (equal_nulp(phi, w3[p->ic[2] + 1], 3)) {
        double dphi = atan2(p->v[1], p->v[0]) - w3[p->ic[2] + 1];
        if (dphi < -PI) dphi = dphi + 2.0 * PI;
        if (dphi > 0.0) { p->on_wall[2] = -1; p->ic[2]++; if (p->ic[2] == st->n3) p->ic[2] = 0; }
        else p->on_wall[2] = +1;
    }
}

/* adjust_wall: spherical :301-466, cylindrical :238-353 */
static void adjust_wall_polar(const orc_state *st, photon_t *p)
{
    p->on_wall[0] = p->on_wall[1] = p->on_wall[2] = 0;
    const double *r = p->r, *v = p->v;
    double w_sq = r[0] * r[0] + r[1] * r[1];
    double phi = polar_phi(r, v, w_sq);
    if (st->grid_type == GRID_CYL) {
        if (r[0] * v[0] + r[1] * v[1] >= 0.0) {
            if (equal_nulp(w_sq, st->wr2[p->ic[0]], 3)) p->on_wall[0] = -1;
            else if (equal_nulp(w_sq, st->wr2[p->ic[0] + 1], 3)) { p->on_wall[0] = -1; p->ic[0]++; }
        } else {
            if (equal_nulp(w_sq, st->wr2[p->ic[0]], 3)) { p->on_wall[0] = +1; p->ic[0]--; }
            else if (equal_nulp(w_sq, st->wr2[p->ic[0] + 1], 3)) p->on_wall[0] = +1;
        }
        const double *w2 = st->w[1];
        if (v[2] > 0.0) {
            if (equal_nulp(r[2], w2[p->ic[1]], 3)) p->on_wall[1] = -1;
            else if (equal_nulp(r[2], w2[p->ic[1] + 1], 3)) { p->on_wall[1] = -1; p->ic[1]++; }
        } else if (v[2] < 0.0) {
            if (equal_nulp(r[2], w2[p->ic[1]], 3)) { p->on_wall[1] = +1; p->ic[1]--; }
            else if (equal_nulp(r[2], w2[p->ic[1] + 1], 3)) p->on_wall[1] = +1;
        }
        adjust_wall_phi(st, p, phi);
        return;
    }
    double r_sq = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
    double theta = polar_theta(r, v, r_sq);
    /* radial wall :330-347 */
    if ((r[0] * v[0] + r[1] * v[1]) + r[2] * v[2] >= 0.0) {
        if (equal_nulp(r_sq, st->wr2[p->ic[0]], 3)) p->on_wall[0] = -1;
        else if (equal_nulp(r_sq, st->wr2[p->ic[0] + 1], 3)) { p->on_wall[0] = -1; p->ic[0]++; }
    } else {
        if (equal_nulp(r_sq, st->wr2[p->ic[0]], 3)) { p->on_wall[0] = +1; p->ic[0]--; }
        else if (equal_nulp(r_sq, st->wr2[p->ic[0] + 1], 3)) p->on_wall[0] = +1;
    }
    /* theta wall :349-423 */
    const double *w2 = st->w[1];
    if (r_sq == 0.0) {
        if (fabs(v[2]) < 1.0) {
            double theta_v = atan2(sqrt(v[0] * v[0] + v[1] * v[1]), v[2]);
            if (equal_nulp(theta_v, w2[p->ic[1]], 3)) p->on_wall[1] = -1;
            else if (equal_nulp(theta_v, w2[p->ic[1] + 1], 3)) p->on_wall[1] = +1;
        }
    } else if (p->ic[1] > 0 && equal_nulp(theta, w2[p->ic[1]], 3)) {
        if (p->ic[1] == st->midplane) {
            if (v[2] > 0.0) { p->on_wall[1] = +1; p->ic[1]--; }
            else p->on_wall[1] = -1;
        } else {
            int lhs = sqrt(w_sq) * v[2] * st->wtant[p->ic[1]] - (r[0] * v[0] + r[1] * v[1]) < 0.0;
            if (lhs == (r[2] > 0.0)) p->on_wall[1] = -1;
            else { p->on_wall[1] = +1; p->ic[1]--; }
        }
    } else if (p->ic[1] + 1 < st->n2 && equal_nulp(theta, w2[p->ic[1] + 1], 3)) {
        if (p->ic[1] + 1 == st->midplane) {
            if (v[2] > 0.0) p->on_wall[1] = +1;
            else { p->on_wall[1] = -1; p->ic[1]++; }
        } else {
            int lhs = sqrt(w_sq) * v[2] * st->wtant[p->ic[1] + 1] - (r[0] * v[0] + r[1] * v[1]) < 0.0;
            if (lhs == (r[2] > 0.0)) { p->on_wall[1] = -1; p->ic[1]++; }
            else p->on_wall[1] = +1;
        }
    }
    adjust_wall_phi(st, p, phi);
}

/* in_correct_cell: spherical :553-637, cylindrical :444-516 */
static int in_correct_cell_polar(const orc_state *st, const photon_t *p)
{
    int act[3] = {-1, -1, -1};
    int found = find_cell_polar(st, p->r, p->v, act);
    if (!found) act[0] = act[1] = act[2] = -1;     /* invalid_cell */
    const double thr = 1e-3;
    const double *r = p->r, *v = p->v;
    if (!(p->on_wall[0] || p->on_wall[1] || p->on_wall[2]))
        return act[0] == p->ic[0] && act[1] == p->ic[1] && act[2] == p->ic[2];
    int ok = 1;
    double w_sq = r[0] * r[0] + r[1] * r[1];
    double rad_sq = w_sq;           /* cylindrical: w^2; spherical: r^2 */
    const double *w1 = st->w[0], *w2 = st->w[1], *w3 = st->w[2];
    if (st->grid_type == GRID_SPH) {
        rad_sq = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        if (rad_sq == 0.0) return 1;
    }
    double phi = polar_phi(r, v, w_sq);
    if (p->on_wall[0] == -1) {
        if (w1[p->ic[0]] != sqrt(rad_sq)) ok = ok && fabs(sqrt(rad_sq) / w1[p->ic[0]] - 1.0) < thr;
    } else if (p->on_wall[0] == +1) {
        if (w1[p->ic[0] + 1] != sqrt(rad_sq)) ok = ok && fabs(sqrt(rad_sq) / w1[p->ic[0] + 1] - 1.0) < thr;
    } else ok = ok && act[0] == p->ic[0];
    if (st->grid_type == GRID_SPH) {
        double theta = polar_theta(r, v, rad_sq);
        if (p->on_wall[1] == -1) ok = ok && fabs(theta / w2[p->ic[1]] - 1.0) < thr;
        else if (p->on_wall[1] == +1) ok = ok && fabs(theta / w2[p->ic[1] + 1] - 1.0) < thr;
        else ok = ok && act[1] == p->ic[1];
    } else {
        double dz = w2[p->ic[1] + 1] - w2[p->ic[1]];
        if (p->on_wall[1] == -1) ok = ok && fabs((r[2] - w2[p->ic[1]]) / dz) < thr;
        else if (p->on_wall[1] == +1) ok = ok && fabs((r[2] - w2[p->ic[1] + 1]) / dz) < thr;
        else ok = ok && act[1] == p->ic[1];
    }
    if (p->on_wall[2] != 0) {
        double dphi = phi - w3[p->ic[2] + (p->on_wall[2] == +1 ? 1 : 0)];
        if (dphi > PI) dphi = dphi - 2.0 * PI;
        if (dphi < -PI) dphi = dphi + 2.0 * PI;
        ok = ok && fabs(dphi / (w3[p->ic[2] + 1] - w3[p->ic[2]])) < thr;
    } else ok = ok && act[2] == p->ic[2];
    return ok;
}

/* place_in_cell :234-253 */
static void place_in_cell(const orc_state *st, photon_t *p, acc_t *acc)
{
    if (!find_cell(st, p->r, p->v, p->ic)) {
        acc->killed_geo++; p->killed = 1;
    } else {
        p->in_cell = 1;
        if (st->grid_type == GRID_CAR) adjust_wall(st, p);   /* octree place_in_cell :285-296 has none */
        else if (st->grid_type == GRID_SPH || st->grid_type == GRID_CYL) adjust_wall_polar(st, p);
    }
}

/* in_correct_cell :330-381 */
static int in_correct_cell(const orc_state *st, const photon_t *p)
{
    int act[3];
    if (st->grid_type == GRID_SPH || st->grid_type == GRID_CYL) return in_correct_cell_polar(st, p);
    int found = find_cell(st, p->r, p->v, act);
    const double thr = 1e-3;
    int on_wall = p->on_wall[0] || p->on_wall[1] || p->on_wall[2];
    if (st->grid_type == GRID_VOR) {   /* :274-283: the cell must be one of the two nearest sites */
        int32_t n1 = vor_nearest_from(st, p->r, p->ic[0]);
        if (n1 == p->ic[0]) return 1;
        int32_t n2 = -1; double d2 = DBL_MAX;
        for (int32_t k = st->vidx[n1]; k < st->vidx[n1 + 1]; k++) {
            int32_t nb = st->vneigh[k];
            if (nb < 0) continue;
            double d = vdist2(st, nb, p->r);
            if (d < d2) { d2 = d; n2 = nb; }
        }
        return n2 == p->ic[0];
    }
    if (st->grid_type == GRID_AMR) {   /* grid_geometry_amr.f90:677-726: position within the packet's own grid */
        const amr_grid *g; int ci[3];
        amr_cell_coords(st, (size_t)p->ic[0], &g, ci);
        int ok = 1;
        for (int a = 0; a < 3; a++) {
            int actual = amr_ipos(g->lo[a], g->hi[a], p->r[a], g->n[a]) - 1;
            if (on_wall && p->on_wall[a] == -1) {
                double f = (p->r[a] - g->w[a][ci[a]]) / (g->w[a][ci[a] + 1] - g->w[a][ci[a]]);
                ok = ok && fabs(f) < thr;
            } else if (on_wall && p->on_wall[a] == +1) {
                double f = (p->r[a] - g->w[a][ci[a] + 1]) / (g->w[a][ci[a] + 1] - g->w[a][ci[a]]);
                ok = ok && fabs(f) < thr;
            } else ok = ok && actual == ci[a];
        }
        return ok;
    }
    if (st->grid_type == GRID_OCT) {   /* grid_geometry_octree.f90:366-392 */
        int32_t id = p->ic[0];
        if (on_wall) {
            double f1 = fabs(p->r[0] - st->ox[id]) / st->odx[id];
            double f2 = fabs(p->r[1] - st->oy[id]) / st->ody[id];
            double f3 = fabs(p->r[2] - st->oz[id]) / st->odz[id];
            double frac = 0.0; int ok = 0;
            if (abs(p->on_wall[0]) == 1) { frac = f1 - 1.0; ok = f2 < 1.0 && f3 < 1.0; }
            if (abs(p->on_wall[1]) == 1) { frac = f2 - 1.0; ok = f1 < 1.0 && f3 < 1.0; }
            if (abs(p->on_wall[2]) == 1) { frac = f3 - 1.0; ok = f1 < 1.0 && f2 < 1.0; }
            return fabs(frac) < thr && ok;
        }
        return found && act[0] == id;
    }
    if (on_wall) {
        int ok = 1;
        for (int a = 0; a < 3; a++) {
            const double *w = st->w[a];
            int i = p->ic[a];
            if (p->on_wall[a] == -1) {
                double f = (p->r[a] - w[i]) / (w[i + 1] - w[i]);
                ok = ok && fabs(f) < thr;
            } else if (p->on_wall[a] == +1) {
                double f = (p->r[a] - w[i + 1]) / (w[i + 1] - w[i]);
                ok = ok && fabs(f) < thr;
            } else {
                ok = ok && found && act[a] == i;
            }
        }
        return ok;
    }
    return found && act[0] == p->ic[0] && act[1] == p->ic[1] && act[2] == p->ic[2];
}

/* find_wall :424-521 with insert_t :482-512.  Returns 0 if no wall. */
static int find_wall_oct(const orc_state *st, const photon_t *p, double *tnearest, int id_min[3])
{
    /* grid_geometry_octree.f90:438-537 */
    int32_t id = p->ic[0];
    const double c[3] = {st->ox[id], st->oy[id], st->oz[id]}, h[3] = {st->odx[id], st->ody[id], st->odz[id]};
    double t[3]; int pos[3];
    for (int a = 0; a < 3; a++) {
        pos[a] = p->v[a] > 0.0;
        if (pos[a]) t[a] = (c[a] + h[a] - p->r[a]) / p->v[a];
        else if (p->v[a] < 0.0) t[a] = (c[a] - h[a] - p->r[a]) / p->v[a];
        else t[a] = DBL_MAX;
    }
    id_min[0] = id_min[1] = id_min[2] = 0;
    double tmin; int a;
    if (t[0] < t[2]) { if (t[0] < t[1]) a = 0; else a = 1; }
    else { if (t[2] < t[1]) a = 2; else a = 1; }
    tmin = t[a]; id_min[a] = pos[a] ? +1 : -1;
    if (tmin < 0.0) {
        if (tmin > -10.0 * st->oct_eps) tmin = 0.0;
        else { id_min[0] = id_min[1] = id_min[2] = 0; }
    }
    *tnearest = tmin;
    return id_min[0] || id_min[1] || id_min[2];
}

/* find_wall: grid_geometry_voronoi.f90:322-402.  id_min = (next cell + 1, current cell + 1, 0) */
static int find_wall_vor(const orc_state *st, const photon_t *p, double *tnearest, int id_min[3])
{
    int32_t ic = p->ic[0];
    const double *si = st->vsite + 3 * (size_t)ic;
    int32_t prev = -p->on_wall[1] - 1;      /* cell just left (on_wall_id%w2), -1 if none */
    double tmin = DBL_MAX; int32_t imin = -1; int found = 0;
    for (int32_t k = st->vidx[ic]; k < st->vidx[ic + 1]; k++) {
        int32_t nb = st->vneigh[k];
        double t;
        if (nb < 0) {
            int w = -nb - 1, a = w >> 1, up = w & 1;     /* 0..5 = xmin,xmax,ymin,ymax,zmin,zmax */
            if (up ? !(p->v[a] > 0.0) : !(p->v[a] < 0.0)) continue;
            t = (st->vbox[w] - p->r[a]) / p->v[a];
            if (t > 0.0 && t < tmin) { tmin = t; imin = (int32_t)st->n_cells; found = 1; }
            continue;
        }
        if (nb == prev) continue;
        const double *so = st->vsite + 3 * (size_t)nb;
        double n[3] = {so[0] - si[0], so[1] - si[1], so[2] - si[2]};
        double m[3] = {0.5 * (so[0] + si[0]), 0.5 * (so[1] + si[1]), 0.5 * (so[2] + si[2])};
        t = (n[0] * (m[0] - p->r[0]) + n[1] * (m[1] - p->r[1]) + n[2] * (m[2] - p->r[2])) /
            (n[0] * p->v[0] + n[1] * p->v[1] + n[2] * p->v[2]);
        if (t > 0.0 && t < tmin) { tmin = t; imin = nb; found = 1; }
    }
    *tnearest = tmin;
    id_min[0] = found ? imin + 1 : 0; id_min[1] = ic + 1; id_min[2] = 0;
    return found;
}

/* find_wall: grid_geometry_amr.f90:775-871.  Returns 0 if no wall, -1 on the reference's fatal "negative t". */
static int find_wall_amr(const orc_state *st, const photon_t *p, double *tnearest, int id_min[3])
{
    const amr_grid *g; int ci[3];
    amr_cell_coords(st, (size_t)p->ic[0], &g, ci);
    double t[3]; int pos[3];
    for (int a = 0; a < 3; a++) {
        pos[a] = p->v[a] > 0.0;
        if (pos[a]) t[a] = (g->w[a][ci[a] + 1] - p->r[a]) / p->v[a];
        else if (p->v[a] < 0.0) t[a] = (g->w[a][ci[a]] - p->r[a]) / p->v[a];
        else t[a] = DBL_MAX;
    }
    id_min[0] = id_min[1] = id_min[2] = 0;
    if (fmin(t[0], fmin(t[1], t[2])) < 0.0) return -1;
    int a;
    if (t[0] < t[2]) { if (t[0] < t[1]) a = 0; else a = 1; }
    else { if (t[2] < t[1]) a = 2; else a = 1; }
    *tnearest = t[a]; id_min[a] = pos[a] ? +1 : -1;
    return 1;
}

/* next_cell_int :599-655 */
static int64_t amr_next_cell(const orc_state *st, size_t ic, int axis, int dir, const double r_in[3])
{
    const amr_grid *g; int ci[3];
    amr_cell_coords(st, ic, &g, ci);
    int i[3] = {ci[0] + 1, ci[1] + 1, ci[2] + 1};      /* 1-based like the goto tables */
    i[axis] += dir;
    int32_t go = g->go[amr_go_index(g, i[0], i[1], i[2])];
    if (go == 0) {
        for (int a = 0; a < 3; a++) if (i[a] == 0 || i[a] == g->n[a] + 1) return (int64_t)st->n_cells;   /* outside_cell */
        return (int64_t)(g->start + ((size_t)(i[2] - 1) * g->n[1] + (i[1] - 1)) * g->n[0] + (i[0] - 1));
    }
    double r[3] = {r_in[0], r_in[1], r_in[2]};
    r[axis] = dir > 0 ? r[axis] + st->amr_eps : r[axis] - st->amr_eps;
    return amr_find_position(st, r, go - 1);
}


/* fortranlib quadratic(a, b, c, x1, x2): real roots of a x^2 + b x + c = 0, cancellation-free
 * (q = -(b + sign(b) sqrt(delta))/2, x1 = q/a, x2 = c/q); no real root -> -huge.  Restated from
 * the library's published behaviour like quadratic_pascal_reduced below (parity unpinned at
 * source level; pinned by the reference's spherical goldens). */
static void quadratic_full(double a, double b, double c, double *x1, double *x2)
{
    double delta = b * b - 4.0 * a * c;
    if (delta < 0.0) { *x1 = *x2 = -DBL_MAX; return; }
    double q = b >= 0.0 ? -0.5 * (b + sqrt(delta)) : -0.5 * (b - sqrt(delta));
    *x1 = q / a;
    *x2 = q != 0.0 ? c / q : 0.0;
}
static void quadratic_pascal_reduced(double b, double c, double *t1, double *t2);

typedef struct { double tmin, emin; int imin[3], iext[3]; } wallsel_t;

/* insert_t: spherical_3d.f90:1080-1112 (same in cylindrical_3d.f90) */
static inline void insert_t(wallsel_t *ws, double t, int iw, int i, double e)
{
    if (t > 0.0) {
        double emax = e > ws->emin ? e : ws->emin;
        if (t < ws->tmin - emax) {
            ws->tmin = t; ws->imin[0] = ws->imin[1] = ws->imin[2] = 0; ws->emin = emax; ws->imin[iw] = i;
        } else if (t < ws->tmin + emax) {
            ws->emin = emax; ws->imin[iw] = i;
        }
    }
}

/* both roots of a curved wall unless the packet sits on it, then the one that is not the wall itself */
static inline void insert_pair(wallsel_t *ws, double t1, double t2, int on_it, int iw, int i, double e)
{
    if (on_it) insert_t(ws, fabs(t1) < fabs(t2) ? t2 : t1, iw, i, e);
    else { insert_t(ws, t1, iw, i, e); insert_t(ws, t2, iw, i, e); }
}

/* the phi walls: spherical_3d.f90:985-1064 = cylindrical_3d.f90:691-767 */
static void find_wall_phi(const orc_state *st, const photon_t *p, double r2_xy, wallsel_t *ws)
{
    if (st->n_dim != 3) return;
    const double *w3 = st->w[2], *r = p->r, *v = p->v;
    const int i3 = p->ic[2];
    double dphi = 0.0;
    if (p->on_wall[2] == -1) {
        dphi = atan2(v[1], v[0]) - w3[i3];
        if (dphi > PI) dphi = dphi - 2.0 * PI;
        if (dphi < -PI) dphi = dphi + 2.0 * PI;
    }
    if (p->on_wall[2] == +1) {
        dphi = atan2(v[1], v[0]) - w3[i3 + 1];
        if (dphi > PI) dphi = dphi - 2.0 * PI;
        if (dphi < -PI) dphi = dphi + 2.0 * PI;
    }
    if (p->on_wall[2] == +1 && fabs(dphi) < st->ew[2][i3 + 1]) ws->iext[2] = +1;
    else if (p->on_wall[2] == -1 && fabs(dphi) < st->ew[2][i3]) ws->iext[2] = -1;
    else if (r2_xy > 0.0) {
        for (int side = 0; side < 2; side++) {
            int dir = side ? +1 : -1;
            if (p->on_wall[2] == dir) continue;
            double tp = st->wtanp[i3 + side];
            double t = -(tp * r[0] - r[1]) / (tp * v[0] - v[1]);
            double x_i = r[0] + v[0] * t, y_i = r[1] + v[1] * t;
            double d = fabs(atan2(y_i, x_i) - w3[i3 + side]);
            if (d > PI) d = fabs(d - 2.0 * PI);
            if (d < 0.5 * PI) insert_t(ws, t, 2, dir, 0.0);
        }
    }
}

/* one cone wall (side 0 = lower, 1 = upper): spherical_3d.f90:832-980 */
static void find_wall_cone(const orc_state *st, const photon_t *p, int side, double v2_xy, double v2_z,
                           double rv_xy, double rv_z, double r2_xy, double r2_z, wallsel_t *ws)
{
    const int iw = p->ic[1] + side, dir = side ? +1 : -1;
    const double *r = p->r, *v = p->v;
    const double e = st->ew[1][iw], tt = st->wtant[iw], tt2 = st->wtant2[iw];
    if (p->on_wall[1] == dir && equal_nulp(tt, sqrt(v2_xy) / v[2], 10)
        && equal_nulp(sqrt(r2_xy) * v[2] * tt, rv_xy, 10)) { ws->iext[1] = dir; return; }   /* moving along the wall */
    if (iw == st->midplane && v[2] != 0.0) {
        if (p->on_wall[1] != dir) insert_t(ws, -r[2] / v[2], 1, dir, e);
        return;
    }
    double pA = v2_xy - v2_z * tt2;
    double pB = rv_xy - rv_z * tt2; pB = pB + pB;
    double pC = r2_xy - r2_z * tt2;
    if (fabs(pA) > 0.0) {
        double t1, t2;
        quadratic_full(pA, pB, pC, &t1, &t2);
        double z1 = r[2] + v[2] * t1;
        if ((z1 > 0.0) != (tt > 0.0)) t1 = DBL_MAX;
        double z2 = r[2] + v[2] * t2;
        if ((z2 > 0.0) != (tt > 0.0)) t2 = DBL_MAX;
        insert_pair(ws, t1, t2, p->on_wall[1] == dir, 1, dir, e);
    } else if (fabs(pB) > 0.0) {
        if (p->on_wall[1] != dir) insert_t(ws, -pC / pB, 1, dir, e);
    }
}

/* find_wall: spherical_3d.f90:741-1073 */
static int find_wall_sph(const orc_state *st, const photon_t *p, double *tnearest, int id_min[3])
{
    wallsel_t ws = {DBL_MAX, 0.0, {0, 0, 0}, {0, 0, 0}};
    const double *r = p->r, *v = p->v;
    double v2_xy = v[0] * v[0] + v[1] * v[1], v2_z = v[2] * v[2];
    double rv_xy = r[0] * v[0] + r[1] * v[1], rv_z = r[2] * v[2];
    double r2_xy = r[0] * r[0] + r[1] * r[1], r2_z = r[2] * r[2];
    double pB = rv_xy + rv_z; pB = pB + pB;
    double pC = r2_xy + r2_z, t1, t2;
    const int i1 = p->ic[0];
    if (!p->radial) {
        quadratic_pascal_reduced(pB, pC - st->wr2[i1], &t1, &t2);
        insert_pair(&ws, t1, t2, p->on_wall[0] == -1, 0, -1, st->ew[0][i1]);
    }
    quadratic_pascal_reduced(pB, pC - st->wr2[i1 + 1], &t1, &t2);
    insert_pair(&ws, t1, t2, p->on_wall[0] == +1, 0, +1, st->ew[0][i1 + 1]);
    if (p->ic[1] > 0) find_wall_cone(st, p, 0, v2_xy, v2_z, rv_xy, rv_z, r2_xy, r2_z, &ws);
    if (p->ic[1] < st->n2 - 1) find_wall_cone(st, p, 1, v2_xy, v2_z, rv_xy, rv_z, r2_xy, r2_z, &ws);
    find_wall_phi(st, p, r2_xy, &ws);
    *tnearest = ws.tmin;
    for (int a = 0; a < 3; a++) id_min[a] = ws.imin[a] + ws.iext[a];      /* find_next_wall :1114-1121 */
    return id_min[0] || id_min[1] || id_min[2];
}

/* find_wall: cylindrical_3d.f90:593-771 */
static int find_wall_cyl(const orc_state *st, const photon_t *p, double *tnearest, int id_min[3])
{
    wallsel_t ws = {DBL_MAX, 0.0, {0, 0, 0}, {0, 0, 0}};
    const double *r = p->r, *v = p->v;
    double v2_xy = v[0] * v[0] + v[1] * v[1];
    double rv_xy = r[0] * v[0] + r[1] * v[1];
    double r2_xy = r[0] * r[0] + r[1] * r[1];
    double pB = rv_xy / v2_xy; pB = pB + pB;
    double pC = r2_xy / v2_xy, t1, t2;
    const int i1 = p->ic[0], i2 = p->ic[1];
    quadratic_pascal_reduced(pB, pC - st->wr2[i1] / v2_xy, &t1, &t2);
    insert_pair(&ws, t1, t2, p->on_wall[0] == -1, 0, -1, st->ew[0][i1]);
    quadratic_pascal_reduced(pB, pC - st->wr2[i1 + 1] / v2_xy, &t1, &t2);
    insert_pair(&ws, t1, t2, p->on_wall[0] == +1, 0, +1, st->ew[0][i1 + 1]);
    if (p->on_wall[1] != -1) insert_t(&ws, (st->w[1][i2] - r[2]) / v[2], 1, -1, 0.0);
    if (p->on_wall[1] != +1) insert_t(&ws, (st->w[1][i2 + 1] - r[2]) / v[2], 1, +1, 0.0);
    find_wall_phi(st, p, r2_xy, &ws);
    *tnearest = ws.tmin;
    for (int a = 0; a < 3; a++) id_min[a] = ws.imin[a] + ws.iext[a];
    return id_min[0] || id_min[1] || id_min[2];
}

/* p%icell = next_cell(p%icell, id_min, intersection=p%r); p%on_wall_id = opposite_wall(id_min) */
static void advance_cell(const orc_state *st, photon_t *p, const int id_min[3])
{
    if (st->grid_type == GRID_AMR) {   /* next_cell_wall_id :657-675 */
        int axis = id_min[0] ? 0 : id_min[1] ? 1 : 2;
        int64_t nx = amr_next_cell(st, (size_t)p->ic[0], axis, id_min[axis], p->r);
        p->ic[0] = (int)nx;            /* -1 = invalid_cell: the caller kills the packet */
        for (int a = 0; a < 3; a++) p->on_wall[a] = -id_min[a];
        return;
    }
    if (st->grid_type == GRID_VOR) {   /* next_cell_wall_id :266-272: wall id = cell id */
        p->ic[0] = id_min[0] - 1;
        for (int a = 0; a < 3; a++) p->on_wall[a] = -id_min[a];
        return;
    }
    if (st->grid_type == GRID_OCT) {
        /* next_cell_wall_id :349-364: the first non-zero component decides */
        int wall = id_min[0] ? (id_min[0] > 0 ? 1 : 0) : id_min[1] ? (id_min[1] > 0 ? 3 : 2) : (id_min[2] > 0 ? 5 : 4);
        p->ic[0] = oct_next_cell(st, p->ic[0], wall, p->r);
        for (int a = 0; a < 3; a++) p->on_wall[a] = -id_min[a];
        return;
    }
    for (int a = 0; a < 3; a++) { p->ic[a] += id_min[a]; p->on_wall[a] = -id_min[a]; }
    if (st->grid_type != GRID_CAR) {   /* phi is periodic: spherical_3d.f90:540-546, cylindrical_3d.f90:434-440 */
        if (p->ic[2] == -1) p->ic[2] = st->n3 - 1;
        else if (p->ic[2] == st->n3) p->ic[2] = 0;
    }
}

static int find_wall(const orc_state *st, const photon_t *p, double *tnearest, int id_min[3])
{
    if (st->grid_type == GRID_OCT) return find_wall_oct(st, p, tnearest, id_min);
    if (st->grid_type == GRID_VOR) return find_wall_vor(st, p, tnearest, id_min);
    if (st->grid_type == GRID_AMR) return find_wall_amr(st, p, tnearest, id_min);
    if (st->grid_type == GRID_SPH) return find_wall_sph(st, p, tnearest, id_min);
    if (st->grid_type == GRID_CYL) return find_wall_cyl(st, p, tnearest, id_min);
    double tmin = DBL_MAX, emin = 0.0;
    int imin[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++) {
        const double *w = st->w[a], *ew = st->ew[a];
        int i = p->ic[a];
        for (int side = 0; side < 2; side++) {
            int dir = side ? +1 : -1;
            if (p->on_wall[a] == dir) continue;
            int iw = i + side;
            double t = (w[iw] - p->r[a]) / p->v[a];
            double e = ew[iw];
            if (t > 0.0) {
                double emax = e > emin ? e : emin;
                if (t < tmin - emax) {
                    tmin = t; imin[0] = imin[1] = imin[2] = 0; emin = emax; imin[a] = dir;
                } else if (t < tmin + emax) {
                    emin = emax; imin[a] = dir;
                }
            }
        }
    }
    *tnearest = tmin;
    id_min[0] = imin[0]; id_min[1] = imin[1]; id_min[2] = imin[2];
    return imin[0] || imin[1] || imin[2];
}

/* ------------------------------------------------------------------ */
/* grid_integrate: grid_propagate_3d.f90:35-234 (deposit != NULL) and
 * grid_integrate_noenergy :237-375 (deposit == NULL)                   */
/* ------------------------------------------------------------------ */

static void find_nearest_source(const orc_state *st, const double r[3], const double v[3], double *t_source, int *source_id);

/* find_wall of the AMR grid stops the reference with error("find_wall","negative t") */
static void amr_negative_t(acc_t *acc)
{
    if (!acc->fatal) { acc->fatal = 1; snprintf(acc->err, sizeof acc->err, "negative t"); }
}

static void grid_integrate(const orc_state *st, photon_t *p, double tau_required,
                           rng_t *g, acc_t *acc, double *deposit)
{
    double tau_achieved = 0.0;
    p->reabsorbed = 0;
    p->radial = ((p->r[0] * p->v[0] + p->r[1] * p->v[1]) + p->r[2] * p->v[2]) > 0.0;   /* :73 */
    /* frequency bin of the packet, found once per call (:59-71): locate() - 1-based in the reference, -1 outside */
    int idx = -1;
    if (deposit && acc->sum_spec) idx = locate(st->log_nu_edges, st->n_bins + 1, log10(p->nu));
    if (escaped(st, p->ic)) return;
    if (deposit && acc->nphot) {      /* :90-95 */
        size_t c0 = cell_index(st, p->ic);
        if (acc->last_id[c0] != acc->cur_id) { acc->nphot[c0]++; acc->last_id[c0] = acc->cur_id; }
    }
    if (tau_required == 0.0) return;
    /* distance to the nearest source that can re-absorb the packet: grid_propagate_3d.f90:99-101 */
    double t_source, t_achieved = 0.0; int source_id;
    find_nearest_source(st, p->r, p->v, &t_source, &source_id);
    for (;;) {
        if (g->countdown == 0) {
            g->countdown = rng_check_gap(g, st->check_p, st->check_log1mp);
            if (!in_correct_cell(st, p)) { acc->killed_geo++; p->killed = 1; return; }
        } else g->countdown--;
        double tau_needed = tau_required - tau_achieved;
        double tmin; int id_min[3];
        int fw = find_wall(st, p, &tmin, id_min);
        if (fw < 0) { amr_negative_t(acc); p->killed = 1; return; }
        if (!fw) { acc->killed_geo++; p->killed = 1; return; }
        size_t ic = cell_index(st, p->ic);
        double chi_rho_total = 0.0;
        for (int d = 0; d < st->n_dust; d++) chi_rho_total += p->chi[d] * st->density[(size_t)d * st->n_cells + ic];
        double tau_cell = chi_rho_total * tmin;
        acc->crossings++;
        if (tau_cell < tau_needed) {
            t_achieved += tmin;
            if (t_achieved > t_source) { p->reabsorbed = 1; p->reabsorbed_id = source_id; return; }   /* :139-143 */
            for (int a = 0; a < 3; a++) p->r[a] = p->r[a] + tmin * p->v[a];
            tau_achieved += tau_cell;
            if (deposit)
                for (int d = 0; d < st->n_dust; d++)
                    if (st->density[(size_t)d * st->n_cells + ic] > 0.0) {
                        deposit[(size_t)d * st->n_cells + ic] += tmin * p->kappa[d] * p->energy;
                        if (idx >= 0) acc->sum_spec[((size_t)idx * st->n_dust + d) * st->n_cells + ic] += tmin * p->kappa[d] * p->energy;   /* :155-158 */
                    }
            advance_cell(st, p, id_min);
            if (st->grid_type == GRID_AMR && p->ic[0] < 0) { acc->killed_geo++; p->killed = 1; return; }   /* invalid_cell */
            if (escaped(st, p->ic)) return;
            if (deposit && acc->nphot) {      /* :175-180 */
                size_t c1 = cell_index(st, p->ic);
                if (acc->last_id[c1] != acc->cur_id) { acc->nphot[c1]++; acc->last_id[c1] = acc->cur_id; }
            }
        } else {
            double tact = tmin * (tau_needed / tau_cell);
            t_achieved += tact;
            if (t_achieved > t_source) { p->reabsorbed = 1; p->reabsorbed_id = source_id; return; }   /* :184-188 */
            for (int a = 0; a < 3; a++) p->r[a] = p->r[a] + tact * p->v[a];
            tau_achieved += tau_needed;
            p->on_wall[0] = p->on_wall[1] = p->on_wall[2] = 0;
            if (deposit)
                for (int d = 0; d < st->n_dust; d++)
                    if (st->density[(size_t)d * st->n_cells + ic] > 0.0) {
                        deposit[(size_t)d * st->n_cells + ic] += tact * p->kappa[d] * p->energy;
                        if (idx >= 0) acc->sum_spec[((size_t)idx * st->n_dust + d) * st->n_cells + ic] += tact * p->kappa[d] * p->energy;   /* :214-222 */
                    }
            return;
        }
    }
}

/* grid_escape_tau: grid_propagate_3d.f90:377-480 -- optical depth from the
 * packet to the grid edge (or tmax) along its direction; works on a copy. */
static double grid_escape_tau(const orc_state *st, const photon_t *p_orig, double tmax,
                              rng_t *g, acc_t *acc, int *killed)
{
    photon_t p = *p_orig;
    p.radial = ((p.r[0] * p.v[0] + p.r[1] * p.v[1]) + p.r[2] * p.v[2]) > 0.0;   /* grid_propagate_3d.f90:400,505 */
    double tau = 0.0, t_achieved = 0.0;
    *killed = 0;
    if (escaped(st, p.ic)) return 0.0;
    {   /* a source in the way: no point in going on (grid_propagate_3d.f90:414-420) */
        double t_source; int sid;
        find_nearest_source(st, p.r, p.v, &t_source, &sid);
        if (t_source < tmax) { *killed = 1; return 0.0; }
    }
    for (;;) {
        if (g->countdown == 0) {
            g->countdown = rng_check_gap(g, st->check_p, st->check_log1mp);
            if (!in_correct_cell(st, &p)) { acc->killed_geo++; *killed = 1; return tau; }
        } else g->countdown--;
        double tmin; int id_min[3];
        int fw = find_wall(st, &p, &tmin, id_min);
        if (fw < 0) { amr_negative_t(acc); *killed = 1; return tau; }
        if (!fw) { acc->killed_geo++; *killed = 1; return tau; }
        size_t ic = cell_index(st, p.ic);
        int finished = 0;
        if (t_achieved + tmin > tmax) { tmin = tmax - t_achieved; finished = 1; }
        for (int a = 0; a < 3; a++) p.r[a] = p.r[a] + tmin * p.v[a];
        t_achieved += tmin;
        for (int d = 0; d < st->n_dust; d++) tau += p.chi[d] * st->density[(size_t)d * st->n_cells + ic] * tmin;
        acc->crossings++;
        if (finished) return tau;
        advance_cell(st, &p, id_min);
        if (st->grid_type == GRID_AMR && p.ic[0] < 0) { acc->killed_geo++; *killed = 1; return tau; }   /* invalid_cell */
        if (escaped(st, p.ic)) return tau;
    }
}

/* ------------------------------------------------------------------ */
/* Sources: source.f90:100-179, source_type.f90:398-564                 */
/* ------------------------------------------------------------------ */

/* fortranlib random_planck_frequency: Carter & Cashwell (1975) sampling of
 * x = h nu / k T from the Planck function (5 uniforms). */
static double random_planck_frequency(rng_t *g, double T)
{
    double xi0 = rng_uniform(g);
    double target = xi0 * (PI * PI * PI * PI / 90.0);
    double sum = 0.0; int m = 0;
    do { m++; sum += 1.0 / ((double)m * m * m * m); } while (sum < target && m < 1000);
    double x1 = rng_uniform(g), x2 = rng_uniform(g), x3 = rng_uniform(g), x4 = rng_uniform(g);
    double x = -log((1.0 - x1) * (1.0 - x2) * (1.0 - x3) * (1.0 - x4)) / (double)m;
    return x * K_CGS * T / H_CGS;
}

/* inward normals of the six box faces as written in source_type.f90:864-899
 * (note the negative sin(theta) used for the "max" faces) */
static void box_face_normal(int face, angle_t *a)
{
    static const double tab[6][4] = {{0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, 1}, {0, -1, 0, 1}, {1, 0, 1, 0}, {-1, 0, 1, 0}};
    a->cost = tab[face][0]; a->sint = tab[face][1]; a->cosp = tab[face][2]; a->sinp = tab[face][3];
}

static int random_position_cell(const orc_state *st, size_t ic, photon_t *p, rng_t *g);
static double dust_sample_j_nu(const dust_t *d, int id, double frac, double xi);
static double dust_sample_emit_probability(const dust_t *d, int id, double frac, double nu);
static int emit_from_nu(const orc_state *st, photon_t *p, rng_t *g, acc_t *acc, int reemit_id, double reemit_energy, int inu);
static int emit_from(const orc_state *st, photon_t *p, rng_t *g, acc_t *acc, int reemit_id, double reemit_energy) { return emit_from_nu(st, p, g, acc, reemit_id, reemit_energy, -1); }
static int emit(const orc_state *st, photon_t *p, rng_t *g, acc_t *acc) { return emit_from(st, p, g, acc, -1, 0.0); }

/* ran_mu_limb(a, b): source_type.f90:982-1086 -- mu from the pdf a mu^2 + b mu by the real root of the cubic */
static double ran_mu_limb(double a, double b, double xi_in)
{
    double s = a * (1.0 / 3.0), t = b * 0.5;
    double norm = s + t;
    s = s / norm; t = t / norm;
    double xi = -xi_in;
    double bb = t / s, dd = xi / s;
    const double alpha = 1.0 / 3.0, gamma = 1.0 / 27.0;
    double pp = -bb * bb * alpha * alpha;
    double q = (dd + 2.0 * bb * bb * bb * gamma) * 0.5;
    double p3 = pp * pp * pp, q2 = q * q;
    double delta = q2 + p3;
    if (delta < 0) {
        double phi = acos(-q / sqrt(fabs(p3)));
        double y = 2 * sqrt(fabs(pp)) * cos(phi * alpha);
        return y - bb * alpha;
    }
    delta = sqrt(delta);
    return cbrt(-q + delta) + cbrt(-q - delta) - bb * alpha;
}

/* quadratic_pascal_reduced(b, c, t1, t2) (fortranlib): real roots of t^2 + b t + c = 0 in the
 * cancellation-free form; no real root -> both set to -huge (never selected by the callers) */
static void quadratic_pascal_reduced(double b, double c, double *t1, double *t2)
{
    double delta = b * b - 4.0 * c;
    if (delta < 0.0) { *t1 = *t2 = -DBL_MAX; return; }
    double q = b >= 0.0 ? -0.5 * (b + sqrt(delta)) : -0.5 * (b - sqrt(delta));
    *t1 = q;
    *t2 = q != 0.0 ? c / q : 0.0;
}

/* source_distance :324-357 + find_nearest_source source.f90:206-227 */
static void find_nearest_source(const orc_state *st, const double r[3], const double v[3], double *t_source, int *source_id)
{
    *source_id = -1; *t_source = INFINITY;
    if (!st->any_intersect) return;
    for (int is = 0; is < st->n_sources; is++) {
        const source_t *s = &st->src[is];
        if (s->type != 2) continue;
        double dr[3] = {r[0] - s->position[0], r[1] - s->position[1], r[2] - s->position[2]};
        double pB = 2.0 * (dr[0] * v[0] + dr[1] * v[1] + dr[2] * v[2]);
        double pC = (dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]) - s->radius * s->radius;
        double t1, t2, dist = INFINITY;
        quadratic_pascal_reduced(pB, pC, &t1, &t2);
        const double tol = 1.e-8;
        if (t1 < dist && t1 > tol * s->radius) dist = t1;
        if (t2 < dist && t2 > tol * s->radius) dist = t2;
        if (dist < *t_source) { *t_source = dist; *source_id = is; }
    }
}

/* emit: source.f90:100-179; reemit_id >= 0 re-emits from that source with the given energy */
/* interpolate_pdf(pdf, x, bounds_error=.false., fill_value=0) of a log pdf (fortranlib type_pdf): the
 * normalised pdf interpolated in log-log */
static double pdf_interp_log(const pdf_t *q, double xv)
{
    if (!(xv >= q->x[0]) || !(xv <= q->x[q->n - 1])) return 0.0;
    return interp1d_loglog(q->x, q->pdf, q->n, xv);
}

/* normalized_B_nu: source_type.f90:1088-1096 */
static double normalized_B_nu(double nu, double T)
{
    const double a = 2.0 * H_CGS / 29979245800.0 / 29979245800.0 / 5.67051e-5 * PI, b = H_CGS / K_CGS;
    const double T4 = T * T * T * T;
    return a * nu * nu * nu / (exp(b * nu / T) - 1.0) / T4;
}

/* inu >= 0: emit(p, inu=inu), the frequency is frequencies(inu) and the energy carries the
 * probability of emission there (source_type.f90:440-468, source.f90:145-161) */
static int emit_from_nu(const orc_state *st, photon_t *p, rng_t *g, acc_t *acc, int reemit_id, double reemit_energy, int inu)
{
    const uint32_t keep_seq = reemit_id >= 0 ? p->peel_seq : 0u;      /* a re-emitted packet is still the same packet */
    memset(p, 0, sizeof(*p));
    p->peel_seq = keep_seq;
    int is = 0;
    if (st->n_sources > 1) {
        double xi = rng_uniform(g);
        if (st->cfg.sample_sources_evenly) is = (int)(xi * st->n_sources);
        else is = sample_discrete(st->lum_cdf, st->n_sources, xi);
    }
    if (reemit_id >= 0) is = reemit_id;
    p->source_id = is;
    const source_t *s = &st->src[is];
    int ispot = -1;
    if (s->type == 2) {
        /* emit_from_sphere :604-690 */
        angle_t a_coord, a_local;
        if (s->n_spots > 0) {   /* source_emit case(3) :421-427: a spot or the rest of the sphere, by luminosity */
            int k = sample_discrete(s->spot_cdf, s->n_spots + 1, rng_uniform(g));
            if (k < s->n_spots) ispot = k;
        }
        if (ispot >= 0) {       /* emit_from_sphere(spot) :632-636: rejection until the position falls inside the spot */
            for (;;) {
                random_sphere_angle(g, &a_coord);
                double n1[3], n2[3]; angle_to_vector(&a_coord, n1); angle_to_vector(&s->spot_a[ispot], n2);
                if ((n1[0] * n2[0] + n1[1] * n2[1]) + n1[2] * n2[2] > s->spot_cost[ispot]) break;
            }
        } else random_sphere_angle(g, &a_coord);
        double phi = TWOPI * rng_uniform(g);
        a_local.cosp = cos(phi); a_local.sinp = sin(phi);
        if (s->limb_darkening) a_local.cost = ran_mu_limb(1.5, 1.0, rng_uniform(g));
        else a_local.cost = sqrt(rng_uniform(g));
        a_local.sint = sqrt(1.0 - a_local.cost * a_local.cost);
        rotate_angle(&a_local, &a_coord, &p->a);
        double n[3]; angle_to_vector(&a_coord, n);
        for (int k = 0; k < 3; k++) p->r[k] = n[k] * s->radius + s->position[k];
        p->last_isotropic = 0;
        p->source_a = a_coord;
    } else if (s->type == 4) {
        /* emit_from_map :713-741: cell from the luminosity map, uniform position in it, isotropic direction */
        const double xi = rng_uniform(g);
        size_t lo = 0, hi = st->n_cells - 1;
        while (lo < hi) { size_t mid = (lo + hi) >> 1; if (xi < s->map_cdf[mid]) hi = mid; else lo = mid + 1; }
        if (random_position_cell(st, lo, p, g)) {
            if (!acc->fatal) { acc->fatal = 1; snprintf(acc->err, sizeof acc->err, "map sources are not available for this grid type"); }
            return -1;
        }
        random_sphere_angle(g, &p->a);
        p->last_isotropic = 1;
    } else if (s->type == 8) {
        /* emit_from_point_collection :570-598 */
        int k = sample_discrete(s->point_cdf, s->n_points, rng_uniform(g));
        p->r[0] = s->points[3 * k]; p->r[1] = s->points[3 * k + 1]; p->r[2] = s->points[3 * k + 2];
        random_sphere_angle(g, &p->a);
        p->last_isotropic = 1;
    } else if (s->type == 7) {
        /* emit_from_plane_parallel :935-975 */
        double rr = pow(rng_uniform(g), 0.5) * s->radius;
        double phi = 360.0 * rng_uniform(g) * PI / 180.0;      /* random_uni(phi, 0, 360) then angle3d_deg(90, phi) */
        angle_t a_local, a_final;
        a_local.cost = cos(90.0 * PI / 180.0); a_local.sint = sin(90.0 * PI / 180.0); a_local.cosp = cos(phi); a_local.sinp = sin(phi);
        rotate_angle(&a_local, &s->direction, &a_final);
        double n[3]; angle_to_vector(&a_final, n);
        for (int k = 0; k < 3; k++) p->r[k] = n[k] * rr + s->position[k];
        p->a = s->direction;
        p->last_isotropic = 0;
    } else if (s->type == 1) {
        /* emit_from_point :539-564 */
        p->r[0] = s->position[0]; p->r[1] = s->position[1]; p->r[2] = s->position[2];
        random_sphere_angle(g, &p->a);
        p->last_isotropic = 1;
    } else if (s->type == 5) {
        /* emit_from_extern_sph :748-809 */
        angle_t a_coord, a_local;
        random_sphere_angle(g, &a_coord);
        double phi = TWOPI * rng_uniform(g);
        a_local.cosp = cos(phi); a_local.sinp = sin(phi);
        a_local.cost = sqrt(rng_uniform(g));
        a_local.sint = sqrt(1.0 - a_local.cost * a_local.cost);
        rotate_angle(&a_local, &a_coord, &p->a);
        p->a.cost = -p->a.cost; p->a.cosp = -p->a.cosp; p->a.sinp = -p->a.sinp;   /* point inwards */
        double n[3]; angle_to_vector(&a_coord, n);
        for (int k = 0; k < 3; k++) p->r[k] = n[k] * s->radius + s->position[k];
        p->last_isotropic = 0;
        p->source_a = a_coord;
        p->source_a.cost = -a_coord.cost; p->source_a.cosp = -a_coord.cosp; p->source_a.sinp = -a_coord.sinp;
    } else {
        /* emit_from_extern_box :822-907 */
        int face = sample_discrete(s->face_cdf, 6, rng_uniform(g));
        angle_t a_local, a_coord;
        double phi = TWOPI * rng_uniform(g);
        a_local.cosp = cos(phi); a_local.sinp = sin(phi);
        a_local.cost = sqrt(rng_uniform(g));
        a_local.sint = sqrt(1.0 - a_local.cost * a_local.cost);
        int axis = face >> 1, up = face & 1;
        for (int k = 0; k < 3; k++) {
            if (k == axis) p->r[k] = s->box[2 * k + up];
            else p->r[k] = s->box[2 * k] + (s->box[2 * k + 1] - s->box[2 * k]) * rng_uniform(g);
        }
        box_face_normal(face, &a_coord);
        rotate_angle(&a_local, &a_coord, &p->a);
        p->last_isotropic = 0;
        p->face_id = face;
    }
    p->s[0] = 1.0; p->s[1] = p->s[2] = p->s[3] = 0.0;
    p->energy = 1.0;
    p->inu = inu;
    p->emiss_type = s->spectrum_type;
    if (s->spectrum_type == 3) {
        /* 'lte' (source_type.f90:455-459, 486-491): select_dust_specific_energy_rho (grid_physics_3d.f90:101-109) in the
         * emitting cell, then the emissivity of that dust */
        size_t ic = cell_index(st, p->ic);
        int id = 0;
        double cdf[ORC_MAX_DUST], c = 0.0;
        for (int d = 0; d < st->n_dust; d++) { c += st->specific_energy[(size_t)d * st->n_cells + ic] * st->density[(size_t)d * st->n_cells + ic]; cdf[d] = c; }
        for (int d = 0; d < st->n_dust; d++) cdf[d] /= c;
        id = sample_discrete(cdf, st->n_dust, rng_uniform(g));
        size_t k = (size_t)id * st->n_cells + ic;
        p->dust_id = id; p->emiss_var_id = st->jnu_var_id[k]; p->emiss_var_frac = st->jnu_var_frac[k];
    }
    if (ispot >= 0) {      /* the spot's own spectrum: source_type.f90:447-461, 480-492 */
        const int sty = s->spot_stype[ispot];
        if (inu >= 0) {
            p->nu = st->frequencies[inu];
            p->energy = sty == 1 ? pdf_interp_log(&s->spot_spectrum[ispot], p->nu) : normalized_B_nu(p->nu, s->spot_T[ispot]);
        } else if (sty == 1) p->nu = pdf_sample_log(&s->spot_spectrum[ispot], rng_uniform(g));
        else p->nu = random_planck_frequency(g, s->spot_T[ispot]);
    } else
    if (inu >= 0) {
        p->nu = st->frequencies[inu];
        p->energy = s->spectrum_type == 1 ? pdf_interp_log(&s->spectrum, p->nu)
                  : s->spectrum_type == 2 ? normalized_B_nu(p->nu, s->temperature)
                  : dust_sample_emit_probability(&st->dust[p->dust_id], p->emiss_var_id, p->emiss_var_frac, p->nu);
    } else if (s->spectrum_type == 1) p->nu = pdf_sample_log(&s->spectrum, rng_uniform(g));
    else if (s->spectrum_type == 2) p->nu = random_planck_frequency(g, s->temperature);
    else p->nu = dust_sample_j_nu(&st->dust[p->dust_id], p->emiss_var_id, p->emiss_var_frac, rng_uniform(g));
    angle_to_vector(&p->a, p->v);
    if (reemit_id >= 0) p->energy = reemit_energy;
    else {
        if (inu >= 0) p->energy = p->energy * st->energy_total;
        if (st->cfg.sample_sources_evenly) p->energy = p->energy * st->lum_pdf[is] * st->n_sources;
        acc->energy_current += p->energy;
    }
    if (update_optconsts(st, p, acc)) return -1;
    p->last = LAST_SR;
    p->a_prev = p->a; memcpy(p->s_prev, p->s, sizeof p->s); memcpy(p->v_prev, p->v, sizeof p->v);
    g->countdown = rng_check_gap(g, st->check_p, st->check_log1mp);
    place_in_cell(st, p, acc);
    if (p->killed) {
        if (!acc->fatal) {
            acc->fatal = 1;
            snprintf(acc->err, sizeof acc->err,
                     "photon was not emitted inside a cell - this usually indicates that a source is not inside the grid");
        }
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Dust interaction: dust_interact.f90:22-79, dust_type_4elem.f90       */
/* ------------------------------------------------------------------ */

/* dust_sample_j_nu :379-398 */
static double dust_sample_j_nu(const dust_t *d, int id, double frac, double xi)
{
    double nu1 = pdf_sample_log(&d->j_nu[id], xi);
    double nu2 = pdf_sample_log(&d->j_nu[id + 1], xi);
    return pow(10.0, log10(nu1) + frac * (log10(nu2) - log10(nu1)));
}

/* scatter_stokes :603-690 */
static void scatter_stokes(double s[4], const angle_t *a_coord, const angle_t *a_scat,
                           const angle_t *a_final, double P1, double P2, double P3, double P4)
{
    double cos_a = a_coord->cost, sin_a = a_coord->sint;
    double cos_b = a_scat->cost, sin_b = a_scat->sint;
    double cos_c = a_final->cost, sin_c = a_final->sint;
    double cos_big_b = a_coord->cosp * a_final->cosp + a_coord->sinp * a_final->sinp;
    double cos_big_c = a_scat->cosp, sin_big_c = fabs(a_scat->sinp);
    double cos_big_a, sin_big_a;
    if (sin_big_c < 10.0 * DBL_MIN && sin_c < 10.0 * DBL_MIN) {
        cos_big_a = -cos_big_b * cos_big_c;
        sin_big_a = sqrt(1.0 - cos_big_a * cos_big_a);
    } else {
        cos_big_a = (cos_a - cos_b * cos_c) / (sin_b * sin_c);
        sin_big_a = sin_big_c * sin_a / sin_c;
    }
    double cos_i2 = cos_big_a, sin_i2 = sin_big_a;
    double cos_2_i2 = 1.0 - 2.0 * sin_i2 * sin_i2;
    double sin_2_i2 = 2.0 * sin_i2 * cos_i2;
    double cos_2_alpha = 1.0 - 2.0 * a_scat->sinp * a_scat->sinp;
    double sin_2_alpha = -2.0 * a_scat->sinp * a_scat->cosp;
    double cos_2_beta = cos_2_i2, sin_2_beta;
    if (a_scat->sinp < 0.0) sin_2_beta = sin_2_i2; else sin_2_beta = -sin_2_i2;
    double I = s[0], Q = s[1], U = s[2], V = s[3];
    double RLS1 = P1 * I + P2 * (cos_2_alpha * Q + sin_2_alpha * U);
    double RLS2 = P2 * I + P1 * (cos_2_alpha * Q + sin_2_alpha * U);
    double RLS3 = -P4 * V + P3 * (-sin_2_alpha * Q + cos_2_alpha * U);
    double RLS4 = P3 * V + P4 * (-sin_2_alpha * Q + cos_2_alpha * U);
    s[0] = RLS1;
    s[1] = cos_2_beta * RLS2 + sin_2_beta * RLS3;
    s[2] = -sin_2_beta * RLS2 + cos_2_beta * RLS3;
    s[3] = RLS4;
}

/* dust_scatter :446-566 */
static void dust_scatter(const dust_t *d, double nu, angle_t *a, double s[4], rng_t *g)
{
    angle_t a_scat, a_final;
    random_sphere_angle(g, &a_scat);
    double sin_2_i1 = 2.0 * a_scat.sinp * a_scat.cosp;
    double cos_2_i1 = 1.0 - 2.0 * a_scat.sinp * a_scat.sinp;
    double c1 = s[0], c2 = cos_2_i1 * s[1] - sin_2_i1 * s[2];
    double ctot = c1 + c2;
    c1 /= ctot; c2 /= ctot;
    int nm = d->n_mu;
    int inu = locate(d->nu, d->n_nu, nu);
    double P1, P2, P3, P4;
    if (inu == -1) {
        P1 = 1.0; P2 = 0.0; P3 = 1.0; P4 = 0.0;
    } else {
        double xi = rng_uniform(g);
        const double *C1 = d->P1_cdf + (size_t)inu * nm, *C2 = d->P2_cdf + (size_t)inu * nm;
        int imin = 0, imax = nm - 1, imu = 0;
        double cdf1 = 0, cdf2 = 1;
        /* bisection of :508-536 (1-based imin=1, imax=n_mu there) */
        for (int it = 0; it < 1000000; it++) {
            imu = ((imax + 1) + (imin + 1)) / 2 - 1;
            if (d->zero_p2) { cdf1 = C1[imu]; cdf2 = C1[imu + 1]; }
            else { cdf1 = c1 * C1[imu] + c2 * C2[imu]; cdf2 = c1 * C1[imu + 1] + c2 * C2[imu + 1]; }
            if (xi > cdf2) imin = imu;
            else if (xi < cdf1) imax = imu;
            else break;
            if (imin == imax) break;
        }
        a_scat.cost = (xi - cdf1) / (cdf2 - cdf1) * (d->mu[imu + 1] - d->mu[imu]) + d->mu[imu];
        a_scat.sint = sqrt(1.0 - a_scat.cost * a_scat.cost);
        P1 = interp2d(d->mu, nm, d->nu, d->n_nu, d->P1, a_scat.cost, nu);
        P2 = interp2d(d->mu, nm, d->nu, d->n_nu, d->P2, a_scat.cost, nu);
        P3 = interp2d(d->mu, nm, d->nu, d->n_nu, d->P3, a_scat.cost, nu);
        P4 = interp2d(d->mu, nm, d->nu, d->n_nu, d->P4, a_scat.cost, nu);
    }
    rotate_angle(&a_scat, a, &a_final);
    scatter_stokes(s, a, &a_scat, &a_final, P1, P2, P3, P4);
    *a = a_final;
    double norm = 1.0 / s[0];
    s[0] = 1.0; s[1] *= norm; s[2] *= norm; s[3] *= norm;
}

/* dust_scatter_peeloff :421-444 */
static void dust_scatter_peeloff(const dust_t *d, double nu, angle_t *a, double s[4], const angle_t *a_req)
{
    angle_t a_scat;
    difference_angle(a, a_req, &a_scat);
    if (a_scat.cost < d->mu_min || a_scat.cost > d->mu_max) {
        s[0] = s[1] = s[2] = s[3] = 0.0;
    } else {
        double P1 = interp2d(d->mu, d->n_mu, d->nu, d->n_nu, d->P1, a_scat.cost, nu);
        double P2 = interp2d(d->mu, d->n_mu, d->nu, d->n_nu, d->P2, a_scat.cost, nu);
        double P3 = interp2d(d->mu, d->n_mu, d->nu, d->n_nu, d->P3, a_scat.cost, nu);
        double P4 = interp2d(d->mu, d->n_mu, d->nu, d->n_nu, d->P4, a_scat.cost, nu);
        scatter_stokes(s, a, &a_scat, a_req, P1, P2, P3, P4);
    }
    *a = *a_req;
}

/* interact: dust_interact.f90:22-79 (+ select_dust_chi_rho grid_physics_3d.f90:87-99) */
static void interact(const orc_state *st, photon_t *p, rng_t *g, acc_t *acc)
{
    size_t ic = cell_index(st, p->ic);
    int id = 0;
    if (st->n_dust > 1) {
        double cdf[ORC_MAX_DUST], c = 0.0;
        for (int d = 0; d < st->n_dust; d++) { c += p->chi[d] * st->density[(size_t)d * st->n_cells + ic]; cdf[d] = c; }
        for (int d = 0; d < st->n_dust; d++) cdf[d] /= c;
        id = sample_discrete(cdf, st->n_dust, rng_uniform(g));
    }
    double albedo = p->albedo[id];
    p->a_prev = p->a; memcpy(p->v_prev, p->v, sizeof p->v); memcpy(p->s_prev, p->s, sizeof p->s);
    double xi = rng_uniform(g);
    acc->interactions++;
    if (xi > albedo) {
        /* dust_emit :334-354 */
        size_t k = (size_t)id * st->n_cells + ic;
        p->nu = dust_sample_j_nu(&st->dust[id], st->jnu_var_id[k], st->jnu_var_frac[k], rng_uniform(g));
        p->s[0] = 1.0; p->s[1] = p->s[2] = p->s[3] = 0.0;
        random_sphere_angle(g, &p->a);
        update_optconsts(st, p, acc);
        p->scattered = 0; p->reprocessed = 1; p->last_isotropic = 1; p->dust_id = id; p->last = LAST_DE;
    } else {
        dust_scatter(&st->dust[id], p->nu, &p->a, p->s, g);
        p->scattered = 1; p->last_isotropic = 0; p->dust_id = id; p->last = LAST_DS; p->n_scat++;
    }
    angle_to_vector(&p->a, p->v);
}

static void peeloff_photon(const orc_state *st, const photon_t *p_orig, rng_t *g, acc_t *acc, int polychromatic);

/* ------------------------------------------------------------------ */
/* Modified random walk: grid_mrw_3d.f90 (Min et al. 2009)              */
/* ------------------------------------------------------------------ */

/* initialize_cumulative :157-195 */
static void mrw_initialize_cumulative(orc_state *st)
{
    const int n = 100;
    for (int i = 0; i < n; i++) {
        st->mrw_x[i] = (double)i / (double)(n - 1);
        double y = 0.0;
        if (i == n - 1) y = 0.5;
        else {
            for (long long j = 1;; j++) {
                double term = pow(st->mrw_x[i], (double)(j * j));
                if (term == 0.0) break;
                if (j % 2 == 0) y -= term; else y += term;
            }
        }
        st->mrw_y[i] = y * 2.0;
    }
}

/* prepare_mrw :29-53 + update_alpha_inv_planck grid_physics_3d.f90:397-418 */
static int prepare_mrw(orc_state *st)
{
    if (st->grid_type == GRID_VOR) { snprintf(st->err, sizeof st->err, "distance_to_closest_wall: not implemented for Voronoi grid"); return 1; }
    for (int d = 0; d < st->n_dust; d++)
        if (!st->dust[d].mo_kappa_planck || !st->dust[d].mo_chi_inv_planck || !st->dust[d].b_nu) {
            snprintf(st->err, sizeof st->err, "MRW needs the kappa_planck and chi_inv_planck mean opacities of every dust type"); return 1;
        }
    if (!st->alpha_inv_planck) {
        st->alpha_inv_planck = malloc(sizeof(double) * st->n_cells);
        st->diff_coeff = malloc(sizeof(double) * st->n_cells);
        mrw_initialize_cumulative(st);
    }
    for (size_t ic = 0; ic < st->n_cells; ic++) {
        double a = 0.0, tot = 0.0;
        for (int d = 0; d < st->n_dust; d++) {
            const dust_t *du = &st->dust[d];
            size_t k = (size_t)d * st->n_cells + ic;
            double c = interp1d_loglog(du->mo_e, du->mo_chi_inv_planck, du->n_e, st->specific_energy[k]);
            if (st->density[k] > 0.0) a += st->density[k] * c;
            tot += st->density[k] * c;
        }
        st->alpha_inv_planck[ic] = a;
        st->diff_coeff[ic] = 1.0 / 3.0 / tot;
    }
    return 0;
}

/* distance_to_closest_wall: cartesian_3d.f90:396-430, octree.f90:410-437, amr.f90:743-773 */
static double distance_to_closest_wall(const orc_state *st, const photon_t *p)
{
    double lo[3], hi[3];
    if (st->grid_type == GRID_SPH || st->grid_type == GRID_CYL) {   /* spherical_3d.f90:675-739, cylindrical_3d.f90:552-591 */
        const double *r = p->r;
        const int i1 = p->ic[0], i2 = p->ic[1], i3 = p->ic[2];
        double rcyl = sqrt(r[0] * r[0] + r[1] * r[1]);
        double d1, d2, d3, d4, d5 = DBL_MAX, d6 = DBL_MAX;
        if (st->grid_type == GRID_SPH) {
            double rad = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
            d1 = rad - st->w[0][i1]; d2 = st->w[0][i1 + 1] - rad;
            if (fabs(d1) < st->ew[0][i1]) d1 = 0.0;
            if (fabs(d2) < st->ew[0][i1 + 1]) d2 = 0.0;
            d3 = fabs(-rcyl + st->wtant[i2] * r[2]) / sqrt(1 + st->wtant[i2] * st->wtant[i2]);
            d4 = fabs(-rcyl + st->wtant[i2 + 1] * r[2]) / sqrt(1 + st->wtant[i2 + 1] * st->wtant[i2 + 1]);
        } else {
            d1 = rcyl - st->w[0][i1]; d2 = st->w[0][i1 + 1] - rcyl;
            d3 = r[2] - st->w[1][i2]; d4 = st->w[1][i2 + 1] - r[2];
        }
        if (st->n_dim == 3) {
            d5 = fabs(st->wtanp[i3] * r[0] - r[1]) / sqrt(st->wtanp[i3] * st->wtanp[i3] + 1.0);
            d6 = fabs(st->wtanp[i3 + 1] * r[0] - r[1]) / sqrt(st->wtanp[i3 + 1] * st->wtanp[i3 + 1] + 1.0);
        }
        double d = fmin(fmin(fmin(d1, d2), fmin(d3, d4)), fmin(d5, d6));
        return d < 0.0 ? 0.0 : d;
    }
    if (st->grid_type == GRID_CAR) {
        for (int a = 0; a < 3; a++) { lo[a] = st->w[a][p->ic[a]]; hi[a] = st->w[a][p->ic[a] + 1]; }
    } else if (st->grid_type == GRID_OCT) {
        int32_t id = p->ic[0];
        const double c[3] = {st->ox[id], st->oy[id], st->oz[id]}, h[3] = {st->odx[id], st->ody[id], st->odz[id]};
        double d = DBL_MAX;
        for (int a = 0; a < 3; a++) {
            double d1 = p->r[a] - c[a] + h[a], d2 = c[a] + h[a] - p->r[a];
            if (d1 < d) d = d1;
            if (d2 < d) d = d2;
        }
        return d < 0.0 ? 0.0 : d;
    } else {
        const amr_grid *g; int ci[3];
        amr_cell_coords(st, (size_t)p->ic[0], &g, ci);
        for (int a = 0; a < 3; a++) { lo[a] = g->w[a][ci[a]]; hi[a] = g->w[a][ci[a] + 1]; }
    }
    double d = DBL_MAX;
    for (int a = 0; a < 3; a++) {
        double d1 = p->r[a] - lo[a], d2 = hi[a] - p->r[a];
        if (d1 < d) d = d1;
        if (d2 < d) d = d2;
    }
    return d < 0.0 ? 0.0 : d;
}

/* grid_do_mrw :55-107 (deposit != NULL) and grid_do_mrw_noenergy :109-148 */
static void grid_do_mrw(const orc_state *st, photon_t *p, rng_t *g, double *deposit, double *sum_spec)
{
    size_t ic = cell_index(st, p->ic);
    double R0 = distance_to_closest_wall(st, p);
    if (deposit) {
        /* sample_cumulative :197-202: interp1d(ycdf, xcdf, xi) */
        double xi = rng_uniform(g);
        int j = locate(st->mrw_y, 100, xi);
        double y = (j < 0) ? NAN : st->mrw_x[j] + (xi - st->mrw_y[j]) / (st->mrw_y[j + 1] - st->mrw_y[j]) * (st->mrw_x[j + 1] - st->mrw_x[j]);
        double q = R0 / PI;
        double ct = -log(y) / st->diff_coeff[ic] * (q * q);     /* (R0/pi)**2. */
        for (int d = 0; d < st->n_dust; d++) {
            size_t k = (size_t)d * st->n_cells + ic;
            if (st->density[k] > 0.0) {
                const dust_t *du = &st->dust[d];
                double e = p->energy * ct * interp1d_loglog(du->mo_e, du->mo_kappa_planck, du->n_e, st->specific_energy[k]);
                deposit[k] += e;
                if (sum_spec) {     /* deposit_specific_energy_spectrum: grid_physics_3d.f90:367-395 */
                    const int iv = st->jnu_var_id[k]; const double fr = st->jnu_var_frac[k];
                    const double *f0 = st->jnu_bin_frac + ((size_t)d * st->nj_max + iv) * st->n_bins, *f1 = f0 + st->n_bins;
                    for (int b = 0; b < st->n_bins; b++)
                        sum_spec[((size_t)b * st->n_dust + d) * st->n_cells + ic] += e * ((1.0 - fr) * f0[b] + fr * f1[b]);
                }
            }
        }
    }
    /* random_sphere_vector3d + new position on the sphere of radius R0 */
    angle_t ar; random_sphere_angle(g, &ar);
    double dr[3]; angle_to_vector(&ar, dr);
    for (int a = 0; a < 3; a++) p->r[a] = p->r[a] + dr[a] * R0;
    p->a_prev = p->a; memcpy(p->v_prev, p->v, sizeof p->v); memcpy(p->s_prev, p->s, sizeof p->s);
    random_sphere_angle(g, &p->a);
    angle_to_vector(&p->a, p->v);
    int id = 0;
    if (st->n_dust > 1) {      /* select_dust_chi_rho: grid_physics_3d.f90:87-99 */
        double cdf[ORC_MAX_DUST], c = 0.0;
        for (int d = 0; d < st->n_dust; d++) { c += p->chi[d] * st->density[(size_t)d * st->n_cells + ic]; cdf[d] = c; }
        for (int d = 0; d < st->n_dust; d++) cdf[d] /= c;
        id = sample_discrete(cdf, st->n_dust, rng_uniform(g));
    }
    {   /* dust_sample_b_nu :400-419 */
        const dust_t *du = &st->dust[id];
        size_t k = (size_t)id * st->n_cells + ic;
        double xi = rng_uniform(g);
        double nu1 = pdf_sample_log(&du->b_nu[st->jnu_var_id[k]], xi), nu2 = pdf_sample_log(&du->b_nu[st->jnu_var_id[k] + 1], xi);
        p->nu = pow(10.0, log10(nu1) + st->jnu_var_frac[k] * (log10(nu2) - log10(nu1)));
    }
    /* the opacities of the packet are NOT refreshed here (the reference calls update_optconsts only
     * in emit and interact): the next grid_integrate runs with those of the previous frequency */
    p->last_isotropic = 1; p->dust_id = id; p->last = LAST_DE;
}

/* the loop of iter_lucy.f90:133-152 / iter_final.f90:165-183; returns 1 if the packet was killed */
static int mrw_steps(const orc_state *st, photon_t *p, rng_t *g, acc_t *acc, double *deposit, int peel)
{
    int64_t k;
    for (k = 1; k <= st->cfg.n_inter_mrw_max; k++) {
        size_t ic = cell_index(st, p->ic);
        if (st->alpha_inv_planck[ic] * distance_to_closest_wall(st, p) > st->cfg.mrw_gamma) {
            grid_do_mrw(st, p, g, deposit, deposit ? acc->sum_spec : NULL);
            if (peel) peeloff_photon(st, p, g, acc, 0);
        } else break;
    }
    if (k == st->cfg.n_inter_mrw_max + 1) { acc->killed_int++; p->killed = 1; return 1; }
    return 0;
}

/* ------------------------------------------------------------------ */
/* do_lucy: iter_lucy.f90:66-237                                        */
/* ------------------------------------------------------------------ */

static void lucy_packet(const orc_state *st, uint64_t id, int iter, acc_t *acc)
{
    rng_t g; photon_t p;
    rng_init(&g, st->cfg.seed, (uint32_t)iter, id);
    acc->cur_id = id + 1;
    if (emit(st, &p, &g, acc)) return;
    for (int64_t inter = 1; inter <= st->cfg.n_inter_max + 1; inter++) {
        if (st->cfg.mrw && inter > 1 && mrw_steps(st, &p, &g, acc, acc->sum, 0)) break;
        double tau = rng_exp(&g);
        grid_integrate(st, &p, tau, &g, acc, acc->sum);
        if (p.reabsorbed) {     /* iter_lucy.f90:155-185: re-emit from the absorbing source until the packet gets away */
            int64_t ia;
            for (ia = 1; ia <= st->cfg.n_reabs_max; ia++) {
                int rid = p.reabsorbed_id; double re = p.energy;
                if (emit_from(st, &p, &g, acc, rid, re)) return;
                tau = rng_exp(&g);
                grid_integrate(st, &p, tau, &g, acc, acc->sum);
                if (!p.reabsorbed) break;
            }
            if (ia == st->cfg.n_reabs_max + 1) { acc->killed_int++; p.killed = 1; break; }
        }
        if (p.killed || escaped(st, p.ic)) break;
        if (inter == st->cfg.n_inter_max + 1) { acc->killed_int++; p.killed = 1; break; }
        interact(st, &p, &g, acc);
        if (p.killed) break; /* fatal frequency-range error */
        p.killed = (st->cfg.kill_on_scatter && p.scattered) || (st->cfg.kill_on_absorb && !p.scattered);
        if (p.killed) break;
    }
}

static int resolve_threads(int n_threads)
{
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    return n_threads;
#else
    (void)n_threads; return 1;
#endif
}

int orc_lucy_accumulate(orc_state *st, uint64_t first_id, uint64_t n_local, int iter,
                        int n_threads, orc_iter_stats *stats)
{
    size_t ntot = (size_t)st->n_dust * st->n_cells;
    int nt = resolve_threads(n_threads);
    if ((uint64_t)nt > n_local && n_local > 0) nt = (int)n_local;
    if (nt < 1) nt = 1;
    memset(st->specific_energy_sum, 0, sizeof(double) * ntot);   /* grid_reset_energy */
    precompute_jnu_var(st);                                      /* iter_lucy.f90:107 */
    if (st->cfg.mrw && prepare_mrw(st)) return 1;                /* iter_lucy.f90:109-112 */
    acc_t *accs = calloc(nt, sizeof(acc_t));
    const size_t nspec = (size_t)st->n_bins * ntot;
    if (st->n_photons) memset(st->n_photons, 0, sizeof(int64_t) * st->n_cells);
    if (nspec) memset(st->spec_sum, 0, sizeof(double) * nspec);
    for (int t = 0; t < nt; t++) {
        accs[t].sum = (t == 0) ? st->specific_energy_sum : calloc(ntot ? ntot : 1, sizeof(double));
        /* every thread keeps its own n_photons / last_photon_id like an MPI rank does (mpi_routines.f90:303-310 sums them) */
        if (st->n_photons) { accs[t].nphot = calloc(st->n_cells, sizeof(int64_t)); accs[t].last_id = calloc(st->n_cells, sizeof(uint64_t)); }
        if (nspec) accs[t].sum_spec = (t == 0) ? st->spec_sum : calloc(nspec, sizeof(double));
    }
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        acc_t *acc = &accs[t];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int64_t i = 0; i < (int64_t)n_local; i++) {
            if (acc->fatal) continue;
            lucy_packet(st, first_id + (uint64_t)i, iter, acc);
        }
    }
    memset(&st->pending, 0, sizeof st->pending);
    int fatal = 0;
    /* the threads' copies summed in thread order, cell by cell (the order of the additions, and so the result, does not
     * depend on how the cells are spread over the threads that do it) */
#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (int64_t k = 0; k < (int64_t)ntot; k++) {
        double v = st->specific_energy_sum[k];
        for (int t = 1; t < nt; t++) v += accs[t].sum[k];
        st->specific_energy_sum[k] = v;
    }
    for (int t = 0; t < nt; t++) {
        if (t > 0) {
            free(accs[t].sum);
            if (nspec) { for (size_t k = 0; k < nspec; k++) st->spec_sum[k] += accs[t].sum_spec[k]; free(accs[t].sum_spec); }
        }
        if (st->n_photons) {
            for (size_t k = 0; k < st->n_cells; k++) st->n_photons[k] += accs[t].nphot[k];
            free(accs[t].nphot); free(accs[t].last_id);
        }
        st->pending.energy_current += accs[t].energy_current;
        st->pending.killed_geo += accs[t].killed_geo;
        st->pending.killed_int += accs[t].killed_int;
        st->pending.crossings += accs[t].crossings;
        st->pending.interactions += accs[t].interactions;
        if (accs[t].fatal && !fatal) { fatal = 1; snprintf(st->err, sizeof st->err, "%s", accs[t].err); }
    }
    st->pending.n_packets = n_local;
    free(accs);
    if (stats) *stats = st->pending;
    return fatal;
}

int orc_lucy_finish(orc_state *st, double *specific_energy_out, orc_iter_stats *stats)
{
    if (!(st->pending.energy_current > 0.0)) { snprintf(st->err, sizeof st->err, "no energy emitted"); return 1; }
    update_energy_abs(st, st->energy_total / st->pending.energy_current); /* iter_lucy.f90:224 */
    if (st->cfg.pda && solve_pda(st)) return 1;                            /* :227 */
    sublimate_dust(st);                                                    /* :235 */
    for (int d = 0; d < st->n_dust; d++) st->pending.energy_abs_tot[d] = st->energy_abs_tot[d];
    if (specific_energy_out) memcpy(specific_energy_out, st->specific_energy, sizeof(double) * st->n_dust * st->n_cells);
    if (stats) *stats = st->pending;
    return 0;
}

/* overwrite the pending accumulators (used by the 2-rank gloo test after the
 * all-reduce): block = [sum (n_dust*n_cells) | energy_current | killed_geo |
 * killed_int | crossings | interactions] */
int orc_set_accumulators(orc_state *st, const double *block)
{
    size_t ntot = (size_t)st->n_dust * st->n_cells;
    memcpy(st->specific_energy_sum, block, sizeof(double) * ntot);
    st->pending.energy_current = block[ntot];
    st->pending.killed_geo = (uint64_t)block[ntot + 1];
    st->pending.killed_int = (uint64_t)block[ntot + 2];
    st->pending.crossings = (uint64_t)block[ntot + 3];
    st->pending.interactions = (uint64_t)block[ntot + 4];
    /* optional extensions of the block: [8 tail doubles][n_photons as doubles (n_cells)][spectrum sums (n_bins*n_dust*n_cells)] */
    const double *q = block + ntot + 8;
    if (st->n_photons) { for (size_t k = 0; k < st->n_cells; k++) st->n_photons[k] = (int64_t)q[k]; q += st->n_cells; }
    if (st->n_bins) memcpy(st->spec_sum, q, sizeof(double) * st->n_bins * ntot);
    return 0;
}

const int64_t *orc_n_photons(const orc_state *st) { return st->n_photons; }
const double *orc_specific_energy_spectrum(const orc_state *st) { return st->spec; }
const double *orc_specific_energy_sum_spectrum(const orc_state *st) { return st->spec_sum; }
int orc_pda_last_cells(const orc_state *st) { return st->pda_last_cells; }

/* fortranlib `quantile` (lib_statistics; source absent): element of the sorted sample at rank
 * nint(percent / 100 * (n - 1)) + 1 -- nint rounds half away from zero. */
static int cmp_double(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }
static double quantile(double *v, size_t n, double percent)
{
    if (!n) return 0.0;
    qsort(v, n, sizeof(double), cmp_double);
    long ipos = (long)floor(percent / 100.0 * (double)(n - 1) + 0.5);
    if (ipos < 0) ipos = 0;
    if ((size_t)ipos > n - 1) ipos = (long)(n - 1);
    return v[ipos];
}

/* specific_energy_converged: src/grid/grid_physics_3d.f90:637-689 -- the value whose evolution is tested: the
 * `percentile` quantile of max(a/b, b/a) over the (cell, dust) pairs that changed and are positive before and after.
 * status: 0 value computed, 1 nothing changed (value 0), 2 could not check (only zero cells changed). */
int orc_convergence_value(const orc_state *st, const double *prev, double percentile, double *value)
{
    const size_t n = (size_t)st->n_dust * st->n_cells;
    const double *cur = st->specific_energy;
    int all_same = 1, only_zero = 1;
    size_t m = 0;
    for (size_t k = 0; k < n; k++) {
        if (prev[k] != cur[k]) { all_same = 0; if (prev[k] != 0.0 && cur[k] != 0.0) only_zero = 0; }
        if (prev[k] > 0.0 && cur[k] > 0.0 && prev[k] != cur[k]) m++;
    }
    if (all_same) { *value = 0.0; return 1; }
    if (only_zero) { *value = 0.0; return 2; }
    double *v = malloc(sizeof(double) * (m ? m : 1));
    size_t j = 0;
    for (size_t k = 0; k < n; k++)
        if (prev[k] > 0.0 && cur[k] > 0.0 && prev[k] != cur[k]) { double a = prev[k] / cur[k], b = cur[k] / prev[k]; v[j++] = a > b ? a : b; }
    *value = quantile(v, m, percentile);
    free(v);
    return 0;
}

int orc_lucy_iteration(orc_state *st, uint64_t n_packets, int iter, int n_threads,
                       double *specific_energy_out, orc_iter_stats *stats)
{
    if (n_packets == 0) return 0;
    int rc = orc_lucy_accumulate(st, 0, n_packets, iter, n_threads, NULL);
    if (rc) return rc;
    return orc_lucy_finish(st, specific_energy_out, stats);
}

/* ------------------------------------------------------------------ */
/* Peel-off imaging: images_peeled.f90:95-270, image_type.f90          */
/* ------------------------------------------------------------------ */

#define C_CGS 29979245800.0
#define STEF_BOLTZ 5.67051e-5


/* Spectra on the frequency bins of an image group, cached for the raytracing iteration:
 * get_source_spectrum / get_dust_emissivity / get_dust_extinction (images_peeled.f90:422-538) with
 * get_spectrum_binned (source_type.f90:1118-1172), get_j_nu_binned and get_chi_nu_binned
 * (dust_type_4elem.f90:722-750, 793-818). */
static void raytracing_caches(const orc_state *st)
{
    if (st->n_peeled == 0) return;
    int nj_max = 1;
    for (int d = 0; d < st->n_dust; d++) if (st->dust[d].n_jnu > nj_max) nj_max = st->dust[d].n_jnu;
    double **lo = malloc(sizeof(double *) * st->n_peeled), **hi = malloc(sizeof(double *) * st->n_peeled);
    for (int ig = 0; ig < st->n_peeled; ig++) {
        peeled_t *p = &st->peeled[ig];
        const int nn = p->d.n_nu;
        lo[ig] = malloc(sizeof(double) * nn); hi[ig] = malloc(sizeof(double) * nn);
        for (int i = 0; i < nn; i++) {
            lo[ig][i] = pow(10.0, p->log10_nu_min + (p->log10_nu_max - p->log10_nu_min) * (double)i / (double)nn);
            hi[ig][i] = pow(10.0, p->log10_nu_min + (p->log10_nu_max - p->log10_nu_min) * (double)(i + 1) / (double)nn);
        }
        p->src_spec = calloc((size_t)(st->n_sources ? st->n_sources : 1) * nn, sizeof(double));
        p->dust_log10_em = calloc((size_t)(st->n_dust ? st->n_dust : 1) * nj_max * nn, sizeof(double));
        p->dust_chi = calloc((size_t)(st->n_dust ? st->n_dust : 1) * nn, sizeof(double));
        p->nj_stride = nj_max;
    }
    if (st->cfg.monochromatic) {
        /* use_exact_nu: get_spectrum_interp (source_type.f90:1098-1116), get_j_nu_interp and get_chi_nu_interp
         * (dust_type_4elem.f90:708-720, 780-791) at the group's frequencies */
        for (int ig = 0; ig < st->n_peeled; ig++) {
            peeled_t *p = &st->peeled[ig];
            const int nn = p->d.n_nu;
            const double *nu = st->frequencies + (p->d.inu_min - 1);
            for (int is = 0; is < st->n_sources; is++) {
                const source_t *src = &st->src[is];
                for (int i = 0; i < nn; i++)
                    p->src_spec[(size_t)is * nn + i] = src->spectrum_type == 1 ? pdf_interp_log(&src->spectrum, nu[i])
                                                        : src->spectrum_type == 2 ? normalized_B_nu(nu[i], src->temperature) : 0.0;   /* lte: the packets carry the dust emissivity */
            }
            for (int d = 0; d < st->n_dust; d++) {
                const dust_t *du = &st->dust[d];
                for (int j = 0; j < du->n_jnu; j++)
                    for (int i = 0; i < nn; i++)
                        p->dust_log10_em[((size_t)d * nj_max + j) * nn + i] = log10(pdf_interp_log(&du->j_nu[j], nu[i]));
                for (int i = 0; i < nn; i++) {
                    double c = (nu[i] >= du->nu[0] && nu[i] <= du->nu[du->n_nu - 1]) ? interp1d_loglog(du->nu, du->chi, du->n_nu, nu[i]) : 0.0;
                    p->dust_chi[(size_t)d * nn + i] = c;
                }
            }
            free(lo[ig]); free(hi[ig]);
        }
        free(lo); free(hi);
        return;
    }
    /* blackbody tabulated on 100000 points per decade between 3e9 and 3e16 Hz (:1142-1154), once per source */
    const double l0 = log10(3.e9), l1 = log10(3.e16);
    const int nb = (int)ceil((l1 - l0) * 100000);
    double *bnu = NULL, *bfnu = NULL;
    for (int is = 0; is < st->n_sources; is++) {
        const source_t *src = &st->src[is];
        const double *x, *y; int n;
        if (src->spectrum_type == 3) continue;      /* lte: get_spectrum_binned has no case for it; the packets carry the dust emissivity */
        if (src->spectrum_type == 1) { x = src->spectrum.x; y = src->spectrum.pdf; n = src->spectrum.n; }
        else {
            if (!bnu) {
                bnu = malloc(sizeof(double) * nb); bfnu = malloc(sizeof(double) * nb);
                for (int k = 0; k < nb; k++) bnu[k] = pow(10.0, (double)k / (double)(nb - 1) * (l1 - l0) + l0);
            }
            const double a = 2.0 * H_CGS / C_CGS / C_CGS / STEF_BOLTZ * PI, b = H_CGS / K_CGS;
            const double T = src->temperature, T4 = T * T * T * T;
            for (int k = 0; k < nb; k++) bfnu[k] = a * bnu[k] * bnu[k] * bnu[k] / (exp(b * bnu[k] / T) - 1.0) / T4;   /* normalized_B_nu :1088-1096 */
            x = bnu; y = bfnu; n = nb;
        }
        const double tot = integral_loglog_all(x, y, n);
        for (int ig = 0; ig < st->n_peeled; ig++) {
            peeled_t *p = &st->peeled[ig];
            const int nn = p->d.n_nu;
            for (int i = 0; i < nn; i++) p->src_spec[(size_t)is * nn + i] = integral_loglog_range(x, y, n, lo[ig][i], hi[ig][i]) / tot;
        }
    }
    free(bnu); free(bfnu);
    for (int ig = 0; ig < st->n_peeled; ig++) {
        peeled_t *p = &st->peeled[ig];
        const int nn = p->d.n_nu;
        for (int d = 0; d < st->n_dust; d++) {
            const dust_t *du = &st->dust[d];
            for (int j = 0; j < du->n_jnu; j++) {
                const pdf_t *q = &du->j_nu[j];
                double tot = integral_loglog_all(q->x, q->pdf, q->n);
                for (int i = 0; i < nn; i++)
                    p->dust_log10_em[((size_t)d * nj_max + j) * nn + i] = log10(integral_loglog_range(q->x, q->pdf, q->n, lo[ig][i], hi[ig][i]) / tot);
            }
            for (int i = 0; i < nn; i++)
                p->dust_chi[(size_t)d * nn + i] = integral_loglog_range(du->nu, du->chi, du->n_nu, lo[ig][i], hi[ig][i]) / (hi[ig][i] - lo[ig][i]);
        }
        free(lo[ig]); free(hi[ig]);
    }
    free(lo); free(hi);
}

static int peeled_setup(orc_state *st, peeled_t *p, const orc_peeled_desc *in)
{
    memset(p, 0, sizeof *p);
    p->d = *in;
    if (in->inside_observer) {      /* images_peeled.f90:312-315, 356-363 */
        if (p->d.d_min < 0.0) p->d.d_min = 0.0;
        if (in->compute_image && in->x_min < in->x_max) { snprintf(g_error, sizeof g_error, "longitudes should increase towards the left for inside observers"); return 1; }
        if (in->compute_sed) { snprintf(g_error, sizeof g_error, "computing SEDs for inside observers is not supported"); return 1; }
    }
    p->theta = dup(in->theta, in->n_view); p->phi = dup(in->phi, in->n_view);
    p->d.theta = p->theta; p->d.phi = p->phi;
    p->view = malloc(sizeof(angle_t) * in->n_view);
    for (int i = 0; i < in->n_view; i++) {
        /* angle3d_deg(theta, phi) */
        double t = in->theta[i] * PI / 180.0, f = in->phi[i] * PI / 180.0;
        p->view[i].cost = cos(t); p->view[i].sint = sin(t); p->view[i].cosp = cos(f); p->view[i].sinp = sin(f);
    }
    if (st->cfg.monochromatic) {    /* image_type.f90:243-258 */
        if (in->inu_min < 1 || in->inu_min > st->cfg.n_frequencies) { snprintf(g_error, sizeof g_error, "inu_min value is out of range"); return 1; }
        if (in->inu_max < 1 || in->inu_max > st->cfg.n_frequencies) { snprintf(g_error, sizeof g_error, "inu_max value is out of range"); return 1; }
        p->d.n_nu = in->inu_max - in->inu_min + 1;
    }
    if (in->use_filters) {        /* image_type.f90:173-181,285-291; images_peeled.f90:349 */
        if (st->cfg.monochromatic) { snprintf(g_error, sizeof g_error, "cannot use filters in monochromatic mode"); return 1; }
        if (st->cfg.raytracing) { snprintf(g_error, sizeof g_error, "filter convolution cannot be used with raytracing"); return 1; }
        p->filt_off = malloc(sizeof(int) * (in->n_nu + 1));
        p->filt_off[0] = 0;
        for (int i = 0; i < in->n_nu; i++) p->filt_off[i + 1] = p->filt_off[i] + in->filt_n[i];
        p->filt_nu = dup(in->filt_nu, p->filt_off[in->n_nu]); p->filt_tr = dup(in->filt_tr, p->filt_off[in->n_nu]);
    }
    p->n_stokes = in->compute_stokes ? 4 : 1;
    /* image_type.f90:283-300 */
    switch (in->track_origin) {
    case 0: p->n_orig = 1; break;
    case 1: p->n_orig = 4; break;
    case 2: p->n_orig = 2 * (st->n_sources + st->n_dust); break;
    case 3: p->n_orig = 2 * (in->track_n_scat + 2); break;
    default: snprintf(g_error, sizeof g_error, "unknown track_origin"); return 1;
    }
    p->log10_nu_min = log10(in->nu_min); p->log10_nu_max = log10(in->nu_max);
    if (in->compute_sed) {
        p->log10_ap_min = log10(in->ap_min); p->log10_ap_max = log10(in->ap_max);
        p->sed_size = (size_t)p->n_stokes * p->n_orig * in->n_view * in->n_ap * in->n_nu;
        p->sed = calloc(p->sed_size, sizeof(double)); p->sed2 = calloc(p->sed_size, sizeof(double));
    }
    if (in->compute_image) {
        p->img_size = (size_t)p->n_stokes * p->n_orig * in->n_view * in->n_y * in->n_x * in->n_nu;
        p->img = calloc(p->img_size, sizeof(double)); p->img2 = calloc(p->img_size, sizeof(double));
    }
    return 0;
}

static void peeled_free(peeled_t *p)
{
    free(p->theta); free(p->phi); free(p->view); free(p->sed); free(p->sed2); free(p->img); free(p->img2);
    free(p->src_spec); free(p->dust_log10_em); free(p->dust_chi);
    free(p->filt_off); free(p->filt_nu); free(p->filt_tr);
}

int orc_peeled_n_orig(const orc_state *st, int g) { return st->peeled[g].n_orig; }
const double *orc_peeled_sed(const orc_state *st, int g) { return st->peeled[g].sed; }
const double *orc_peeled_img(const orc_state *st, int g) { return st->peeled[g].img; }
const double *orc_peeled_sed2(const orc_state *st, int g) { return st->peeled[g].sed2; }
const double *orc_peeled_img2(const orc_state *st, int g) { return st->peeled[g].img2; }
double *orc_peeled_sed_rw(orc_state *st, int g) { return st->peeled[g].sed; }
double *orc_peeled_img_rw(orc_state *st, int g) { return st->peeled[g].img; }

/* fortranlib ipos(xmin, xmax, x, n): 1-based bin of x in n equal bins; here
 * 0-based, -1 / n when outside (callers test the range). */
static int ipos0(double xmin, double xmax, double x, int n)
{
    double f = (x - xmin) / (xmax - xmin);
    if (f < 0.0) return -1;
    int i = (int)floor(f * n);
    if (f == 1.0) i = n - 1;
    return i;
}

/* origin slot (0-based): image_type.f90:112-134 orig() and :442-465 */
static int origin_slot(const orc_state *st, const peeled_t *pg, const photon_t *p)
{
    int o; /* 1 source, 2 dust, 3 scattered source, 4 scattered dust */
    if (p->scattered) o = p->reprocessed ? 4 : 3; else o = p->reprocessed ? 2 : 1;
    switch (pg->d.track_origin) {
    case 0: return 0;
    case 1: return o - 1;
    case 2:
        /* detailed: [source emit per source | dust emit per dust | source scat per source | dust scat per dust] */
        switch (o) {
        case 1: return p->source_id;
        case 2: return st->n_sources + p->dust_id;
        case 3: return st->n_sources + st->n_dust + p->source_id;
        default: return 2 * st->n_sources + st->n_dust + p->dust_id;
        }
    case 3: {
        int ns = p->n_scat < pg->d.track_n_scat + 1 ? p->n_scat : pg->d.track_n_scat + 1;
        return (p->reprocessed ? (pg->d.track_n_scat + 2) : 0) + ns;
    }
    }
    return 0;
}

/* image_bin_single :478-522 */
static void image_bin_single(const orc_state *st, int ig, const photon_t *p, double x_image, double y_image,
                             int iv, acc_t *acc, int inu, int io, double transmission);

/* image_bin :408-476 */
static void image_bin(const orc_state *st, int ig, const photon_t *p, double x_image, double y_image,
                      int iv, acc_t *acc)
{
    const peeled_t *pg = &st->peeled[ig];
    const orc_peeled_desc *d = &pg->d;
    if (p->energy != p->energy || p->s[0] != p->s[0]) return;   /* :421-429 NaN energy / flux ignored */
    int io = origin_slot(st, pg, p);
    if (d->use_filters) {      /* :467-475: interp1d (linear) of the transmission curve, 0 outside it */
        for (int f = 0; f < d->n_nu; f++) {
            const double *fx = pg->filt_nu + pg->filt_off[f], *ft = pg->filt_tr + pg->filt_off[f];
            const int n = pg->filt_off[f + 1] - pg->filt_off[f];
            const int j = locate(fx, n, p->nu);
            if (j < 0) continue;
            const double tr = ft[j] + (p->nu - fx[j]) / (fx[j + 1] - fx[j]) * (ft[j + 1] - ft[j]);
            if (tr > 0.0) image_bin_single(st, ig, p, x_image, y_image, iv, acc, f, io, tr);
        }
        return;
    }
    int inu = st->cfg.monochromatic ? p->inu - (d->inu_min - 1)      /* image_type.f90:435-436 */
                                    : ipos0(pg->log10_nu_min, pg->log10_nu_max, log10(p->nu), d->n_nu);
    image_bin_single(st, ig, p, x_image, y_image, iv, acc, inu, io, 1.0);
}

static void image_bin_single(const orc_state *st, int ig, const photon_t *p, double x_image, double y_image,
                             int iv, acc_t *acc, int inu, int io, double transmission)
{
    const peeled_t *pg = &st->peeled[ig];
    const orc_peeled_desc *d = &pg->d;
    (void)st;
    if (inu < 0 || inu >= d->n_nu) return;
    int ns = pg->n_stokes;
    if (d->compute_image) {
        int ix = ipos0(d->x_min, d->x_max, x_image, d->n_x);
        int iy = ipos0(d->y_min, d->y_max, y_image, d->n_y);
        if (ix >= 0 && ix < d->n_x && iy >= 0 && iy < d->n_y) {
            for (int is = 0; is < ns; is++) {
                size_t k = (((((size_t)is * pg->n_orig + io) * d->n_view + iv) * d->n_y + iy) * d->n_x + ix) * d->n_nu + inu;
                double val = p->s[is] * p->energy * transmission;
                acc->img[ig][k] += val;
                if (d->uncertainties) acc->img2[ig][k] += val * val;
            }
        }
    }
    if (d->compute_sed) {
        /* find_sed_bin :337-356 */
        double lr = log10(sqrt(x_image * x_image + y_image * y_image));
        int ir;
        if (lr < pg->log10_ap_min || d->n_ap == 1) ir = 0;
        else ir = ipos0(pg->log10_ap_min, pg->log10_ap_max, lr, d->n_ap - 1) + 1;
        if (ir >= 0 && ir < d->n_ap) {
            for (int is = 0; is < ns; is++) {
                size_t k = ((((size_t)is * pg->n_orig + io) * d->n_view + iv) * d->n_ap + ir) * d->n_nu + inu;
                double val = p->s[is] * p->energy * transmission;
                acc->sed[ig][k] += val;
                if (d->uncertainties) acc->sed2[ig][k] += val * val;
            }
        }
    }
}

/* in_image :374-406 */
static int in_image(const peeled_t *pg, double x, double y)
{
    const orc_peeled_desc *d = &pg->d;
    if (d->compute_image) {
        if ((x >= d->x_min && x <= d->x_max) || (x <= d->x_min && x >= d->x_max))
            if ((y >= d->y_min && y <= d->y_max) || (y <= d->y_min && y >= d->y_max)) return 1;
    }
    if (d->compute_sed && x * x + y * y <= d->ap_max * d->ap_max) return 1;
    return 0;
}

/* peeloff_photon :95-270 (external observers) */
/* image_bin_raytraced: image_type.f90:527-606 -- the whole spectrum goes into Stokes I of one pixel */
static void image_bin_raytraced(const orc_state *st, int ig, const photon_t *p, double x_image, double y_image,
                                int iv, const double *spectrum, acc_t *acc)
{
    const peeled_t *pg = &st->peeled[ig];
    const orc_peeled_desc *d = &pg->d;
    if (p->energy != p->energy || p->s[0] != p->s[0]) return;
    int io = origin_slot(st, pg, p);
    if (d->compute_image) {
        int ix = ipos0(d->x_min, d->x_max, x_image, d->n_x);
        int iy = ipos0(d->y_min, d->y_max, y_image, d->n_y);
        if (ix >= 0 && ix < d->n_x && iy >= 0 && iy < d->n_y)
            for (int iw = 0; iw < d->n_nu; iw++) {
                size_t k = ((((size_t)io * d->n_view + iv) * d->n_y + iy) * d->n_x + ix) * d->n_nu + iw;
                acc->img[ig][k] += spectrum[iw];
                if (d->uncertainties) acc->img2[ig][k] += spectrum[iw] * spectrum[iw];
            }
    }
    if (d->compute_sed) {
        double lr = log10(sqrt(x_image * x_image + y_image * y_image));
        int ir;
        if (lr < pg->log10_ap_min || d->n_ap == 1) ir = 0;
        else ir = ipos0(pg->log10_ap_min, pg->log10_ap_max, lr, d->n_ap - 1) + 1;
        if (ir >= 0 && ir < d->n_ap)
            for (int iw = 0; iw < d->n_nu; iw++) {
                size_t k = (((size_t)io * d->n_view + iv) * d->n_ap + ir) * d->n_nu + iw;
                acc->sed[ig][k] += spectrum[iw];
                if (d->uncertainties) acc->sed2[ig][k] += spectrum[iw] * spectrum[iw];
            }
    }
}

/* grid_escape_column_density: grid_propagate_3d.f90:482-582 */
static void grid_escape_column_density(const orc_state *st, const photon_t *p_orig, double tmax, double *col,
                                       rng_t *g, acc_t *acc, int *killed)
{
    photon_t p = *p_orig;
    p.radial = ((p.r[0] * p.v[0] + p.r[1] * p.v[1]) + p.r[2] * p.v[2]) > 0.0;   /* grid_propagate_3d.f90:400,505 */
    double t_current = 0.0;
    *killed = 0;
    for (int d = 0; d < st->n_dust; d++) col[d] = 0.0;
    if (escaped(st, p.ic)) return;
    {   /* grid_propagate_3d.f90:516-523 */
        double t_source; int sid;
        find_nearest_source(st, p.r, p.v, &t_source, &sid);
        if (t_source < tmax) { *killed = 1; return; }
    }
    for (;;) {
        if (g->countdown == 0) {
            g->countdown = rng_check_gap(g, st->check_p, st->check_log1mp);
            if (!in_correct_cell(st, &p)) { acc->killed_geo++; *killed = 1; return; }
        } else g->countdown--;
        double tmin; int id_min[3];
        int fw = find_wall(st, &p, &tmin, id_min);
        if (fw < 0) { amr_negative_t(acc); *killed = 1; return; }
        if (!fw) { acc->killed_geo++; *killed = 1; return; }
        size_t ic = cell_index(st, p.ic);
        int finished = 0;
        if (t_current + tmin > tmax) { tmin = tmax - t_current; finished = 1; }
        for (int a = 0; a < 3; a++) p.r[a] = p.r[a] + tmin * p.v[a];
        t_current += tmin;
        for (int d = 0; d < st->n_dust; d++) col[d] += st->density[(size_t)d * st->n_cells + ic] * tmin;
        acc->crossings++;
        if (finished) return;
        advance_cell(st, &p, id_min);
        if (st->grid_type == GRID_AMR && p.ic[0] < 0) { acc->killed_geo++; *killed = 1; return; }
        if (escaped(st, p.ic)) return;
    }
}

static void peeloff_photon(const orc_state *st, const photon_t *p_orig, rng_t *g, acc_t *acc, int polychromatic)
{
    for (int ig = 0; ig < st->n_peeled; ig++) {
        const peeled_t *pg = &st->peeled[ig];
        for (int iv = 0; iv < pg->d.n_view; iv++) {
            photon_t p = *p_orig;
            memcpy(p.s, p.s_prev, sizeof p.s); p.a = p.a_prev; memcpy(p.v, p.v_prev, sizeof p.v);
            angle_t a_req = pg->view[iv];
            const int inside = pg->d.inside_observer;
            if (inside) {   /* a_peeloff :410-421: vector3d_to_angle3d(r_peeloff - r) */
                double w[3] = {pg->d.peeloff_origin[0] - p.r[0], pg->d.peeloff_origin[1] - p.r[1], pg->d.peeloff_origin[2] - p.r[2]};
                double rxy = sqrt(w[0] * w[0] + w[1] * w[1]), rr = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
                a_req.cost = w[2] / rr; a_req.sint = rxy / rr;
                if (rxy > 0.0) { a_req.cosp = w[0] / rxy; a_req.sinp = w[1] / rxy; } else { a_req.cosp = 1.0; a_req.sinp = 0.0; }
            }
            double v_req[3]; angle_to_vector(&a_req, v_req);
            if (p.last_isotropic) {
                p.s[0] = 1.0; p.s[1] = p.s[2] = p.s[3] = 0.0;
                p.a = a_req; memcpy(p.v, v_req, sizeof v_req);
            } else {
                if (p.last == LAST_SR) {
                    /* source_emit_peeloff :512-533, emit_from_extern_*_peeloff :811-820,909-933 */
                    const source_t *src = &st->src[p.source_id];
                    if (src->peeloff) {
                        angle_t nrm;
                        if (src->type == 5 || src->type == 2) nrm = p.source_a; else box_face_normal(p.face_id, &nrm);
                        double vn[3]; angle_to_vector(&nrm, vn);
                        double mu = v_req[0] * vn[0] + v_req[1] * vn[1] + v_req[2] * vn[2];
                        if (mu < 0.0) mu = 0.0;
                        /* emit_from_sphere_peeloff :692-707 (pdfs normalised to 4 pi) */
                        if (src->type == 2 && src->limb_darkening) p.s[0] = 2.0 * (1.5 * mu * mu + mu);
                        else p.s[0] = 4.0 * mu;
                        p.s[1] = p.s[2] = p.s[3] = 0.0;
                    } else { p.s[0] = p.s[1] = p.s[2] = p.s[3] = 0.0; }
                    p.a = a_req;
                } else if (p.last == LAST_DS) {
                    dust_scatter_peeloff(&st->dust[p.dust_id], p.nu, &p.a, p.s, &a_req);
                } else {
                    p.a = a_req;
                }
                angle_to_vector(&p.a, p.v);
            }
            p.killed = 0;
            place_in_cell(st, &p, acc);
            if (p.killed) continue;
            double dr[3] = {p.r[0] - pg->d.peeloff_origin[0], p.r[1] - pg->d.peeloff_origin[1], p.r[2] - pg->d.peeloff_origin[2]};
            double dd, tmax = DBL_MAX, x_image, y_image;
            if (inside) { dd = sqrt((dr[0] * dr[0] + dr[1] * dr[1]) + dr[2] * dr[2]); tmax = dd; }       /* :158-165 */
            else dd = -(v_req[0] * p.r[0] + v_req[1] * p.r[1] + v_req[2] * p.r[2]);
            if (dd < pg->d.d_min || dd > pg->d.d_max) continue;
            if (inside) {
                /* sky position for an observer looking along the group's viewing angle: :169-205 */
                const angle_t *av = &pg->view[iv];
                double va[3]; angle_to_vector(&p.a, va);
                double sx = (va[0] * av->cosp + va[1] * av->sinp) * av->sint + va[2] * av->cost;
                double sy = -va[0] * av->sinp + va[1] * av->cosp;
                double sz = -(va[0] * av->cosp + va[1] * av->sinp) * av->cost + va[2] * av->sint;
                const double rad2deg = 180.0 / PI;
                x_image = atan2(sy, sx) * rad2deg;
                y_image = atan2(sqrt(sx * sx + sy * sy), sz) * rad2deg - 90.0;
                /* Fortran modulo(a, 360) = a - 360 floor(a / 360) */
                double ax = x_image - pg->d.x_max, ay = y_image - pg->d.y_min;
                x_image = pg->d.x_max + (ax - 360.0 * floor(ax / 360.0));
                y_image = pg->d.y_min + (ay - 360.0 * floor(ay / 360.0));
            } else {
                x_image = dr[1] * p.a.cosp - dr[0] * p.a.sinp;
                y_image = dr[2] * p.a.sint - dr[1] * p.a.cost * p.a.sinp - dr[0] * p.a.cost * p.a.cosp;
            }
            if (!in_image(pg, x_image, y_image)) continue;
            /* the 1/d^2 flux dilution of inside observers (:236) is applied after the optical depth is known */
            const double dilute = inside ? 1.0 / (4.0 * PI * pow(dd, 2.0)) : 1.0;
            if (polychromatic) {
                /* images_peeled.f90:218-254: the packet carries the whole spectrum of its emitter */
                double col[ORC_MAX_DUST]; int killed_c = 0;
                for (int d = 0; d < st->n_dust; d++) col[d] = 0.0;
                if (!pg->d.ignore_optical_depth) grid_escape_column_density(st, &p, tmax, col, g, acc, &killed_c);
                if (killed_c) continue;
                if (inside) for (int k = 0; k < 4; k++) p.s[k] = p.s[k] * dilute;
                const int nn = pg->d.n_nu;
                double spec[nn];
                if (p.emiss_type == 3) {      /* get_dust_emissivity :451-505 */
                    const double *le = pg->dust_log10_em + ((size_t)p.dust_id * pg->nj_stride + p.emiss_var_id) * nn;
                    for (int i = 0; i < nn; i++) {
                        double v = pow(10.0, (le[nn + i] - le[i]) * p.emiss_var_frac + le[i]);
                        spec[i] = (v != v) ? 0.0 : v;
                    }
                } else {
                    const double *ss = pg->src_spec + (size_t)p.source_id * nn;
                    for (int i = 0; i < nn; i++) spec[i] = ss[i];
                }
                for (int i = 0; i < nn; i++) spec[i] = spec[i] * p.s[0] * p.energy;
                for (int d = 0; d < st->n_dust; d++) {
                    const double *chi = pg->dust_chi + (size_t)d * nn;
                    for (int i = 0; i < nn; i++) spec[i] = spec[i] * exp(-col[d] * chi[i]);
                }
                image_bin_raytraced(st, ig, &p, x_image, y_image, iv, spec, acc);
                continue;
            }
            double tau = 0.0; int killed = 0;
            /* The propagation checks of a peel-off walk draw from their own stream, keyed by (packet, peel-off event,
             * view), not from the packet's: the walk then is a function of the event alone and can be done anywhere, in any
             * order (the device defers it to a separate kernel).  16 blocks per walk: one check per ~1/p steps. */
            rng_t gp = *g;
            gp.stream_b = 2u; gp.blk_b = (p_orig->peel_seq * (uint32_t)st->n_views_total + (uint32_t)(st->view_base[ig] + iv)) * 16u;
            gp.countdown = rng_check_gap(&gp, st->check_p, st->check_log1mp);
            if (!pg->d.ignore_optical_depth) tau = grid_escape_tau(st, &p, tmax, &gp, acc, &killed);
            if (killed) continue;
            if (inside) for (int k = 0; k < 4; k++) p.s[k] = p.s[k] * dilute;
            double att = exp(-tau);
            for (int k = 0; k < 4; k++) p.s[k] *= att;
            image_bin(st, ig, &p, x_image, y_image, iv, acc);
        }
    }
    if (!polychromatic) ((photon_t *)p_orig)->peel_seq++;
}

/* forced_interaction_wr99: forced_interaction.f90:23-58 */
static void forced_interaction_wr99(double tau_escape, double xi, double *tau, double *weight)
{
    double ome = tau_escape > 1e-7 ? 1.0 - exp(-tau_escape) : tau_escape;
    *tau = -log(1.0 - xi * ome);
    *weight = ome;
}

/* forced_interaction_baes16: forced_interaction.f90:60-133 */
static void forced_interaction_baes16(double tau_escape, double bxi, double xi, double *tau, double *weight)
{
    double ome = tau_escape > 1e-7 ? 1.0 - exp(-tau_escape) : tau_escape;
    double alpha = (1.0 - bxi) / ome, beta = bxi / tau_escape;
    double tmin = 0.0, tmax = tau_escape, t = 0.0;
    for (int i = 0; i < 60; i++) {
        t = 0.5 * (tmin + tmax);
        double test = t > 1e-7 ? alpha * (1.0 - exp(-t)) + beta * t : alpha * t + beta * t;
        if (test > xi) tmax = t; else tmin = t;
    }
    t = 0.5 * (tmin + tmax);
    *tau = t;
    *weight = 1.0 / (alpha + beta * exp(t));
}

/* do_final + propagate: iter_final.f90:60-273 */
static void final_packet(const orc_state *st, uint64_t id, acc_t *acc)
{
    rng_t g; photon_t p;
    rng_init(&g, st->cfg.seed, 0x10000u, id);
    if (emit(st, &p, &g, acc)) return;
    const int scattering_only = st->cfg.raytracing;      /* do_final(..., peeloff_scattering_only=use_raytracing) */
    if (st->n_peeled && !scattering_only) peeloff_photon(st, &p, &g, acc, 0);
    for (int64_t inter = 1; inter <= st->cfg.n_inter_max + 1; inter++) {
        double tau;
        if (st->cfg.mrw && inter > 1 && mrw_steps(st, &p, &g, acc, NULL, st->n_peeled && !scattering_only)) break;
        if (inter == 1 && st->cfg.forced_first_interaction) {
            int killed = 0;
            double tau_escape = grid_escape_tau(st, &p, DBL_MAX, &g, acc, &killed);
            if (tau_escape > 1e-10 && !killed) {
                double weight, xi = rng_uniform(&g);
                if (st->cfg.forced_first_interaction_algorithm == 2)
                    forced_interaction_baes16(tau_escape, st->cfg.baes16_xi, xi, &tau, &weight);
                else
                    forced_interaction_wr99(tau_escape, xi, &tau, &weight);
                p.energy *= weight;
            } else tau = rng_exp(&g);
        } else tau = rng_exp(&g);
        grid_integrate(st, &p, tau, &g, acc, NULL);
        if (p.reabsorbed) {     /* iter_final.f90:213-243 */
            int64_t ia;
            for (ia = 1; ia <= st->cfg.n_reabs_max; ia++) {
                int rid = p.reabsorbed_id; double re = p.energy;
                if (emit_from(st, &p, &g, acc, rid, re)) return;
                /* peeled even in scattering-only mode: "a kind of scattering" not seen by the raytracing */
                if (st->n_peeled) peeloff_photon(st, &p, &g, acc, 0);
                tau = rng_exp(&g);
                grid_integrate(st, &p, tau, &g, acc, NULL);
                if (!p.reabsorbed) break;
            }
            if (ia == st->cfg.n_reabs_max + 1) { acc->killed_int++; p.killed = 1; break; }
        }
        if (p.killed || escaped(st, p.ic)) break;
        if (inter == st->cfg.n_inter_max + 1) { acc->killed_int++; p.killed = 1; break; }
        interact(st, &p, &g, acc);
        if (p.killed) break;
        p.killed = (st->cfg.kill_on_scatter && p.scattered) || (st->cfg.kill_on_absorb && !p.scattered);
        if (p.killed) break;
        if (st->n_peeled && (p.scattered || !scattering_only)) peeloff_photon(st, &p, &g, acc, 0);
    }
    /* iter_final.f90:127-129 + binned_images_bin_photon (images_binned.f90:58-81): packets that were not killed */
    if (st->has_binned && !p.killed) {
        double phi = atan2(p.a.sinp, p.a.cosp);
        if (phi < 0.0) phi = phi + 2.0 * PI;
        int it = ipos0(-1.0, 1.0, p.a.cost, st->n_theta), ip = ipos0(0.0, 2.0 * PI, phi, st->n_phi);
        if (it >= 0 && it < st->n_theta && ip >= 0 && ip < st->n_phi) {
            double x_image = p.r[1] * p.a.cosp - p.r[0] * p.a.sinp;
            double y_image = p.r[2] * p.a.sint - p.r[1] * p.a.cost * p.a.sinp - p.r[0] * p.a.cost * p.a.cosp;
            image_bin(st, st->n_peeled, &p, x_image, y_image, st->n_phi * it + ip, acc);
        }
    }
}

static uint64_t g_final_first_id = 0;   /* debugging aid: id offset of the next final iteration */
void orc_set_final_first_id(uint64_t first) { g_final_first_id = first; }

/* Runs `fn` for n_packets packet ids on n_threads threads with thread-local image cubes and
 * merges them into the state's cubes (zeroed first if `zero`). */
typedef void (*image_packet_fn)(const orc_state *st, uint64_t id, acc_t *acc, const void *ctx);

static int image_run(orc_state *st, uint64_t n_packets, int n_threads, uint64_t first_id, image_packet_fn fn,
                     const void *ctx, int zero, orc_iter_stats *tot_out)
{
    int nt = resolve_threads(n_threads);
    if ((uint64_t)nt > n_packets && n_packets > 0) nt = (int)n_packets;
    if (nt < 1) nt = 1;
    acc_t *accs = calloc(nt, sizeof(acc_t));
    int ng = st->n_groups;
    for (int t = 0; t < nt; t++) {
        accs[t].sed = calloc(ng ? ng : 1, sizeof(double *)); accs[t].sed2 = calloc(ng ? ng : 1, sizeof(double *));
        accs[t].img = calloc(ng ? ng : 1, sizeof(double *)); accs[t].img2 = calloc(ng ? ng : 1, sizeof(double *));
        for (int g = 0; g < ng; g++) {
            peeled_t *pg = &st->peeled[g];
            if (t == 0) {
                if (zero) {
                    if (pg->sed) { memset(pg->sed, 0, sizeof(double) * pg->sed_size); memset(pg->sed2, 0, sizeof(double) * pg->sed_size); }
                    if (pg->img) { memset(pg->img, 0, sizeof(double) * pg->img_size); memset(pg->img2, 0, sizeof(double) * pg->img_size); }
                }
                accs[t].sed[g] = pg->sed; accs[t].sed2[g] = pg->sed2; accs[t].img[g] = pg->img; accs[t].img2[g] = pg->img2;
            } else {
                if (pg->sed) { accs[t].sed[g] = calloc(pg->sed_size, sizeof(double)); accs[t].sed2[g] = calloc(pg->sed_size, sizeof(double)); }
                if (pg->img) { accs[t].img[g] = calloc(pg->img_size, sizeof(double)); accs[t].img2[g] = calloc(pg->img_size, sizeof(double)); }
            }
        }
    }
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        acc_t *acc = &accs[t];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int64_t i = 0; i < (int64_t)n_packets; i++) {
            if (acc->fatal) continue;
            fn(st, first_id + (uint64_t)i, acc, ctx);
        }
    }
    orc_iter_stats tot; memset(&tot, 0, sizeof tot);
    int fatal = 0;
    for (int t = 0; t < nt; t++) {
        for (int g = 0; g < ng; g++) {
            peeled_t *pg = &st->peeled[g];
            if (t > 0) {
                if (pg->sed) for (size_t k = 0; k < pg->sed_size; k++) { pg->sed[k] += accs[t].sed[g][k]; pg->sed2[k] += accs[t].sed2[g][k]; }
                if (pg->img) for (size_t k = 0; k < pg->img_size; k++) { pg->img[k] += accs[t].img[g][k]; pg->img2[k] += accs[t].img2[g][k]; }
                free(accs[t].sed[g]); free(accs[t].sed2[g]); free(accs[t].img[g]); free(accs[t].img2[g]);
            }
        }
        free(accs[t].sed); free(accs[t].sed2); free(accs[t].img); free(accs[t].img2);
        tot.energy_current += accs[t].energy_current;
        tot.killed_geo += accs[t].killed_geo; tot.killed_int += accs[t].killed_int;
        tot.crossings += accs[t].crossings; tot.interactions += accs[t].interactions;
        if (accs[t].fatal && !fatal) { fatal = 1; snprintf(st->err, sizeof st->err, "%s", accs[t].err); }
    }
    tot.n_packets = n_packets;
    free(accs);
    *tot_out = tot;
    return fatal;
}

static void final_packet_fn(const orc_state *st, uint64_t id, acc_t *acc, const void *ctx) { (void)ctx; final_packet(st, id, acc); }

/* The final iteration in two halves, for the sharded runs of tests/test_distributed_cpu.py (mp_collect_images,
 * src/mpi/mpi_routines.f90:381-459): raw flux sums of the packet ids [first_id, first_id + n_local) into zeroed cubes, then --
 * after the cubes and the emitted energy have been summed over the ranks -- the scaling of image_type.f90:136-151. */
int orc_final_accumulate(orc_state *st, uint64_t first_id, uint64_t n_local, int n_threads, orc_iter_stats *stats)
{
    precompute_jnu_var(st); /* iter_final.f90:99 */
    if (st->cfg.mrw && prepare_mrw(st)) return 1;   /* iter_final.f90:93-96 */
    orc_iter_stats tot;
    if (image_run(st, n_local, n_threads, first_id, final_packet_fn, NULL, 1, &tot)) return 1;
    if (stats) *stats = tot;
    return 0;
}

int orc_final_scale(orc_state *st, double energy_current)
{
    /* peeled_images_adjust_scale(energy_total/energy_current): image_type.f90:136-151 */
    if (energy_current > 0.0) {
        double scale = st->energy_total / energy_current;
        for (int g = 0; g < st->n_groups; g++) {
            peeled_t *pg = &st->peeled[g];
            /* binned_images_adjust_scale :34-38: x n_theta x n_phi (flux per bin -> 4 pi normalisation of the peeled images) */
            const double sc = g == st->n_peeled ? scale * (double)st->n_theta * (double)st->n_phi : scale;
            if (pg->sed) for (size_t k = 0; k < pg->sed_size; k++) { pg->sed[k] *= sc; pg->sed2[k] *= sc * sc; }
            if (pg->img) for (size_t k = 0; k < pg->img_size; k++) { pg->img[k] *= sc; pg->img2[k] *= sc * sc; }
        }
    }
    return 0;
}

int orc_final_iteration(orc_state *st, uint64_t n_packets, int n_threads, orc_iter_stats *stats)
{
    orc_iter_stats tot;
    if (orc_final_accumulate(st, g_final_first_id, n_packets, n_threads, &tot)) return 1;
    orc_final_scale(st, tot.energy_current);
    if (stats) *stats = tot;
    return 0;
}

/* ------------------------------------------------------------------ */
/* do_raytracing: iter_raytracing.f90:30-143                            */
/* ------------------------------------------------------------------ */

typedef struct { uint64_t n_total; } ray_ctx;

/* source part :56-76: emit, weight energy_total / n_photons_sources, polychromatic peel-off */
static void ray_source_packet(const orc_state *st, uint64_t id, acc_t *acc, const void *ctx)
{
    const ray_ctx *c = ctx;
    rng_t g; photon_t p;
    rng_init(&g, st->cfg.seed, 0x20000u, id);
    if (emit(st, &p, &g, acc)) return;
    p.energy = p.energy * st->energy_total / (double)c->n_total;
    peeloff_photon(st, &p, &g, acc, 1);
}

/* random_position_cell of each geometry (cartesian_3d.f90:383-394, octree.f90:397-408, amr.f90:728-741) */
static int random_position_cell(const orc_state *st, size_t ic, photon_t *p, rng_t *g)
{
    double x = rng_uniform(g), y = rng_uniform(g), z = rng_uniform(g);
    if (st->grid_type == GRID_CAR) {
        int i1 = (int)(ic % st->n1), i2 = (int)((ic / st->n1) % st->n2), i3 = (int)(ic / ((size_t)st->n1 * st->n2));
        p->ic[0] = i1; p->ic[1] = i2; p->ic[2] = i3;
        p->r[0] = x * (st->w[0][i1 + 1] - st->w[0][i1]) + st->w[0][i1];
        p->r[1] = y * (st->w[1][i2 + 1] - st->w[1][i2]) + st->w[1][i2];
        p->r[2] = z * (st->w[2][i3 + 1] - st->w[2][i3]) + st->w[2][i3];
        return 0;
    }
    if (st->grid_type == GRID_SPH || st->grid_type == GRID_CYL) {   /* spherical_3d.f90:639-673, cylindrical_3d.f90:518-550 */
        int i1 = (int)(ic % st->n1), i2 = (int)((ic / st->n1) % st->n2), i3 = (int)(ic / ((size_t)st->n1 * st->n2));
        p->ic[0] = i1; p->ic[1] = i2; p->ic[2] = i3;
        const double *w1 = st->w[0], *w2 = st->w[1], *w3 = st->w[2];
        double rr, tz, ph = z * (w3[i3 + 1] - w3[i3]) + w3[i3];
        if (st->grid_type == GRID_SPH) {
            double a3 = w1[i1] * w1[i1] * w1[i1], b3 = w1[i1 + 1] * w1[i1 + 1] * w1[i1 + 1];
            rr = pow(x * (b3 - a3) + a3, 1.0 / 3.0);
            tz = acos(y * (st->wcost[i2 + 1] - st->wcost[i2]) + st->wcost[i2]);
        } else {
            double a2 = pow(w1[i1], 2.0), b2 = pow(w1[i1 + 1], 2.0);
            rr = sqrt(x * (b2 - a2) + a2);
            tz = y * (w2[i2 + 1] - w2[i2]) + w2[i2];
        }
        if (rr <= w1[i1] || rr >= w1[i1 + 1]) rr = 0.5 * (w1[i1] + w1[i1 + 1]);
        if (tz <= w2[i2] || tz >= w2[i2 + 1]) tz = 0.5 * (w2[i2] + w2[i2 + 1]);
        if (ph <= w3[i3] || ph >= w3[i3 + 1]) ph = 0.5 * (w3[i3] + w3[i3 + 1]);
        if (st->grid_type == GRID_SPH) { p->r[0] = rr * sin(tz) * cos(ph); p->r[1] = rr * sin(tz) * sin(ph); p->r[2] = rr * cos(tz); }
        else { p->r[0] = rr * cos(ph); p->r[1] = rr * sin(ph); p->r[2] = tz; }
        return 0;
    }
    p->ic[0] = (int)ic; p->ic[1] = p->ic[2] = 0;
    if (st->grid_type == GRID_OCT) {
        p->r[0] = (2.0 * x - 1.0) * st->odx[ic] + st->ox[ic];
        p->r[1] = (2.0 * y - 1.0) * st->ody[ic] + st->oy[ic];
        p->r[2] = (2.0 * z - 1.0) * st->odz[ic] + st->oz[ic];
        return 0;
    }
    if (st->grid_type == GRID_AMR) {
        const amr_grid *gr; int ci[3];
        amr_cell_coords(st, ic, &gr, ci);
        const double u[3] = {x, y, z};
        for (int a = 0; a < 3; a++) p->r[a] = u[a] * (gr->w[a][ci[a] + 1] - gr->w[a][ci[a]]) + gr->w[a][ci[a]];
        return 0;
    }
    if (st->grid_type == GRID_VOR && st->vbb) {
        /* grid_geometry_voronoi.f90:285-310: positions uniform in the cell's bounding box until one lies in the cell.  The
         * reference asks its kd-tree for the nearest site; a point is in cell ic iff no neighbour's site is closer. */
        const double *bb = st->vbb + 6 * ic;
        for (int trial = 0; trial < 1000000; trial++) {
            if (trial) { x = rng_uniform(g); y = rng_uniform(g); z = rng_uniform(g); }
            p->r[0] = bb[0] + x * (bb[3] - bb[0]); p->r[1] = bb[1] + y * (bb[4] - bb[1]); p->r[2] = bb[2] + z * (bb[5] - bb[2]);   /* random_uni */
            const double d0 = vdist2(st, (int32_t)ic, p->r);
            int inside = 1;
            for (int32_t k = st->vidx[ic]; k < st->vidx[ic + 1] && inside; k++)
                if (st->vneigh[k] >= 0 && vdist2(st, st->vneigh[k], p->r) < d0) inside = 0;
            if (inside) return 0;
        }
        return -1;   /* "too many samples" */
    }
    return -1;   /* voronoi without bounding boxes */
}

/* thermal part :96-126 with emit_from_grid (grid_physics_3d.f90:691-753) */
static void ray_dust_packet(const orc_state *st, uint64_t id, acc_t *acc, const void *ctx)
{
    const ray_ctx *c = ctx;
    rng_t g; photon_t p;
    rng_init(&g, st->cfg.seed, 0x30000u, id);
    memset(&p, 0, sizeof p);
    double xi = rng_uniform(&g);
    int d = (int)ceil(xi * (double)st->n_dust); if (d < 1) d = 1;
    p.dust_id = d - 1;
    xi = rng_uniform(&g);       /* random_masked_cell: grid_geometry_common_3d.f90:104-115 */
    long long im = (long long)ceil(xi * (double)st->n_masked); if (im < 1) im = 1;
    size_t ic = st->mask_map[im - 1];
    if (random_position_cell(st, ic, &p, &g)) {
        if (!acc->fatal) { acc->fatal = 1; snprintf(acc->err, sizeof acc->err, "raytracing of dust emission is not available for this grid type"); }
        return;
    }
    p.in_cell = 1;
    random_sphere_angle(&g, &p.a);
    angle_to_vector(&p.a, p.v);
    p.s[0] = 1.0;
    size_t k = (size_t)p.dust_id * st->n_cells + ic;
    if (st->energy_abs_tot[p.dust_id] > 0.0) {
        double mass = st->density[k] * st->volume[ic];
        p.energy = st->specific_energy[k] * mass * (double)st->n_masked / st->energy_abs_tot[p.dust_id];
    } else p.energy = 0.0;
    p.emiss_type = 3; p.emiss_var_id = st->jnu_var_id[k]; p.emiss_var_frac = st->jnu_var_frac[k];
    p.scattered = 0; p.reprocessed = 1; p.last_isotropic = 1; p.last = LAST_DE;
    g.countdown = rng_check_gap(&g, st->check_p, st->check_log1mp);
    if (p.energy > 0.0) {
        p.energy = p.energy * st->energy_abs_tot[p.dust_id] / (double)c->n_total * (double)st->n_dust;
        peeloff_photon(st, &p, &g, acc, 1);
    }
}

/* one id range of one part (which = 0 sources, 1 dust); zero_first clears the cubes before */
int orc_raytracing_accumulate(orc_state *st, int which, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first,
                              int n_threads, orc_iter_stats *stats)
{
    if (!st->cfg.raytracing) { snprintf(st->err, sizeof st->err, "raytracing was not requested in the configuration"); return 1; }
    if (which == 0) precompute_jnu_var(st);
    orc_iter_stats a; memset(&a, 0, sizeof a);
    ray_ctx c; c.n_total = n_total;
    if (zero_first)
        for (int g = 0; g < st->n_peeled; g++) {
            peeled_t *pg = &st->peeled[g];
            if (pg->sed) { memset(pg->sed, 0, sizeof(double) * pg->sed_size); memset(pg->sed2, 0, sizeof(double) * pg->sed_size); }
            if (pg->img) { memset(pg->img, 0, sizeof(double) * pg->img_size); memset(pg->img2, 0, sizeof(double) * pg->img_size); }
        }
    if (n_local > 0 && n_total > 0 && (which == 0 || st->n_dust > 0))
        if (image_run(st, n_local, n_threads, first_id, which == 0 ? ray_source_packet : ray_dust_packet, &c, 0, &a)) return 1;
    if (stats) *stats = a;
    return 0;
}

int orc_raytracing_iteration(orc_state *st, uint64_t n_sources, uint64_t n_dust, int n_threads, orc_iter_stats *stats)
{
    if (!st->cfg.raytracing) { snprintf(st->err, sizeof st->err, "raytracing was not requested in the configuration"); return 1; }
    precompute_jnu_var(st);   /* iter_raytracing.f90:49 */
    orc_iter_stats a, b; memset(&a, 0, sizeof a); memset(&b, 0, sizeof b);
    ray_ctx c;
    if (n_sources > 0) { c.n_total = n_sources; if (image_run(st, n_sources, n_threads, 0, ray_source_packet, &c, 0, &a)) return 1; }
    if (n_dust > 0 && st->n_dust > 0) { c.n_total = n_dust; if (image_run(st, n_dust, n_threads, 0, ray_dust_packet, &c, 0, &b)) return 1; }
    if (stats) {
        *stats = a;
        stats->killed_geo += b.killed_geo; stats->killed_int += b.killed_int; stats->crossings += b.crossings;
        stats->n_packets += b.n_packets;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* do_final_mono: iter_final_mono.f90:58-343, grid_monochromatic.f90    */
/* ------------------------------------------------------------------ */

/* dust_sample_emit_probability: dust_type_4elem.f90:356-377 */
static double dust_sample_emit_probability(const dust_t *d, int id, double frac, double nu)
{
    double p1 = pdf_interp_log(&d->j_nu[id], nu), p2 = pdf_interp_log(&d->j_nu[id + 1], nu);
    if (p1 == 0.0 || p2 == 0.0) return 0.0;
    double lp = log10(p1) + frac * (log10(p2) - log10(p1));
    return pow(10.0, lp);
}

/* setup_monochromatic_grid_pdfs :51-117: per dust type a discrete pdf over ALL cells, weight =
 * (emission probability at nu) x (energy emitted in the cell x n_cells / energy_abs_tot);
 * mean_prob = mean of the weights.  Returns 1 if nothing emits at this frequency. */
static int setup_monochromatic_grid_pdfs(orc_state *st, int inu)
{
    const double nu = st->frequencies[inu];
    const size_t nc = st->n_cells;
    if (!st->mono_cdf) st->mono_cdf = malloc(sizeof(double) * nc * (st->n_dust ? st->n_dust : 1));
    double total = 0.0;
    for (int d = 0; d < st->n_dust; d++) {
        double *cdf = st->mono_cdf + (size_t)d * nc;
        double sum = 0.0;
        for (size_t ic = 0; ic < nc; ic++) {
            size_t k = (size_t)d * nc + ic;
            double energy = 0.0;
            if (st->energy_abs_tot[d] > 0.0)
                energy = st->specific_energy[k] * st->density[k] * st->volume[ic] * (double)nc / st->energy_abs_tot[d];
            int valid = 1;
            if (st->grid_type == GRID_OCT) valid = !st->orefined[ic];
            else if (st->grid_type == GRID_AMR) valid = !amr_covered(st, ic);
            else if (st->grid_type == GRID_VOR) valid = st->volume[ic] > 0.0;
            if (!valid) energy = 0.0;
            double prob = dust_sample_emit_probability(&st->dust[d], st->jnu_var_id[k], st->jnu_var_frac[k], nu);
            sum += prob * energy;
            cdf[ic] = sum;          /* running sum; normalised below (set_pdf of a discrete pdf) */
        }
        st->mono_mean_prob[d] = sum / (double)nc;
        if (sum > 0.0) for (size_t ic = 0; ic < nc; ic++) cdf[ic] /= sum;
        total += st->mono_mean_prob[d];
    }
    st->mono_inu = inu;
    return total == 0.0;
}

/* propagate :232-341: like do_final's, with forced scattering (interact(force_scatter=.true.): the
 * packet keeps its frequency and loses (1 - albedo) of its energy) and the energy threshold */
static void mono_propagate(const orc_state *st, photon_t *p, rng_t *g, acc_t *acc)
{
    const double energy_initial = p->energy;
    for (int64_t inter = 1; inter <= st->cfg.n_inter_max + 1; inter++) {
        double tau;
        if (inter == 1 && st->cfg.forced_first_interaction) {
            int killed = 0;
            double tau_escape = grid_escape_tau(st, p, DBL_MAX, g, acc, &killed);
            if (tau_escape > 1e-10 && !killed) {
                double weight, xi = rng_uniform(g);
                if (st->cfg.forced_first_interaction_algorithm == 2)
                    forced_interaction_baes16(tau_escape, st->cfg.baes16_xi, xi, &tau, &weight);
                else
                    forced_interaction_wr99(tau_escape, xi, &tau, &weight);
                p->energy *= weight;
            } else tau = rng_exp(g);
        } else tau = rng_exp(g);
        grid_integrate(st, p, tau, g, acc, NULL);
        if (p->reabsorbed) {     /* :281-313 */
            int64_t ia;
            for (ia = 1; ia <= st->cfg.n_reabs_max; ia++) {
                int rid = p->reabsorbed_id; double re = p->energy; int inu = p->inu;
                if (emit_from_nu(st, p, g, acc, rid, re, inu)) return;
                if (st->n_peeled) peeloff_photon(st, p, g, acc, 0);
                tau = rng_exp(g);
                grid_integrate(st, p, tau, g, acc, NULL);
                if (!p->reabsorbed) break;
            }
            if (ia == st->cfg.n_reabs_max + 1) { acc->killed_int++; p->killed = 1; break; }
        }
        if (p->killed || escaped(st, p->ic)) break;
        if (inter == st->cfg.n_inter_max + 1) { acc->killed_int++; p->killed = 1; break; }
        /* interact(p, force_scatter=.true.): dust_interact.f90:22-79 with xi = 0 */
        {
            size_t ic = cell_index(st, p->ic);
            int id = 0;
            if (st->n_dust > 1) {
                double cdf[ORC_MAX_DUST], c = 0.0;
                for (int d = 0; d < st->n_dust; d++) { c += p->chi[d] * st->density[(size_t)d * st->n_cells + ic]; cdf[d] = c; }
                for (int d = 0; d < st->n_dust; d++) cdf[d] /= c;
                id = sample_discrete(cdf, st->n_dust, rng_uniform(g));
            }
            double albedo = p->albedo[id];
            p->a_prev = p->a; memcpy(p->v_prev, p->v, sizeof p->v); memcpy(p->s_prev, p->s, sizeof p->s);
            acc->interactions++;
            if (0.0 > albedo) { p->killed = 1; break; }     /* cannot happen: albedo >= 0 */
            dust_scatter(&st->dust[id], p->nu, &p->a, p->s, g);
            p->scattered = 1; p->last_isotropic = 0; p->dust_id = id; p->last = LAST_DS; p->n_scat++;
            angle_to_vector(&p->a, p->v);
            p->energy = p->energy * albedo;
        }
        p->killed = (st->cfg.kill_on_scatter && p->scattered) || (p->energy < energy_initial * st->cfg.monochromatic_energy_threshold);
        if (p->killed) break;
        if (st->n_peeled) peeloff_photon(st, p, g, acc, 0);
    }
}

typedef struct { uint64_t n_total; int inu; } mono_ctx;

/* source part :84-133 */
static void mono_source_packet(const orc_state *st, uint64_t id, acc_t *acc, const void *ctx)
{
    const mono_ctx *c = ctx;
    rng_t g; photon_t p;
    rng_init(&g, st->cfg.seed, 0x40000u + (uint32_t)c->inu, id);
    if (emit_from_nu(st, &p, &g, acc, -1, 0.0, c->inu)) return;
    p.energy = p.energy / (double)c->n_total;
    if (st->n_peeled && !st->cfg.raytracing) peeloff_photon(st, &p, &g, acc, 0);
    mono_propagate(st, &p, &g, acc);
}

/* dust part :146-217 with emit_from_monochromatic_grid_pdf (grid_monochromatic.f90:119-174) */
static void mono_dust_packet(const orc_state *st, uint64_t id, acc_t *acc, const void *ctx)
{
    const mono_ctx *c = ctx;
    rng_t g; photon_t p;
    rng_init(&g, st->cfg.seed, 0x50000u + (uint32_t)c->inu, id);
    memset(&p, 0, sizeof p);
    p.nu = st->frequencies[c->inu]; p.inu = c->inu;
    if (update_optconsts(st, &p, acc)) return;
    double xi = rng_uniform(&g);
    int d = (int)ceil(xi * (double)st->n_dust); if (d < 1) d = 1;
    d -= 1;
    if (st->mono_mean_prob[d] == 0.0) return;
    /* grid_sample_pdf_map: sample_pdf of the discrete pdf = first cell whose cumulative exceeds xi */
    const double *cdf = st->mono_cdf + (size_t)d * st->n_cells;
    xi = rng_uniform(&g);
    size_t lo = 0, hi = st->n_cells - 1;
    while (lo < hi) { size_t mid = (lo + hi) >> 1; if (xi < cdf[mid]) hi = mid; else lo = mid + 1; }
    size_t ic = lo;
    if (random_position_cell(st, ic, &p, &g)) {
        if (!acc->fatal) { acc->fatal = 1; snprintf(acc->err, sizeof acc->err, "monochromatic dust emission is not available for this grid type"); }
        return;
    }
    p.in_cell = 1;
    random_sphere_angle(&g, &p.a);
    angle_to_vector(&p.a, p.v);
    p.s[0] = 1.0;
    p.energy = st->mono_mean_prob[d];
    p.scattered = 0; p.reprocessed = 1; p.last_isotropic = 1; p.dust_id = d; p.last = LAST_DE;
    p.a_prev = p.a; memcpy(p.s_prev, p.s, sizeof p.s); memcpy(p.v_prev, p.v, sizeof p.v);
    g.countdown = rng_check_gap(&g, st->check_p, st->check_log1mp);
    if (p.energy > 0.0) {
        p.energy = p.energy * st->energy_abs_tot[d] / (double)c->n_total * (double)st->n_dust;
        if (st->n_peeled && !st->cfg.raytracing) peeloff_photon(st, &p, &g, acc, 0);
        mono_propagate(st, &p, &g, acc);
    }
}

int orc_mono_accumulate(orc_state *st, int which, int inu, uint64_t first_id, uint64_t n_local, uint64_t n_total, int zero_first,
                        int n_threads, orc_iter_stats *stats)
{
    if (!st->cfg.monochromatic) { snprintf(st->err, sizeof st->err, "monochromatic mode was not requested in the configuration"); return 1; }
    if (inu < 0 || inu >= st->cfg.n_frequencies) { snprintf(st->err, sizeof st->err, "incorrect inu"); return 1; }
    orc_iter_stats a; memset(&a, 0, sizeof a);
    if (zero_first)
        for (int g = 0; g < st->n_peeled; g++) {
            peeled_t *pg = &st->peeled[g];
            if (pg->sed) { memset(pg->sed, 0, sizeof(double) * pg->sed_size); memset(pg->sed2, 0, sizeof(double) * pg->sed_size); }
            if (pg->img) { memset(pg->img, 0, sizeof(double) * pg->img_size); memset(pg->img2, 0, sizeof(double) * pg->img_size); }
        }
    if (inu == 0) precompute_jnu_var(st);    /* iter_final_mono.f90:79 (cheap; the sharded callers have no other hook) */
    mono_ctx c; c.n_total = n_total; c.inu = inu;
    if (which == 1) {
        if (st->n_dust == 0 || setup_monochromatic_grid_pdfs(st, inu)) { if (stats) *stats = a; return 0; }   /* "No emission at this frequency" */
    }
    if (n_local > 0 && n_total > 0)
        if (image_run(st, n_local, n_threads, first_id, which == 0 ? mono_source_packet : mono_dust_packet, &c, 0, &a)) return 1;
    if (stats) *stats = a;
    return 0;
}

int orc_mono_iteration(orc_state *st, uint64_t n_sources, uint64_t n_dust, int n_threads, orc_iter_stats *stats)
{
    if (!st->cfg.monochromatic) { snprintf(st->err, sizeof st->err, "monochromatic mode was not requested in the configuration"); return 1; }
    precompute_jnu_var(st);    /* iter_final_mono.f90:79 */
    orc_iter_stats tot, a; memset(&tot, 0, sizeof tot);
    int first = 1;
    for (int which = 0; which < 2; which++) {
        uint64_t n = which == 0 ? n_sources : n_dust;
        for (int inu = 0; inu < st->cfg.n_frequencies; inu++) {
            if (orc_mono_accumulate(st, which, inu, 0, n, n, first, n_threads, &a)) return 1;
            first = 0;
            tot.energy_current += a.energy_current; tot.killed_geo += a.killed_geo; tot.killed_int += a.killed_int;
            tot.crossings += a.crossings; tot.interactions += a.interactions; tot.n_packets += a.n_packets;
        }
    }
    if (stats) *stats = tot;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Probes for unit tests                                               */
/* ------------------------------------------------------------------ */

int orc_walk_ray(const orc_state *st, const double r0[3], const double v[3], double *path_out)
{
    photon_t p; acc_t acc; memset(&p, 0, sizeof p); memset(&acc, 0, sizeof acc);
    memcpy(p.r, r0, sizeof p.r); memcpy(p.v, v, sizeof p.v);
    place_in_cell(st, &p, &acc);
    if (p.killed) return -1;
    int n = 0; double path = 0.0;
    while (!escaped(st, p.ic)) {
        double tmin; int id_min[3];
        if (find_wall(st, &p, &tmin, id_min) <= 0) return -2;
        for (int a = 0; a < 3; a++) p.r[a] = p.r[a] + tmin * p.v[a];
        advance_cell(st, &p, id_min);
        path += tmin; n++;
        if (n > 100000000) return -3;
    }
    if (path_out) *path_out = path;
    return n;
}

double orc_probe_uniform(int64_t seed, int iter, uint64_t packet_id, int k)
{
    rng_t g; rng_init(&g, seed, (uint32_t)iter, packet_id);
    double x = 0; for (int i = 0; i <= k; i++) x = rng_uniform(&g);
    return x;
}

void orc_probe_scatter(const orc_state *st, int dust, double nu, const double a_in[4], const double s_in[4],
                       int64_t seed, uint64_t packet_id, double a_out[4], double s_out[4])
{
    rng_t g; rng_init(&g, seed, 1u, packet_id);
    angle_t a = {a_in[0], a_in[1], a_in[2], a_in[3]};
    double s[4] = {s_in[0], s_in[1], s_in[2], s_in[3]};
    dust_scatter(&st->dust[dust], nu, &a, s, &g);
    a_out[0] = a.cost; a_out[1] = a.sint; a_out[2] = a.cosp; a_out[3] = a.sinp;
    memcpy(s_out, s, sizeof s);
}

double orc_probe_sample_jnu(const orc_state *st, int dust, int jid, double frac, double xi)
{
    return dust_sample_j_nu(&st->dust[dust], jid, frac, xi);
}

double orc_probe_planck(double T, int64_t seed, uint64_t packet_id)
{
    rng_t g; rng_init(&g, seed, 1u, packet_id);
    return random_planck_frequency(&g, T);
}

void orc_probe_optconsts(const orc_state *st, int dust, double nu, double out3[3])
{
    const dust_t *du = &st->dust[dust];
    out3[0] = interp1d_loglog(du->nu, du->chi, du->n_nu, nu);
    out3[1] = interp1d_loglog(du->nu, du->albedo, du->n_nu, nu);
    out3[2] = out3[0] * (1.0 - out3[1]);
}

/* rotate_angle followed by difference_angle (must return the local angle) */
void orc_probe_rotate(const double loc_in[4], const double co_in[4], double fin_out[4], double loc_back[4])
{
    angle_t loc = {loc_in[0], loc_in[1], loc_in[2], loc_in[3]}, co = {co_in[0], co_in[1], co_in[2], co_in[3]}, fin, back;
    rotate_angle(&loc, &co, &fin);
    difference_angle(&co, &fin, &back);
    fin_out[0] = fin.cost; fin_out[1] = fin.sint; fin_out[2] = fin.cosp; fin_out[3] = fin.sinp;
    loc_back[0] = back.cost; loc_back[1] = back.sint; loc_back[2] = back.cosp; loc_back[3] = back.sinp;
}
