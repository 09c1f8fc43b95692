// hyp_kernels.h -- HIP kernels of the photon-packet engine (gfx950 / MI355X).
//
// Mapping: ONE PACKET PER LANE, 64 packets per wavefront, persistent waves that
// pull packet ids from a global dispenser in chunks.  Lanes of a wave are in one
// of three phases (walk / interact / emit); the expensive, rare phases
// (interaction, emission) are deferred until enough lanes of the wave wait for
// them (wave ballot + popcount) so that the hot cell-walk loop runs with little
// divergence.  Grid walls live in LDS; density / accumulators / dust tables in
// HBM (L2 + Infinity-Cache resident at 128^3).  Energy deposits are FP64
// hardware atomics (global_atomic_add_f64) into one of `n_copies` accumulator
// replicas selected by the XCD the workgroup runs on.
#pragma once

#include "hyp_device.h"

enum { ST_NEED_EMIT = 0, ST_WALK = 1, ST_NEED_INTERACT = 2, ST_DONE = 3, ST_PLACED = 4, ST_NEED_REEMIT = 5, ST_MRW = 6,
       ST_ESCAPED = 7 };      // imaging iteration only: left the grid alive (binned images), then ST_NEED_EMIT
enum { LAST_SR = 0, LAST_DS = 1, LAST_DE = 2 };

#define GEOM_CAR 0
#define GEOM_OCT 1
#define GEOM_VOR 2
#define GEOM_AMR 3
#define GEOM_SPH 4
#define GEOM_CYL 5

// Where a packet is: Cartesian = three cell indices; octree = cell id plus a
// register copy of the leaf's record (centre, level, parent, sub-cell).
template <int GEOM> struct Cell;
template <> struct Cell<GEOM_CAR> { int ic[3], ow[3]; };
template <> struct Cell<GEOM_OCT> { int id, ow[3]; double c[3]; int parent, level, subcell; };
template <> struct Cell<GEOM_VOR> { int id, ow[3]; };   // ow[1] = -(previous cell + 1)
template <> struct Cell<GEOM_SPH> { int ic[3], ow[3], radial; };      // radial: (r.v) > 0 at the start of the integration (find_wall skips the inner sphere)
template <> struct Cell<GEOM_CYL> { int ic[3], ow[3]; };
template <> struct Cell<GEOM_AMR> { int id, ow[3], grid, i[3]; };   // id = unique cell id (n_cells: outside, -1: invalid); i = 0-based position in the grid

template <int NDT, int GEOM>
struct Packet {
    double r[3], v[3];          // position, direction
    Angle a;
    double s[4];
    double nu, energy;
    double tau_req, tau_ach;
    double chi[NDT], albedo[NDT], kappa[NDT];
    Cell<GEOM> cell;
    int inter;
    int emiss_dust;             // 'lte' sources: dust type whose emissivity the packet was emitted with (-1: the source's own spectrum)
    // re-absorption by sources (grid_propagate_3d.f90:99-101,139-143): distance to the nearest
    // intersecting source along v, distance covered in this grid_integrate, that source, and the
    // number of successive re-emissions (iter_lucy.f90:155-185)
    double t_src, t_ach;
    int reabs_id, reabs;
    int spec_idx;               // frequency bin of the specific-energy spectrum during this grid_integrate (-1: none)
    unsigned int peel_seq;      // peel-off events of this packet so far (keys the check stream of the peel-off walks)
    int n_visited;              // cells this packet has been counted in (count_photon)
    double e_init;              // monochromatic launches on the deferred schedule: energy at emission (iter_final_mono.f90:247)
};

// extra state carried only by the imaging (final) iteration
struct PeelState {
    Angle a_prev;
    double s_prev[4], v_prev[3];
    int last, last_isotropic, scattered, reprocessed, n_scat, dust_id, source_id;
};

struct Counters {
    double energy_current;
    unsigned long long crossings;
    unsigned int killed_geo, killed_int, interactions;
};

template <int NDT>
__device__ __forceinline__ int ndust(const DProblem &P) { return NDT == HYP_MAXD ? P.n_dust : NDT; }

__device__ __forceinline__ void raise_error(const DProblem &P, int code, double d0, double d1, double d2)
{
    if (atomicCAS(P.err, 0, code) == 0) { P.err_data[0] = d0; P.err_data[1] = d1; P.err_data[2] = d2; }
}

// dust.f90:64-79
template <int NDT, int GEOM>
__device__ __forceinline__ bool update_optconsts(const DProblem &P, Packet<NDT, GEOM> &p)
{
    const int nd = ndust<NDT>(P);
    double lnu = log10(p.nu);
#pragma unroll
    for (int d = 0; d < NDT; d++) {
        if (d < nd) {
            const DDust &D = P.dust[d];
            if (p.nu < D.nu_min || p.nu > D.nu_max) {
                raise_error(P, ERR_NU_RANGE, p.nu, D.nu_min, D.nu_max);
                return false;
            }
            int j = locate_g(D.nu, D.n_nu, p.nu);
            double chi = interp_loglog_at(D.nu, D.log10_nu, D.chi, D.log10_chi, j, p.nu, lnu);
            double alb = interp_loglog_at(D.nu, D.log10_nu, D.albedo, D.log10_albedo, j, p.nu, lnu);
            p.chi[d] = chi; p.albedo[d] = alb; p.kappa[d] = chi * (1.0 - alb);
        }
    }
    return true;
}

struct Walls {            // LDS-resident copies of the wall tables (Cartesian)
    const double *w[3];
    const double *ew[3];
    int n[3];
};

// ----------------------------- Cartesian -----------------------------------
__device__ __forceinline__ bool geo_escaped(const DProblem &P, const Cell<GEOM_CAR> &c)
{
    return c.ic[0] < 0 || c.ic[0] >= P.n1 || c.ic[1] < 0 || c.ic[1] >= P.n2 || c.ic[2] < 0 || c.ic[2] >= P.n3;
}

__device__ __forceinline__ size_t geo_index(const DProblem &P, const Cell<GEOM_CAR> &c)
{
    return ((size_t)c.ic[2] * P.n2 + c.ic[1]) * P.n1 + c.ic[0];
}

// grid_geometry_cartesian_3d.f90:143-166
__device__ __forceinline__ bool find_cell_car(const Walls &W, const double r[3], int ic[3])
{
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int i = locate(W.w[a], W.n[a] + 1, r[a]);
        if (i < 0 || i >= W.n[a]) return false;
        ic[a] = i;
    }
    return true;
}

// place_in_cell + adjust_wall, grid_geometry_cartesian_3d.f90:168-253
__device__ __forceinline__ bool geo_place(const DProblem &P, const Walls &W, const double r[3], const double v[3], Cell<GEOM_CAR> &c)
{
    if (!find_cell_car(W, r, c.ic)) return false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        c.ow[a] = 0;
        int i = c.ic[a];
        double wl = W.w[a][i], wu = W.w[a][i + 1];
        if (v[a] > 0.0) {
            if (r[a] == wl) c.ow[a] = -1;
            else if (r[a] == wu) { c.ow[a] = -1; c.ic[a] = i + 1; }
        } else if (v[a] < 0.0) {
            if (r[a] == wl) { c.ow[a] = +1; c.ic[a] = i - 1; }
            else if (r[a] == wu) c.ow[a] = +1;
        }
    }
    return true;
}

// grid_geometry_cartesian_3d.f90:330-381.  The reference locates the position in the three wall arrays (find_cell) and
// compares the cell it finds with the packet's; only "is it the packet's cell" is used, so the three binary searches
// (7 dependent reads each at 128 cells) are replaced by the two walls of the packet's own cell: locate() returns i exactly
// when w[i] <= r < w[i + 1], or r sits on the last wall and i is the last cell; `found` = inside the grid on every axis
// (find_cell stops at the first axis outside: the result is then false whatever the other axes say, as here).
__device__ __forceinline__ bool geo_in_correct_cell(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_CAR> &c)
{
    bool found = true, same[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int i = c.ic[a], n = W.n[a];
        const double lo = W.w[a][0], hi = W.w[a][n];
        found = found && (r[a] >= lo) && (r[a] <= hi);
        const bool inside = i >= 0 && i < n;          // a packet that has left the grid is not asked
        const double wl = W.w[a][inside ? i : 0], wu = W.w[a][inside ? i + 1 : 1];
        same[a] = inside && ((r[a] >= wl && r[a] < wu) || (r[a] == hi && i == n - 1));
    }
    const double thr = 1e-3;
    if (c.ow[0] | c.ow[1] | c.ow[2]) {
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            int i = c.ic[a];
            double wl = W.w[a][i], wu = W.w[a][i + 1];
            if (c.ow[a] == -1) ok = ok && fabs((r[a] - wl) / (wu - wl)) < thr;
            else if (c.ow[a] == +1) ok = ok && fabs((r[a] - wu) / (wu - wl)) < thr;
            else ok = ok && found && same[a];
        }
        return ok;
    }
    return found && same[0] && same[1] && same[2];
}

// find_wall + insert_t, grid_geometry_cartesian_3d.f90:424-521.  The six
// candidate distances of the reference are (w - r)/v for the lower and upper
// wall of each axis, kept only when positive; a quotient is positive exactly
// when numerator and denominator have the same sign, so the (IEEE) divide is
// issued only for candidates that can win (normally one per axis).  The walk
// arithmetic is kept bit-identical to the reference formulation (true division,
// no FMA contraction: the library is built with -ffp-contract=off) because
// degenerate set-ups -- sources on cell vertices, views along cell diagonals --
// sit on knife edges that 1-ulp differences tip over.
__device__ __forceinline__ bool geo_find_wall(const DProblem &P, const Walls &W, const double r[3], const double v[3],
                                              const Cell<GEOM_CAR> &c, double &tnear, int im[3])
{
    double tmin = HYP_DBL_MAX, emin = 0.0;
    im[0] = im[1] = im[2] = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        int i = c.ic[a];
        double wl = W.w[a][i], wu = W.w[a][i + 1];
        double d1 = wl - r[a], d2 = wu - r[a], va = v[a];
        bool c1 = (c.ow[a] != -1) && ((d1 > 0.0 && va > 0.0) || (d1 < 0.0 && va < 0.0));
        bool c2 = (c.ow[a] != +1) && ((d2 > 0.0 && va > 0.0) || (d2 < 0.0 && va < 0.0));
        if (c1 || c2) {
            double t = (c1 ? d1 : d2) / va;
            double e = W.ew[a][i + (c1 ? 0 : 1)];
            int dir = c1 ? -1 : +1;
            double emax = fmax(e, emin);
            if (t < tmin - emax) { tmin = t; im[0] = im[1] = im[2] = 0; emin = emax; im[a] = dir; }
            else if (t < tmin + emax) { emin = emax; im[a] = dir; }
            if (c1 && c2) {   // both walls ahead: only after round-off misplacement
                t = d2 / va; e = W.ew[a][i + 1];
                emax = fmax(e, emin);
                if (t < tmin - emax) { tmin = t; im[0] = im[1] = im[2] = 0; emin = emax; im[a] = +1; }
                else if (t < tmin + emax) { emin = emax; im[a] = +1; }
            }
        }
    }
    tnear = tmin;
    return (im[0] | im[1] | im[2]) != 0;
}

// next_cell_wall_id :303-328 + opposite_wall
__device__ __forceinline__ void geo_advance(const DProblem &P, const double r[3], Cell<GEOM_CAR> &c, const int im[3])
{
#pragma unroll
    for (int a = 0; a < 3; a++) { c.ic[a] += im[a]; c.ow[a] = -im[a]; }
}

// max of two values neither of which is a NaN: ONE v_max_f64.  (fmax() costs three: the compiler first makes each operand
// "canonical" -- v_max_f64 x, x, x -- because a signalling NaN would have to be quieted.)
__device__ __forceinline__ double max_no_nan(double a, double b)
{
    double m;
    asm("v_max_f64 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m;
}

// find_wall for the common case that, on every axis, the wall *behind* the packet is not a
// candidate of geo_find_wall (it is one only if round-off left the packet outside its cell on
// an axis where it is not flagged as sitting on that wall).  Then at most the wall ahead is a
// candidate on each axis, with exactly geo_find_wall's condition, and the candidates are merged
// in the same order with the same epsilon rules.  Written without control flow: the wall ahead
// is picked by index arithmetic (iu = 1 where v > 0), so that the three quotients can be
// scheduled together and no lane-divergent branch is left in the step.  Returns false when the
// precondition fails; the caller then uses geo_find_wall.
//
// The three quotients d / v are formed with the reciprocals inv = RN(1 / v) that the lane
// computed (with a true IEEE division) when it took the packet -- the direction is fixed during
// a visit: q0 = RN(d inv), rem = d - q0 v (exact in one FMA), t = RN(q0 + rem inv).  By
// Markstein's theorem (IBM J. Res. Dev. 34, 1990; Muller et al., Handbook of Floating-Point
// Arithmetic, 2nd ed., Thm 4.8) t is then the correctly rounded quotient RN(d / v), i.e.
// bit-for-bit the IEEE division of the reference formulation, for 3 instructions instead of
// the ~14 of a division (div_scale x2, rcp, 8 FMA steps, div_fmas, div_fixup).  The theorem
// needs no underflow/overflow in q0, rem and inv: the caller enables this path (v_ok) only when
// every non-zero direction component is at least 2^-400 in magnitude, the distances d are
// differences of wall and position coordinates (zero or >= one ulp of a coordinate), and an axis
// with v = 0 is never a candidate, so its inf/NaN quotient is not looked at.
// tools/ubench/markstein_check.c compares the sequence with the division on 4e8 operand pairs
// including all-ones / near-power-of-two / short mantissas: no mismatch.
//
// The sign tests of the reference -- a wall is a candidate when (w - r) and v have the same sign -- cost no arithmetic of
// their own (round 5; they were two FP64 products with sgn(v) per axis):
//   * wall AHEAD: d and v have the same sign and are both non-zero exactly when the quotient t = RN(d / v) is > 0: the exact
//     quotient is then positive and at least |d| (|v| <= 1: no underflow to zero); opposite signs give t <= -0, d = 0 gives
//     t = +-0, and v = 0 gives inv = inf, rem = NaN, t = NaN -- every comparison with it false, as the reference's `v > 0`.
//   * wall BEHIND: db with its sign flipped where v <= 0 (one XOR of the high word with smask, the sign bit where v <= 0) is
//     > 0 exactly when db and v have the same sign; where v = 0 the flipped value is > 0 only for a packet beyond the upper
//     wall of its cell, which then takes the general search like any packet outside its cell (the same result, slower).
// The first axis meets tmin = DBL_MAX, emin = 0: t0 < DBL_MAX -+ e0 holds for every finite t0 (hyp_create refuses walls beyond
// 2^300, and |1 / v| <= 2^400 under v_ok, so t0 < 2^701), and max(e0, 0) = e0 (an epsilon is 3 x spacing(w) > 0), so its
// candidate is taken without the two sums and comparisons.
__device__ __forceinline__ bool find_wall_ahead(const Walls &W, const double r[3], const double v[3], const double inv[3],
                                                const int iu[3], const int smask[3],
                                                const Cell<GEOM_CAR> &c, double &tnear, int im[3], bool &found)
{
    double tmin = HYP_DBL_MAX, emin = 0.0;
    int m0 = 0, m1 = 0, m2 = 0;
    bool simple = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int ia = c.ic[a] + iu[a], ib = c.ic[a] + 1 - iu[a];
        const double d = W.w[a][ia] - r[a], db = W.w[a][ib] - r[a];
        const int dir = 2 * iu[a] - 1, ow = c.ow[a];
        const double q0 = d * inv[a];
        const double t = __builtin_fma(__builtin_fma(-q0, v[a], d), inv[a], q0);
        // wall ahead: c2 = (ow != +1) && d2 > 0 for v > 0;  c1 = (ow != -1) && d1 < 0 for v < 0
        const bool cand = (ow != dir) & (t > 0.0);
        // wall behind: c1 = (ow != -1) && d1 > 0 for v > 0;  c2 = (ow != +1) && d2 < 0 for v < 0
        const double db_s = __hiloint2double(__double2hiint(db) ^ smask[a], __double2loint(db));
        simple = simple & !((ow != -dir) & (db_s > 0.0));
        const double e = W.ew[a][ia];
        if (a == 0) {
            tmin = cand ? t : tmin;
            emin = cand ? e : emin;
            m0 = cand ? dir : 0;
        } else {
            const double emax = max_no_nan(e, emin);
            const bool lt = cand & (t < tmin - emax);
            const bool any = cand & (t < tmin + emax);          // lt or within epsilon of the current minimum
            tmin = lt ? t : tmin;
            emin = any ? emax : emin;
            const int mine = any ? dir : 0;
            if (a == 1) { m0 = lt ? 0 : m0; m1 = mine; }
            if (a == 2) { m0 = lt ? 0 : m0; m1 = lt ? 0 : m1; m2 = mine; }
        }
    }
    tnear = tmin;
    im[0] = m0; im[1] = m1; im[2] = m2;
    found = (m0 | m1 | m2) != 0;
    return simple;
}

// geo_find_wall for a direction that stays fixed over many steps (a peel-off walk, a packet between two interactions), inv =
// RN(1 / v) per axis computed once (the caller checks v_ok: every non-zero component at least 2^-400 in magnitude): the branch-free
// search above where it applies, geo_find_wall otherwise -- the same wall, the same t, bit for bit.
__device__ __forceinline__ bool car_find_wall_inv(const DProblem &P, const Walls &W, const double r[3], const double v[3], const double inv[3],
                                                  const Cell<GEOM_CAR> &c, double &tnear, int im[3])
{
    int iu[3], smask[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { iu[a] = v[a] > 0.0 ? 1 : 0; smask[a] = v[a] > 0.0 ? 0 : (int)0x80000000; }
    bool found;
    if (find_wall_ahead(W, r, v, inv, iu, smask, c, tnear, im, found)) return found;
    return geo_find_wall(P, W, r, v, c, tnear, im);
}

// ------------------------------- octree -------------------------------------
// grid_geometry_octree.f90.  Cell records are 32 B (centre, parent, sub-cell,
// level, refined); half-widths are root half-width * 2^-level (exact).
__device__ __forceinline__ void oct_load(const DProblem &P, int id, Cell<GEOM_OCT> &c)
{
    const OctCell &o = P.oct_cells[id];
    c.id = id; c.c[0] = o.x; c.c[1] = o.y; c.c[2] = o.z;
    c.parent = o.parent; c.level = o.level; c.subcell = o.subcell;
}

// locate_cell :135-146: descend from `id` to the leaf containing r
__device__ __forceinline__ int oct_locate(const DProblem &P, const double r[3], int id)
{
    for (;;) {
        const OctCell &o = P.oct_cells[id];
        if (!o.refined) return id;
        int sub = (r[0] < o.x ? 0 : 1) | (r[1] < o.y ? 0 : 2) | (r[2] < o.z ? 0 : 4);
        id = P.oct_children[8 * (size_t)id + sub];
    }
}

// locate_cell from `id` + the record of the leaf, one 32-B record per level
__device__ __forceinline__ void oct_descend(const DProblem &P, const double r[3], int id, Cell<GEOM_OCT> &c)
{
    for (;;) {
        const OctCell o = oct_cell_ldg(P.oct_cells + id);
        if (!o.refined) {
            c.id = id; c.c[0] = o.x; c.c[1] = o.y; c.c[2] = o.z;
            c.parent = o.parent; c.level = o.level; c.subcell = o.subcell;
            return;
        }
        int sub = (r[0] < o.x ? 0 : 1) | (r[1] < o.y ? 0 : 2) | (r[2] < o.z ? 0 : 4);
        id = hyp_ldg(P.oct_children + 8 * (size_t)id + sub);
    }
}

__device__ __forceinline__ bool geo_escaped(const DProblem &P, const Cell<GEOM_OCT> &c) { return (unsigned long long)c.id == P.n_cells; }
__device__ __forceinline__ size_t geo_index(const DProblem &P, const Cell<GEOM_OCT> &c) { return (size_t)c.id; }

// find_cell :260-283 + place_in_cell :285-296 (no wall adjustment in the octree)
__device__ __forceinline__ bool geo_place(const DProblem &P, const Walls &W, const double r[3], const double v[3], Cell<GEOM_OCT> &c)
{
    if (r[0] < P.oct_box[0] || r[0] > P.oct_box[1] || r[1] < P.oct_box[2] || r[1] > P.oct_box[3] ||
        r[2] < P.oct_box[4] || r[2] > P.oct_box[5]) return false;
    oct_descend(P, r, 0, c);
    return true;
}

// :366-392
__device__ __forceinline__ bool geo_in_correct_cell(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_OCT> &c)
{
    if (c.ow[0] | c.ow[1] | c.ow[2]) {
        double f[3];
#pragma unroll
        for (int a = 0; a < 3; a++) f[a] = fabs(r[a] - c.c[a]) / ldexp(P.oct_half[a], -c.level);
        double frac = 0.0; bool ok = false;
        if (c.ow[0] != 0) { frac = f[0] - 1.0; ok = f[1] < 1.0 && f[2] < 1.0; }
        if (c.ow[1] != 0) { frac = f[1] - 1.0; ok = f[0] < 1.0 && f[2] < 1.0; }
        if (c.ow[2] != 0) { frac = f[2] - 1.0; ok = f[0] < 1.0 && f[1] < 1.0; }
        return fabs(frac) < 1e-3 && ok;
    }
    bool inside = !(r[0] < P.oct_box[0] || r[0] > P.oct_box[1] || r[1] < P.oct_box[2] || r[1] > P.oct_box[3] ||
                    r[2] < P.oct_box[4] || r[2] > P.oct_box[5]);
    return inside && oct_locate(P, r, 0) == c.id;
}

// find_wall :438-537: nearest of the three faces ahead
__device__ __forceinline__ bool geo_find_wall(const DProblem &P, const Walls &W, const double r[3], const double v[3],
                                              const Cell<GEOM_OCT> &c, double &tnear, int im[3])
{
    double t[3]; bool pos[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double h = ldexp(P.oct_half[a], -c.level);
        pos[a] = v[a] > 0.0;
        if (pos[a]) t[a] = (c.c[a] + h - r[a]) / v[a];
        else if (v[a] < 0.0) t[a] = (c.c[a] - h - r[a]) / v[a];
        else t[a] = HYP_DBL_MAX;
    }
    im[0] = im[1] = im[2] = 0;
    int a;
    if (t[0] < t[2]) a = (t[0] < t[1]) ? 0 : 1;
    else a = (t[2] < t[1]) ? 2 : 1;
    double tmin = a == 0 ? t[0] : a == 1 ? t[1] : t[2];
    bool up = a == 0 ? pos[0] : a == 1 ? pos[1] : pos[2];
    if (a == 0) im[0] = up ? 1 : -1; else if (a == 1) im[1] = up ? 1 : -1; else im[2] = up ? 1 : -1;
    if (tmin < 0.0) {
        if (tmin > -10.0 * P.oct_eps) tmin = 0.0;
        else { im[0] = im[1] = im[2] = 0; }
    }
    tnear = tmin;
    return (im[0] | im[1] | im[2]) != 0;
}

// find_wall :438-537 with the three quotients (wall - r) / v formed from the reciprocals inv = RN(1 / v) of a direction that
// stays fixed over many steps (a peel-off walk): q0 = RN(d inv), rem = d - q0 v (exact in one FMA), t = RN(q0 + rem inv) is
// the correctly rounded quotient RN(d / v) (Markstein; see find_wall_ahead in hyp_tiled.h for the conditions), i.e. bit for
// bit the IEEE division of geo_find_wall, for 3 instructions instead of ~14.  v_ok = every non-zero component of v is at
// least 2^-400 in magnitude (else the caller's lane takes geo_find_wall).
__device__ __forceinline__ bool oct_find_wall_inv(const DProblem &P, const double r[3], const double v[3], const double inv[3],
                                                  const Cell<GEOM_OCT> &c, double &tnear, int im[3])
{
    double t[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double h = ldexp(P.oct_half[a], -c.level);
        const double wall = v[a] > 0.0 ? c.c[a] + h : c.c[a] - h;
        const double d = wall - r[a];
        const double q0 = d * inv[a];
        const double tq = __builtin_fma(__builtin_fma(-q0, v[a], d), inv[a], q0);
        t[a] = v[a] == 0.0 ? HYP_DBL_MAX : tq;
    }
    im[0] = im[1] = im[2] = 0;
    int a;
    if (t[0] < t[2]) a = (t[0] < t[1]) ? 0 : 1;
    else a = (t[2] < t[1]) ? 2 : 1;
    double tmin = a == 0 ? t[0] : a == 1 ? t[1] : t[2];
    const double va = a == 0 ? v[0] : a == 1 ? v[1] : v[2];
    const int dir = va > 0.0 ? 1 : -1;
    if (a == 0) im[0] = dir; else if (a == 1) im[1] = dir; else im[2] = dir;
    if (tmin < 0.0) {
        if (tmin > -10.0 * P.oct_eps) tmin = 0.0;
        else { im[0] = im[1] = im[2] = 0; }
    }
    tnear = tmin;
    return (im[0] | im[1] | im[2]) != 0;
}

// next_cell_int :328-347 (climb until a sibling exists on that side, then
// descend to the leaf that contains the intersection point) + opposite_wall
__device__ __forceinline__ void geo_advance(const DProblem &P, const double r[3], Cell<GEOM_OCT> &c, const int im[3])
{
    const int axis = im[0] ? 0 : im[1] ? 1 : 2;
    const int up = (im[0] + im[1] + im[2]) > 0 ? 1 : 0;
    c.ow[0] = -im[0]; c.ow[1] = -im[1]; c.ow[2] = -im[2];
    // The climb and the descent are chains of dependent loads whose length differs from lane to lane (the wave waits
    // for the longest: up to 2 x depth L2 round trips per crossing).  oct_neigh holds, for every cell and face, where that
    // climb-and-descend ends when it is stopped at the cell's own level (or at a leaf above it): the descent from there
    // makes the same comparisons as the one from the sibling as long as r lies inside the cell's extent on the two
    // other axes.  Within 1e-6 of an edge (round-off may have put r in the neighbour's column) take the reference's route.
    if (P.oct_neigh) {
        bool fast = true;
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const double h = ldexp(P.oct_half[b], -c.level), d = fabs(r[b] - c.c[b]);
            if (b != axis && !(d < h * (1.0 - 1e-6) && h * 1e-6 > 1e-14 * (fabs(c.c[b]) + h))) fast = false;
        }
        if (fast) {
            const int n = hyp_ldg(P.oct_neigh + 6 * (size_t)c.id + 2 * axis + up);
            if ((unsigned long long)n == P.n_cells) { c.id = n; return; }
            oct_descend(P, r, n, c);
            return;
        }
    }
    int id = c.id, parent = c.parent, sub = c.subcell;
    for (;;) {
        if (id == 0) { c.id = (int)P.n_cells; return; }
        int bit = (sub >> axis) & 1;
        if (bit != up) {
            int sib = up ? (sub | (1 << axis)) : (sub & ~(1 << axis));
            oct_descend(P, r, P.oct_children[8 * (size_t)parent + sib], c);
            return;
        }
        id = parent;
        const OctCell &o = P.oct_cells[id];
        parent = o.parent; sub = o.subcell;
    }
}

// ------------------------------- voronoi ------------------------------------
// grid_geometry_voronoi.f90: walls are the bisector planes with the neighbouring
// sites (CSR lists), the next cell is the neighbour itself.
__device__ __forceinline__ double vor_dist2(const DProblem &P, int i, const double r[3])
{
    const double *s = P.vor_sites + 3 * (size_t)i;
    double dx = s[0] - r[0], dy = s[1] - r[1], dz = s[2] - r[2];
    return dx * dx + dy * dy + dz * dz;
}

// nearest site by steepest descent over the neighbour graph (the reference asks
// a kd-tree, kdtree2_n_nearest :224; same answer)
__device__ __forceinline__ int vor_nearest_from(const DProblem &P, const double r[3], int seed)
{
    int cur = seed;
    double dcur = vor_dist2(P, cur, r);
    for (;;) {
        int best = cur; double dbest = dcur;
        // the gathered wall records hold the neighbour AND its site: one stream of independent 32-byte loads per hop
        const int k0 = P.vor_idx[cur], k1 = P.vor_idx[cur + 1];
#pragma unroll 4
        for (int k = k0; k < k1; k++) {
            const VorWall w = P.vor_walls[k];
            if (w.nb < 0) continue;
            const double dx = w.x - r[0], dy = w.y - r[1], dz = w.z - r[2];
            const double d = dx * dx + dy * dy + dz * dz;
            if (d < dbest) { dbest = d; best = w.nb; }
        }
        if (best == cur) return cur;
        cur = best; dcur = dbest;
    }
}

__device__ __forceinline__ int vor_nearest(const DProblem &P, const double r[3])
{
    int g = P.vor_g, id[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double f = (r[a] - P.vor_box[2 * a]) / (P.vor_box[2 * a + 1] - P.vor_box[2 * a]);
        int i = (int)(f * g);
        id[a] = i < 0 ? 0 : (i >= g ? g - 1 : i);
    }
    return vor_nearest_from(P, r, P.vor_seed[(id[2] * g + id[1]) * g + id[0]]);
}

__device__ __forceinline__ bool geo_escaped(const DProblem &P, const Cell<GEOM_VOR> &c) { return (unsigned long long)c.id == P.n_cells; }
__device__ __forceinline__ size_t geo_index(const DProblem &P, const Cell<GEOM_VOR> &c) { return (size_t)c.id; }

// find_cell :196-229 + place_in_cell :231-242
__device__ __forceinline__ bool geo_place(const DProblem &P, const Walls &W, const double r[3], const double v[3], Cell<GEOM_VOR> &c)
{
    if (r[0] < P.vor_box[0] || r[0] > P.vor_box[1] || r[1] < P.vor_box[2] || r[1] > P.vor_box[3] ||
        r[2] < P.vor_box[4] || r[2] > P.vor_box[5]) return false;
    c.id = vor_nearest(P, r);
    return true;
}

// :274-283: the cell must be one of the two nearest sites
__device__ __forceinline__ bool geo_in_correct_cell(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_VOR> &c)
{
    int n1 = vor_nearest_from(P, r, c.id);
    if (n1 == c.id) return true;
    int n2 = -1; double d2 = HYP_DBL_MAX;
    for (int k = P.vor_idx[n1]; k < P.vor_idx[n1 + 1]; k++) {
        int nb = P.vor_neigh[k];
        if (nb < 0) continue;
        double d = vor_dist2(P, nb, r);
        if (d < d2) { d2 = d; n2 = nb; }
    }
    return n2 == c.id;
}

// find_wall :322-402; im = (next cell + 1, current cell + 1, 0)
__device__ __forceinline__ bool geo_find_wall(const DProblem &P, const Walls &W, const double r[3], const double v[3],
                                              const Cell<GEOM_VOR> &c, double &tnear, int im[3])
{
    const int ic = c.id;
    const double *si = P.vor_sites + 3 * (size_t)ic;
    const double s0 = si[0], s1 = si[1], s2 = si[2];
    const int prev = -c.ow[1] - 1;
    double tmin = HYP_DBL_MAX; int imin = -1; bool found = false;
    // vor_walls holds, per CSR entry, the neighbour's id AND its site (gathered at set-up): one 32-B record at an address
    // that depends on k only, instead of the id and then the site it points to -- the loop is a stream of independent
    // loads, not ~15 chains of two dependent ones.
    const int k0 = P.vor_idx[ic], k1 = P.vor_idx[ic + 1];
#pragma unroll 4
    for (int k = k0; k < k1; k++) {
        const VorWall w = P.vor_walls[k];
        const int nb = w.nb;
        double t; int cand; bool ahead;
        if (nb < 0) {
            const int iw = -nb - 1, a = iw >> 1, up = iw & 1;
            const double va = a == 0 ? v[0] : a == 1 ? v[1] : v[2];
            const double ra = a == 0 ? r[0] : a == 1 ? r[1] : r[2];
            ahead = up ? (va > 0.0) : (va < 0.0);
            t = (P.vor_box[iw] - ra) / va;
            cand = (int)P.n_cells;
        } else {
            const double n0 = w.x - s0, n1 = w.y - s1, n2 = w.z - s2;
            const double m0 = 0.5 * (w.x + s0), m1 = 0.5 * (w.y + s1), m2 = 0.5 * (w.z + s2);
            t = (n0 * (m0 - r[0]) + n1 * (m1 - r[1]) + n2 * (m2 - r[2])) / (n0 * v[0] + n1 * v[1] + n2 * v[2]);
            ahead = nb != prev;
            cand = nb;
        }
        if (ahead && t > 0.0 && t < tmin) { tmin = t; imin = cand; found = true; }
    }
    tnear = tmin;
    im[0] = found ? imin + 1 : 0; im[1] = ic + 1; im[2] = 0;
    return found;
}

// next_cell_wall_id :266-272 (wall id = cell id) + opposite_wall
__device__ __forceinline__ void geo_advance(const DProblem &P, const double r[3], Cell<GEOM_VOR> &c, const int im[3])
{
    c.id = im[0] - 1;
    c.ow[0] = -im[0]; c.ow[1] = -im[1]; c.ow[2] = -im[2];
}


// ------------------------------- AMR ----------------------------------------
// grid_geometry_amr.f90: levels of uniform grids; each grid carries a goto table
// (one ghost layer included) that says in which grid a position continues.
__device__ __forceinline__ bool geo_escaped(const DProblem &P, const Cell<GEOM_AMR> &c) { return (unsigned long long)c.id == P.n_cells; }
__device__ __forceinline__ bool geo_invalid(const DProblem &P, const Cell<GEOM_AMR> &c) { return c.id < 0; }
__device__ __forceinline__ size_t geo_index(const DProblem &P, const Cell<GEOM_AMR> &c) { return (size_t)c.id; }

// ipos (fortranlib): 1-based bin of x in n equal bins of [xmin, xmax]; 0 below, n+1 above
__device__ __forceinline__ int amr_ipos(double xmin, double xmax, double x, int n)
{
    if (x < xmin) return 0;
    if (x > xmax) return n + 1;
    if (x < xmax) { int i = (int)((x - xmin) / (xmax - xmin) * (double)n) + 1; return i > n ? n : i; }
    return n;
}

// ipos2 :510-519
__device__ __forceinline__ int amr_ipos2(double xmin, double xmax, double x, int n)
{
    const double eps = (xmax - xmin) * 1.e-10;
    int i = amr_ipos(xmin, xmax, x, n);
    if (i == 0 && fabs(x - xmin) < eps) i = 1;
    if (i == n + 1 && fabs(x - xmax) < eps) i = n;
    return i;
}

// find_position_in_grid :521-545 (the recursion is a loop over the goto tables)
__device__ __forceinline__ void amr_find_position(const DProblem &P, const double r[3], int k, Cell<GEOM_AMR> &c)
{
    for (;;) {
        const AmrGrid &g = P.amr_grids[k];
        int i[3];
#pragma unroll
        for (int a = 0; a < 3; a++) i[a] = amr_ipos2(g.lo[a], g.hi[a], r[a], g.n[a]);
        const int go = P.amr_go[g.go_off + (i[2] * (g.n[1] + 2) + i[1]) * (g.n[0] + 2) + i[0]];
        if (go == 0) {
            if (i[0] < 1 || i[0] > g.n[0] || i[1] < 1 || i[1] > g.n[1] || i[2] < 1 || i[2] > g.n[2]) { c.id = -1; return; }
            c.grid = k; c.i[0] = i[0] - 1; c.i[1] = i[1] - 1; c.i[2] = i[2] - 1;
            c.id = (int)(g.start + (unsigned)((c.i[2] * g.n[1] + c.i[1]) * g.n[0] + c.i[0]));
            return;
        }
        k = go - 1;
    }
}

// find_cell_position :560-572 + place_in_cell :574-585 (no wall adjustment)
__device__ __forceinline__ bool geo_place(const DProblem &P, const Walls &W, const double r[3], const double v[3], Cell<GEOM_AMR> &c)
{
    c.ow[0] = c.ow[1] = c.ow[2] = 0;
    for (int k = 0; k < P.n_amr_level1; k++) {
        const AmrGrid &g = P.amr_grids[k];
        if (r[0] < g.lo[0] || r[0] > g.hi[0] || r[1] < g.lo[1] || r[1] > g.hi[1] || r[2] < g.lo[2] || r[2] > g.hi[2]) continue;
        amr_find_position(P, r, k, c);
        return c.id >= 0;
    }
    return false;
}

// in_correct_cell :677-726: position within the packet's own grid
__device__ __forceinline__ bool geo_in_correct_cell(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_AMR> &c)
{
    const AmrGrid &g = P.amr_grids[c.grid];
    const bool on_wall = (c.ow[0] | c.ow[1] | c.ow[2]) != 0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double wl = P.amr_walls[g.w_off[a] + c.i[a]], wu = P.amr_walls[g.w_off[a] + c.i[a] + 1];
        if (on_wall && c.ow[a] == -1) ok = ok && fabs((r[a] - wl) / (wu - wl)) < 1e-3;
        else if (on_wall && c.ow[a] == +1) ok = ok && fabs((r[a] - wu) / (wu - wl)) < 1e-3;
        else ok = ok && (amr_ipos(g.lo[a], g.hi[a], r[a], g.n[a]) - 1 == c.i[a]);
    }
    return ok;
}

// find_wall :775-871: nearest of the three faces ahead; a negative distance stops the reference
__device__ __forceinline__ bool geo_find_wall(const DProblem &P, const Walls &W, const double r[3], const double v[3],
                                              const Cell<GEOM_AMR> &c, double &tnear, int im[3])
{
    const AmrGrid &g = P.amr_grids[c.grid];
    double t[3]; bool pos[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        pos[a] = v[a] > 0.0;
        if (pos[a]) t[a] = (P.amr_walls[g.w_off[a] + c.i[a] + 1] - r[a]) / v[a];
        else if (v[a] < 0.0) t[a] = (P.amr_walls[g.w_off[a] + c.i[a]] - r[a]) / v[a];
        else t[a] = HYP_DBL_MAX;
    }
    im[0] = im[1] = im[2] = 0;
    if (fmin(t[0], fmin(t[1], t[2])) < 0.0) { raise_error(P, ERR_NEGATIVE_T, t[0], t[1], t[2]); return false; }
    int a;
    if (t[0] < t[2]) a = (t[0] < t[1]) ? 0 : 1;
    else a = (t[2] < t[1]) ? 2 : 1;
    tnear = a == 0 ? t[0] : a == 1 ? t[1] : t[2];
    const bool up = a == 0 ? pos[0] : a == 1 ? pos[1] : pos[2];
    if (a == 0) im[0] = up ? 1 : -1; else if (a == 1) im[1] = up ? 1 : -1; else im[2] = up ? 1 : -1;
    return true;
}

// next_cell_int :599-655 + opposite_wall
__device__ __forceinline__ void geo_advance(const DProblem &P, const double r[3], Cell<GEOM_AMR> &c, const int im[3])
{
    const AmrGrid &g = P.amr_grids[c.grid];
    const int axis = im[0] ? 0 : im[1] ? 1 : 2;
    const int dir = im[0] + im[1] + im[2];
    c.ow[0] = -im[0]; c.ow[1] = -im[1]; c.ow[2] = -im[2];
    int i[3] = {c.i[0] + 1 + im[0], c.i[1] + 1 + im[1], c.i[2] + 1 + im[2]};      // 1-based, ghost layer 0 and n+1
    const int go = P.amr_go[g.go_off + (i[2] * (g.n[1] + 2) + i[1]) * (g.n[0] + 2) + i[0]];
    if (go == 0) {
        if (i[0] == 0 || i[0] == g.n[0] + 1 || i[1] == 0 || i[1] == g.n[1] + 1 || i[2] == 0 || i[2] == g.n[2] + 1) { c.id = (int)P.n_cells; return; }
        c.i[0] = i[0] - 1; c.i[1] = i[1] - 1; c.i[2] = i[2] - 1;
        c.id = (int)(g.start + (unsigned)((c.i[2] * g.n[1] + c.i[1]) * g.n[0] + c.i[0]));
        return;
    }
    double rr[3] = {r[0], r[1], r[2]};
    if (axis == 0) rr[0] = dir > 0 ? rr[0] + P.amr_eps : rr[0] - P.amr_eps;
    else if (axis == 1) rr[1] = dir > 0 ? rr[1] + P.amr_eps : rr[1] - P.amr_eps;
    else rr[2] = dir > 0 ? rr[2] + P.amr_eps : rr[2] - P.amr_eps;
    amr_find_position(P, rr, go - 1, c);
}

#include "hyp_polar.h"

// in_correct_cell with the direction at hand (the polar grids read theta / phi off the direction where the position cannot tell)
template <int GEOM>
__device__ __forceinline__ bool geo_check_cell(const DProblem &P, const Walls &W, const double r[3], const double v[3], const Cell<GEOM> &c)
{
    return geo_in_correct_cell(P, W, r, c);
}
__device__ __forceinline__ bool geo_check_cell(const DProblem &P, const Walls &W, const double r[3], const double v[3], const Cell<GEOM_SPH> &c) { return polar_in_correct_cell<GEOM_SPH>(P, r, v, c); }
__device__ __forceinline__ bool geo_check_cell(const DProblem &P, const Walls &W, const double r[3], const double v[3], const Cell<GEOM_CYL> &c) { return polar_in_correct_cell<GEOM_CYL>(P, r, v, c); }

// start of an integration along v from r: radial = (p%r .dot. p%v) > 0 (grid_propagate_3d.f90:73,262,400,505)
template <int GEOM>
__device__ __forceinline__ void geo_begin(const double r[3], const double v[3], Cell<GEOM> &c) {}
__device__ __forceinline__ void geo_begin(const double r[3], const double v[3], Cell<GEOM_SPH> &c)
{
    c.radial = ((r[0] * v[0] + r[1] * v[1]) + r[2] * v[2]) > 0.0;
}

// geometries whose next_cell cannot fail
template <int GEOM>
__device__ __forceinline__ bool geo_invalid(const DProblem &P, const Cell<GEOM> &c) { return false; }

template <int GEOM>
__device__ __forceinline__ void geo_clear_wall(Cell<GEOM> &c) { c.ow[0] = c.ow[1] = c.ow[2] = 0; }

// ran_mu_limb(a, b): source_type.f90:982-1086
__device__ __forceinline__ double ran_mu_limb(double a, double b, double xi_in)
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

// source_distance (source_type.f90:324-357) over the sources that can intersect: source.f90:206-227
__device__ __forceinline__ void find_nearest_source(const DProblem &P, const double r[3], const double v[3], double &t_source, int &source_id)
{
    source_id = -1; t_source = HYP_INF;
    if (!P.any_intersect) return;
    for (int is = 0; is < P.n_sources; is++) {
        const DSource &S = P.sources[is];
        if (S.type != 2) continue;
        const double dr0 = r[0] - S.pos[0], dr1 = r[1] - S.pos[1], dr2 = r[2] - S.pos[2];
        const double pB = 2.0 * (dr0 * v[0] + dr1 * v[1] + dr2 * v[2]);
        const double pC = (dr0 * dr0 + dr1 * dr1 + dr2 * dr2) - S.radius * S.radius;
        // quadratic_pascal_reduced (fortranlib): cancellation-free roots of t^2 + pB t + pC = 0
        double t1 = -HYP_DBL_MAX, t2 = -HYP_DBL_MAX;
        const double delta = pB * pB - 4.0 * pC;
        if (!(delta < 0.0)) {
            const double q = pB >= 0.0 ? -0.5 * (pB + sqrt(delta)) : -0.5 * (pB - sqrt(delta));
            t1 = q; t2 = q != 0.0 ? pC / q : 0.0;
        }
        double dist = HYP_INF;
        if (t1 < dist && t1 > 1.e-8 * S.radius) dist = t1;
        if (t2 < dist && t2 > 1.e-8 * S.radius) dist = t2;
        if (dist < t_source) { t_source = dist; source_id = is; }
    }
}

// start of a grid_integrate call: grid_propagate_3d.f90:99-101
template <int NDT, int GEOM>
__device__ __forceinline__ void begin_integrate(const DProblem &P, Packet<NDT, GEOM> &p)
{
    p.t_ach = 0.0;
    find_nearest_source(P, p.r, p.v, p.t_src, p.reabs_id);
    geo_begin(p.r, p.v, p.cell);
}

// n_photons(ic) += 1 unless this packet has been counted in the cell before (grid_propagate_3d.f90:90-95,175-180).  The
// reference runs its packets one after the other, so its "the last packet here was not this one" IS "this packet has not
// been here before": n_photons = number of distinct packets per cell.  Here packets interleave, so every LANE keeps the set
// of cells its current packet has been counted in: an open-addressing table of HYP_VISIT_SLOTS (tag, cell) words in HBM
// (32 KB per lane, 4.3 GB for the 131 072 lanes of a launch, cleared per iteration), entries of earlier packets
// recognised by their tag and overwritten.  Exact and independent of the order packets run in; a packet that visits
// more than 3/4 HYP_VISIT_SLOTS distinct cells (never seen) is counted on every entry from then on and raises
// DProblem::nphot_inexact.
constexpr int HYP_VISIT_SLOTS = 4096;
__device__ __forceinline__ void count_photon(const DProblem &P, size_t ic, unsigned int tag, int &n_visited)
{
    unsigned long long *tab = P.visit_tab + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * HYP_VISIT_SLOTS;
    const unsigned long long key = ((unsigned long long)tag << 32) | (unsigned long long)(unsigned int)ic;
    if (n_visited >= HYP_VISIT_SLOTS * 3 / 4) { atomicAdd(&P.n_photons[ic], 1u); *P.nphot_inexact = 1; return; }
    unsigned int h = ((unsigned int)ic * 2654435761u) >> 20;       // 12 bits
    for (int probe = 0; probe < HYP_VISIT_SLOTS; probe++) {
        const unsigned long long e = tab[h];
        if (e == key) return;
        if ((unsigned int)(e >> 32) != tag) {       // empty, or left by an earlier packet of this lane
            tab[h] = key; n_visited++;
            atomicAdd(&P.n_photons[ic], 1u);
            return;
        }
        h = (h + 1u) & (HYP_VISIT_SLOTS - 1u);
    }
    atomicAdd(&P.n_photons[ic], 1u); *P.nphot_inexact = 1;       // table full of this packet's own entries: cannot happen below 3/4 load
}
// never 0 (the cleared-table marker): id 2^32 - 1 (mod 2^32) shares the tag of an id 2^31 away
__device__ __forceinline__ unsigned int photon_tag(const Rng &g) { const unsigned int t = g.id_lo + 1u; return t ? t : 0x80000000u; }

// start of a grid_integrate call of the Lucy iteration: also the frequency bin of the packet (:59-71) and the
// count of the starting cell (:90-95)
template <int NDT, int GEOM>
__device__ __forceinline__ void begin_integrate_lucy(const DProblem &P, Packet<NDT, GEOM> &p, const Rng &g)
{
    begin_integrate(P, p);
    p.spec_idx = -1;
    if (P.n_bins) p.spec_idx = locate_g(P.log_nu_edges, P.n_bins + 1, log10(p.nu));
    if (P.count_photons && !geo_escaped(P, p.cell)) count_photon(P, geo_index(P, p.cell), photon_tag(g), p.n_visited);
}

// One iteration of the big loop of grid_integrate (grid_propagate_3d.f90:106-232)
// / grid_integrate_noenergy (:237-375).  Returns the lane's next phase;
// ST_NEED_EMIT means the packet left the grid or was killed.
template <int NDT, int GEOM, bool DEPOSIT>
__device__ __forceinline__ int walk_step(const DProblem &P, const Walls &W, Packet<NDT, GEOM> &p, Rng &g,
                                         double *__restrict__ sum, Counters &cnt)
{
    const int nd = ndust<NDT>(P);
    if (g.countdown == 0) {
        g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp);
        if (!geo_check_cell(P, W, p.r, p.v, p.cell)) { cnt.killed_geo++; return ST_NEED_EMIT; }
    } else g.countdown--;
    double tmin; int im[3];
    if (!geo_find_wall(P, W, p.r, p.v, p.cell, tmin, im)) { cnt.killed_geo++; return ST_NEED_EMIT; }
    const size_t base = geo_index(P, p.cell) * (size_t)nd;
    double rho[NDT];
    double chi_rho = 0.0;
#pragma unroll
    for (int d = 0; d < NDT; d++) {
        rho[d] = 0.0;
        if (d < nd) { rho[d] = hyp_ldg(P.density + base + d); chi_rho += p.chi[d] * rho[d]; }
    }
    double tau_cell = chi_rho * tmin;
    double tau_needed = p.tau_req - p.tau_ach;
    cnt.crossings++;
    if (tau_cell < tau_needed) {
        if (P.any_intersect) { p.t_ach += tmin; if (p.t_ach > p.t_src) return ST_NEED_REEMIT; }     // re-absorbed by a source :139-143
#pragma unroll
        for (int a = 0; a < 3; a++) p.r[a] = p.r[a] + tmin * p.v[a];
        p.tau_ach += tau_cell;
        if (DEPOSIT) {
#pragma unroll
            for (int d = 0; d < NDT; d++)
                if (d < nd && rho[d] > 0.0) {
                    hyp_atomic_add_g(&sum[base + d], tmin * p.kappa[d] * p.energy);
                    if (P.n_bins && p.spec_idx >= 0)
                        unsafeAtomicAdd(&P.sum_spec[(size_t)p.spec_idx * P.n_cells * nd + base + d], tmin * p.kappa[d] * p.energy);
                }
        }
        geo_advance(P, p.r, p.cell, im);
        if (geo_invalid(P, p.cell)) { cnt.killed_geo++; return ST_NEED_EMIT; }     // amr: invalid_cell
        // the imaging iteration (no deposits) tells packets that left the grid from killed ones: iter_final.f90:127-129
        if (geo_escaped(P, p.cell)) return DEPOSIT ? ST_NEED_EMIT : ST_ESCAPED;
        if (DEPOSIT && P.count_photons) count_photon(P, geo_index(P, p.cell), photon_tag(g), p.n_visited);
        return ST_WALK;
    } else {
        double tact = tmin * (tau_needed / tau_cell);
        if (P.any_intersect) { p.t_ach += tact; if (p.t_ach > p.t_src) return ST_NEED_REEMIT; }     // :184-188
#pragma unroll
        for (int a = 0; a < 3; a++) p.r[a] = p.r[a] + tact * p.v[a];
        p.tau_ach += tau_needed;
        geo_clear_wall(p.cell);
        if (DEPOSIT) {
#pragma unroll
            for (int d = 0; d < NDT; d++)
                if (d < nd && rho[d] > 0.0) {
                    hyp_atomic_add_g(&sum[base + d], tact * p.kappa[d] * p.energy);
                    if (P.n_bins && p.spec_idx >= 0)
                        unsafeAtomicAdd(&P.sum_spec[(size_t)p.spec_idx * P.n_cells * nd + base + d], tact * p.kappa[d] * p.energy);
                }
        }
        return ST_NEED_INTERACT;
    }
}

// fortranlib random_planck_frequency (Carter & Cashwell 1975)
__device__ __forceinline__ double random_planck_frequency(Rng &g, double T)
{
    double target = rng_uniform(g) * (HYP_PI * HYP_PI * HYP_PI * HYP_PI / 90.0);
    double sum = 0.0; int m = 0;
    do { m++; double dm = (double)m; sum += 1.0 / (dm * dm * dm * dm); } while (sum < target && m < 1000);
    double x1 = rng_uniform(g), x2 = rng_uniform(g), x3 = rng_uniform(g), x4 = rng_uniform(g);
    double x = -log((1.0 - x1) * (1.0 - x2) * (1.0 - x3) * (1.0 - x4)) / (double)m;
    return x * HYP_K_CGS * T / HYP_H_CGS;
}

template <int GEOM>
__device__ __forceinline__ bool random_position_cell(const DProblem &P, size_t ic, double x, double y, double z, double r[3], Rng &g);
__device__ __forceinline__ double dust_sample_j_nu(const DDust &D, int jid, double frac, double xi);
__device__ __forceinline__ double dust_emit_probability(const DProblem &P, const DDust &D, int jid, double frac);

// emit: source.f90:100-179 + source_emit/emit_from_point source_type.f90:398-564.
// Returns false on a fatal error (flag raised).
// CLASS (checked by the host): 0 any source; 1 ("simple") every source is a point source with a tabulated or blackbody spectrum and
// the launch is not monochromatic; 2 the same plus external spherical / box sources (configs[4]'s point + extern_box): the other
// emitters stay out of the kernel, whose registers then allow another wave per SIMD.
template <int NDT, int GEOM, int CLASS = 0>
__device__ __forceinline__ bool emit_packet(const DProblem &P, const Walls &W, Packet<NDT, GEOM> &p, Rng &g,
                                            Counters &cnt, int &source_id, Angle &src_normal, int reemit_id = -1, double reemit_energy = 0.0)
{
    int is = 0;
    if (P.n_sources > 1) {
        double xi = rng_uniform(g);
        if (P.sample_sources_evenly) is = (int)(xi * P.n_sources);
        else {
            is = P.n_sources - 1;
            for (int i = 0; i < P.n_sources - 1; i++) if (xi < P.sources[i].lum_cdf) { is = i; break; }
        }
    }
    if (reemit_id >= 0) is = reemit_id;      // emit(reemit=.true., reemit_id=...): source.f90:135-141
    source_id = is;
    const DSource &S = P.sources[is];
    size_t map_cell = 0;
    int ispot = -1;
    p.emiss_dust = -1;
    src_normal.cost = 1.0; src_normal.sint = 0.0; src_normal.cosp = 1.0; src_normal.sinp = 0.0;
    constexpr bool SIMPLE = CLASS == 1 || CLASS == 3;
    constexpr bool FEW = CLASS != 0;          // no sphere / spot / map / point-collection / plane-parallel emitters, no 'lte' spectrum, not monochromatic
    constexpr bool MONO_OK = CLASS == 0 || CLASS == 3;      // CLASS 3: the SIMPLE emitter in a monochromatic launch (final_defer_kernel<.., true, true>)
    if (SIMPLE) {
        p.r[0] = S.pos[0]; p.r[1] = S.pos[1]; p.r[2] = S.pos[2];
        random_sphere_angle(g, p.a);
    } else if (!FEW && S.type == 2) {
        // emit_from_sphere: source_type.f90:604-690
        Angle a_coord, a_local;
        if (S.n_spots > 0) {        // source_emit case(3), source_type.f90:421-427: a spot or the rest of the sphere, by luminosity
            const double xi = rng_uniform(g);
            int k = S.n_spots;
            for (int i = S.n_spots - 1; i >= 0; i--) if (xi < S.spot_tab[i]) k = i;
            if (k < S.n_spots) ispot = k;
        }
        if (ispot >= 0) {           // emit_from_sphere(spot) :632-636: rejection until the position lies inside the spot
            const double *q = S.spot_tab + (S.n_spots + 1) + (size_t)ispot * SPOT_STRIDE;
            for (;;) {
                random_sphere_angle(g, a_coord);
                double n0, n1, n2;
                angle_to_vector(a_coord, n0, n1, n2);
                if ((n0 * q[0] + n1 * q[1]) + n2 * q[2] > q[3]) break;
            }
        } else random_sphere_angle(g, a_coord);
        double sp, cp;
        sincos(HYP_TWOPI * rng_uniform(g), &sp, &cp);
        a_local.cosp = cp; a_local.sinp = sp;
        if (S.limb_darkening) a_local.cost = ran_mu_limb(1.5, 1.0, rng_uniform(g));
        else a_local.cost = sqrt(rng_uniform(g));
        a_local.sint = sqrt(1.0 - a_local.cost * a_local.cost);
        rotate_angle(a_local, a_coord, p.a);
        double n0, n1, n2;
        angle_to_vector(a_coord, n0, n1, n2);
        p.r[0] = n0 * S.radius + S.pos[0]; p.r[1] = n1 * S.radius + S.pos[1]; p.r[2] = n2 * S.radius + S.pos[2];
        src_normal = a_coord;       // outward normal (p%source_a)
    } else if (!FEW && S.type == 4) {
        // emit_from_map: source_type.f90:713-741 -- cell from the luminosity map, uniform position in it, isotropic direction
        const double xi = rng_uniform(g);
        size_t lo = 0, hi = (size_t)P.n_cells - 1;
        while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (xi < S.map_cdf[mid]) hi = mid; else lo = mid + 1; }
        map_cell = lo;
        const double x = rng_uniform(g), y = rng_uniform(g), z = rng_uniform(g);
        if (!random_position_cell<GEOM>(P, lo, x, y, z, p.r, g)) { raise_error(P, ERR_RAY_GRID, 0.0, 0.0, 0.0); return false; }
        random_sphere_angle(g, p.a);
    } else if (!FEW && S.type == 8) {
        // emit_from_point_collection: source_type.f90:570-598
        const double xi = rng_uniform(g);
        int k = S.n_points - 1;
        for (int i = 0; i < S.n_points - 1; i++) if (xi < S.point_cdf[i]) { k = i; break; }
        p.r[0] = S.points[3 * k]; p.r[1] = S.points[3 * k + 1]; p.r[2] = S.points[3 * k + 2];
        random_sphere_angle(g, p.a);
    } else if (!FEW && S.type == 7) {
        // emit_from_plane_parallel: source_type.f90:935-975 (not peeled: source_emit_peeloff has no case for it)
        const double rr = pow(rng_uniform(g), 0.5) * S.radius;
        const double phi = 360.0 * rng_uniform(g) * HYP_PI / 180.0;
        Angle a_local, a_dir, a_final;
        a_local.cost = cos(90.0 * HYP_PI / 180.0); a_local.sint = sin(90.0 * HYP_PI / 180.0);
        double sp, cp;
        sincos(phi, &sp, &cp);
        a_local.cosp = cp; a_local.sinp = sp;
        a_dir.cost = S.dir_cost; a_dir.sint = S.dir_sint; a_dir.cosp = S.dir_cosp; a_dir.sinp = S.dir_sinp;
        rotate_angle(a_local, a_dir, a_final);
        double n0, n1, n2;
        angle_to_vector(a_final, n0, n1, n2);
        p.r[0] = n0 * rr + S.pos[0]; p.r[1] = n1 * rr + S.pos[1]; p.r[2] = n2 * rr + S.pos[2];
        p.a = a_dir;
        src_normal = a_dir;
    } else if (S.type == 1) {
        // emit_from_point: source_type.f90:539-564
        p.r[0] = S.pos[0]; p.r[1] = S.pos[1]; p.r[2] = S.pos[2];
        random_sphere_angle(g, p.a);
    } else if (S.type == 5) {
        // emit_from_extern_sph: source_type.f90:748-809
        Angle a_coord, a_local;
        random_sphere_angle(g, a_coord);
        double sp, cp;
        sincos(HYP_TWOPI * rng_uniform(g), &sp, &cp);
        a_local.cosp = cp; a_local.sinp = sp;
        a_local.cost = sqrt(rng_uniform(g));
        a_local.sint = sqrt(1.0 - a_local.cost * a_local.cost);
        rotate_angle(a_local, a_coord, p.a);
        p.a.cost = -p.a.cost; p.a.cosp = -p.a.cosp; p.a.sinp = -p.a.sinp;
        double n0, n1, n2;
        angle_to_vector(a_coord, n0, n1, n2);
        p.r[0] = n0 * S.radius + S.pos[0]; p.r[1] = n1 * S.radius + S.pos[1]; p.r[2] = n2 * S.radius + S.pos[2];
        src_normal = a_coord;
        src_normal.cost = -a_coord.cost; src_normal.cosp = -a_coord.cosp; src_normal.sinp = -a_coord.sinp;
    } else {
        // emit_from_extern_box: source_type.f90:822-907
        double xi = rng_uniform(g);
        int face = 5;
        for (int k = 0; k < 5; k++) if (xi < S.face_cdf[k]) { face = k; break; }
        Angle a_local, a_coord;
        double sp, cp;
        sincos(HYP_TWOPI * rng_uniform(g), &sp, &cp);
        a_local.cosp = cp; a_local.sinp = sp;
        a_local.cost = sqrt(rng_uniform(g));
        a_local.sint = sqrt(1.0 - a_local.cost * a_local.cost);
        const int axis = face >> 1, up = face & 1;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k == axis) p.r[k] = S.box[2 * k + up];
            else p.r[k] = S.box[2 * k] + (S.box[2 * k + 1] - S.box[2 * k]) * rng_uniform(g);
        }
        // inward normals as written in source_type.f90:864-899 (negative sin(theta) on the max faces)
        a_coord.cost = axis == 2 ? (up ? -1.0 : 1.0) : 0.0;
        a_coord.sint = axis == 2 ? 0.0 : (up ? -1.0 : 1.0);
        a_coord.cosp = axis == 1 ? 0.0 : 1.0;
        a_coord.sinp = axis == 1 ? 1.0 : 0.0;
        rotate_angle(a_local, a_coord, p.a);
        src_normal = a_coord;
    }
    p.s[0] = 1.0; p.s[1] = 0.0; p.s[2] = 0.0; p.s[3] = 0.0;
    p.energy = 1.0;
    int lte_jid = 0; double lte_frac = 0.0;
    if (!FEW && S.spectrum_type == 3) {
        // 'lte' (source_type.f90:455-459, 486-491): select_dust_specific_energy_rho (grid_physics_3d.f90:101-109) in the
        // emitting cell, then the emissivity of that dust
        const int nd = ndust<NDT>(P);
        const size_t base = map_cell * (size_t)nd;
        double c = 0.0;
        for (int d = 0; d < nd; d++) c += P.specific_energy[base + d] * hyp_ldg(P.density + base + d);
        const double xi = rng_uniform(g);
        int id = nd - 1; double run = 0.0; bool found = false;
        for (int d = 0; d < nd - 1; d++) {
            run += P.specific_energy[base + d] * hyp_ldg(P.density + base + d);
            if (!found && xi < run / c) { id = d; found = true; }
        }
        p.emiss_dust = id;
        lte_jid = P.jnu_id[base + id]; lte_frac = P.jnu_frac[base + id];
    }
    if (!FEW && ispot >= 0) {       // the spot's own spectrum: source_type.f90:447-461, 480-492
        const double *q = S.spot_tab + (S.n_spots + 1) + (size_t)ispot * SPOT_STRIDE;
        if (P.mono_which) {
            p.nu = P.mono_nu;
            p.energy = S.spot_blob[(size_t)q[10] + P.mono_inu];
        } else if (q[4] == 1.0)
            p.nu = sample_log_pdf(S.spot_blob + (size_t)q[7], S.spot_blob + (size_t)q[8], S.spot_blob + (size_t)q[9], (int)q[6], rng_uniform(g));
        else p.nu = random_planck_frequency(g, q[5]);
    } else
    if (MONO_OK && P.mono_which) {     // emit(p, inu=inu): source_type.f90:440-468, the energy carries the emission probability at nu
        p.nu = P.mono_nu;
        p.energy = (!FEW && S.spectrum_type == 3) ? dust_emit_probability(P, P.dust[p.emiss_dust], lte_jid, lte_frac)
                                                  : P.mono_src_prob[(size_t)is * P.n_frequencies + P.mono_inu];
    } else if (!FEW && S.spectrum_type == 3) p.nu = dust_sample_j_nu(P.dust[p.emiss_dust], lte_jid, lte_frac, rng_uniform(g));
    else if (S.spectrum_type == 1) p.nu = sample_log_pdf(S.spec_x, S.spec_cdf, S.spec_bp1, S.n_spec, rng_uniform(g));
    else p.nu = random_planck_frequency(g, S.temperature);
    angle_to_vector(p.a, p.v[0], p.v[1], p.v[2]);
    if (reemit_id >= 0) p.energy = reemit_energy;
    else {
        if (MONO_OK && P.mono_which) p.energy = p.energy * P.energy_total;      // source.f90:161
        if (P.sample_sources_evenly) p.energy = p.energy * S.lum_pdf * P.n_sources;
        cnt.energy_current += p.energy;
    }
    if (!update_optconsts<NDT, GEOM>(P, p)) return false;
    g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp);
    geo_clear_wall(p.cell);
    bool placed = false;
    if constexpr (GEOM == GEOM_VOR) {       // a point source sits in one cell, found at set-up (same search, same answer)
        if (S.type == 1 && S.vor_cell1 > 0) { p.cell.id = S.vor_cell1 - 1; placed = true; }
    }
    if (!placed && !geo_place(P, W, p.r, p.v, p.cell)) {
        cnt.killed_geo++;
        raise_error(P, ERR_NOT_IN_CELL, p.r[0], p.r[1], p.r[2]);
        return false;
    }
    p.inter = 1;
    if (reemit_id < 0) { p.peel_seq = 0; p.n_visited = 0; }      // a re-emitted packet is still the same packet
    return true;
}

// dust_sample_j_nu: dust_type_4elem.f90:379-398
__device__ __forceinline__ double dust_sample_j_nu(const DDust &D, int jid, double frac, double xi)
{
    const size_t o = (size_t)jid * D.n_enu;
    const size_t oc = (size_t)jid * D.n_ecoarse;
    double nu1, nu2;
    sample_log_pdf_pair(D.emiss_x, D.emiss_cdf + o, D.emiss_cdf + o + D.n_enu, D.emiss_bp1 + o, D.emiss_bp1 + o + D.n_enu,
                        D.emiss_coarse + oc, D.emiss_coarse + oc + D.n_ecoarse, D.n_enu, D.n_ecoarse, xi, nu1, nu2);
    double l1 = log10(nu1);
    return exp10(l1 + frac * (log10(nu2) - l1));
}

__device__ __forceinline__ void interp_P(const DDust &D, double mu, double nu, double &P1, double &P2, double &P3, double &P4)
{
    int i = locate_g(D.mu, D.n_mu, mu), j = locate_g(D.nu, D.n_nu, nu);
    if (i < 0 || j < 0) { P1 = P2 = P3 = P4 = __builtin_nan(""); return; }
    double fx = (mu - D.mu[i]) / (D.mu[i + 1] - D.mu[i]);
    double fy = (nu - D.nu[j]) / (D.nu[j + 1] - D.nu[j]);
    P1 = bilinear(D.P1, D.n_mu, i, j, fx, fy);
    P2 = bilinear(D.P2, D.n_mu, i, j, fx, fy);
    P3 = bilinear(D.P3, D.n_mu, i, j, fx, fy);
    P4 = bilinear(D.P4, D.n_mu, i, j, fx, fy);
}

// dust_scatter: dust_type_4elem.f90:446-566
__device__ __forceinline__ void dust_scatter(const DDust &D, double nu, Angle &a, double s[4], Rng &g)
{
    Angle a_scat, a_final;
    random_sphere_angle(g, a_scat);
    double sin_2_i1 = 2.0 * a_scat.sinp * a_scat.cosp;
    double cos_2_i1 = 1.0 - 2.0 * a_scat.sinp * a_scat.sinp;
    double c1 = s[0], c2 = cos_2_i1 * s[1] - sin_2_i1 * s[2];
    double ctot = c1 + c2;
    c1 /= ctot; c2 /= ctot;
    const int nm = D.n_mu;
    int inu = locate_g(D.nu, D.n_nu, nu);
    double P1, P2, P3, P4;
    if (inu == -1) {
        P1 = 1.0; P2 = 0.0; P3 = 1.0; P4 = 0.0;
    } else {
        double xi = rng_uniform(g);
        const double *C1 = D.P1_cdf + (size_t)inu * nm, *C2 = D.P2_cdf + (size_t)inu * nm;
        int imin = 0, imax = nm - 1, imu = 0;
        double cdf1 = 0.0, cdf2 = 1.0;
        for (int it = 0; it < 64; it++) {
            imu = ((imax + 1) + (imin + 1)) / 2 - 1;
            if (D.zero_p2) { cdf1 = C1[imu]; cdf2 = C1[imu + 1]; }
            else { cdf1 = c1 * C1[imu] + c2 * C2[imu]; cdf2 = c1 * C1[imu + 1] + c2 * C2[imu + 1]; }
            if (xi > cdf2) imin = imu;
            else if (xi < cdf1) imax = imu;
            else break;
            if (imin == imax) break;
        }
        a_scat.cost = (xi - cdf1) / (cdf2 - cdf1) * (D.mu[imu + 1] - D.mu[imu]) + D.mu[imu];
        a_scat.sint = sqrt(1.0 - a_scat.cost * a_scat.cost);
        interp_P(D, a_scat.cost, nu, P1, P2, P3, P4);
    }
    rotate_angle(a_scat, a, a_final);
    scatter_stokes(s, a, a_scat, a_final, P1, P2, P3, P4);
    a = a_final;
    double norm = 1.0 / s[0];
    s[0] = 1.0; s[1] *= norm; s[2] *= norm; s[3] *= norm;
}

// interact: dust_interact.f90:22-79 (+ select_dust_chi_rho grid_physics_3d.f90:87-99).
// Returns false on a fatal error.  `scattered`/`dust_id` report what happened.
template <int NDT, int GEOM>
__device__ __forceinline__ bool interact(const DProblem &P, Packet<NDT, GEOM> &p, Rng &g, Counters &cnt,
                                         int &scattered, int &dust_id, bool force_scatter = false)
{
    const int nd = ndust<NDT>(P);
    const size_t base = geo_index(P, p.cell) * (size_t)nd;
    int id = 0;
    double albedo = p.albedo[0];
    if (NDT > 1 && nd > 1) {
        double cdf[NDT], c = 0.0;
#pragma unroll
        for (int d = 0; d < NDT; d++) { if (d < nd) c += p.chi[d] * hyp_ldg(P.density + base + d); cdf[d] = c; }
        double xi = rng_uniform(g);
        id = nd - 1; albedo = 0.0;
        bool found = false;
#pragma unroll
        for (int d = 0; d < NDT; d++) {
            if (d < nd) {
                if (!found && d < nd - 1 && xi < cdf[d] / c) { id = d; found = true; }
            }
        }
#pragma unroll
        for (int d = 0; d < NDT; d++) if (d == id) albedo = p.albedo[d];
    }
    dust_id = id;
    cnt.interactions++;
    double xi = force_scatter ? 0.0 : rng_uniform(g);      // interact(p, force_scatter): dust_interact.f90:49-53
    if (xi > albedo) {
        const DDust &D = P.dust[id];
        int jid = P.jnu_id[base + id];
        double frac = P.jnu_frac[base + id];
        p.nu = dust_sample_j_nu(D, jid, frac, rng_uniform(g));
        p.s[0] = 1.0; p.s[1] = 0.0; p.s[2] = 0.0; p.s[3] = 0.0;
        random_sphere_angle(g, p.a);
        scattered = 0;
        if (!update_optconsts<NDT, GEOM>(P, p)) return false;
    } else {
        dust_scatter(P.dust[id], p.nu, p.a, p.s, g);
        scattered = 1;
    }
    angle_to_vector(p.a, p.v[0], p.v[1], p.v[2]);
    if (force_scatter) p.energy = p.energy * albedo;       // :75-77
    return true;
}

// ---------------------------------------------------------------------------
// Modified random walk (grid_mrw_3d.f90; Min et al. 2009)
// ---------------------------------------------------------------------------

// distance_to_closest_wall: grid_geometry_cartesian_3d.f90:396-422, _octree.f90:410-437, _amr.f90:743-773
__device__ __forceinline__ double geo_closest_wall(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_CAR> &c)
{
    double d = HYP_DBL_MAX;
#pragma unroll
    for (int a = 0; a < 3; a++) d = fmin(d, fmin(r[a] - W.w[a][c.ic[a]], W.w[a][c.ic[a] + 1] - r[a]));
    return d < 0.0 ? 0.0 : d;
}
__device__ __forceinline__ double geo_closest_wall(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_OCT> &c)
{
    double d = HYP_DBL_MAX;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double h = ldexp(P.oct_half[a], -c.level);
        d = fmin(d, fmin(r[a] - c.c[a] + h, c.c[a] + h - r[a]));
    }
    return d < 0.0 ? 0.0 : d;
}
__device__ __forceinline__ double geo_closest_wall(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_VOR> &c)
{
    return 0.0;     // not defined for Voronoi grids: the engine refuses the combination
}
__device__ __forceinline__ double geo_closest_wall(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_AMR> &c)
{
    const AmrGrid &g = P.amr_grids[c.grid];
    double d = HYP_DBL_MAX;
#pragma unroll
    for (int a = 0; a < 3; a++)
        d = fmin(d, fmin(r[a] - P.amr_walls[g.w_off[a] + c.i[a]], P.amr_walls[g.w_off[a] + c.i[a] + 1] - r[a]));
    return d < 0.0 ? 0.0 : d;
}

// tau_inv_planck_to_closest_wall(p) > mrw_gamma: grid_physics_3d.f90:81-85
template <int NDT, int GEOM>
__device__ __forceinline__ bool mrw_wanted(const DProblem &P, const Walls &W, const Packet<NDT, GEOM> &p)
{
    return P.mrw_alpha[geo_index(P, p.cell)] * geo_closest_wall(P, W, p.r, p.cell) > P.mrw_gamma;
}

// dust_sample_b_nu: dust_type_4elem.f90:400-419
__device__ __forceinline__ double dust_sample_b_nu(const DDust &D, int jid, double frac, double xi)
{
    const size_t o = (size_t)jid * D.n_enu;
    const size_t oc = (size_t)jid * D.n_ecoarse;
    double nu1, nu2;
    sample_log_pdf_pair(D.emiss_x, D.bnu_cdf + o, D.bnu_cdf + o + D.n_enu, D.bnu_bp1 + o, D.bnu_bp1 + o + D.n_enu,
                        D.bnu_coarse + oc, D.bnu_coarse + oc + D.n_ecoarse, D.n_enu, D.n_ecoarse, xi, nu1, nu2);
    double l1 = log10(nu1);
    return exp10(l1 + frac * (log10(nu2) - l1));
}

// grid_do_mrw :55-107 (DEPOSIT) / grid_do_mrw_noenergy :109-148.  The packet's opacities are
// deliberately left at those of the previous frequency: the reference refreshes them only in
// emit and interact.  Returns the dust species the new frequency was drawn from.
template <int NDT, int GEOM, bool DEPOSIT>
__device__ __forceinline__ int mrw_step(const DProblem &P, const Walls &W, Packet<NDT, GEOM> &p, Rng &g, double *__restrict__ sum)
{
    const int nd = ndust<NDT>(P);
    const size_t ic = geo_index(P, p.cell);
    const size_t base = ic * (size_t)nd;
    const double R0 = geo_closest_wall(P, W, p.r, p.cell);
    if (DEPOSIT) {
        // sample_cumulative :197-202: interp1d(ycdf, xcdf, xi)
        const double xi = rng_uniform(g);
        const int j = locate_g(P.mrw_y, 100, xi);
        const double y = (j < 0) ? __builtin_nan("")
                                 : P.mrw_x[j] + (xi - P.mrw_y[j]) / (P.mrw_y[j + 1] - P.mrw_y[j]) * (P.mrw_x[j + 1] - P.mrw_x[j]);
        const double q = R0 / HYP_PI;
        const double ct = -log(y) / P.mrw_diff[ic] * (q * q);
#pragma unroll
        for (int d = 0; d < NDT; d++)
            if (d < nd && hyp_ldg(P.density + base + d) > 0.0) {
                const double e = p.energy * ct * P.mrw_kp[base + d];
                hyp_atomic_add_g(&sum[base + d], e);
                if (P.n_bins) {      // deposit_specific_energy_spectrum: grid_physics_3d.f90:367-395
                    const int iv = P.jnu_id[base + d]; const double fr = P.jnu_frac[base + d];
                    const double *f0 = P.jnu_bin_frac + ((size_t)d * P.nj_max + iv) * P.n_bins, *f1 = f0 + P.n_bins;
                    for (int b = 0; b < P.n_bins; b++)
                        unsafeAtomicAdd(&P.sum_spec[(size_t)b * P.n_cells * nd + base + d], e * ((1.0 - fr) * f0[b] + fr * f1[b]));
                }
            }
    }
    Angle ar;
    random_sphere_angle(g, ar);
    double dx, dy, dz;
    angle_to_vector(ar, dx, dy, dz);
    p.r[0] = p.r[0] + dx * R0; p.r[1] = p.r[1] + dy * R0; p.r[2] = p.r[2] + dz * R0;
    random_sphere_angle(g, p.a);
    angle_to_vector(p.a, p.v[0], p.v[1], p.v[2]);
    int id = 0;
    if (NDT > 1 && nd > 1) {       // select_dust_chi_rho: grid_physics_3d.f90:87-99
        double cdf[NDT], c = 0.0;
#pragma unroll
        for (int d = 0; d < NDT; d++) { if (d < nd) c += p.chi[d] * hyp_ldg(P.density + base + d); cdf[d] = c; }
        const double xi = rng_uniform(g);
        id = nd - 1;
        bool found = false;
#pragma unroll
        for (int d = 0; d < NDT; d++)
            if (d < nd && !found && d < nd - 1 && xi < cdf[d] / c) { id = d; found = true; }
    }
    p.nu = dust_sample_b_nu(P.dust[id], P.jnu_id[base + id], P.jnu_frac[base + id], rng_uniform(g));
    return id;
}

// the loop of iter_lucy.f90:138-152; true = the packet was killed (n_inter_mrw_max steps without leaving the regime)
template <int NDT, int GEOM>
__device__ __forceinline__ bool mrw_loop_lucy(const DProblem &P, const Walls &W, Packet<NDT, GEOM> &p, Rng &g, double *__restrict__ sum,
                                              Counters &cnt)
{
    long long k;
#pragma unroll 1
    for (k = 1; k <= P.n_inter_mrw_max; k++) {
        if (!mrw_wanted(P, W, p)) break;
        mrw_step<NDT, GEOM, true>(P, W, p, g, sum);
    }
    if (k == P.n_inter_mrw_max + 1) { cnt.killed_int++; return true; }
    return false;
}

// wave-level sum of a double (64 lanes)
__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__device__ __forceinline__ unsigned int xcc_id()
{
    unsigned int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

// Hands out packet ids to the lanes of a wave that need one.  The wave owns a
// private range [next, end) refilled `chunk` ids at a time from the global
// dispenser, so the global atomic is hit once per `chunk` packets.
struct Dispenser {
    unsigned long long next, end;
};

__device__ __forceinline__ bool take_id(const DProblem &P, const LaunchParams &L, Dispenser &dsp, bool need,
                                        unsigned long long &id)
{
    bool got = false;
    unsigned long long mask = __ballot(need);
    const unsigned int lane = __lane_id();
    while (mask) {
        if (dsp.next >= dsp.end) {
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(P.counter, (unsigned long long)L.chunk);
            b = __shfl(b, 0, 64);
            if (b >= L.end_id) break;       // pool exhausted
            dsp.next = b;
            dsp.end = (b + (unsigned long long)L.chunk < L.end_id) ? b + (unsigned long long)L.chunk : L.end_id;
        }
        unsigned long long avail = dsp.end - dsp.next;
        unsigned int rank = __popcll(mask & ((1ull << lane) - 1ull));
        bool mine = ((mask >> lane) & 1ull) && rank < avail;
        if (mine) { id = dsp.next + rank; got = true; }
        unsigned long long taken = __ballot(mine);
        dsp.next += __popcll(taken);
        mask &= ~taken;
    }
    return got;
}

// Copies the Cartesian wall tables into LDS (nothing to stage for the octree).
template <int GEOM>
__device__ __forceinline__ void stage_walls(const DProblem &P, double *lds, Walls &W)
{
    if (GEOM == GEOM_CAR) {
        const int m1 = P.n1 + 1, m2 = P.n2 + 1, m3 = P.n3 + 1;
        double *w0 = lds, *w1 = w0 + m1, *w2 = w1 + m2;
        double *e0 = w2 + m3, *e1 = e0 + m1, *e2 = e1 + m2;
        for (int i = threadIdx.x; i < m1; i += blockDim.x) { w0[i] = P.w[0][i]; e0[i] = P.ew[0][i]; }
        for (int i = threadIdx.x; i < m2; i += blockDim.x) { w1[i] = P.w[1][i]; e1[i] = P.ew[1][i]; }
        for (int i = threadIdx.x; i < m3; i += blockDim.x) { w2[i] = P.w[2][i]; e2[i] = P.ew[2][i]; }
        W.w[0] = w0; W.w[1] = w1; W.w[2] = w2; W.ew[0] = e0; W.ew[1] = e1; W.ew[2] = e2;
        W.n[0] = P.n1; W.n[1] = P.n2; W.n[2] = P.n3;
        __syncthreads();
    } else {
        W.w[0] = W.w[1] = W.w[2] = nullptr; W.ew[0] = W.ew[1] = W.ew[2] = nullptr;
        W.n[0] = W.n[1] = W.n[2] = 0;
    }
}

// ---------------------------------------------------------------------------
// Lucy iteration: do_lucy packet loop, iter_lucy.f90:119-209
// ---------------------------------------------------------------------------
// Workgroups per CU the register budget of the persistent kernels is set for (x 4 waves / 4 SIMDs = waves per SIMD).
// The Cartesian walk is atomic-bound and keeps everything in registers at 2; the Voronoi walk waits on memory (a stream
// of wall records per crossing) and gains from more waves even though they spill (measured: 167 / 141 / 138 ms at 2 / 3 / 4).
constexpr int HYP_LUCY_WAVES = 2;
constexpr int HYP_LUCY_WAVES_VOR = 4;
constexpr int HYP_LUCY_WAVES_OCT = 2;
constexpr int HYP_FINAL_WAVES = 2;
template <int GEOM> constexpr int lucy_waves() { return GEOM == GEOM_VOR ? HYP_LUCY_WAVES_VOR : GEOM == GEOM_OCT ? HYP_LUCY_WAVES_OCT : HYP_LUCY_WAVES; }
constexpr int HYP_WALK_STEPS = 4;        // cell crossings between two looks at the lanes' states, Cartesian grid
constexpr int HYP_WALK_STEPS_TREE = 16;  // octree, Voronoi: a crossing is several dependent loads, fewer state checks pay (configs[3]                          // imaging 52.7 -> 50 ms; the Lucy kernels do not care)
#ifndef HYP_WALK_STEPS_OTHER_N
#define HYP_WALK_STEPS_OTHER_N 8
#endif
constexpr int HYP_WALK_STEPS_OTHER = HYP_WALK_STEPS_OTHER_N;  // AMR, polar grids
template <int GEOM> constexpr int walk_steps()
{
    return GEOM == GEOM_CAR ? HYP_WALK_STEPS : (GEOM == GEOM_OCT || GEOM == GEOM_VOR) ? HYP_WALK_STEPS_TREE : HYP_WALK_STEPS_OTHER;
}
// the imaging kernels deposit nothing, so on a Cartesian grid too a longer run between state checks pays (64^3, tau = 1:
// inline 50.6 -> 47.9 ms, deferred 32.5 -> 31.5 ms; tau = 5 with three views: 533 -> 469, 251 -> 245 ms)
constexpr int HYP_WALK_STEPS_FINAL_CAR = 16;
template <int GEOM> constexpr int final_walk_steps() { return GEOM == GEOM_CAR ? HYP_WALK_STEPS_FINAL_CAR : walk_steps<GEOM>(); }
template <int NDT, int GEOM>
__global__ __launch_bounds__(256, lucy_waves<GEOM>()) void lucy_kernel(const DProblem *__restrict__ Pp, LaunchParams L)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    double *sum = P.sum;
    // accumulator replica of this workgroup: replicas are spread first over the
    // XCDs (no line is shared between two L2s), then over workgroups of an XCD
    if (P.n_copies > 1) {
        unsigned c = xcc_id();
        if (P.n_copies > 8) c += 8u * ((blockIdx.x >> 3) % (unsigned)(P.n_copies >> 3));
        sum += (size_t)(c % (unsigned)P.n_copies) * P.copy_stride;
    }
    constexpr bool kDeposit = true;

    Packet<NDT, GEOM> p;
    Rng g;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    Dispenser dsp; dsp.next = 0; dsp.end = 0;
    int st = ST_NEED_EMIT;
    bool pool_empty = false;
    rng_init(g, P.seed_key, L.iter_tag, 0);
    p.inter = 1; p.tau_req = 0.0; p.tau_ach = 0.0;

    p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
    for (;;) {
        unsigned long long m_walk = __ballot(st == ST_WALK);
        unsigned long long m_int = __ballot(st == ST_NEED_INTERACT);
        unsigned long long m_emit = __ballot(st == ST_NEED_EMIT);
        unsigned long long m_re = P.any_intersect ? __ballot(st == ST_NEED_REEMIT) : 0ull;
        if (!(m_walk | m_int | m_emit | m_re)) break;

        // ---- packets re-absorbed by a source are re-emitted from it: iter_lucy.f90:155-185 ----
        if (m_re && (__popcll(m_re) >= L.emit_threshold || !m_walk)) {
            if (st == ST_NEED_REEMIT) {
                if ((long long)p.reabs == P.n_reabs_max) { cnt.killed_int++; st = ST_NEED_EMIT; }
                else {
                    const int inter = p.inter, reabs = p.reabs + 1, rid = p.reabs_id;
                    const double e = p.energy;
                    int source_id; Angle src_normal;
                    bool ok = emit_packet<NDT, GEOM>(P, W, p, g, cnt, source_id, src_normal, rid, e);
                    p.inter = inter; p.reabs = reabs;
                    if (!ok || geo_escaped(P, p.cell)) st = ST_NEED_EMIT;
                    else {
                        p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                        begin_integrate_lucy(P, p, g);
                        st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                    }
                }
            }
            m_walk = __ballot(st == ST_WALK);
            m_int = __ballot(st == ST_NEED_INTERACT);
            m_emit = __ballot(st == ST_NEED_EMIT);
        }

        // ---- interaction phase (deferred until enough lanes wait) ----
        if (m_int && (__popcll(m_int) >= L.interact_threshold || !m_walk)) {
            if (st == ST_NEED_INTERACT) {
                p.reabs = 0;
                if ((long long)p.inter == P.n_inter_max + 1) {
                    cnt.killed_int++; st = ST_NEED_EMIT;
                } else {
                    int scattered, dust_id;
                    bool ok = interact<NDT, GEOM>(P, p, g, cnt, scattered, dust_id);
                    bool killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered);
                    if (killed) st = ST_NEED_EMIT;
                    else if (P.mrw && mrw_loop_lucy<NDT, GEOM>(P, W, p, g, sum, cnt)) st = ST_NEED_EMIT;    // iter_lucy.f90:138-152
                    else {
                        p.inter++;
                        p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                        begin_integrate_lucy(P, p, g);
                        st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                    }
                }
            }
            m_walk = __ballot(st == ST_WALK);
            m_emit = __ballot(st == ST_NEED_EMIT);
        }

        // ---- emission phase ----
        if (m_emit && !pool_empty && (__popcll(m_emit) >= L.emit_threshold || !m_walk)) {
            unsigned long long id = 0;
            bool got = take_id(P, L, dsp, st == ST_NEED_EMIT, id);
            if (st == ST_NEED_EMIT) {
                if (!got) st = ST_DONE;
                else {
                    rng_init(g, P.seed_key, L.iter_tag, id);
                    int source_id;
                    Angle src_normal;
                    bool ok = emit_packet<NDT, GEOM>(P, W, p, g, cnt, source_id, src_normal);
                    p.reabs = 0;
                    if (!ok) st = ST_NEED_EMIT;
                    else if (geo_escaped(P, p.cell)) st = ST_NEED_EMIT;
                    else {
                        p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                        begin_integrate_lucy(P, p, g);
                        st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                    }
                }
            }
            if (__ballot(st == ST_DONE)) pool_empty = true;
            if (*((volatile int *)P.err) != 0) { if (st == ST_NEED_EMIT) st = ST_DONE; pool_empty = true; }
        } else if (m_emit && pool_empty) {
            if (st == ST_NEED_EMIT) st = ST_DONE;
        }

        // ---- walk phase: a few cell crossings per outer iteration ----
#pragma unroll 1
        for (int k = 0; k < walk_steps<GEOM>(); k++) {
            if (st == ST_WALK) st = walk_step<NDT, GEOM, kDeposit>(P, W, p, g, sum, cnt);
                if (!kDeposit && st == ST_ESCAPED) st = ST_NEED_EMIT;      // HYP_NO_DEPOSIT probe builds
        }
    }

    // per-wave reduction of the scalar tallies, one atomic per wave and slot
    double e = wave_sum(cnt.energy_current);
    double c = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    double ki = wave_sum((double)cnt.killed_int);
    double ni = wave_sum((double)cnt.interactions);
    if (__lane_id() == 0) {
        unsafeAtomicAdd(&P.tail[TAIL_ENERGY], e);
        unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], c);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
        if (ki != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], ki);
        unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], ni);
    }
}

// ---------------------------------------------------------------------------
// Imaging iteration: do_final / propagate (iter_final.f90:60-273) with
// peeloff_photon (images_peeled.f90:95-270)
// ---------------------------------------------------------------------------

// One walk along a fixed direction (grid_escape_tau, grid_escape_column_density): on Cartesian and cylindrical grids and octrees the wall search with one
// reciprocal per direction (car_find_wall_inv / oct_find_wall_inv: the same wall, the same t bit for bit, three operations per quotient
// instead of an IEEE division); inv = RN(1 / v), v_ok = every non-zero component at least 2^-400 in magnitude.
template <int GEOM>
__device__ __forceinline__ void walk_reciprocals(const DProblem &P, const double v[3], double inv[3], bool &v_ok)
{
    v_ok = false;
    if constexpr (GEOM == GEOM_CAR || GEOM == GEOM_OCT) {
        v_ok = true;
#pragma unroll
        for (int a = 0; a < 3; a++) { inv[a] = 1.0 / v[a]; v_ok = v_ok & ((v[a] == 0.0) | (fabs(v[a]) >= 0x1p-400)); }
    }
    if constexpr (GEOM == GEOM_CYL) {
        // cyl_find_wall_inv (hyp_polar.h): inv = (1 / v_xy^2, v_xy^2, 1 / v_z); its range conditions
        const double v2 = v[0] * v[0] + v[1] * v[1], r_out = P.w[0][P.n1];
        v_ok = r_out >= 0x1p-250 && r_out <= 0x1p250 && v2 >= 0x1p-200 && fabs(v[2]) >= 0x1p-200;
        inv[0] = 1.0 / v2; inv[1] = v2; inv[2] = 1.0 / v[2];
    }
}
template <int GEOM>
__device__ __forceinline__ bool find_wall_fixed_dir(const DProblem &P, const Walls &W, const double r[3], const double v[3], const double inv[3], bool v_ok,
                                                    const Cell<GEOM> &c, double &tmin, int im[3])
{
    if constexpr (GEOM == GEOM_OCT) return v_ok ? oct_find_wall_inv(P, r, v, inv, c, tmin, im) : geo_find_wall(P, W, r, v, c, tmin, im);
    else if constexpr (GEOM == GEOM_CAR) return v_ok ? car_find_wall_inv(P, W, r, v, inv, c, tmin, im) : geo_find_wall(P, W, r, v, c, tmin, im);
    else if constexpr (GEOM == GEOM_CYL) return v_ok ? cyl_find_wall_inv(P, r, v, c, inv[1], inv[0], inv[2], tmin, im) : geo_find_wall(P, W, r, v, c, tmin, im);
    // (spherical grids: dismissing the cone walls before solving them, as the tiled walk does for packets that have not interacted yet, was
    // measured in the peel-off walks too -- 703.7 -> 766.1 ms on the 400 x 200 imaging iteration: not done)
    else return geo_find_wall(P, W, r, v, c, tmin, im);
}

// grid_escape_tau: grid_propagate_3d.f90:377-480 -- optical depth from (r, ic, ow)
// along v to the edge of the grid (external observers: tmax = huge).
template <int NDT, int GEOM>
__device__ __forceinline__ double escape_tau(const DProblem &P, const Walls &W, const double r0[3], const double v[3],
                                             const Cell<GEOM> &cell0,
                                             const double chi[NDT], Rng &g, Counters &cnt, bool &killed, double tmax = HYP_DBL_MAX,
                                             uint32_t stream = 1u)
{
    const int nd = ndust<NDT>(P);
    double r[3] = {r0[0], r0[1], r0[2]};
    Cell<GEOM> c = cell0;
    geo_begin(r0, v, c);
    double tau = 0.0;
    killed = false;
    if (geo_escaped(P, c)) return 0.0;
    if (P.any_intersect) {      // a source in the way: grid_propagate_3d.f90:414-420 (tmax = huge for external observers)
        double t_source; int sid;
        find_nearest_source(P, r0, v, t_source, sid);
        if (t_source < tmax) { killed = true; return 0.0; }
    }
    double t_achieved = 0.0;       // inside observers stop at the observer: grid_propagate_3d.f90:446-452
    double inv[3] = {1.0, 1.0, 1.0}; bool v_ok;
    walk_reciprocals<GEOM>(P, v, inv, v_ok);
    for (;;) {
        if (g.countdown == 0) {
            g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp, stream);
            if (!geo_check_cell(P, W, r, v, c)) { cnt.killed_geo++; killed = true; return tau; }
        } else g.countdown--;
        double tmin; int im[3];
        if (!find_wall_fixed_dir<GEOM>(P, W, r, v, inv, v_ok, c, tmin, im)) { cnt.killed_geo++; killed = true; return tau; }
        const size_t base = geo_index(P, c) * (size_t)nd;
        bool finished = false;
        if (tmax < HYP_DBL_MAX) {
            if (t_achieved + tmin > tmax) { tmin = tmax - t_achieved; finished = true; }
            t_achieved += tmin;
        }
#pragma unroll
        for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
#pragma unroll
        for (int d = 0; d < NDT; d++) if (d < nd) tau += chi[d] * hyp_ldg(P.density + base + d) * tmin;
        cnt.crossings++;
        if (finished) return tau;
        geo_advance(P, r, c, im);
        if (geo_invalid(P, c)) { cnt.killed_geo++; killed = true; return tau; }     // amr: invalid_cell
        if (geo_escaped(P, c)) return tau;
    }
}

// fortranlib ipos: 0-based bin of x in n equal bins over [xmin, xmax]; -1 / n outside
__device__ __forceinline__ int ipos0(double xmin, double xmax, double x, int n)
{
    double f = (x - xmin) / (xmax - xmin);
    if (f < 0.0) return -1;
    if (f == 1.0) return n - 1;
    if (!(f < 1.0)) return n;
    return (int)floor(f * n);
}

struct PeelFlags { int scattered, reprocessed, n_scat, dust_id, source_id; };

// The direct light of a point source lands on ONE pixel per view and frequency bin, once per packet: atomics on one
// address serialise in L2 (~35 ns each; 85 of 146 ms of the configs[3] imaging iteration before this cache).  Each
// workgroup therefore keeps partial sums of the addresses its waves hit together in a small LDS table (open hashing
// without probing: an address that finds its slot taken goes to HBM directly) and adds them to the cube once, when the
// workgroup ends.
constexpr int HYP_IMG_CACHE = 256;       // entries (4 KB of LDS)
struct ImgCache { unsigned long long *keys; double *vals; };

__device__ __forceinline__ void img_cache_init(ImgCache &ic, unsigned long long *keys, double *vals)
{
    ic.keys = keys; ic.vals = vals;
    for (int i = threadIdx.x; i < HYP_IMG_CACHE; i += blockDim.x) { keys[i] = 0ull; vals[i] = 0.0; }
    __syncthreads();
}

// all waves of the workgroup, after their last deposit
__device__ __forceinline__ void img_cache_flush(const ImgCache &ic)
{
    __syncthreads();
    for (int i = threadIdx.x; i < HYP_IMG_CACHE; i += blockDim.x)
        if (ic.keys[i]) unsafeAtomicAdd((double *)ic.keys[i], ic.vals[i]);
}

// *addr += v; `claim`: the address may take a free slot of the table (it was hit by several lanes at once)
__device__ __forceinline__ void img_add(const ImgCache *ic, double *addr, double v, bool claim)
{
    if (v == 0.0) return;       // Q, U, V of unpolarised light
    if (ic) {
        const unsigned long long key = (unsigned long long)addr;
        const unsigned int slot = ((unsigned int)(key >> 3) * 2654435761u) >> 24;
        unsigned long long cur = ic->keys[slot];
        if (cur == 0ull && claim) { cur = atomicCAS(&ic->keys[slot], 0ull, key); if (cur == 0ull) cur = key; }
        if (cur == key) { atomicAdd(&ic->vals[slot], v); return; }
    }
    unsafeAtomicAdd(addr, v);
}

// Wave-cooperative accumulation of one value per lane into cube[key + i*stride],
// i = 0..n-1 (the Stokes components).  Lanes that hit the same element are summed
// in registers first (up to three distinct keys per call; the direct light of a
// point source puts every lane of the wave on ONE pixel, which would otherwise
// serialise as same-address atomics), the rest falls back to one atomic per lane.
// Must be called with all 64 lanes active; `key < 0` = nothing to add.
__device__ __forceinline__ void wave_accumulate(double *__restrict__ cube, double *__restrict__ cube2, long long key,
                                                size_t stride, int n, const double val[4], const ImgCache *ic = nullptr, bool serial = false)
{
    if (serial) {
        // option "reproducible": one lane at a time, in lane order, each lane's additions finished before the next lane's start -- the
        // sums in the cubes are then made in the program order of the (single) wave of the launch
        for (unsigned long long m = __ballot(key >= 0); m; m &= m - 1ull) {
            if ((int)__lane_id() == __ffsll((long long)m) - 1) {
                for (int i = 0; i < n; i++) {
                    unsafeAtomicAdd(&cube[key + (long long)i * (long long)stride], val[i]);
                    if (cube2) unsafeAtomicAdd(&cube2[key + (long long)i * (long long)stride], val[i] * val[i]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);
        }
        return;
    }
    unsigned long long todo = __ballot(key >= 0);
    for (int round = 0; round < 3 && todo; round++) {
        int leader = __ffsll((long long)todo) - 1;
        long long k0 = __shfl(key, leader, 64);
        bool same = key == k0;
        unsigned long long grp = __ballot(same);
        if (__popcll(grp) > 1) {
            for (int i = 0; i < n; i++) {
                double v = same ? val[i] : 0.0;
                double t = wave_sum(v);
                double t2 = cube2 ? wave_sum(v * v) : 0.0;
                if ((int)__lane_id() == leader) {
                    img_add(ic, &cube[k0 + (long long)i * (long long)stride], t, true);
                    if (cube2) img_add(ic, &cube2[k0 + (long long)i * (long long)stride], t2, true);
                }
            }
            if (same) key = -1;
        } else {
            break;   // keys are scattered: no point in grouping further
        }
        todo = __ballot(key >= 0);
    }
    if (key >= 0) {
        for (int i = 0; i < n; i++) {
            img_add(ic, &cube[key + (long long)i * (long long)stride], val[i], false);
            if (cube2) img_add(ic, &cube2[key + (long long)i * (long long)stride], val[i] * val[i], false);
        }
    }
}

// image_bin / image_bin_single: image_type.f90:408-524 -- element indices of the
// Stokes-I entry in the image and SED cubes (-1 = not binned)
__device__ __forceinline__ void image_bin_keys(const DProblem &P, const DPeeled &G, double nu, double energy, double s0,
                                               const PeelFlags &f, double x_image, double y_image, int iv,
                                               long long &k_img, long long &k_sed, int inu_filter = -1)
{
    k_img = -1; k_sed = -1;
    int inu = inu_filter >= 0 ? inu_filter
            : P.mono_which ? P.mono_inu - (G.inu_min - 1)       // image_type.f90:435-436
                           : ipos0(G.log10_nu_min, G.log10_nu_max, log10(nu), G.n_nu);
    if (inu < 0 || inu >= G.n_nu) return;
    if (energy != energy || s0 != s0) return;
    int o = f.scattered ? (f.reprocessed ? 4 : 3) : (f.reprocessed ? 2 : 1);
    int io = 0;
    if (G.track_origin == 1) io = o - 1;
    else if (G.track_origin == 2) {
        io = (o == 1) ? f.source_id : (o == 2) ? P.n_sources + f.dust_id
           : (o == 3) ? P.n_sources + P.n_dust + f.source_id : 2 * P.n_sources + P.n_dust + f.dust_id;
    } else if (G.track_origin == 3) {
        int ns = f.n_scat < G.track_n_scat + 1 ? f.n_scat : G.track_n_scat + 1;
        io = (f.reprocessed ? (G.track_n_scat + 2) : 0) + ns;
    }
    if (G.compute_image) {
        int ix = ipos0(G.x_min, G.x_max, x_image, G.n_x);
        int iy = ipos0(G.y_min, G.y_max, y_image, G.n_y);
        if (ix >= 0 && ix < G.n_x && iy >= 0 && iy < G.n_y)
            k_img = (long long)((((((size_t)0 * G.n_orig + io) * G.n_view + iv) * G.n_y + iy) * G.n_x + ix) * G.n_nu + inu);
    }
    if (G.compute_sed) {
        double lr = log10(sqrt(x_image * x_image + y_image * y_image));
        int ir;
        if (lr < G.log10_ap_min || G.n_ap == 1) ir = 0;
        else ir = ipos0(G.log10_ap_min, G.log10_ap_max, lr, G.n_ap - 1) + 1;
        if (ir >= 0 && ir < G.n_ap)
            k_sed = (long long)(((((size_t)0 * G.n_orig + io) * G.n_view + iv) * G.n_ap + ir) * G.n_nu + inu);
    }
}

// image_bin (image_type.f90:408-476) for all lanes of the wave: `live` lanes bin their packet (Stokes vector s already
// attenuated, energy, frequency nu) at (x_image, y_image) of view iv.  With filters the packet goes into every filter
// whose transmission at nu (linear interpolation, 0 outside the curve) is positive, with that weight (:467-475).
// Must be called by all 64 lanes.
template <bool PLAIN = false>
__device__ __forceinline__ void deposit_images(const DProblem &P, const DPeeled &G, bool live, double nu, double energy, const double s[4],
                                               const PeelFlags &f, double x_image, double y_image, int iv, const ImgCache *ic = nullptr)
{
    const size_t stride_img = (size_t)G.n_orig * G.n_view * G.n_y * G.n_x * G.n_nu, stride_sed = (size_t)G.n_orig * G.n_view * G.n_ap * G.n_nu;
    const bool use_filters = G.use_filters;
    const int n_pass = use_filters ? G.n_nu : 1;
    for (int pass = 0; pass < n_pass; pass++) {
        long long k_img = -1, k_sed = -1;
        double val[4] = {0.0, 0.0, 0.0, 0.0};
        if (live) {
            double tr = 1.0;
            if (use_filters) {
                const int o0 = (int)G.filt_off[pass], o1 = (int)G.filt_off[pass + 1];
                const double *fx = G.filt_nu + o0, *ft = G.filt_tr + o0;
                const int j = locate(fx, o1 - o0, nu);
                tr = j < 0 ? 0.0 : ft[j] + (nu - fx[j]) / (fx[j + 1] - fx[j]) * (ft[j + 1] - ft[j]);
            }
            if (tr > 0.0) {
                image_bin_keys(P, G, nu, energy, s[0], f, x_image, y_image, iv, k_img, k_sed, use_filters ? pass : -1);
                val[0] = s[0] * energy; val[1] = s[1] * energy; val[2] = s[2] * energy; val[3] = s[3] * energy;
                if (use_filters) { val[0] *= tr; val[1] *= tr; val[2] *= tr; val[3] *= tr; }
            }
        }
        // wave-uniform from here: combine lanes that hit the same pixel / SED bin
        if (G.compute_image) wave_accumulate(G.img, G.uncertainties ? G.img2 : nullptr, k_img, stride_img, G.n_stokes, val, ic, G.serial != 0);
        if (G.compute_sed) wave_accumulate(G.sed, G.uncertainties ? G.sed2 : nullptr, k_sed, stride_sed, G.n_stokes, val, ic, G.serial != 0);
    }
}

// Inside observers (images_peeled.f90:158-205, 410-421): the peel-off direction is towards the observer's position
// (vector3d_to_angle3d(r_peeloff - r)), d = distance to it.
__device__ __forceinline__ void inside_direction(const DPeeled &G, const double r[3], Angle &a_req, double &d)
{
    const double w0 = G.origin[0] - r[0], w1 = G.origin[1] - r[1], w2 = G.origin[2] - r[2];
    const double rxy = sqrt(w0 * w0 + w1 * w1);
    d = sqrt((w0 * w0 + w1 * w1) + w2 * w2);
    a_req.cost = w2 / d; a_req.sint = rxy / d;
    if (rxy > 0.0) { a_req.cosp = w0 / rxy; a_req.sinp = w1 / rxy; } else { a_req.cosp = 1.0; a_req.sinp = 0.0; }
}

// sky position (longitude, latitude in degrees around the group's viewing direction) of a packet travelling along a_req
__device__ __forceinline__ void inside_sky_position(const DPeeled &G, int iv, const Angle &a_req, double &x_image, double &y_image)
{
    const double vt = G.view[4 * iv + 0], vs = G.view[4 * iv + 1], vcp = G.view[4 * iv + 2], vsp = G.view[4 * iv + 3];
    double va0, va1, va2;
    angle_to_vector(a_req, va0, va1, va2);
    const double sx = (va0 * vcp + va1 * vsp) * vs + va2 * vt;
    const double sy = -va0 * vsp + va1 * vcp;
    const double sz = -(va0 * vcp + va1 * vsp) * vt + va2 * vs;
    const double rad2deg = 180.0 / HYP_PI;
    x_image = atan2(sy, sx) * rad2deg;
    y_image = atan2(sqrt(sx * sx + sy * sy), sz) * rad2deg - 90.0;
    const double ax = x_image - G.x_max, ay = y_image - G.y_min;       // Fortran modulo(a, 360)
    x_image = G.x_max + (ax - 360.0 * floor(ax / 360.0));
    y_image = G.y_min + (ay - 360.0 * floor(ay / 360.0));
}

// The propagation checks of a peel-off walk draw from their own stream, keyed by (packet, peel-off event, view), not from
// the packet's: the walk then is a function of the event alone and can be done by any lane, in any order, in another kernel.
// 16 blocks per walk (one check every ~1 / propagation_check_frequency steps).
__device__ __forceinline__ void peel_rng(const DProblem &P, Rng &gp, uint32_t key0, uint32_t key1, unsigned long long id, unsigned int peel_seq,
                                         int view_global)
{
    gp.key0 = key0; gp.key1 = key1; gp.id_lo = (uint32_t)id; gp.id_hi = (uint32_t)(id >> 32);
    gp.blk_a = 0; gp.have_a = 0; gp.buf_a = 0.0;
    gp.blk_b = (peel_seq * (unsigned int)P.n_views_total + (unsigned int)view_global) * 16u;
    gp.countdown = rng_check_gap(gp, P.check_p, P.check_log1mp, 2u);
}

// peeloff_photon, external observers: images_peeled.f90:95-270.  Called by ALL
// lanes of the wave (`active` = this lane has a packet to peel) so that the image
// deposits can be combined across lanes.
template <int NDT, int GEOM, bool PLAIN = false, bool LEAN = false>
__device__ __forceinline__ void peeloff(const DProblem &P, const Walls &W, const Packet<NDT, GEOM> &p, bool active,
                                        const Angle &a_prev, const double s_prev[4], int last, bool last_isotropic,
                                        const PeelFlags &f, Rng &g, Counters &cnt, const ImgCache *ic = nullptr)
{
    for (int ig = 0; ig < P.n_peeled; ig++) {
        const DPeeled &G = P.peeled[ig];
        for (int iv = 0; iv < G.n_view; iv++) {
            bool live = false;
            double s_out[4] = {0.0, 0.0, 0.0, 0.0}, x_out = 0.0, y_out = 0.0;
            if (active) {
                Angle a_req;
                a_req.cost = G.view[4 * iv + 0]; a_req.sint = G.view[4 * iv + 1];
                a_req.cosp = G.view[4 * iv + 2]; a_req.sinp = G.view[4 * iv + 3];
                double d_obs = 0.0;
                if ((!PLAIN && !LEAN && G.inside_observer)) inside_direction(G, p.r, a_req, d_obs);
                double s[4];
                if (last_isotropic) {
                    s[0] = 1.0; s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
                } else if (last == LAST_SR) {
                    // source_emit_peeloff + emit_from_extern_{sph,box}_peeloff: source_type.f90:512-533,811-820,909-933
                    // (a_prev holds the inward surface normal at the emission point)
                    double mu = 0.0;
                    if (P.sources[f.source_id].peeloff) {
                        double n0, n1, n2, q0, q1, q2;
                        angle_to_vector(a_prev, n0, n1, n2);
                        angle_to_vector(a_req, q0, q1, q2);
                        mu = q0 * n0 + q1 * n1 + q2 * n2;
                        if (mu < 0.0) mu = 0.0;
                    }
                    // emit_from_sphere_peeloff :692-707 for limb-darkened spheres, 4 mu otherwise
                    const DSource &S = P.sources[f.source_id];
                    s[0] = (S.type == 2 && S.limb_darkening) ? 2.0 * (1.5 * mu * mu + mu) : 4.0 * mu;
                    s[1] = 0.0; s[2] = 0.0; s[3] = 0.0;
                } else if (last != LAST_DS) {
                    s[0] = s_prev[0]; s[1] = s_prev[1]; s[2] = s_prev[2]; s[3] = s_prev[3];
                } else {
                    // dust_scatter_peeloff: dust_type_4elem.f90:421-444
                    const DDust &D = P.dust[f.dust_id];
                    s[0] = s_prev[0]; s[1] = s_prev[1]; s[2] = s_prev[2]; s[3] = s_prev[3];
                    Angle a_scat;
                    difference_angle(a_prev, a_req, a_scat);
                    if (a_scat.cost < D.mu_min || a_scat.cost > D.mu_max) { s[0] = s[1] = s[2] = s[3] = 0.0; }
                    else {
                        double P1, P2, P3, P4;
                        interp_P(D, a_scat.cost, p.nu, P1, P2, P3, P4);
                        scatter_stokes(s, a_prev, a_scat, a_req, P1, P2, P3, P4);
                    }
                }
                double v[3];
                angle_to_vector(a_req, v[0], v[1], v[2]);
                // the copy keeps the packet's wall flags; Cartesian place_in_cell resets them
                Cell<GEOM> c = p.cell;
                bool ok = geo_place(P, W, p.r, v, c);
                if (!ok) cnt.killed_geo++;
                double d = (!PLAIN && !LEAN && G.inside_observer) ? d_obs : -(v[0] * p.r[0] + v[1] * p.r[1] + v[2] * p.r[2]);
                ok = ok && !(d < G.d_min || d > G.d_max);
                double dr0 = p.r[0] - G.origin[0], dr1 = p.r[1] - G.origin[1], dr2 = p.r[2] - G.origin[2];
                double x_image = dr1 * a_req.cosp - dr0 * a_req.sinp;
                double y_image = dr2 * a_req.sint - dr1 * a_req.cost * a_req.sinp - dr0 * a_req.cost * a_req.cosp;
                if ((!PLAIN && !LEAN && G.inside_observer)) inside_sky_position(G, iv, a_req, x_image, y_image);
                bool inside = false;
                if (G.compute_image)
                    inside = ((x_image >= G.x_min && x_image <= G.x_max) || (x_image <= G.x_min && x_image >= G.x_max)) &&
                             ((y_image >= G.y_min && y_image <= G.y_max) || (y_image <= G.y_min && y_image >= G.y_max));
                if (!inside && G.compute_sed) inside = x_image * x_image + y_image * y_image <= G.ap_max * G.ap_max;
                ok = ok && inside;
                if (ok) {
                    double tau = 0.0; bool killed = false;
                    Rng gp;
                    peel_rng(P, gp, g.key0, g.key1, ((unsigned long long)g.id_hi << 32) | g.id_lo, p.peel_seq, G.view_base + iv);
                    if (!G.ignore_optical_depth)
                        tau = escape_tau<NDT, GEOM>(P, W, p.r, v, c, p.chi, gp, cnt, killed, (!PLAIN && !LEAN && G.inside_observer) ? d_obs : HYP_DBL_MAX, 2u);
                    if (!killed) {
                        if ((!PLAIN && !LEAN && G.inside_observer)) {        // 1 / (4 pi d^2) flux dilution: images_peeled.f90:236
                            const double dil = 1.0 / (4.0 * HYP_PI * (d_obs * d_obs));
                            s[0] = s[0] * dil; s[1] = s[1] * dil; s[2] = s[2] * dil; s[3] = s[3] * dil;
                        }
                        double att = exp(-tau);
                        s[0] *= att; s[1] *= att; s[2] *= att; s[3] *= att;
                        live = true; s_out[0] = s[0]; s_out[1] = s[1]; s_out[2] = s[2]; s_out[3] = s[3]; x_out = x_image; y_out = y_image;
                    }
                }
            }
            deposit_images<PLAIN>(P, G, live, p.nu, p.energy, s_out, f, x_out, y_out, iv, ic);
        }
    }
}

// geo%volume of every geometry
__device__ __forceinline__ double cell_volume(const DProblem &P, size_t ic)
{
    if (P.grid_type == 3) return P.vor_volume[ic];
    if (P.grid_type == 4) {
        const AmrGrid &g = P.amr_grids[P.amr_cell_grid[ic]];      // grid%volume, grid_geometry_amr.f90:143-151
        return ((g.hi[0] - g.lo[0]) / (double)g.n[0]) * ((g.hi[1] - g.lo[1]) / (double)g.n[1]) * ((g.hi[2] - g.lo[2]) / (double)g.n[2]);
    }
    if (P.grid_type == 2) {
        int lev = P.oct_cells[ic].level;
        return ldexp(P.oct_half[0], -lev) * ldexp(P.oct_half[1], -lev) * ldexp(P.oct_half[2], -lev) * 8.0;
    }
    int i1 = (int)(ic % P.n1);
    size_t t = ic / P.n1;
    int i2 = (int)(t % P.n2), i3 = (int)(t / P.n2);
    if (P.grid_type == 5) {     // grid_geometry_spherical_3d.f90:138-155: dr3 * dcost * dphi / 3
        const double a = P.w[0][i1], b = P.w[0][i1 + 1];
        return (b * b * b - a * a * a) * (P.wcost[i2] - P.wcost[i2 + 1]) * (P.w[2][i3 + 1] - P.w[2][i3]) / 3.0;
    }
    if (P.grid_type == 6) {     // grid_geometry_cylindrical_3d.f90:135-147: dw2 * dz * dphi / 2
        const double a = P.w[0][i1], b = P.w[0][i1 + 1];
        return (b * b - a * a) * (P.w[1][i2 + 1] - P.w[1][i2]) * (P.w[2][i3 + 1] - P.w[2][i3]) / 2.0;
    }
    return (P.w[0][i1 + 1] - P.w[0][i1]) * (P.w[1][i2 + 1] - P.w[1][i2]) * (P.w[2][i3 + 1] - P.w[2][i3]);
}

// ---------------------------------------------------------------------------
// Raytracing iteration: do_raytracing (iter_raytracing.f90:30-143) with the
// polychromatic branch of peeloff_photon (images_peeled.f90:218-254)
// ---------------------------------------------------------------------------

// grid_escape_column_density: grid_propagate_3d.f90:482-582
template <int NDT, int GEOM>
__device__ __forceinline__ void escape_column(const DProblem &P, const Walls &W, const double r0[3], const double v[3],
                                              const Cell<GEOM> &cell0, double col[NDT], Rng &g, Counters &cnt, bool &killed, double tmax = HYP_DBL_MAX)
{
    const int nd = ndust<NDT>(P);
    double r[3] = {r0[0], r0[1], r0[2]};
    Cell<GEOM> c = cell0;
    geo_begin(r0, v, c);
    killed = false;
#pragma unroll
    for (int d = 0; d < NDT; d++) col[d] = 0.0;
    if (geo_escaped(P, c)) return;
    if (P.any_intersect) {      // grid_propagate_3d.f90:516-523
        double t_source; int sid;
        find_nearest_source(P, r0, v, t_source, sid);
        if (t_source < tmax) { killed = true; return; }
    }
    double t_current = 0.0;
    double inv[3] = {1.0, 1.0, 1.0}; bool v_ok;
    walk_reciprocals<GEOM>(P, v, inv, v_ok);
    for (;;) {
        if (g.countdown == 0) {
            g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp);
            if (!geo_check_cell(P, W, r, v, c)) { cnt.killed_geo++; killed = true; return; }
        } else g.countdown--;
        double tmin; int im[3];
        if (!find_wall_fixed_dir<GEOM>(P, W, r, v, inv, v_ok, c, tmin, im)) { cnt.killed_geo++; killed = true; return; }
        const size_t base = geo_index(P, c) * (size_t)nd;
        bool finished = false;
        if (tmax < HYP_DBL_MAX) {       // inside observers: grid_propagate_3d.f90:551-557
            if (t_current + tmin > tmax) { tmin = tmax - t_current; finished = true; }
            t_current += tmin;
        }
#pragma unroll
        for (int a = 0; a < 3; a++) r[a] = r[a] + tmin * v[a];
#pragma unroll
        for (int d = 0; d < NDT; d++) if (d < nd) col[d] += hyp_ldg(P.density + base + d) * tmin;
        cnt.crossings++;
        if (finished) return;
        geo_advance(P, r, c, im);
        if (geo_invalid(P, c)) { cnt.killed_geo++; killed = true; return; }
        if (geo_escaped(P, c)) return;
    }
}

// random_position_cell: cartesian_3d.f90:383-394, octree.f90:397-408, amr.f90:728-741, voronoi.f90:285-310.
// (x, y, z) are three uniforms already drawn by the caller; only the Voronoi rejection loop draws more from g.
template <int GEOM>
__device__ __forceinline__ bool random_position_cell(const DProblem &P, size_t ic, double x, double y, double z, double r[3], Rng &g)
{
    if (GEOM == GEOM_CAR) {
        int i1 = (int)(ic % P.n1);
        size_t t = ic / P.n1;
        int i2 = (int)(t % P.n2), i3 = (int)(t / P.n2);
        r[0] = x * (P.w[0][i1 + 1] - P.w[0][i1]) + P.w[0][i1];
        r[1] = y * (P.w[1][i2 + 1] - P.w[1][i2]) + P.w[1][i2];
        r[2] = z * (P.w[2][i3 + 1] - P.w[2][i3]) + P.w[2][i3];
        return true;
    }
    if (GEOM == GEOM_OCT) {
        const OctCell &o = P.oct_cells[ic];
        r[0] = (2.0 * x - 1.0) * ldexp(P.oct_half[0], -o.level) + o.x;
        r[1] = (2.0 * y - 1.0) * ldexp(P.oct_half[1], -o.level) + o.y;
        r[2] = (2.0 * z - 1.0) * ldexp(P.oct_half[2], -o.level) + o.z;
        return true;
    }
    if (GEOM == GEOM_AMR) {
        const AmrGrid &g = P.amr_grids[P.amr_cell_grid[ic]];
        const size_t l = ic - g.start;
        const int i[3] = {(int)(l % g.n[0]), (int)((l / g.n[0]) % g.n[1]), (int)(l / ((size_t)g.n[0] * g.n[1]))};
        const double u[3] = {x, y, z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double wl = P.amr_walls[g.w_off[a] + i[a]], wu = P.amr_walls[g.w_off[a] + i[a] + 1];
            r[a] = u[a] * (wu - wl) + wl;
        }
        return true;
    }
    if (GEOM == GEOM_SPH || GEOM == GEOM_CYL) { polar_random_position<GEOM>(P, ic, x, y, z, r); return true; }
    if (GEOM == GEOM_VOR && P.vor_bb) {
        // positions uniform in the cell's bounding box until one lies in the cell (the reference asks its kd-tree for the
        // nearest site; a point is in cell ic iff none of the cell's neighbours has its site closer)
        const double *bb = P.vor_bb + 6 * ic;
        const double s0 = P.vor_sites[3 * ic], s1 = P.vor_sites[3 * ic + 1], s2 = P.vor_sites[3 * ic + 2];
        for (int trial = 0; trial < 1000000; trial++) {
            if (trial) { x = rng_uniform(g); y = rng_uniform(g); z = rng_uniform(g); }
            r[0] = bb[0] + x * (bb[3] - bb[0]); r[1] = bb[1] + y * (bb[4] - bb[1]); r[2] = bb[2] + z * (bb[5] - bb[2]);
            const double a0 = r[0] - s0, a1 = r[1] - s1, a2 = r[2] - s2;
            const double d0 = (a0 * a0 + a1 * a1) + a2 * a2;
            bool inside = true;
            for (int k = P.vor_idx[ic]; k < P.vor_idx[ic + 1] && inside; k++) {
                const int j = P.vor_neigh[k];
                if (j < 0) continue;
                const double b0 = r[0] - P.vor_sites[3 * j], b1 = r[1] - P.vor_sites[3 * j + 1], b2 = r[2] - P.vor_sites[3 * j + 2];
                if ((b0 * b0 + b1 * b1) + b2 * b2 < d0) inside = false;
            }
            if (inside) return true;
        }
        return false;      // "too many samples"
    }
    return false;      // voronoi without bounding boxes
}

// Polychromatic peel-off of a freshly emitted packet: the whole binned spectrum of its emitter,
// attenuated per frequency bin by the column densities along the line of sight, goes into
// Stokes I of one pixel / aperture (image_bin_raytraced, image_type.f90:527-606).  Called by all
// lanes of the wave.  emiss_dust < 0: source packet (spectrum of source f.source_id).
template <int NDT, int GEOM>
__device__ __forceinline__ void peeloff_poly(const DProblem &P, const Walls &W, const double r[3], bool active, double energy,
                                             bool isotropic, const Angle &src_normal, int emiss_dust, int var_id, double var_frac,
                                             const PeelFlags &f, Rng &g, Counters &cnt, const ImgCache *ic = nullptr)
{
    const int nd = ndust<NDT>(P);
    for (int ig = 0; ig < P.n_peeled; ig++) {
        const DPeeled &G = P.peeled[ig];
        for (int iv = 0; iv < G.n_view; iv++) {
            long long k_img = -1, k_sed = -1;
            double col[NDT], s0 = 0.0;
#pragma unroll
            for (int d = 0; d < NDT; d++) col[d] = 0.0;
            if (active) {
                Angle a_req;
                a_req.cost = G.view[4 * iv + 0]; a_req.sint = G.view[4 * iv + 1];
                a_req.cosp = G.view[4 * iv + 2]; a_req.sinp = G.view[4 * iv + 3];
                double d_obs = 0.0;
                if (G.inside_observer) inside_direction(G, r, a_req, d_obs);
                if (isotropic) s0 = 1.0;
                else {      // source_emit_peeloff of the external sources: source_type.f90:512-533
                    double mu = 0.0;
                    if (P.sources[f.source_id].peeloff) {
                        double n0, n1, n2, q0, q1, q2;
                        angle_to_vector(src_normal, n0, n1, n2);
                        angle_to_vector(a_req, q0, q1, q2);
                        mu = q0 * n0 + q1 * n1 + q2 * n2;
                        if (mu < 0.0) mu = 0.0;
                    }
                    const DSource &S = P.sources[f.source_id];
                    s0 = (S.type == 2 && S.limb_darkening) ? 2.0 * (1.5 * mu * mu + mu) : 4.0 * mu;
                }
                double v[3];
                angle_to_vector(a_req, v[0], v[1], v[2]);
                Cell<GEOM> c;
                geo_clear_wall(c);
                bool ok = geo_place(P, W, r, v, c);
                if (!ok) cnt.killed_geo++;
                double d = G.inside_observer ? d_obs : -(v[0] * r[0] + v[1] * r[1] + v[2] * r[2]);
                ok = ok && !(d < G.d_min || d > G.d_max);
                double dr0 = r[0] - G.origin[0], dr1 = r[1] - G.origin[1], dr2 = r[2] - G.origin[2];
                double x_image = dr1 * a_req.cosp - dr0 * a_req.sinp;
                double y_image = dr2 * a_req.sint - dr1 * a_req.cost * a_req.sinp - dr0 * a_req.cost * a_req.cosp;
                if (G.inside_observer) inside_sky_position(G, iv, a_req, x_image, y_image);
                bool inside = false;
                if (G.compute_image)
                    inside = ((x_image >= G.x_min && x_image <= G.x_max) || (x_image <= G.x_min && x_image >= G.x_max)) &&
                             ((y_image >= G.y_min && y_image <= G.y_max) || (y_image <= G.y_min && y_image >= G.y_max));
                if (!inside && G.compute_sed) inside = x_image * x_image + y_image * y_image <= G.ap_max * G.ap_max;
                ok = ok && inside;
                if (ok) {
                    bool killed = false;
                    if (!G.ignore_optical_depth) escape_column<NDT, GEOM>(P, W, r, v, c, col, g, cnt, killed, G.inside_observer ? d_obs : HYP_DBL_MAX);
                    if (G.inside_observer) s0 = s0 * (1.0 / (4.0 * HYP_PI * (d_obs * d_obs)));      // images_peeled.f90:236
                    if (!killed && energy == energy) {
                        // origin slot and pixel / aperture of the first frequency bin
                        int o = f.scattered ? (f.reprocessed ? 4 : 3) : (f.reprocessed ? 2 : 1);
                        int io = 0;
                        if (G.track_origin == 1) io = o - 1;
                        else if (G.track_origin == 2) {
                            io = (o == 1) ? f.source_id : (o == 2) ? P.n_sources + f.dust_id
                               : (o == 3) ? P.n_sources + P.n_dust + f.source_id : 2 * P.n_sources + P.n_dust + f.dust_id;
                        } else if (G.track_origin == 3) {
                            int ns = f.n_scat < G.track_n_scat + 1 ? f.n_scat : G.track_n_scat + 1;
                            io = (f.reprocessed ? (G.track_n_scat + 2) : 0) + ns;
                        }
                        if (G.compute_image) {
                            int ix = ipos0(G.x_min, G.x_max, x_image, G.n_x);
                            int iy = ipos0(G.y_min, G.y_max, y_image, G.n_y);
                            if (ix >= 0 && ix < G.n_x && iy >= 0 && iy < G.n_y)
                                k_img = (long long)(((((size_t)io * G.n_view + iv) * G.n_y + iy) * G.n_x + ix) * G.n_nu);
                        }
                        if (G.compute_sed) {
                            double lr = log10(sqrt(x_image * x_image + y_image * y_image));
                            int ir;
                            if (lr < G.log10_ap_min || G.n_ap == 1) ir = 0;
                            else ir = ipos0(G.log10_ap_min, G.log10_ap_max, lr, G.n_ap - 1) + 1;
                            if (ir >= 0 && ir < G.n_ap) k_sed = (long long)((((size_t)io * G.n_view + iv) * G.n_ap + ir) * G.n_nu);
                        }
                    }
                }
            }
            // wave-uniform from here: one frequency bin at a time, lanes on the same pixel combined
            const double *le = nullptr, *ss = nullptr;
            if (k_img >= 0 || k_sed >= 0) {
                if (emiss_dust >= 0) le = G.dust_log10_em + ((size_t)emiss_dust * G.nj_stride + var_id) * G.n_nu;
                else ss = G.src_spec + (size_t)f.source_id * G.n_nu;
            }
            for (int iw = 0; iw < G.n_nu; iw++) {
                double val[4] = {0.0, 0.0, 0.0, 0.0};
                if (k_img >= 0 || k_sed >= 0) {
                    double sp;
                    if (le) {       // get_dust_emissivity: images_peeled.f90:451-505
                        sp = exp10((le[G.n_nu + iw] - le[iw]) * var_frac + le[iw]);
                        if (sp != sp) sp = 0.0;
                    } else sp = ss[iw];
                    sp = sp * s0 * energy;
#pragma unroll
                    for (int d = 0; d < NDT; d++) if (d < nd) sp = sp * exp(-col[d] * G.dust_chi[(size_t)d * G.n_nu + iw]);
                    val[0] = sp;
                }
                if (G.compute_image)
                    wave_accumulate(G.img, G.uncertainties ? G.img2 : nullptr, k_img >= 0 ? k_img + iw : -1, 0, 1, val, ic, G.serial != 0);
                if (G.compute_sed)
                    wave_accumulate(G.sed, G.uncertainties ? G.sed2 : nullptr, k_sed >= 0 ? k_sed + iw : -1, 0, 1, val, ic, G.serial != 0);
            }
        }
    }
}

// which = 0: packets from the sources (:56-76); which = 1: thermal packets from the grid
// (:96-126 with emit_from_grid, grid_physics_3d.f90:691-753).  L.iter_tag keys the RNG streams.
template <int NDT, int GEOM>
__global__ __launch_bounds__(256, 2) void ray_kernel(const DProblem *__restrict__ Pp, LaunchParams L, int which, double n_total)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    // every source packet of a point source puts its whole spectrum on one pixel: all frequency bins of it are hot
    __shared__ unsigned long long img_keys[HYP_IMG_CACHE];
    __shared__ double img_vals[HYP_IMG_CACHE];
    ImgCache ic;
    img_cache_init(ic, img_keys, img_vals);
    Rng g;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    Dispenser dsp; dsp.next = 0; dsp.end = 0;
    rng_init(g, P.seed_key, L.iter_tag, 0);
    for (;;) {
        unsigned long long id = 0;
        bool got = take_id(P, L, dsp, true, id);
        if (!__ballot(got)) break;
        bool active = got;
        double r[3] = {0.0, 0.0, 0.0}, energy = 0.0, var_frac = 0.0;
        bool isotropic = true;
        Angle src_normal; src_normal.cost = 1.0; src_normal.sint = 0.0; src_normal.cosp = 1.0; src_normal.sinp = 0.0;
        int emiss_dust = -1, var_id = 0;
        PeelFlags f; f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = 0;
        if (active) {
            rng_init(g, P.seed_key, L.iter_tag, id);
            if (which == 0) {
                Packet<NDT, GEOM> p;
                int source_id = 0;
                bool ok = emit_packet<NDT, GEOM>(P, W, p, g, cnt, source_id, src_normal);
                f.source_id = source_id;
                isotropic = P.sources[source_id].type == 1 || P.sources[source_id].type == 8 || P.sources[source_id].type == 4;
                if (ok && p.emiss_dust >= 0) {     // 'lte' source: the packet carries the emissivity of the dust of its cell
                    emiss_dust = p.emiss_dust;
                    const size_t k = geo_index(P, p.cell) * (size_t)ndust<NDT>(P) + (size_t)emiss_dust;
                    var_id = P.jnu_id[k]; var_frac = P.jnu_frac[k];
                    f.dust_id = 0;
                }
                r[0] = p.r[0]; r[1] = p.r[1]; r[2] = p.r[2];
                energy = p.energy * P.energy_total / n_total;
                active = ok;
            } else {
                const int nd = ndust<NDT>(P);
                double xi = rng_uniform(g);
                int d = (int)ceil(xi * (double)nd); if (d < 1) d = 1;
                f.dust_id = d - 1; f.reprocessed = 1; emiss_dust = d - 1;
                xi = rng_uniform(g);        // random_masked_cell: grid_geometry_common_3d.f90:104-115
                long long im = (long long)ceil(xi * (double)P.n_masked); if (im < 1) im = 1;
                const size_t ic = P.mask_map[im - 1];
                const double x = rng_uniform(g), y = rng_uniform(g), z = rng_uniform(g);
                if (!random_position_cell<GEOM>(P, ic, x, y, z, r, g)) { raise_error(P, ERR_RAY_GRID, 0.0, 0.0, 0.0); active = false; }
                Angle a; random_sphere_angle(g, a);         // drawn like the reference, not used by the peel-off
                const size_t k = ic * (size_t)nd + (size_t)(d - 1);
                const double eat = P.energy_abs_tot[d - 1];
                if (eat > 0.0) {
                    const double mass = P.density[k] * cell_volume(P, ic);
                    energy = P.specific_energy[k] * mass * (double)P.n_masked / eat;
                } else energy = 0.0;
                var_id = P.jnu_id[k]; var_frac = P.jnu_frac[k];
                g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp);
                if (energy > 0.0) energy = energy * eat / n_total * (double)nd;
                else active = false;
            }
        }
        peeloff_poly<NDT, GEOM>(P, W, r, active, energy, isotropic, src_normal, emiss_dust, var_id, var_frac, f, g, cnt, &ic);
        if (*((volatile int *)P.err) != 0) break;
    }
    img_cache_flush(ic);
    double c = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    if (__lane_id() == 0) {
        unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], c);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
    }
}

// dust_sample_emit_probability (dust_type_4elem.f90:356-377) from the per-row table of the run's frequencies
__device__ __forceinline__ double dust_emit_probability(const DProblem &P, const DDust &D, int jid, double frac)
{
    const double l1 = D.mono_log10_prob[(size_t)jid * P.n_frequencies + P.mono_inu];
    const double l2 = D.mono_log10_prob[(size_t)(jid + 1) * P.n_frequencies + P.mono_inu];
    if (l1 == -HYP_INF || l2 == -HYP_INF) return 0.0;
    return exp10(l1 + frac * (l2 - l1));
}

// emit_from_monochromatic_grid_pdf: grid_monochromatic.f90:119-174.  Returns false when the packet carries no
// energy (nothing emits at this frequency in the chosen dust type) or on a fatal error.
template <int NDT, int GEOM>
__device__ __forceinline__ bool emit_mono_dust(const DProblem &P, const Walls &W, Packet<NDT, GEOM> &p, Rng &g, Counters &cnt, int &dust_id)
{
    const int nd = ndust<NDT>(P);
    p.nu = P.mono_nu;
    if (!update_optconsts<NDT, GEOM>(P, p)) return false;
    int d = (int)ceil(rng_uniform(g) * (double)nd);
    if (d < 1) d = 1;
    d -= 1;
    dust_id = d;
    if (P.mono_mean_prob[d] == 0.0) return false;
    // grid_sample_pdf_map: first cell whose cumulative exceeds xi
    const double *cdf = P.mono_cdf + (size_t)d * P.n_cells;
    const double xi = rng_uniform(g);
    size_t lo = 0, hi = (size_t)P.n_cells - 1;
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (xi < cdf[mid]) hi = mid; else lo = mid + 1; }
    const double x = rng_uniform(g), y = rng_uniform(g), z = rng_uniform(g);
    if (!random_position_cell<GEOM>(P, lo, x, y, z, p.r, g)) { raise_error(P, ERR_RAY_GRID, 0.0, 0.0, 0.0); return false; }
    random_sphere_angle(g, p.a);
    angle_to_vector(p.a, p.v[0], p.v[1], p.v[2]);
    p.s[0] = 1.0; p.s[1] = 0.0; p.s[2] = 0.0; p.s[3] = 0.0;
    g.countdown = rng_check_gap(g, P.check_p, P.check_log1mp);
    geo_clear_wall(p.cell);
    if (!geo_place(P, W, p.r, p.v, p.cell)) { cnt.killed_geo++; return false; }
    p.inter = 1; p.peel_seq = 0;
    p.energy = P.mono_mean_prob[d] * P.energy_abs_tot[d] / P.mono_n_total * (double)nd;
    return p.energy > 0.0;
}

// binned_images_bin_photon: images_binned.f90:58-81.  Called by ALL lanes of the wave (`active` = this lane's packet
// just left the grid); the deposit goes through the same wave-combined path as the peel-off.
template <int NDT, int GEOM>
__device__ __forceinline__ void bin_escaped(const DProblem &P, const Packet<NDT, GEOM> &p, bool active, const PeelFlags &f)
{
    const DPeeled &G = P.peeled[P.binned];
    bool live = false;
    double x_image = 0.0, y_image = 0.0;
    int iv = 0;
    if (active) {
        double phi = atan2(p.a.sinp, p.a.cosp);
        if (phi < 0.0) phi = phi + HYP_TWOPI;
        const int it = ipos0(-1.0, 1.0, p.a.cost, P.n_bin_theta), ip = ipos0(0.0, HYP_TWOPI, phi, P.n_bin_phi);
        if (it >= 0 && it < P.n_bin_theta && ip >= 0 && ip < P.n_bin_phi) {
            x_image = p.r[1] * p.a.cosp - p.r[0] * p.a.sinp;
            y_image = p.r[2] * p.a.sint - p.r[1] * p.a.cost * p.a.sinp - p.r[0] * p.a.cost * p.a.cosp;
            iv = P.n_bin_phi * it + ip;
            live = true;
        }
    }
    deposit_images(P, G, live, p.nu, p.energy, p.s, f, x_image, y_image, iv);
}

// forced first interaction: forced_interaction.f90:23-133
__device__ __forceinline__ void forced_interaction(const DProblem &P, double tau_escape, double xi, double &tau, double &weight)
{
    double ome = tau_escape > 1e-7 ? 1.0 - exp(-tau_escape) : tau_escape;
    if (P.forced_algo == 2) {
        double alpha = (1.0 - P.baes16_xi) / ome, beta = P.baes16_xi / tau_escape;
        double tlo = 0.0, thi = tau_escape, t = 0.0;
        for (int i = 0; i < 60; i++) {
            t = 0.5 * (tlo + thi);
            double test = t > 1e-7 ? alpha * (1.0 - exp(-t)) + beta * t : alpha * t + beta * t;
            if (test > xi) thi = t; else tlo = t;
        }
        t = 0.5 * (tlo + thi);
        tau = t; weight = 1.0 / (alpha + beta * exp(t));
    } else {
        tau = -log(1.0 - xi * ome); weight = ome;
    }
}

// PLAIN: the polychromatic peel-off iteration without anything optional -- point sources only, no monochromatic launch,
// no modified random walk, no re-absorbing sources, no binned images, no inside observers (the host checks; filters only change the deposit;
// with an inside observer the problem still qualifies for the deferred schedule, whose peel kernel handles it, but not for this kernel).
// Those paths cost registers even where a problem never takes them; this is the imaging kernel of BASELINE configs[3].
// LEAN (with PLAIN = false): any sources (spheres with limb darkening / spots / re-absorption, maps, external and plane-parallel ones),
// but no modified random walk, no monochromatic launch, no binned images and no inside observers (the host checks) -- what a model
// lit by a star with a radius needs.  Those four paths are what the general kernel spills for: 350 -> 125 spilled VGPRs on octrees.
template <int NDT, int GEOM, bool PLAIN, bool LEAN = false>
__global__ __launch_bounds__(256, HYP_FINAL_WAVES) void final_kernel(const DProblem *__restrict__ Pp, LaunchParams L)
{
    extern __shared__ double lds[];
    const DProblem &P = *Pp;
    Walls W;
    stage_walls<GEOM>(P, lds, W);
    __shared__ unsigned long long img_keys[HYP_IMG_CACHE];
    __shared__ double img_vals[HYP_IMG_CACHE];
    ImgCache ic;
    img_cache_init(ic, img_keys, img_vals);
    Packet<NDT, GEOM> p;
    Rng g;
    const bool has_mrw = !PLAIN && !LEAN && P.mrw, has_reabs = !PLAIN && P.any_intersect;
    const int mono = (PLAIN || LEAN) ? 0 : P.mono_which;
    Counters cnt;
    cnt.energy_current = 0.0; cnt.crossings = 0; cnt.killed_geo = 0; cnt.killed_int = 0; cnt.interactions = 0;
    Dispenser dsp; dsp.next = 0; dsp.end = 0;
    PeelFlags f; f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = 0;
    int st = ST_NEED_EMIT;
    bool pool_empty = false;
    rng_init(g, P.seed_key, L.iter_tag, 0);
    p.inter = 1; p.tau_req = 0.0; p.tau_ach = 0.0;
    p.t_src = HYP_INF; p.t_ach = 0.0; p.reabs_id = -1; p.reabs = 0;
    long long mrw_k = 1;       // MRW steps of the current interaction (state ST_MRW)
    double e_init = 0.0;       // monochromatic: energy at emission (the packet dies below mono_threshold of it)

    for (;;) {
        // packets that left the grid alive go into the binned images (iter_final.f90:127-129), then their lane is free
        if (__ballot(st == ST_ESCAPED)) {
            if (!PLAIN && !LEAN && P.binned >= 0) bin_escaped<NDT, GEOM>(P, p, st == ST_ESCAPED, f);
            if (st == ST_ESCAPED) st = ST_NEED_EMIT;
        }
        unsigned long long m_walk = __ballot(st == ST_WALK);
        unsigned long long m_int = __ballot(st == ST_NEED_INTERACT);
        unsigned long long m_emit = __ballot(st == ST_NEED_EMIT);
        unsigned long long m_re = has_reabs ? __ballot(st == ST_NEED_REEMIT) : 0ull;
        const unsigned long long m_mrw = has_mrw ? __ballot(st == ST_MRW) : 0ull;
        if (!(m_walk | m_int | m_emit | m_re | m_mrw)) break;

        // peel: 0 none, 1 after emission, 2 after interaction, 3 after re-emission by a source
        int peel = 0;
        Angle a_prev = p.a;
        double s_prev[4] = {p.s[0], p.s[1], p.s[2], p.s[3]};
        int last = LAST_SR; bool last_iso = true;

        // ---- packets re-absorbed by a source are re-emitted from it: iter_final.f90:213-243 ----
        if (!PLAIN && m_re && (__popcll(m_re) >= L.emit_threshold || !m_walk)) {
            if (st == ST_NEED_REEMIT) {
                if ((long long)p.reabs == P.n_reabs_max) { cnt.killed_int++; st = ST_NEED_EMIT; }
                else {
                    const int inter = p.inter, reabs = p.reabs + 1, rid = p.reabs_id;
                    const double e = p.energy;
                    int source_id = 0; Angle src_normal;
                    bool ok = emit_packet<NDT, GEOM>(P, W, p, g, cnt, source_id, src_normal, rid, e);
                    p.inter = inter; p.reabs = reabs;
                    f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
                    if (!ok) st = ST_NEED_EMIT;
                    else { peel = 3; last = LAST_SR; st = ST_PLACED; last_iso = false; a_prev = src_normal; }
                }
            }
            m_walk = __ballot(st == ST_WALK);
            m_int = __ballot(st == ST_NEED_INTERACT);
            m_emit = __ballot(st == ST_NEED_EMIT);
        }

        // ---- modified random walk, one step per pass, each peeled off as isotropic emission:
        //      iter_final.f90:165-183 ----
        if (!PLAIN && !LEAN && m_mrw) {
            if (st == ST_MRW) {
                if (mrw_k == P.n_inter_mrw_max + 1) { cnt.killed_int++; st = ST_NEED_EMIT; }
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
                p.reabs = 0;
                if ((long long)p.inter == P.n_inter_max + 1) {
                    cnt.killed_int++; st = ST_NEED_EMIT;
                } else {
                    a_prev = p.a; s_prev[0] = p.s[0]; s_prev[1] = p.s[1]; s_prev[2] = p.s[2]; s_prev[3] = p.s[3];
                    int scattered, dust_id;
                    // monochromatic: always scatter, the energy decreases by the albedo (iter_final_mono.f90:330-336)
                    bool ok = interact<NDT, GEOM>(P, p, g, cnt, scattered, dust_id, mono != 0);
                    f.dust_id = dust_id;
                    if (scattered) { f.scattered = 1; f.n_scat++; last = LAST_DS; last_iso = false; }
                    else { f.scattered = 0; f.reprocessed = 1; last = LAST_DE; last_iso = true; }
                    bool killed = !ok || (P.kill_on_scatter && scattered) || (P.kill_on_absorb && !scattered && !mono);
                    if (mono && p.energy < e_init * P.mono_threshold) killed = true;
                    if (killed) st = ST_NEED_EMIT;
                    else { p.inter++; peel = 2; }
                }
            }
            m_walk = __ballot(st == ST_WALK);
            m_emit = __ballot(st == ST_NEED_EMIT);
        }

        if (m_emit && !pool_empty && (__popcll(m_emit) >= L.emit_threshold || !m_walk)) {
            unsigned long long id = 0;
            bool got = take_id(P, L, dsp, st == ST_NEED_EMIT, id);
            if (st == ST_NEED_EMIT) {
                if (!got) st = ST_DONE;
                else {
                    rng_init(g, P.seed_key, L.iter_tag, id);
                    int source_id = 0;
                    Angle src_normal;
                    if (mono == 2) {
                        // thermal packets of the monochromatic iteration: iter_final_mono.f90:176-196
                        int dust_id = 0;
                        bool ok = emit_mono_dust<NDT, GEOM>(P, W, p, g, cnt, dust_id);
                        f.scattered = 0; f.reprocessed = 1; f.n_scat = 0; f.dust_id = dust_id; f.source_id = 0;
                        if (!ok) st = ST_NEED_EMIT;
                        else { peel = 1; last = LAST_DE; last_iso = true; st = ST_PLACED; p.reabs = 0; e_init = p.energy; }
                    } else {
                    bool ok = emit_packet<NDT, GEOM, PLAIN>(P, W, p, g, cnt, source_id, src_normal);
                    f.scattered = 0; f.reprocessed = 0; f.n_scat = 0; f.dust_id = 0; f.source_id = source_id;
                    if (!ok) st = ST_NEED_EMIT;
                    else {
                        if (mono) { p.energy = p.energy / P.mono_n_total; e_init = p.energy; }     // iter_final_mono.f90:113-116
                        peel = 1; last = LAST_SR; st = ST_PLACED;   // placed, awaiting tau
                        p.reabs = 0;
                        last_iso = PLAIN || P.sources[source_id].type == 1 || P.sources[source_id].type == 8 || P.sources[source_id].type == 4;
                        // external sources: a_prev carries the inward normal for emit_peeloff
                        if (!last_iso) a_prev = src_normal;
                    }
                    }
                }
            }
            if (__ballot(st == ST_DONE)) pool_empty = true;
            if (*((volatile int *)P.err) != 0) { if (st == ST_NEED_EMIT) st = ST_DONE; pool_empty = true; }
        } else if (m_emit && pool_empty) {
            if (st == ST_NEED_EMIT) st = ST_DONE;
        }

        // ---- peel-off + optical depth sampling for lanes that just emitted / interacted ----
        if (__ballot(peel != 0)) {
            // with raytracing on only scattered packets are peeled here (iter_final.f90:120,268); direct
            // and thermal emission come from the raytracing iteration
            // (a re-emission by a source is peeled in any case: "a kind of scattering", :226-227)
            const bool do_peel = peel != 0 && (!P.peel_scattered_only || (peel == 2 && last == LAST_DS) || peel == 3);   // peel 4 (MRW): :171-173
            if (P.n_peeled > 0 && __ballot(do_peel)) {
                peeloff<NDT, GEOM, PLAIN, LEAN>(P, W, p, do_peel, a_prev, s_prev, last, last_iso, f, g, cnt, &ic);
                if (do_peel) p.peel_seq++;
            }
            if (peel != 0) {
                if (peel == 1) {
                    // first propagation after emission: iter_final.f90:191-209
                    if (geo_escaped(P, p.cell)) st = ST_ESCAPED;
                    else {
                        bool sampled = false;
                        if (P.forced_first) {
                            bool killed = false;
                            double tau_escape = escape_tau<NDT, GEOM>(P, W, p.r, p.v, p.cell, p.chi, g, cnt, killed);
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
                    }
                } else if (peel == 3 && geo_escaped(P, p.cell)) {
                    st = ST_ESCAPED;
                } else if (peel == 4) {
                    // stays in ST_MRW: the next pass decides on another step
                } else if (peel == 2 && has_mrw && !mono) {
                    st = ST_MRW; mrw_k = 1;
                } else {
                    p.tau_req = rng_exp(g); p.tau_ach = 0.0;
                    begin_integrate(P, p);
                    st = (p.tau_req == 0.0) ? ST_NEED_INTERACT : ST_WALK;
                }
            }
        }

#pragma unroll 1
        for (int k = 0; k < final_walk_steps<GEOM>(); k++) {
            if (st == ST_WALK) st = walk_step<NDT, GEOM, false>(P, W, p, g, nullptr, cnt);
        }
    }

    img_cache_flush(ic);
    double e = wave_sum(cnt.energy_current);
    double c = wave_sum((double)cnt.crossings);
    double kg = wave_sum((double)cnt.killed_geo);
    double ki = wave_sum((double)cnt.killed_int);
    double ni = wave_sum((double)cnt.interactions);
    if (__lane_id() == 0) {
        unsafeAtomicAdd(&P.tail[TAIL_ENERGY], e);
        unsafeAtomicAdd(&P.tail[TAIL_CROSSINGS], c);
        if (kg != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_GEO], kg);
        if (ki != 0.0) unsafeAtomicAdd(&P.tail[TAIL_KILLED_INT], ki);
        unsafeAtomicAdd(&P.tail[TAIL_INTERACTIONS], ni);
    }
}

#ifndef HYP_GEOM_TU   // the geometry-independent kernels: static, each host unit (hyp_engine.h) compiles the ones it launches
// image_scale: image_type.f90:136-151 -- x *= scale over [0,n), x *= scale^2 over [n,2n)
static __global__ void image_scale_kernel(double *__restrict__ a, size_t n, double scale)
{
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < 2 * n; k += step)
        a[k] *= (k < n) ? scale : scale * scale;
}

// ---------------------------------------------------------------------------
// Elementwise kernels of the iteration epilogue
// ---------------------------------------------------------------------------

// sum[0][k] += sum[c][k], c = 1..n_copies-1  (before the collective)
static __global__ void reduce_copies_kernel(double *__restrict__ sum, size_t n, size_t stride, int n_copies)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        double s = sum[i];
        for (int c = 1; c < n_copies; c++) s += sum[i + (size_t)c * stride];
        sum[i] = s;
    }
}

// setup_monochromatic_grid_pdfs (grid_monochromatic.f90:51-117), part 1: w[d][ic] = emission probability at the
// frequency x energy emitted in the cell x n_cells / energy_abs_tot(d).  Masked cells carry no density.
static __global__ void mono_weight_kernel(const DProblem *__restrict__ Pp, double *__restrict__ w)
{
    const DProblem &P = *Pp;
    const size_t nc = (size_t)P.n_cells, n = nc * (size_t)P.n_dust;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        const size_t ic = k / P.n_dust; const int d = (int)(k - ic * P.n_dust);
        double energy = 0.0;
        const double eat = P.energy_abs_tot[d];
        if (eat > 0.0) energy = P.specific_energy[k] * P.density[k] * cell_volume(P, ic) * (double)nc / eat;
        const double prob = dust_emit_probability(P, P.dust[d], P.jnu_id[k], P.jnu_frac[k]);
        w[(size_t)d * nc + ic] = prob * energy;
    }
}

// part 2: in-place cumulative sum over the cells of dust type blockIdx.x, normalised to 1 (set_pdf of a discrete
// pdf); mean[d] = total / n_cells.  One 1024-thread block per dust type: each thread owns a contiguous chunk.
static __global__ __launch_bounds__(1024) void mono_scan_kernel(double *__restrict__ w, size_t nc, double *__restrict__ mean)
{
    __shared__ double part[1024];
    double *a = w + (size_t)blockIdx.x * nc;
    const size_t chunk = (nc + 1023) / 1024;
    const size_t lo = (size_t)threadIdx.x * chunk, hi = lo + chunk < nc ? lo + chunk : nc;
    double s = 0.0;
    for (size_t i = lo; i < hi; i++) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int t = 0; t < 1024; t++) { const double v = part[t]; part[t] = run; run += v; }
        mean[blockIdx.x] = run / (double)nc;
        mean[HYP_MAXD + blockIdx.x] = run;
    }
    __syncthreads();
    const double total = mean[HYP_MAXD + blockIdx.x];
    double run = part[threadIdx.x];
    for (size_t i = lo; i < hi; i++) { run += a[i]; a[i] = total > 0.0 ? run / total : 0.0; }
}

struct FinishParams {
    double scale;           // energy_total / energy_current
    int enforce_energy_range, additional, write_out, pad;
};

// dust_jnu_var_pos_frac: dust_type_4elem.f90:295-320
__device__ __forceinline__ void jnu_var_pos_frac(const DDust &D, double e, int &id, double &frac)
{
    int n = D.n_jnu;
    if (e < D.jnu_var[0]) { id = 0; frac = 0.0; }
    else if (e > D.jnu_var[n - 1]) { id = n - 2; frac = 1.0; }
    else {
        int j = locate_g(D.jnu_var, n, e);
        id = j;
        frac = (log10(e) - D.log10_jnu_var[j]) / (D.log10_jnu_var[j + 1] - D.log10_jnu_var[j]);
    }
}

__device__ __forceinline__ double clamp_energy(const DDust &D, double e, int enforce)
{
    if (e < D.minimum_specific_energy) e = D.minimum_specific_energy;
    if (enforce && D.have_e_range) {
        if (e < D.e_min) e = D.e_min;
        if (e > D.e_max) e = D.e_max;
    }
    return e;
}

__device__ __forceinline__ double chi_rosseland(const DDust &D, double e)
{
    int j = locate_g(D.mo_e, D.n_e, e);
    if (j < 0) return __builtin_nan("");
    double y1 = D.mo_chi_ross[j], y2 = D.mo_chi_ross[j + 1];
    if (y1 > 0.0 && y2 > 0.0) {
        double f = (log10(e) - log10(D.mo_e[j])) / (log10(D.mo_e[j + 1]) - log10(D.mo_e[j]));
        return exp10(log10(y1) + f * (log10(y2) - log10(y1)));
    }
    return y1 + (e - D.mo_e[j]) / (D.mo_e[j + 1] - D.mo_e[j]) * (y2 - y1);
}

// interp1d_loglog on a mean-opacity table (dust.f90:81-100)
__device__ __forceinline__ double mean_opacity(const DDust &D, const double *__restrict__ y, double e)
{
    int j = locate_g(D.mo_e, D.n_e, e);
    if (j < 0) return __builtin_nan("");
    double y1 = y[j], y2 = y[j + 1];
    if (y1 > 0.0 && y2 > 0.0) {
        double f = (log10(e) - log10(D.mo_e[j])) / (log10(D.mo_e[j + 1]) - log10(D.mo_e[j]));
        return exp10(log10(y1) + f * (log10(y2) - log10(y1)));
    }
    return y1 + (e - D.mo_e[j]) / (D.mo_e[j + 1] - D.mo_e[j]) * (y2 - y1);
}

// prepare_mrw (grid_mrw_3d.f90:29-53) + update_alpha_inv_planck (grid_physics_3d.f90:397-418),
// plus kappa_planck(specific_energy) of every (cell, dust) for the deposits of grid_do_mrw.
// One thread per cell.
static __global__ void mrw_prepare_kernel(const DProblem *__restrict__ Pp, const double *__restrict__ specific_energy,
                                   const double *__restrict__ density, double *__restrict__ alpha, double *__restrict__ diff,
                                   double *__restrict__ kp)
{
    const DProblem &P = *Pp;
    const int nd = P.n_dust;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t ic = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ic < (size_t)P.n_cells; ic += step) {
        double a = 0.0, tot = 0.0;
        for (int d = 0; d < nd; d++) {
            const DDust &D = P.dust[d];
            const size_t k = ic * nd + d;
            const double e = specific_energy[k], rho = density[k];
            const double c = mean_opacity(D, D.mo_chi_inv_planck, e);
            if (rho > 0.0) a += rho * c;
            tot += rho * c;
            kp[k] = mean_opacity(D, D.mo_kappa_planck, e);
        }
        alpha[ic] = a;
        diff[ic] = 1.0 / 3.0 / tot;
    }
}

// update_energy_abs + check_energy_abs + sublimate_dust + update_energy_abs_tot
// + precompute_jnu_var fused (grid_physics_3d.f90:420-629).  One thread per
// (cell, dust) element of the cell-major arrays.  mode 0: full update from the
// accumulators; mode 1: only clamp + jnu_var + totals (used at create time);
// mode 2: update_energy_abs only (no sublimation: solve_pda comes in between, iter_lucy.f90:224-235);
// mode 3: sublimate_dust from the current specific energy.  `spec` = the frequency-resolved specific
// energy [n_bins][n_cells][n_dust], rescaled or reset where dust sublimates (:441-447,463-464,479-480).
static __global__ void finish_kernel(const DProblem *__restrict__ Pp, FinishParams F, int mode,
                              double *__restrict__ specific_energy, double *__restrict__ density,
                              const double *__restrict__ additional, int *__restrict__ jnu_id,
                              double *__restrict__ jnu_frac, double *__restrict__ energy_abs_tot,
                              double *__restrict__ out_ref_layout, double *__restrict__ spec, int n_bins)
{
    const bool from_sum = mode == 0 || mode == 2, sublimate = mode == 0 || mode == 3;
    const DProblem &P = *Pp;
    const int nd = P.n_dust;
    const size_t n = (size_t)P.n_cells * nd;
    double local_tot[HYP_MAXD];
#pragma unroll
    for (int d = 0; d < HYP_MAXD; d++) local_tot[d] = 0.0;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        size_t ic = k / nd;
        int d = (int)(k - ic * nd);
        const double vol = cell_volume(P, ic);
        const DDust &D = P.dust[d];
        double e;
        if (from_sum) {
            e = P.sum[k] * F.scale / vol;
            if (vol == 0.0) e = 0.0;
            if (F.additional) e += additional[k];
        } else {
            e = specific_energy[k];
        }
        e = clamp_energy(D, e, F.enforce_energy_range);
        double rho = density[k];
        if (sublimate && D.sublimation_mode != 0) {
            double es = D.sublimation_specific_energy;
            if (e > es) {
                if (n_bins) {
                    for (int b = 0; b < n_bins; b++) {
                        double &q = spec[(size_t)b * n + k];
                        q = D.sublimation_mode == 1 ? D.minimum_specific_energy : q * (es / e);
                    }
                }
                if (D.sublimation_mode == 1) { rho = 0.0; e = D.minimum_specific_energy; }
                else if (D.sublimation_mode == 2) {
                    double r = chi_rosseland(D, e) / chi_rosseland(D, es);
                    rho = rho * es / e * r * r; e = es;
                } else e = es;
                density[k] = rho;
            }
            e = clamp_energy(D, e, F.enforce_energy_range);
        }
        specific_energy[k] = e;
        if (out_ref_layout) out_ref_layout[(size_t)d * P.n_cells + ic] = e;
        int id; double fr;
        jnu_var_pos_frac(D, e, id, fr);
        jnu_id[k] = id; jnu_frac[k] = fr;
#pragma unroll
        for (int dd = 0; dd < HYP_MAXD; dd++) if (dd == d) local_tot[dd] += e * rho * vol;
    }
    for (int d = 0; d < nd; d++) {
        double s = wave_sum(local_tot[d]);
        if (__lane_id() == 0 && s != 0.0) unsafeAtomicAdd(&energy_abs_tot[d], s);
    }
}

// [n_dust][n_cells] (reference layout) <-> [n_cells][n_dust] (device layout)
static __global__ void to_cell_major_kernel(const double *__restrict__ in, double *__restrict__ out, size_t n_cells, int nd)
{
    size_t n = n_cells * nd, step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        size_t ic = k / nd; int d = (int)(k - ic * nd);
        out[k] = in[(size_t)d * n_cells + ic];
    }
}

static __global__ void to_ref_layout_kernel(const double *__restrict__ in, double *__restrict__ out, size_t n_cells, int nd)
{
    size_t n = n_cells * nd, step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        size_t ic = k / nd; int d = (int)(k - ic * nd);
        out[(size_t)d * n_cells + ic] = in[k];
    }
}
#endif  // HYP_GEOM_TU
