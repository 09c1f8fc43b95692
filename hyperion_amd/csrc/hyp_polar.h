// hyp_polar.h -- spherical and cylindrical polar grids on the device
// (src/grid/grid_geometry_spherical_3d.f90, src/grid/grid_geometry_cylindrical_3d.f90).
//
// Cells are (i1, i2, i3) = (r, theta, phi) or (w, z, phi), 0-based, cell id as on Cartesian grids.
// The walls are spheres / cylinders (a reduced quadratic each), cones (a full quadratic, or a plane
// for the mid-plane wall) and phi half-planes; candidates go through the reference's epsilon-merged
// insert_t.  Operation order follows the reference formulation (the library is built with
// -ffp-contract=off), so the walk agrees with the CPU oracle to the last bit wherever the device
// and host libm agree (atan2 / sqrt).  The tables (wr2 = w1^2, tan(theta), tan(phi), cos(theta))
// are built on the host like setup_grid_geometry does and stay L1/L2 resident: these grids have a
// few hundred walls per axis.
#pragma once

// equal_nulp: spherical_3d.f90:49-59
__device__ __forceinline__ bool equal_nulp(double x, double y, int n)
{
    if (x == y) return true;
    return fabs(x - y) <= n * spacing_d(x > y ? x : y);
}

__device__ __forceinline__ double polar_theta(const double r[3], const double v[3], double r_sq)
{
    if (r_sq == 0.0) return atan2(sqrt(v[0] * v[0] + v[1] * v[1]), v[2]);
    return atan2(sqrt(r[0] * r[0] + r[1] * r[1]), r[2]);
}

__device__ __forceinline__ double polar_phi(const double r[3], const double v[3], double w_sq)
{
    double phi = w_sq == 0.0 ? atan2(v[1], v[0]) : atan2(r[1], r[0]);
    if (phi < 0.0) phi = phi + HYP_TWOPI;
    return phi;
}

template <int GEOM>
__device__ __forceinline__ bool polar_escaped(const DProblem &P, const Cell<GEOM> &c)
{
    if (GEOM == GEOM_SPH) return c.ic[0] < 0 || c.ic[0] >= P.n1;                                // spherical_3d.f90:483-490
    return c.ic[0] < 0 || c.ic[0] >= P.n1 || c.ic[1] < 0 || c.ic[1] >= P.n2;                   // cylindrical_3d.f90:381-390
}
__device__ __forceinline__ bool geo_escaped(const DProblem &P, const Cell<GEOM_SPH> &c) { return polar_escaped<GEOM_SPH>(P, c); }
__device__ __forceinline__ bool geo_escaped(const DProblem &P, const Cell<GEOM_CYL> &c) { return polar_escaped<GEOM_CYL>(P, c); }
__device__ __forceinline__ size_t geo_index(const DProblem &P, const Cell<GEOM_SPH> &c) { return ((size_t)c.ic[2] * P.n2 + c.ic[1]) * P.n1 + c.ic[0]; }
__device__ __forceinline__ size_t geo_index(const DProblem &P, const Cell<GEOM_CYL> &c) { return ((size_t)c.ic[2] * P.n2 + c.ic[1]) * P.n1 + c.ic[0]; }

// find_cell: spherical :226-299, cylindrical :183-236
template <int GEOM>
__device__ __forceinline__ bool polar_find_cell(const DProblem &P, const double r[3], const double v[3], int ic[3])
{
    const double w_sq = r[0] * r[0] + r[1] * r[1];
    const double phi = polar_phi(r, v, w_sq);
    int i1, i2;
    if (GEOM == GEOM_SPH) {
        const double r_sq = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        i1 = locate(P.wr2, P.n1 + 1, r_sq);
        i2 = locate(P.w[1], P.n2 + 1, polar_theta(r, v, r_sq));
    } else {
        i1 = locate(P.wr2, P.n1 + 1, w_sq);
        i2 = locate(P.w[1], P.n2 + 1, r[2]);
    }
    const int i3 = locate(P.w[2], P.n3 + 1, phi);
    if (i1 < 0 || i1 >= P.n1 || i2 < 0 || i2 >= P.n2 || i3 < 0 || i3 >= P.n3) return false;
    ic[0] = i1; ic[1] = i2; ic[2] = i3;
    return true;
}

// azimuthal part of adjust_wall: spherical :425-462 = cylindrical :312-349
template <int GEOM>
__device__ __forceinline__ void polar_adjust_phi(const DProblem &P, const double r[3], const double v[3], double phi, Cell<GEOM> &c)
{
    const double *w3 = P.w[2];
    if (r[0] == 0.0 && r[1] == 0.0 && v[0] == 0.0 && v[1] == 0.0) return;
    if (equal_nulp(phi, w3[c.ic[2]], 3)) {
        double dphi = atan2(v[1], v[0]) - w3[c.ic[2]];
        if (dphi < -HYP_PI) dphi = dphi + HYP_TWOPI;
        if (dphi > 0.0) c.ow[2] = -1;
        else { c.ow[2] = +1; c.ic[2]--; if (c.ic[2] == -1) c.ic[2] = P.n3 - 1; }
    } else if (equal_nulp(phi, w3[c.ic[2] + 1], 3)) {
        double dphi = atan2(v[1], v[0]) - w3[c.ic[2] + 1];
        if (dphi < -HYP_PI) dphi = dphi + HYP_TWOPI;
        if (dphi > 0.0) { c.ow[2] = -1; c.ic[2]++; if (c.ic[2] == P.n3) c.ic[2] = 0; }
        else c.ow[2] = +1;
    }
}

// place_in_cell + adjust_wall: spherical :301-489, cylindrical :238-376
template <int GEOM>
__device__ __forceinline__ bool polar_place(const DProblem &P, const double r[3], const double v[3], Cell<GEOM> &c)
{
    if (!polar_find_cell<GEOM>(P, r, v, c.ic)) return false;
    c.ow[0] = c.ow[1] = c.ow[2] = 0;
    const double w_sq = r[0] * r[0] + r[1] * r[1];
    const double phi = polar_phi(r, v, w_sq);
    const double *w2 = P.w[1];
    if (GEOM == GEOM_CYL) {
        if (r[0] * v[0] + r[1] * v[1] >= 0.0) {
            if (equal_nulp(w_sq, P.wr2[c.ic[0]], 3)) c.ow[0] = -1;
            else if (equal_nulp(w_sq, P.wr2[c.ic[0] + 1], 3)) { c.ow[0] = -1; c.ic[0]++; }
        } else {
            if (equal_nulp(w_sq, P.wr2[c.ic[0]], 3)) { c.ow[0] = +1; c.ic[0]--; }
            else if (equal_nulp(w_sq, P.wr2[c.ic[0] + 1], 3)) c.ow[0] = +1;
        }
        if (v[2] > 0.0) {
            if (equal_nulp(r[2], w2[c.ic[1]], 3)) c.ow[1] = -1;
            else if (equal_nulp(r[2], w2[c.ic[1] + 1], 3)) { c.ow[1] = -1; c.ic[1]++; }
        } else if (v[2] < 0.0) {
            if (equal_nulp(r[2], w2[c.ic[1]], 3)) { c.ow[1] = +1; c.ic[1]--; }
            else if (equal_nulp(r[2], w2[c.ic[1] + 1], 3)) c.ow[1] = +1;
        }
        polar_adjust_phi<GEOM>(P, r, v, phi, c);
        return true;
    }
    const double r_sq = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
    const double theta = polar_theta(r, v, r_sq);
    if ((r[0] * v[0] + r[1] * v[1]) + r[2] * v[2] >= 0.0) {
        if (equal_nulp(r_sq, P.wr2[c.ic[0]], 3)) c.ow[0] = -1;
        else if (equal_nulp(r_sq, P.wr2[c.ic[0] + 1], 3)) { c.ow[0] = -1; c.ic[0]++; }
    } else {
        if (equal_nulp(r_sq, P.wr2[c.ic[0]], 3)) { c.ow[0] = +1; c.ic[0]--; }
        else if (equal_nulp(r_sq, P.wr2[c.ic[0] + 1], 3)) c.ow[0] = +1;
    }
    if (r_sq == 0.0) {
        if (fabs(v[2]) < 1.0) {
            const double theta_v = atan2(sqrt(v[0] * v[0] + v[1] * v[1]), v[2]);
            if (equal_nulp(theta_v, w2[c.ic[1]], 3)) c.ow[1] = -1;
            else if (equal_nulp(theta_v, w2[c.ic[1] + 1], 3)) c.ow[1] = +1;
        }
    } else if (c.ic[1] > 0 && equal_nulp(theta, w2[c.ic[1]], 3)) {
        if (c.ic[1] == P.midplane) {
            if (v[2] > 0.0) { c.ow[1] = +1; c.ic[1]--; }
            else c.ow[1] = -1;
        } else {
            const bool lhs = sqrt(w_sq) * v[2] * P.wtant[c.ic[1]] - (r[0] * v[0] + r[1] * v[1]) < 0.0;
            if (lhs == (r[2] > 0.0)) c.ow[1] = -1;
            else { c.ow[1] = +1; c.ic[1]--; }
        }
    } else if (c.ic[1] + 1 < P.n2 && equal_nulp(theta, w2[c.ic[1] + 1], 3)) {
        if (c.ic[1] + 1 == P.midplane) {
            if (v[2] > 0.0) c.ow[1] = +1;
            else { c.ow[1] = -1; c.ic[1]++; }
        } else {
            const bool lhs = sqrt(w_sq) * v[2] * P.wtant[c.ic[1] + 1] - (r[0] * v[0] + r[1] * v[1]) < 0.0;
            if (lhs == (r[2] > 0.0)) { c.ow[1] = -1; c.ic[1]++; }
            else c.ow[1] = +1;
        }
    }
    polar_adjust_phi<GEOM>(P, r, v, phi, c);
    return true;
}
__device__ __forceinline__ bool geo_place(const DProblem &P, const Walls &W, const double r[3], const double v[3], Cell<GEOM_SPH> &c) { return polar_place<GEOM_SPH>(P, r, v, c); }
__device__ __forceinline__ bool geo_place(const DProblem &P, const Walls &W, const double r[3], const double v[3], Cell<GEOM_CYL> &c) { return polar_place<GEOM_CYL>(P, r, v, c); }

// in_correct_cell: spherical :553-637, cylindrical :444-516
template <int GEOM>
__device__ __forceinline__ bool polar_in_correct_cell(const DProblem &P, const double r[3], const double v[3], const Cell<GEOM> &c)
{
    int act[3] = {-1, -1, -1};
    if (!polar_find_cell<GEOM>(P, r, v, act)) act[0] = act[1] = act[2] = -1;
    const double thr = 1e-3;
    if (!(c.ow[0] | c.ow[1] | c.ow[2])) return act[0] == c.ic[0] && act[1] == c.ic[1] && act[2] == c.ic[2];
    bool ok = true;
    const double w_sq = r[0] * r[0] + r[1] * r[1];
    double rad_sq = w_sq;
    const double *w1 = P.w[0], *w2 = P.w[1], *w3 = P.w[2];
    if (GEOM == GEOM_SPH) {
        rad_sq = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        if (rad_sq == 0.0) return true;
    }
    const double phi = polar_phi(r, v, w_sq);
    if (c.ow[0] == -1) {
        if (w1[c.ic[0]] != sqrt(rad_sq)) ok = ok && fabs(sqrt(rad_sq) / w1[c.ic[0]] - 1.0) < thr;
    } else if (c.ow[0] == +1) {
        if (w1[c.ic[0] + 1] != sqrt(rad_sq)) ok = ok && fabs(sqrt(rad_sq) / w1[c.ic[0] + 1] - 1.0) < thr;
    } else ok = ok && act[0] == c.ic[0];
    if (GEOM == GEOM_SPH) {
        const double theta = polar_theta(r, v, rad_sq);
        if (c.ow[1] == -1) ok = ok && fabs(theta / w2[c.ic[1]] - 1.0) < thr;
        else if (c.ow[1] == +1) ok = ok && fabs(theta / w2[c.ic[1] + 1] - 1.0) < thr;
        else ok = ok && act[1] == c.ic[1];
    } else {
        const double dz = w2[c.ic[1] + 1] - w2[c.ic[1]];
        if (c.ow[1] == -1) ok = ok && fabs((r[2] - w2[c.ic[1]]) / dz) < thr;
        else if (c.ow[1] == +1) ok = ok && fabs((r[2] - w2[c.ic[1] + 1]) / dz) < thr;
        else ok = ok && act[1] == c.ic[1];
    }
    if (c.ow[2] != 0) {
        double dphi = phi - w3[c.ic[2] + (c.ow[2] == +1 ? 1 : 0)];
        if (dphi > HYP_PI) dphi = dphi - HYP_TWOPI;
        if (dphi < -HYP_PI) dphi = dphi + HYP_TWOPI;
        ok = ok && fabs(dphi / (w3[c.ic[2] + 1] - w3[c.ic[2]])) < thr;
    } else ok = ok && act[2] == c.ic[2];
    return ok;
}

// ---- find_wall ------------------------------------------------------------------------------
struct WallSel { double tmin, emin; int imin[3], iext[3]; };

// insert_t: spherical_3d.f90:1080-1112
__device__ __forceinline__ void insert_t(WallSel &ws, double t, int iw, int i, double e)
{
    if (t > 0.0) {
        const double emax = max_no_nan(e, ws.emin);      // (neither is a NaN: one v_max_f64 instead of fmax's three)
        if (t < ws.tmin - emax) {
            ws.tmin = t; ws.emin = emax;
            ws.imin[0] = iw == 0 ? i : 0; ws.imin[1] = iw == 1 ? i : 0; ws.imin[2] = iw == 2 ? i : 0;
        } else if (t < ws.tmin + emax) {
            ws.emin = emax;
            if (iw == 0) ws.imin[0] = i; else if (iw == 1) ws.imin[1] = i; else ws.imin[2] = i;
        }
    }
}

// both roots of a curved wall unless the packet sits on it: then the one that is not the wall itself
__device__ __forceinline__ void insert_pair(WallSel &ws, double t1, double t2, bool on_it, int iw, int i, double e)
{
    if (on_it) insert_t(ws, fabs(t1) < fabs(t2) ? t2 : t1, iw, i, e);
    else { insert_t(ws, t1, iw, i, e); insert_t(ws, t2, iw, i, e); }
}

// fortranlib quadratic_pascal_reduced / quadratic: cancellation-free real roots, -huge if none
__device__ __forceinline__ void quad_reduced(double b, double c, double &t1, double &t2)
{
    const double delta = b * b - 4.0 * c;
    if (delta < 0.0) { t1 = t2 = -HYP_DBL_MAX; return; }
    const double q = b >= 0.0 ? -0.5 * (b + sqrt(delta)) : -0.5 * (b - sqrt(delta));
    t1 = q; t2 = q != 0.0 ? c / q : 0.0;
}
__device__ __forceinline__ void quad_full(double a, double b, double c, double &x1, double &x2)
{
    const double delta = b * b - 4.0 * a * c;
    if (delta < 0.0) { x1 = x2 = -HYP_DBL_MAX; return; }
    const double q = b >= 0.0 ? -0.5 * (b + sqrt(delta)) : -0.5 * (b - sqrt(delta));
    x1 = q / a; x2 = q != 0.0 ? c / q : 0.0;
}

// phi half-planes: spherical_3d.f90:985-1064 = cylindrical_3d.f90:691-767
template <int GEOM>
__device__ __forceinline__ void polar_wall_phi(const DProblem &P, const double r[3], const double v[3], const Cell<GEOM> &c,
                                               double r2_xy, WallSel &ws)
{
    if (P.n_dim != 3) return;
    const double *w3 = P.w[2];
    const int i3 = c.ic[2];
    double dphi = 0.0;
    if (c.ow[2] != 0) {
        dphi = atan2(v[1], v[0]) - w3[i3 + (c.ow[2] == +1 ? 1 : 0)];
        if (dphi > HYP_PI) dphi = dphi - HYP_TWOPI;
        if (dphi < -HYP_PI) dphi = dphi + HYP_TWOPI;
    }
    if (c.ow[2] == +1 && fabs(dphi) < P.ew[2][i3 + 1]) ws.iext[2] = +1;
    else if (c.ow[2] == -1 && fabs(dphi) < P.ew[2][i3]) ws.iext[2] = -1;
    else if (r2_xy > 0.0) {
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const int dir = side ? +1 : -1;
            if (c.ow[2] == dir) continue;
            const double tp = P.wtanp[i3 + side];
            const double t = -(tp * r[0] - r[1]) / (tp * v[0] - v[1]);
            const double x_i = r[0] + v[0] * t, y_i = r[1] + v[1] * t;
            double d = fabs(atan2(y_i, x_i) - w3[i3 + side]);
            if (d > HYP_PI) d = fabs(d - HYP_TWOPI);
            if (d < 0.5 * HYP_PI) insert_t(ws, t, 2, dir, 0.0);
        }
    }
}

// one cone wall (side 0 = lower, 1 = upper): spherical_3d.f90:832-980
__device__ __forceinline__ void sph_wall_cone(const DProblem &P, const double r[3], const double v[3], const Cell<GEOM_SPH> &c, int side,
                                              double v2_xy, double v2_z, double rv_xy, double rv_z, double r2_xy, double r2_z, WallSel &ws,
                                              double e, double tt, double tt2)
{
    const int iw = c.ic[1] + side, dir = side ? +1 : -1;
    const double pA = v2_xy - v2_z * tt2;
    // The packet sits on this wall: does it fly along the cone (:846-853)?  equal_nulp(tt, s, 10) with s = sqrt(v2_xy) / v_z means
    // tt = s (1 + d), |d| < 10.01 * 2^-52; s carries two roundings, tt2 = RN(tt^2) one, v2_z one, the product one: v2_z tt2 = v2_xy (1 + h)
    // with |h| < 24 * 2^-52, so |pA| <= 2^-47 v2_xy.  A pA above 2^-40 v2_xy (v2_xy normal, so that the bounds hold) therefore says "no"
    // without the square root and the division -- which every wave-step of scattered packets paid twice for the lanes that had come in
    // through a cone (round 6).
    const bool maybe_along = !(v2_xy >= 0x1p-1000 && fabs(pA) > 0x1p-40 * v2_xy);
    if (c.ow[1] == dir && maybe_along && equal_nulp(tt, sqrt(v2_xy) / v[2], 10) && equal_nulp(sqrt(r2_xy) * v[2] * tt, rv_xy, 10)) { ws.iext[1] = dir; return; }
    if (iw == P.midplane && v[2] != 0.0) {
        if (c.ow[1] != dir) insert_t(ws, -r[2] / v[2], 1, dir, e);
        return;
    }
    double pB = rv_xy - rv_z * tt2; pB = pB + pB;
    const double pC = r2_xy - r2_z * tt2;
    // Three coefficients of one strict sign: no positive root (Descartes), and the formulas below say so too -- q has the sign of -pB, so
    // q / pA and pC / q are negative (or -huge without real roots), the side test can only turn them into +huge, and insert_t takes
    // neither a negative value nor +huge.  Exact, and it spares the square root and both divisions.
    if ((pA > 0.0 && pB > 0.0 && pC > 0.0) || (pA < 0.0 && pB < 0.0 && pC < 0.0)) return;
    if (fabs(pA) > 0.0) {
        double t1, t2;
        quad_full(pA, pB, pC, t1, t2);
        const double z1 = r[2] + v[2] * t1;
        if ((z1 > 0.0) != (tt > 0.0)) t1 = HYP_DBL_MAX;
        const double z2 = r[2] + v[2] * t2;
        if ((z2 > 0.0) != (tt > 0.0)) t2 = HYP_DBL_MAX;
        insert_pair(ws, t1, t2, c.ow[1] == dir, 1, dir, e);
    } else if (fabs(pB) > 0.0) {
        if (c.ow[1] != dir) insert_t(ws, -pC / pB, 1, dir, e);
    }
}

// A cone wall that cannot matter, told without solving its quadratic (round 4; the two cones are half of the walk kernel's time,
// profiles/r04_tiled_log.md).  When the cones come up in find_wall the spheres have left ws.tmin: a root of the cone changes
// anything only if insert_t sees 0 < t < ws.tmin + max(e, ws.emin) =: T.  The roots quad_full computes are exact roots of a
// quadratic whose coefficients differ from (pA, pB, pC) by a few units of round-off (the formula is backward stable), so there is
// none in [0, T] when the parabola keeps its sign there by more than that: both ends of the same sign, and the smaller of
// |p(0)|, |p(T)| above the bulge |pA| T^2 / 4 of the parabola over its chord plus 64 eps (|pA| T^2 + |pB| T + |pC|).  A packet that
// flies radially -- every packet until its first interaction, 60 % of the steps -- has pA, pB / 2 rho, pC / rho^2 all equal and the
// (meaningless) roots at the apex: p(0) and p(T) agree in sign by five orders of magnitude more than needed.  Not applied to the
// wall the packet sits on (insert_pair and the extension rule look at both roots) nor to the mid-plane (a plane, one division).
// Measured: applied in EVERY find_wall it costs (528 against 493 ms on the 400 x 200 grid: a wave mixes radial and scattered packets, so both
// cones are solved for some lane in almost every wave-step and the test comes on top, profiles/r04_tiled_log.md); the tiled walk applies it to
// the tasks of packets that have not interacted yet (TileGeom::vsplit, hyp_ptile.h).
__device__ __forceinline__ bool sph_cone_out_of_reach(const DProblem &P, const Cell<GEOM_SPH> &c, int side, double v2_xy, double v2_z, double rv_xy, double rv_z,
                                                      double r2_xy, double r2_z, const WallSel &ws)
{
    const int iw = c.ic[1] + side, dir = side ? +1 : -1;
    if (c.ow[1] == dir || iw == P.midplane) return false;
    const double tt2 = P.wtant2[iw];
    const double pA = v2_xy - v2_z * tt2;
    double pB = rv_xy - rv_z * tt2; pB = pB + pB;
    const double pC = r2_xy - r2_z * tt2;
    const double T = ws.tmin + max_no_nan(P.ew[1][iw], ws.emin);
    const double aT2 = fabs(pA) * T * T, pT = (pA * T + pB) * T + pC;
    const double lim = 0.25 * aT2 + 0x1p-46 * (aT2 + fabs(pB) * T + fabs(pC));
    // (a NaN or an infinity anywhere -- no sphere ahead, ws.tmin = huge -- fails the comparisons: the wall is solved)
    return ((pC > 0.0) == (pT > 0.0)) && fabs(pC) > lim && fabs(pT) > lim;
}

// find_wall: spherical_3d.f90:741-1073.  reach: try to dismiss each cone wall with sph_cone_out_of_reach before solving it (the same
// answer either way; worth it where a whole wave's packets fly radially, hyp_ptile.h)
__device__ __forceinline__ bool sph_find_wall(const DProblem &P, const Walls &W, const double r[3], const double v[3],
                                              const Cell<GEOM_SPH> &c, double &tnear, int im[3], bool reach)
{
    WallSel ws; ws.tmin = HYP_DBL_MAX; ws.emin = 0.0;
    ws.imin[0] = ws.imin[1] = ws.imin[2] = 0; ws.iext[0] = ws.iext[1] = ws.iext[2] = 0;
    const double v2_xy = v[0] * v[0] + v[1] * v[1], v2_z = v[2] * v[2];
    const double rv_xy = r[0] * v[0] + r[1] * v[1], rv_z = r[2] * v[2];
    const double r2_xy = r[0] * r[0] + r[1] * r[1], r2_z = r[2] * r[2];
    double pB = rv_xy + rv_z; pB = pB + pB;
    const double pC = r2_xy + r2_z;
    double t1, t2;
    const int i1 = c.ic[0];
    // every table value of the step loaded up front (the cone walls' behind their `if`s came as further dependent batches)
    const int i2 = c.ic[1];
    const double wr2_a = P.wr2[i1], wr2_b = P.wr2[i1 + 1], e0_a = P.ew[0][i1], e0_b = P.ew[0][i1 + 1];
    double e1_a = P.ew[1][i2], e1_b = P.ew[1][i2 + 1], tt_a = P.wtant[i2], tt_b = P.wtant[i2 + 1], tt2_a = P.wtant2[i2], tt2_b = P.wtant2[i2 + 1];
    asm volatile("" : "+v"(e1_a), "+v"(e1_b), "+v"(tt_a), "+v"(tt_b), "+v"(tt2_a), "+v"(tt2_b));
    // (inner sphere; with pB >= 0 and pC - w^2 >= 0 -- outside it and moving away -- both roots are <= 0 (sum -pB, product >= 0), in
    // quad_reduced's arithmetic too: q = -(pB + sqrt) / 2 <= 0 and c / q <= 0; insert_t takes neither)
    if (!c.radial && !(pB >= 0.0 && pC - wr2_a >= 0.0)) {
        quad_reduced(pB, pC - wr2_a, t1, t2);
        insert_pair(ws, t1, t2, c.ow[0] == -1, 0, -1, e0_a);
    }
    quad_reduced(pB, pC - wr2_b, t1, t2);
    insert_pair(ws, t1, t2, c.ow[0] == +1, 0, +1, e0_b);
    if (c.ic[1] > 0 && !(reach && sph_cone_out_of_reach(P, c, 0, v2_xy, v2_z, rv_xy, rv_z, r2_xy, r2_z, ws))) sph_wall_cone(P, r, v, c, 0, v2_xy, v2_z, rv_xy, rv_z, r2_xy, r2_z, ws, e1_a, tt_a, tt2_a);
    if (c.ic[1] < P.n2 - 1 && !(reach && sph_cone_out_of_reach(P, c, 1, v2_xy, v2_z, rv_xy, rv_z, r2_xy, r2_z, ws))) sph_wall_cone(P, r, v, c, 1, v2_xy, v2_z, rv_xy, rv_z, r2_xy, r2_z, ws, e1_b, tt_b, tt2_b);
    polar_wall_phi<GEOM_SPH>(P, r, v, c, r2_xy, ws);
    tnear = ws.tmin;
#pragma unroll
    for (int a = 0; a < 3; a++) im[a] = ws.imin[a] + ws.iext[a];      // find_next_wall :1114-1121
    return (im[0] | im[1] | im[2]) != 0;
}

__device__ __forceinline__ bool geo_find_wall(const DProblem &P, const Walls &W, const double r[3], const double v[3],
                                              const Cell<GEOM_SPH> &c, double &tnear, int im[3])
{
    return sph_find_wall(P, W, r, v, c, tnear, im, false);
}

// find_wall: cylindrical_3d.f90:593-771
__device__ __forceinline__ bool geo_find_wall(const DProblem &P, const Walls &W, const double r[3], const double v[3],
                                              const Cell<GEOM_CYL> &c, double &tnear, int im[3])
{
    WallSel ws; ws.tmin = HYP_DBL_MAX; ws.emin = 0.0;
    ws.imin[0] = ws.imin[1] = ws.imin[2] = 0; ws.iext[0] = ws.iext[1] = ws.iext[2] = 0;
    const double v2_xy = v[0] * v[0] + v[1] * v[1];
    const double rv_xy = r[0] * v[0] + r[1] * v[1];
    const double r2_xy = r[0] * r[0] + r[1] * r[1];
    double pB = rv_xy / v2_xy; pB = pB + pB;
    const double pC = r2_xy / v2_xy;
    double t1, t2;
    const int i1 = c.ic[0], i2 = c.ic[1];
    // (loading these up front behind a compiler fence, as sph_find_wall does, measured 93.7 against 90.9 ms on a 400 x 200 disc: not done)
    const double wr2_a = P.wr2[i1], wr2_b = P.wr2[i1 + 1], e0_a = P.ew[0][i1], e0_b = P.ew[0][i1 + 1], wz_a = P.w[1][i2], wz_b = P.w[1][i2 + 1];
    quad_reduced(pB, pC - wr2_a / v2_xy, t1, t2);
    insert_pair(ws, t1, t2, c.ow[0] == -1, 0, -1, e0_a);
    quad_reduced(pB, pC - wr2_b / v2_xy, t1, t2);
    insert_pair(ws, t1, t2, c.ow[0] == +1, 0, +1, e0_b);
    if (c.ow[1] != -1) insert_t(ws, (wz_a - r[2]) / v[2], 1, -1, 0.0);
    if (c.ow[1] != +1) insert_t(ws, (wz_b - r[2]) / v[2], 1, +1, 0.0);
    polar_wall_phi<GEOM_CYL>(P, r, v, c, r2_xy, ws);
    tnear = ws.tmin;
#pragma unroll
    for (int a = 0; a < 3; a++) im[a] = ws.imin[a] + ws.iext[a];
    return (im[0] | im[1] | im[2]) != 0;
}

// next_cell_wall_id (phi periodic) + opposite_wall: spherical :519-549, cylindrical :413-442
template <int GEOM>
__device__ __forceinline__ void polar_advance(const DProblem &P, Cell<GEOM> &c, const int im[3])
{
#pragma unroll
    for (int a = 0; a < 3; a++) { c.ic[a] += im[a]; c.ow[a] = -im[a]; }
    if (c.ic[2] == -1) c.ic[2] = P.n3 - 1;
    else if (c.ic[2] == P.n3) c.ic[2] = 0;
}
__device__ __forceinline__ void geo_advance(const DProblem &P, const double r[3], Cell<GEOM_SPH> &c, const int im[3]) { polar_advance<GEOM_SPH>(P, c, im); }
// RN(a / b) from y = RN(1 / b) formed once with a true division (Markstein, as in find_wall_ahead of hyp_kernels.h): q0 = RN(a y),
// rem = a - q0 b exactly (one FMA), q = RN(q0 + rem y) is the correctly rounded quotient -- 3 operations instead of the ~14 of an
// IEEE division.  No underflow or overflow may occur on the way: see cyl_find_wall_inv's caller.
__device__ __forceinline__ double quot_inv(double a, double b, double y)
{
    const double q0 = a * y;
    return __builtin_fma(__builtin_fma(-q0, b, a), y, q0);
}

// geo_find_wall<GEOM_CYL> for a direction that stays fixed over many steps (round 5): six of a step's eight divisions have a divisor that
// belongs to the flight -- v_xy^2 four times, v_z twice -- and are formed from its reciprocal, bit for bit the quotients above.  The inner
// cylinder is left unsolved when the packet is outside it and moving away (pB >= 0 and its constant term >= 0: both roots <= 0 in
// quad_reduced's arithmetic too, see sph_find_wall).  The caller guarantees v2_xy = v_x^2 + v_y^2 >= 2^-200, |v_z| >= 2^-200 and an outer radius
// in [2^-250, 2^250], which keeps every non-zero operand (squares and differences of coordinates, multiples of a coordinate's ulp)
// and every intermediate product hundreds of binades away from underflow and overflow.
__device__ __forceinline__ bool cyl_find_wall_inv(const DProblem &P, const double r[3], const double v[3], const Cell<GEOM_CYL> &c,
                                                  double v2_xy, double inv_v2, double inv_vz, double &tnear, int im[3])
{
    WallSel ws; ws.tmin = HYP_DBL_MAX; ws.emin = 0.0;
    ws.imin[0] = ws.imin[1] = ws.imin[2] = 0; ws.iext[0] = ws.iext[1] = ws.iext[2] = 0;
    const double rv_xy = r[0] * v[0] + r[1] * v[1];
    const double r2_xy = r[0] * r[0] + r[1] * r[1];
    double pB = quot_inv(rv_xy, v2_xy, inv_v2); pB = pB + pB;
    const double pC = quot_inv(r2_xy, v2_xy, inv_v2);
    double t1, t2;
    const int i1 = c.ic[0], i2 = c.ic[1];
    const double wr2_a = P.wr2[i1], wr2_b = P.wr2[i1 + 1], e0_a = P.ew[0][i1], e0_b = P.ew[0][i1 + 1], wz_a = P.w[1][i2], wz_b = P.w[1][i2 + 1];
    const double ca = pC - quot_inv(wr2_a, v2_xy, inv_v2);
    if (!(pB >= 0.0 && ca >= 0.0)) {
        quad_reduced(pB, ca, t1, t2);
        insert_pair(ws, t1, t2, c.ow[0] == -1, 0, -1, e0_a);
    }
    quad_reduced(pB, pC - quot_inv(wr2_b, v2_xy, inv_v2), t1, t2);
    insert_pair(ws, t1, t2, c.ow[0] == +1, 0, +1, e0_b);
    if (c.ow[1] != -1) insert_t(ws, quot_inv(wz_a - r[2], v[2], inv_vz), 1, -1, 0.0);
    if (c.ow[1] != +1) insert_t(ws, quot_inv(wz_b - r[2], v[2], inv_vz), 1, +1, 0.0);
    polar_wall_phi<GEOM_CYL>(P, r, v, c, r2_xy, ws);
    tnear = ws.tmin;
#pragma unroll
    for (int a = 0; a < 3; a++) im[a] = ws.imin[a] + ws.iext[a];
    return (im[0] | im[1] | im[2]) != 0;
}

__device__ __forceinline__ void geo_advance(const DProblem &P, const double r[3], Cell<GEOM_CYL> &c, const int im[3]) { polar_advance<GEOM_CYL>(P, c, im); }

// distance_to_closest_wall: spherical_3d.f90:675-739, cylindrical_3d.f90:552-591
template <int GEOM>
__device__ __forceinline__ double polar_closest_wall(const DProblem &P, const double r[3], const Cell<GEOM> &c)
{
    const int i1 = c.ic[0], i2 = c.ic[1], i3 = c.ic[2];
    const double rcyl = sqrt(r[0] * r[0] + r[1] * r[1]);
    double d1, d2, d3, d4, d5 = HYP_DBL_MAX, d6 = HYP_DBL_MAX;
    if (GEOM == GEOM_SPH) {
        const double rad = sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
        d1 = rad - P.w[0][i1]; d2 = P.w[0][i1 + 1] - rad;
        if (fabs(d1) < P.ew[0][i1]) d1 = 0.0;
        if (fabs(d2) < P.ew[0][i1 + 1]) d2 = 0.0;
        d3 = fabs(-rcyl + P.wtant[i2] * r[2]) / sqrt(1 + P.wtant[i2] * P.wtant[i2]);
        d4 = fabs(-rcyl + P.wtant[i2 + 1] * r[2]) / sqrt(1 + P.wtant[i2 + 1] * P.wtant[i2 + 1]);
    } else {
        d1 = rcyl - P.w[0][i1]; d2 = P.w[0][i1 + 1] - rcyl;
        d3 = r[2] - P.w[1][i2]; d4 = P.w[1][i2 + 1] - r[2];
    }
    if (P.n_dim == 3) {
        d5 = fabs(P.wtanp[i3] * r[0] - r[1]) / sqrt(P.wtanp[i3] * P.wtanp[i3] + 1.0);
        d6 = fabs(P.wtanp[i3 + 1] * r[0] - r[1]) / sqrt(P.wtanp[i3 + 1] * P.wtanp[i3 + 1] + 1.0);
    }
    const double d = fmin(fmin(fmin(d1, d2), fmin(d3, d4)), fmin(d5, d6));
    return d < 0.0 ? 0.0 : d;
}
__device__ __forceinline__ double geo_closest_wall(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_SPH> &c) { return polar_closest_wall<GEOM_SPH>(P, r, c); }
__device__ __forceinline__ double geo_closest_wall(const DProblem &P, const Walls &W, const double r[3], const Cell<GEOM_CYL> &c) { return polar_closest_wall<GEOM_CYL>(P, r, c); }

// random_position_cell: spherical_3d.f90:639-673, cylindrical_3d.f90:518-550
template <int GEOM>
__device__ __forceinline__ void polar_random_position(const DProblem &P, size_t ic, double x, double y, double z, double r[3])
{
    const int i1 = (int)(ic % P.n1);
    const size_t t = ic / P.n1;
    const int i2 = (int)(t % P.n2), i3 = (int)(t / P.n2);
    const double *w1 = P.w[0], *w2 = P.w[1], *w3 = P.w[2];
    double rr, tz, ph = z * (w3[i3 + 1] - w3[i3]) + w3[i3];
    if (GEOM == GEOM_SPH) {
        const double a3 = w1[i1] * w1[i1] * w1[i1], b3 = w1[i1 + 1] * w1[i1 + 1] * w1[i1 + 1];
        rr = pow(x * (b3 - a3) + a3, 1.0 / 3.0);
        tz = acos(y * (P.wcost[i2 + 1] - P.wcost[i2]) + P.wcost[i2]);
    } else {
        const double a2 = w1[i1] * w1[i1], b2 = w1[i1 + 1] * w1[i1 + 1];
        rr = sqrt(x * (b2 - a2) + a2);
        tz = y * (w2[i2 + 1] - w2[i2]) + w2[i2];
    }
    if (rr <= w1[i1] || rr >= w1[i1 + 1]) rr = 0.5 * (w1[i1] + w1[i1 + 1]);
    if (tz <= w2[i2] || tz >= w2[i2 + 1]) tz = 0.5 * (w2[i2] + w2[i2 + 1]);
    if (ph <= w3[i3] || ph >= w3[i3 + 1]) ph = 0.5 * (w3[i3] + w3[i3 + 1]);
    double sp, cp;
    sincos(ph, &sp, &cp);
    if (GEOM == GEOM_SPH) {
        double st, ct;
        sincos(tz, &st, &ct);
        r[0] = rr * st * cp; r[1] = rr * st * sp; r[2] = rr * ct;
    } else { r[0] = rr * cp; r[1] = rr * sp; r[2] = tz; }
}
