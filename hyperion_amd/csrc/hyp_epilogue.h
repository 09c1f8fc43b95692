// hyp_epilogue.h -- what follows the packet loop of a Lucy iteration besides finish_kernel (gfx950):
//   * the n_photons counters and the frequency-resolved specific energy (grid_physics_3d.f90:307-395,500-547),
//   * the partial diffusion approximation, solve_pda (src/grid/grid_pda_3d.f90 with the geometrical factors of
//     grid_pda_{cartesian,spherical,cylindrical}_3d.f90),
//   * the quantity tested by specific_energy_converged (grid_physics_3d.f90:637-689).
// All of it is streaming or small; none of it is on the packet path.  Compiled once, in hyp_engine.hip.
#pragma once

#include "hyp_kernels.h"

// ---------------------------------------------------------------------------
// n_photons: the u32 device counters as doubles in the accumulator block (the block is what the ranks all-reduce,
// mpi_routines.f90:303-310), and back after the collective
// ---------------------------------------------------------------------------
static __global__ void nphot_to_block_kernel(const unsigned int *__restrict__ n, double *__restrict__ out, size_t n_cells)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += step) out[i] = (double)n[i];
}

// ---------------------------------------------------------------------------
// update_energy_abs for the spectrum (:517-524): spec = sum_spec * scale / volume, 0 where the volume is 0
// ---------------------------------------------------------------------------
static __global__ void spectrum_update_kernel(const DProblem *__restrict__ Pp, const double *__restrict__ sum_spec, double *__restrict__ spec,
                                       double scale, int n_bins)
{
    const DProblem &P = *Pp;
    const int nd = P.n_dust;
    const size_t n = (size_t)P.n_cells * nd, step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        const double vol = cell_volume(P, k / nd);
        for (int b = 0; b < n_bins; b++) {
            double e = sum_spec[(size_t)b * n + k] * scale / vol;
            if (vol == 0.0) e = 0.0;
            spec[(size_t)b * n + k] = e;
        }
    }
}

// [n_bins][n_cells][n_dust] (device) -> [n_bins][n_dust][n_cells] (reference layout)
static __global__ void spectrum_to_ref_kernel(const double *__restrict__ in, double *__restrict__ out, size_t n_cells, int nd, int n_bins)
{
    const size_t n = n_cells * nd, step = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n * n_bins; k += step) {
        const size_t b = k / n, r = k - b * n, ic = r / nd;
        const int d = (int)(r - ic * nd);
        out[(b * nd + d) * n_cells + ic] = in[k];
    }
}

// ---------------------------------------------------------------------------
// Partial diffusion approximation
// ---------------------------------------------------------------------------
struct PdaCtl {
    double total_photons;            // sum of n_photons over the grid
    unsigned int n_pda;              // cells the PDA is solved in
    int sweeps;                      // Gauss-Seidel sweeps of the last solve
    unsigned long long maxdiff_bits; // max |s - s_prev| / s_prev of the last update, as the bits of a non-negative double
};

__device__ __forceinline__ void pda_cell_coords(const DProblem &P, size_t ic, int i[3])
{
    i[0] = (int)(ic % P.n1);
    const size_t t = ic / P.n1;
    i[1] = (int)(t % P.n2); i[2] = (int)(t / P.n2);
}

// cell_width: grid_geometry_cartesian_3d.f90:49-61, _spherical_3d.f90:60-72, _cylindrical_3d.f90:60-72
__device__ __forceinline__ double pda_cell_width(const DProblem &P, const int i[3], int dir)
{
    const double *w1 = P.w[0], *w2 = P.w[1], *w3 = P.w[2];
    if (P.grid_type == 1) return P.w[dir][i[dir] + 1] - P.w[dir][i[dir]];
    // centre in the first coordinate: half the outer wall if the inner wall is 0, the geometric mean otherwise
    const double rc = w1[i[0]] == 0.0 ? w1[i[0] + 1] / 2.0 : exp10((log10(w1[i[0]]) + log10(w1[i[0] + 1])) / 2.0);
    if (P.grid_type == 5) {
        if (dir == 0) return w1[i[0] + 1] - w1[i[0]];
        if (dir == 1) return rc * (w2[i[1] + 1] - w2[i[1]]);
        return rc * sin((w2[i[1]] + w2[i[1] + 1]) / 2.0) * (w3[i[2] + 1] - w3[i[2]]);
    }
    if (dir == 0) return w1[i[0] + 1] - w1[i[0]];
    if (dir == 1) return w2[i[1] + 1] - w2[i[1]];
    return rc * (w3[i[2] + 1] - w3[i[2]]);
}

// geometrical_factor of grid_pda_{cartesian,cylindrical,spherical}_3d.f90; wall = 0..5
__device__ __forceinline__ double pda_geom_factor(const DProblem &P, int wall, const int i[3])
{
    const double *w1 = P.w[0], *w2 = P.w[1];
    if (P.grid_type == 6) {
        if (wall == 0) return 2.0 * w1[i[0]] / (w1[i[0]] + w1[i[0] + 1]);
        if (wall == 1) return 2.0 * w1[i[0] + 1] / (w1[i[0]] + w1[i[0] + 1]);
    } else if (P.grid_type == 5) {
        const double sw = w1[i[0]] + w1[i[0] + 1];
        if (wall == 0) return 4.0 * (w1[i[0]] * w1[i[0]]) / (sw * sw);
        if (wall == 1) return 4.0 * (w1[i[0] + 1] * w1[i[0] + 1]) / (sw * sw);
        if (wall == 2) return 2.0 * sin(w2[i[1]]) / (sin(w2[i[1]]) + sin(w2[i[1] + 1]));
        if (wall == 3) return 2.0 * sin(w2[i[1] + 1]) / (sin(w2[i[1]]) + sin(w2[i[1] + 1]));
    }
    return 1.0;
}

__device__ __forceinline__ size_t pda_neighbour(const DProblem &P, const int i[3], int wall, int j[3])    // next_cell_int
{
    j[0] = i[0]; j[1] = i[1]; j[2] = i[2];
    const int dir = wall >> 1;
    j[dir] += (wall & 1) ? 1 : -1;
    if (dir == 2 && P.grid_type != 1) { if (j[2] < 0) j[2] = P.n3 - 1; if (j[2] >= P.n3) j[2] = 0; }    // phi is periodic
    return ((size_t)j[2] * P.n2 + j[1]) * P.n1 + j[0];
}

__device__ __forceinline__ double pda_dtau_rosseland(const DProblem &P, const double *__restrict__ se, const double *__restrict__ rho,
                                                     size_t ic, const int i[3], int dir)
{
    double t = 0.0;
    for (int d = 0; d < P.n_dust; d++) {
        const size_t k = ic * P.n_dust + d;
        t += rho[k] * chi_rosseland(P.dust[d], se[k]) * pda_cell_width(P, i, dir);
    }
    return t;
}

// update_e_mean :72-82
__device__ __forceinline__ double pda_e_mean(const DProblem &P, const double *__restrict__ se, const double *__restrict__ rho, size_t ic)
{
    double sr = 0.0, e = 0.0;
    for (int d = 0; d < P.n_dust; d++) sr += rho[ic * P.n_dust + d];
    if (!(sr > 0.0)) return 0.0;
    for (int d = 0; d < P.n_dust; d++) {
        const size_t k = ic * P.n_dust + d;
        e += rho[k] * se[k] / mean_opacity(P.dust[d], P.dust[d].mo_kappa_planck, se[k]);
    }
    return e / sr;
}

static __global__ void pda_total_kernel(const double *__restrict__ nphot, size_t n_cells, PdaCtl *__restrict__ ctl)
{
    double s = 0.0;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += step) s += nphot[i];
    s = wave_sum(s);
    if (__lane_id() == 0 && s != 0.0) unsafeAtomicAdd(&ctl->total_photons, s);
}

// do_pda = n_photons < threshold and some dust in the cell, minus the cells on the outer faces (check_allowed_pda);
// e_mean of every cell; histogram of the PDA cells over the hyperplanes i1 + i2 + i3
static __global__ void pda_mask_kernel(const DProblem *__restrict__ Pp, const double *__restrict__ nphot, double threshold,
                                const double *__restrict__ se, const double *__restrict__ rho, unsigned char *__restrict__ mask,
                                double *__restrict__ e_mean, unsigned int *__restrict__ hp_count, PdaCtl *__restrict__ ctl)
{
    const DProblem &P = *Pp;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned int mine = 0;
    for (size_t ic = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ic < (size_t)P.n_cells; ic += step) {
        int i[3];
        pda_cell_coords(P, ic, i);
        double sr = 0.0;
        for (int d = 0; d < P.n_dust; d++) sr += rho[ic * P.n_dust + d];
        bool on = nphot[ic] < threshold && sr > 0.0;
        if (i[0] == 0 || i[0] == P.n1 - 1 || i[1] == 0 || i[1] == P.n2 - 1) on = false;
        if (P.grid_type == 1 && (i[2] == 0 || i[2] == P.n3 - 1)) on = false;
        mask[ic] = on ? 1 : 0;
        e_mean[ic] = pda_e_mean(P, se, rho, ic);
        if (on) { mine++; atomicAdd(&hp_count[i[0] + i[1] + i[2]], 1u); }
    }
    if (mine) atomicAdd(&ctl->n_pda, mine);
}

// exclusive scan of the hyperplane histogram (a few hundred entries): one thread
static __global__ void pda_scan_kernel(const unsigned int *__restrict__ hp_count, unsigned int *__restrict__ hp_off, unsigned int *__restrict__ hp_cursor,
                                int n_hp)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned int run = 0;
        for (int s = 0; s < n_hp; s++) { hp_off[s] = run; hp_cursor[s] = 0; run += hp_count[s]; }
        hp_off[n_hp] = run;
    }
}

static __global__ void pda_list_kernel(const DProblem *__restrict__ Pp, const unsigned char *__restrict__ mask, const unsigned int *__restrict__ hp_off,
                                unsigned int *__restrict__ hp_cursor, unsigned int *__restrict__ cells)
{
    const DProblem &P = *Pp;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t ic = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ic < (size_t)P.n_cells; ic += step) {
        if (!mask[ic]) continue;
        int i[3];
        pda_cell_coords(P, ic, i);
        const int s = i[0] + i[1] + i[2];
        cells[hp_off[s] + atomicAdd(&hp_cursor[s], 1u)] = (unsigned int)ic;
    }
}

// start of solve_pda_indiv_*: e_mean of the PDA cells from the current specific energy, and the coefficient of every
// wall of every PDA cell (they depend on the specific energy, which only changes after the solve)
static __global__ void pda_coef_kernel(const DProblem *__restrict__ Pp, const unsigned int *__restrict__ cells, unsigned int n_pda,
                                const double *__restrict__ se, const double *__restrict__ rho, double *__restrict__ e_mean,
                                double *__restrict__ coef, int exact)
{
    const DProblem &P = *Pp;
    const int n_walls = P.grid_type == 1 ? 6 : P.n_dim * 2;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pda; q += step) {
        const size_t ic = cells[q];
        int i[3];
        pda_cell_coords(P, ic, i);
        e_mean[ic] = pda_e_mean(P, se, rho, ic);
        for (int wall = 0; wall < 6; wall++) {
            double c = 0.0;
            if (wall < n_walls) {
                const int dir = wall >> 1;
                int j[3];
                const size_t jc = pda_neighbour(P, i, wall, j);
                double dsum = pda_dtau_rosseland(P, se, rho, ic, i, dir) + pda_dtau_rosseland(P, se, rho, jc, j, dir);
                if (exact && dsum < 1e-100) dsum = 1e-100;
                c = 1. / dsum / pda_cell_width(P, i, dir);
                c = c * pda_geom_factor(P, wall, i);
            }
            coef[6 * q + wall] = c;
        }
    }
}

// Gauss-Seidel sweeps over the PDA cells, solve_pda_indiv_iterative :258-325.  The reference updates the cells one
// after the other in cell order; a cell's neighbours sit on the hyperplanes i1 + i2 + i3 -+ 1 (or across the phi seam),
// so updating hyperplane after hyperplane, all cells of one in parallel, gives every cell exactly the operands the
// sequential loop gives it: same result, bit for bit.  One workgroup (the barrier between hyperplanes is a
// __syncthreads); sweeps until the largest relative change of a sweep is below `tol`.
static __global__ __launch_bounds__(1024) void pda_gs_kernel(const DProblem *__restrict__ Pp, const unsigned int *__restrict__ cells,
                                                      const unsigned int *__restrict__ hp_off, int n_hp, const double *__restrict__ coef,
                                                      double *__restrict__ e_mean, double tol, int max_sweeps, PdaCtl *__restrict__ ctl)
{
    const DProblem &P = *Pp;
    const int n_walls = P.grid_type == 1 ? 6 : P.n_dim * 2;
    __shared__ double red[16];
    __shared__ int done;
    int sweep = 0;
    for (; sweep < max_sweeps; sweep++) {
        double my_max = 0.0;
        for (int s = 0; s < n_hp; s++) {
            const unsigned int q0 = hp_off[s], q1 = hp_off[s + 1];
            if (q0 == q1) continue;          // uniform over the block
            for (unsigned int q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
                const size_t ic = cells[q];
                int i[3];
                pda_cell_coords(P, ic, i);
                double a = 0.0, b = 0.0;
                for (int wall = 0; wall < n_walls; wall++) {
                    int j[3];
                    const size_t jc = pda_neighbour(P, i, wall, j);
                    const double c = coef[6 * (size_t)q + wall];
                    a = a - c;
                    b = b - c * ((volatile double *)e_mean)[jc];
                }
                const double e_old = e_mean[ic], e_new = b / a;
                const double diff = fabs(e_new - e_old) / e_old;
                if (diff > my_max) my_max = diff;
                e_mean[ic] = e_new;
            }
            __threadfence_block();
            __syncthreads();
        }
        double m = my_max;
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
        if (__lane_id() == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (unsigned int w = 0; w < blockDim.x / 64; w++) t = fmax(t, red[w]);
            done = t < tol;
        }
        __syncthreads();
        if (done) { sweep++; break; }
    }
    if (threadIdx.x == 0) ctl->sweeps = sweep;
}

// solve_pda_indiv_exact :185-256 -- fewer than 10 000 PDA cells: the dense system  a x = b  with one row per cell's
// equation  sum_walls c (e_next - e_curr) = 0  (unknown neighbours on the left, Monte Carlo neighbours on the right),
// solved by Gaussian elimination with partial pivoting.  Every row is diagonally dominant (|a_qq| = sum of its coefficients
// >= the sum of its off-diagonal entries), so the pivot search normally confirms the diagonal; it matters where the 1e-100
// clamp of pda_coef leaves rows of wildly different scale.  The matrix is sparse (7-point stencil); rows whose entry in the
// pivot column is zero are skipped, so the work follows the fill-in, not n^3.
static __global__ void pda_dense_build_kernel(const DProblem *__restrict__ Pp, const unsigned int *__restrict__ cells, unsigned int n_pda,
                                       const unsigned int *__restrict__ id_of_cell, const double *__restrict__ coef,
                                       const double *__restrict__ e_mean, double *__restrict__ a, double *__restrict__ b)
{
    const DProblem &P = *Pp;
    const int n_walls = P.grid_type == 1 ? 6 : P.n_dim * 2;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pda; q += step) {
        const size_t ic = cells[q];
        int i[3];
        pda_cell_coords(P, ic, i);
        double diag = 0.0, rhs = 0.0;
        for (int wall = 0; wall < n_walls; wall++) {
            int j[3];
            const size_t jc = pda_neighbour(P, i, wall, j);
            const double c = coef[6 * q + wall];
            diag -= c;
            const unsigned int qn = id_of_cell[jc];
            if (qn != 0xffffffffu) a[q * n_pda + qn] += c;       // (two walls can lead to the same cell across a 2-cell phi seam)
            else rhs -= c * e_mean[jc];
        }
        a[q * n_pda + q] += diag;
        b[q] = rhs;
    }
}

static __global__ void pda_id_kernel(const unsigned int *__restrict__ cells, unsigned int n_pda, unsigned int *__restrict__ id_of_cell)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pda; q += step) id_of_cell[cells[q]] = (unsigned int)q;
}

// elimination step k, partial pivoting (the reference calls fortranlib's lineq_gausselim, grid_pda_3d.f90:246): the row at or
// below k with the largest |a[r][k]| -- the first of them, like the oracle's search -- is found by one workgroup ...
static __global__ __launch_bounds__(1024) void pda_pivot_kernel(const double *__restrict__ a, unsigned int n, unsigned int k, unsigned int *__restrict__ piv)
{
    __shared__ double big_s[16];
    __shared__ unsigned int row_s[16];
    double big = -1.0; unsigned int row = k;
    for (unsigned int r = k + threadIdx.x; r < n; r += blockDim.x) {        // ascending r per thread: a strict > keeps the first maximum
        const double v = fabs(a[(size_t)r * n + k]);
        if (v > big) { big = v; row = r; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(big, off, 64);
        const unsigned int orow = __shfl_xor(row, off, 64);
        if (ob > big || (ob == big && orow < row)) { big = ob; row = orow; }
    }
    if (__lane_id() == 0) { big_s[threadIdx.x >> 6] = big; row_s[threadIdx.x >> 6] = row; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned int w = 1; w < blockDim.x / 64; w++) if (big_s[w] > big || (big_s[w] == big && row_s[w] < row)) { big = big_s[w]; row = row_s[w]; }
        *piv = row;
    }
}
// ... and exchanged with row k (columns >= k: the others are already zero in both; and the right-hand side)
static __global__ void pda_swap_kernel(double *__restrict__ a, double *__restrict__ b, unsigned int n, unsigned int k, const unsigned int *__restrict__ piv)
{
    const unsigned int p = *piv;
    if (p == k) return;
    double *rk = a + (size_t)k * n, *rp = a + (size_t)p * n;
    for (unsigned int c = k + blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) { const double t = rk[c]; rk[c] = rp[c]; rp[c] = t; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { const double t = b[k]; b[k] = b[p]; b[p] = t; }
}
// then the factors f[r] = a[r][k] / a[k][k] for the rows below the pivot ...
static __global__ void pda_elim_factor_kernel(const double *__restrict__ a, unsigned int n, unsigned int k, double *__restrict__ f)
{
    const unsigned int r = k + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) f[r] = a[(size_t)r * n + k] / a[(size_t)k * n + k];
}
// ... and row r -= f[r] * row k (one workgroup row of the grid per matrix row; rows with a zero factor return at once)
static __global__ void pda_elim_update_kernel(double *__restrict__ a, double *__restrict__ b, unsigned int n, unsigned int k, const double *__restrict__ f)
{
    const unsigned int r = k + 1 + blockIdx.y;
    const double fr = f[r];
    if (fr == 0.0) return;
    const double *pk = a + (size_t)k * n;
    double *pr = a + (size_t)r * n;
    for (unsigned int c = k + 1 + blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const double v = pk[c];
        if (v != 0.0) pr[c] -= fr * v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { b[r] -= fr * b[k]; pr[k] = 0.0; }
}
// back substitution, one workgroup: x overwrites b
static __global__ __launch_bounds__(1024) void pda_backsub_kernel(const double *__restrict__ a, double *__restrict__ b, unsigned int n)
{
    __shared__ double red[16];
    for (unsigned int kk = n; kk-- > 0;) {
        const double *row = a + (size_t)kk * n;
        double s = 0.0;
        for (unsigned int c = kk + 1 + threadIdx.x; c < n; c += blockDim.x) { const double v = row[c]; if (v != 0.0) s += v * b[c]; }
        s = wave_sum(s);
        if (__lane_id() == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (unsigned int w = 0; w < blockDim.x / 64; w++) t += red[w];
            b[kk] = (b[kk] - t) / row[kk];
        }
        __threadfence_block();
        __syncthreads();
    }
}
static __global__ void pda_scatter_solution_kernel(const unsigned int *__restrict__ cells, unsigned int n_pda, const double *__restrict__ x,
                                            double *__restrict__ e_mean)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pda; q += step) e_mean[cells[q]] = x[q];
}

// update_specific_energy :36-70 for the PDA cells + the rescaling of their spectrum; the largest relative change
static __global__ void pda_update_kernel(const DProblem *__restrict__ Pp, const unsigned int *__restrict__ cells, unsigned int n_pda,
                                  const double *__restrict__ e_mean, double *__restrict__ se, double *__restrict__ spec, int n_bins,
                                  PdaCtl *__restrict__ ctl)
{
    const DProblem &P = *Pp;
    const int nd = P.n_dust;
    const size_t n = (size_t)P.n_cells * nd, step = (size_t)gridDim.x * blockDim.x;
    double my_max = 0.0;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pda; q += step) {
        const size_t ic = cells[q];
        const double em = e_mean[ic];
        for (int d = 0; d < nd; d++) {
            const DDust &D = P.dust[d];
            const size_t k = ic * nd + d;
            double s = se[k];
            const double s_old = s, smin = D.mo_e[0], smax = D.mo_e[D.n_e - 1];
            if (em < smin / mean_opacity(D, D.mo_kappa_planck, smin)) s = smin;
            else if (em > smax / mean_opacity(D, D.mo_kappa_planck, smax)) s = smax;
            else {
                for (int it = 0; it < 100000; it++) {
                    const double s_prev = s;
                    s = em * mean_opacity(D, D.mo_kappa_planck, s);
                    if (fmax(s / s_prev, s_prev / s) - 1.0 < 1.e-5) break;
                    if (s != s) break;
                }
            }
            se[k] = s;
            if (n_bins && s_old > 0.0) {
                const double f = s / s_old;
                for (int b = 0; b < n_bins; b++) spec[(size_t)b * n + k] *= f;
            }
            const double dv = fabs(s - s_old) / s_old;
            if (dv > my_max) my_max = dv;
        }
    }
    for (int off = 32; off > 0; off >>= 1) my_max = fmax(my_max, __shfl_xor(my_max, off, 64));
    if (__lane_id() == 0 && my_max > 0.0) atomicMax(&ctl->maxdiff_bits, (unsigned long long)__double_as_longlong(my_max));
}

// ---------------------------------------------------------------------------
// specific_energy_converged: grid_physics_3d.f90:637-689
// ---------------------------------------------------------------------------
struct ConvCtl {
    unsigned long long n_valid;      // pairs that changed and are positive before and after
    unsigned long long n_changed;    // pairs that changed at all
    unsigned long long n_changed_nonzero;   // ... with neither value zero
    unsigned long long count;        // scratch of the selection passes
};

// ratio[k] = max(a / b, b / a) of the pairs that count, 0 elsewhere (every valid ratio is > 1)
static __global__ void conv_ratio_kernel(const double *__restrict__ prev, const double *__restrict__ cur, size_t n, double *__restrict__ ratio,
                                  ConvCtl *__restrict__ ctl)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned long long nv = 0, nc = 0, nz = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        const double a = prev[k], b = cur[k];
        double r = 0.0;
        if (a != b) {
            nc++;
            if (a != 0.0 && b != 0.0) nz++;
            if (a > 0.0 && b > 0.0) { r = fmax(a / b, b / a); nv++; }
        }
        ratio[k] = r;
    }
    nv = (unsigned long long)wave_sum((double)nv); nc = (unsigned long long)wave_sum((double)nc); nz = (unsigned long long)wave_sum((double)nz);
    if (__lane_id() == 0) {
        if (nv) atomicAdd(&ctl->n_valid, nv);
        if (nc) atomicAdd(&ctl->n_changed, nc);
        if (nz) atomicAdd(&ctl->n_changed_nonzero, nz);
    }
}

// how many valid ratios have a bit pattern below `limit` (positive doubles order like their bits)
static __global__ void conv_count_kernel(const double *__restrict__ ratio, size_t n, unsigned long long limit, ConvCtl *__restrict__ ctl)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += step) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(ratio[k]);
        if (b != 0ull && b < limit) c++;
    }
    c = (unsigned long long)wave_sum((double)c);
    if (__lane_id() == 0 && c) atomicAdd(&ctl->count, c);
}
