// compares find_wall_inside with geo_find_wall on random inputs (GPU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <string>
#include <vector>
#include "../../hyperion_amd/csrc/hyp_device.h"
#include "../../hyperion_amd/csrc/hyp_kernels.h"
#include "../../hyperion_amd/csrc/hyp_tiled.h"

__device__ double u01(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) * (1.0 / 9007199254740992.0); }

__global__ void k(const double *w, const double *ew, int n, unsigned long long *bad, double *out)
{
    DProblem P; P.n1 = P.n2 = P.n3 = n;
    Walls W; for (int a = 0; a < 3; a++) { W.w[a] = w; W.ew[a] = ew; W.n[a] = n; }
    uint64_t s = 1234567ull + 977ull * (blockIdx.x * blockDim.x + threadIdx.x);
    for (int it = 0; it < 2000; it++) {
        Cell<GEOM_CAR> c; double r[3], v[3];
        for (int a = 0; a < 3; a++) {
            c.ic[a] = (int)(u01(s) * n); c.ow[a] = 0;
            double wl = w[c.ic[a]], wu = w[c.ic[a] + 1];
            r[a] = wl + u01(s) * (wu - wl);
            int kk = (int)(u01(s) * 8);
            if (kk == 0) { r[a] = wl; c.ow[a] = -1; } else if (kk == 1) { r[a] = wu; c.ow[a] = 1; } else if (kk == 2) r[a] = wl; else if (kk == 3) r[a] = wu;
            else if (kk == 4) { r[a] = nextafter(wl, -9.0); c.ow[a] = (int)(u01(s) * 3) - 1; } else if (kk == 5) { r[a] = nextafter(wu, 9.0); c.ow[a] = (int)(u01(s) * 3) - 1; }
            v[a] = 2.0 * u01(s) - 1.0;
        }
        if (u01(s) < 0.1) v[(int)(u01(s) * 3)] = 0.0;
        double t1, t2; int im1[3], im2[3]; bool f2;
        bool f1 = geo_find_wall(P, W, r, v, c, t1, im1);
        int iu[3]; double sg[3];
        for (int a = 0; a < 3; a++) { iu[a] = v[a] > 0.0 ? 1 : 0; sg[a] = v[a] > 0.0 ? 1.0 : (v[a] < 0.0 ? -1.0 : 0.0); }
        bool ins = find_wall_ahead(W, r, v, iu, sg, c, t2, im2, f2);
        if (ins) atomicAdd(bad + 1, 1ull);
        if (ins && (f1 != f2 || (f1 && (t1 != t2 || im1[0] != im2[0] || im1[1] != im2[1] || im1[2] != im2[2])))) {
            unsigned long long b = atomicAdd(bad, 1ull);
            if (b == 0) { out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = v[0]; out[4] = v[1]; out[5] = v[2]; out[6] = t1; out[7] = t2;
                          out[8] = c.ic[0]; out[9] = c.ic[1]; out[10] = c.ic[2]; out[11] = c.ow[0]; out[12] = c.ow[1]; out[13] = c.ow[2];
                          out[14] = im1[0]; out[15] = im1[1]; out[16] = im1[2]; out[17] = im2[0]; out[18] = im2[1]; out[19] = im2[2]; out[20] = f1; out[21] = f2; }
        }
    }
}

int main()
{
    const int n = 8;
    std::vector<double> w(n + 1), ew(n + 1);
    for (int i = 0; i <= n; i++) { w[i] = -1.0 + 2.0 * i / n; ew[i] = 3.0 * fabs(w[i]) * 2.2e-16; }
    double *dw, *dew, *dout; unsigned long long *dbad;
    hipMalloc(&dw, 8 * (n + 1)); hipMalloc(&dew, 8 * (n + 1)); hipMalloc(&dout, 8 * 32); hipMalloc(&dbad, 16);
    hipMemcpy(dw, w.data(), 8 * (n + 1), hipMemcpyHostToDevice); hipMemcpy(dew, ew.data(), 8 * (n + 1), hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 16); hipMemset(dout, 0, 8 * 32);
    k<<<64, 256>>>(dw, dew, n, dbad, dout);
    unsigned long long bad2[2]; double out[32];
    hipMemcpy(bad2, dbad, 16, hipMemcpyDeviceToHost); unsigned long long bad = bad2[0]; hipMemcpy(out, dout, 8 * 32, hipMemcpyDeviceToHost);
    printf("mismatches: %llu of %llu cases on the fast path\n", bad, bad2[1]);
    if (bad) { for (int i = 0; i < 22; i++) printf("%.17g ", out[i]); printf("\n"); }
    return 0;
}
