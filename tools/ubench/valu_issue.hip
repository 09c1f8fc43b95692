// valu_issue.hip -- issue cost (cycles per wave64 instruction, per SIMD) of the vector instructions the cell walk is made of,
// measured on the device: one wave per SIMD and four waves per SIMD, dependent-free streams of 8 independent chains.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue && tools/ubench/valu_issue
// Output: one line per instruction with cycles per instruction at 1 and 4 waves per SIMD (s_memtime ticks = shader cycles,
// /opt/skills/guides/MI355X_MICROARCH.md "Per-instruction cycle constants").  Feeds bench.py's issue_roofline.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 256          // instructions per chain per loop trip (8 chains)
#define TRIPS 2048

#define CHAINS8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int WHICH>
__global__ __launch_bounds__(256) void k(unsigned long long *out, double seed)
{
    double a[8], b = seed * 1.0000001, c = seed * 0.9999999;
    float fa[8], fb = (float)seed, fc = (float)seed * 0.5f;
    int ia[8], ib = (int)seed + 3;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + i; fa[i] = (float)(seed + i); ia[i] = (int)seed + i; }
    unsigned long long t0 = __builtin_readcyclecounter();
    t0 = clock64();
    for (int t = 0; t < TRIPS; t++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define MAX64(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
#define RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#define LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(ib));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ib) : );
#define ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define MADU64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(ib), "v"(ia[i]) : "vcc");
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa[i]) : "v"(fb), "v"(fc));
#define MOV32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(ia[i]) : "v"(ib));
#define AND32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
#define LSHL64(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
            if (WHICH == 0) { CHAINS8(FMA64) }
            if (WHICH == 1) { CHAINS8(MUL64) }
            if (WHICH == 2) { CHAINS8(ADD64) }
            if (WHICH == 3) { CHAINS8(MAX64) }
            if (WHICH == 4) { CHAINS8(CMP64) }
            if (WHICH == 5) { CHAINS8(RCP64) }
            if (WHICH == 6) { CHAINS8(LDEXP64) }
            if (WHICH == 7) { CHAINS8(CNDMASK) }
            if (WHICH == 8) { CHAINS8(ADD32) }
            if (WHICH == 9) { CHAINS8(MADU64) }
            if (WHICH == 10) { CHAINS8(FMA32) }
            if (WHICH == 11) { CHAINS8(MOV32) }
            if (WHICH == 12) { CHAINS8(AND32) }
            if (WHICH == 13) { CHAINS8(LSHL64) }
        }
    }
    unsigned long long t1 = clock64();
    double s = 0.0; float fs = 0.f; int is = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { s += a[i]; fs += fa[i]; is += ia[i]; }
    if (s == 1.2345 && fs == 1.5f && is == 7) out[1] = 1;       // keep the chains alive
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

static double g_ticks = 0.0;
template <int WHICH>
double run(int waves_per_simd, unsigned long long *d_out)
{
    // 256 threads = 4 waves = one per SIMD of a CU; waves_per_simd blocks per CU on every CU
    int dev = 0; hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, dev);
    const int blocks = p.multiProcessorCount * waves_per_simd;
    k<WHICH><<<blocks, 256>>>(d_out, 1.0);       // warm-up
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    k<WHICH><<<blocks, 256>>>(d_out, 1.0);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    unsigned long long t = 0;
    (void)hipMemcpy(&t, d_out, sizeof t, hipMemcpyDeviceToHost);
    g_ticks = (double)t / ((double)REP * TRIPS);      // s_memtime ticks per instruction as seen by one wave
    // every SIMD of the chip issues REP x TRIPS instructions for each of its waves_per_simd waves in `ms`: SIMD time per
    // wave-instruction in cycles of the 2.4 GHz peak clock (the chip may run below it: DVFS)
    return (double)ms * 1e-3 * 2.4e9 / ((double)REP * TRIPS * waves_per_simd);
}

int main()
{
    unsigned long long *d_out = nullptr;
    if (hipMalloc(&d_out, 16) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    (void)hipMemset(d_out, 0, 16);
    const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_cmp_lt_f64", "v_rcp_f64", "v_ldexp_f64", "v_cndmask_b32",
                           "v_add_u32", "v_mad_u64_u32", "v_fma_f32", "v_mov_b32", "v_and_b32", "v_lshlrev_b64"};
    double r1[14], r4[14], k1[14], k4[14];
#define RUN(i) r1[i] = run<i>(1, d_out); k1[i] = g_ticks; r4[i] = run<i>(4, d_out); k4[i] = g_ticks;
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
    printf("SIMD time per wave64 instruction in 2.4 GHz cycles = kernel time (HIP events) x 2.4e9 / (instructions per wave x waves per SIMD); "
           "in brackets: s_memtime ticks per instruction seen by one wave\n\n");
    printf("| instruction | 1 wave per SIMD | 4 waves per SIMD |\n|---|---|---|\n");
    for (int i = 0; i < 14; i++) printf("| `%s` | %.2f (%.2f) | %.2f (%.2f) |\n", names[i], r1[i], k1[i], r4[i], k4[i]);
    (void)hipFree(d_out);
    return 0;
}
