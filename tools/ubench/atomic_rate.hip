// Microbenchmark: throughput of scattered global atomics on MI355X, the
// operation that bounds the Lucy kernel (one FP64 atomic add per cell crossing).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <typename T, int MODE>   // MODE 0: atomic add, 1: plain load (gather), 2: plain store (scatter), 3: load+atomic
__global__ void k(T *buf, size_t mask, int iters, T *sink)
{
    uint32_t s = mix(blockIdx.x * blockDim.x + threadIdx.x + 12345u);
    T acc = 0;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        size_t idx = (size_t)mix(s) & mask;
        if (MODE == 0) unsafeAtomicAdd(&buf[idx], (T)1);
        else if (MODE == 1) acc += buf[idx];
        else if (MODE == 2) buf[idx] = (T)i;
        else { acc += buf[idx ^ 1]; unsafeAtomicAdd(&buf[idx], (T)1); }
    }
    if (acc == (T)-1) *sink = acc;
}

template <typename T, int MODE>
double run(size_t n_elem, int blocks, int iters)
{
    T *buf, *sink;
    hipMalloc(&buf, n_elem * sizeof(T)); hipMemset(buf, 0, n_elem * sizeof(T)); hipMalloc(&sink, sizeof(T));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<T, MODE><<<blocks, 256>>>(buf, n_elem - 1, iters / 8, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<T, MODE><<<blocks, 256>>>(buf, n_elem - 1, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipFree(buf); hipFree(sink);
    return (double)blocks * 256 * iters / (ms * 1e-3);
}

int main()
{
    const char *names[] = {"atomic add", "gather load", "scatter store", "load+atomic"};
    for (size_t mb : {2, 16, 128, 1024}) {
        size_t n64 = mb * 1024 * 1024 / 8;
        printf("footprint %4zu MiB:", mb);
        printf("  f64 atomic %.3e/s", run<double, 0>(n64, 2048, 4096));
        printf("  f32 atomic %.3e/s", run<float, 0>(n64 * 2, 2048, 4096));
        printf("  f64 gather %.3e/s", run<double, 1>(n64, 2048, 4096));
        printf("  f64 scatter-store %.3e/s", run<double, 2>(n64, 2048, 4096));
        printf("  f64 load+atomic %.3e/s\n", run<double, 3>(n64, 2048, 4096));
    }
    (void)names;
    return 0;
}
