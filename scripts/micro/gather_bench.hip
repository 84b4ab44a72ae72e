// Micro-benchmark: cost of a 64-lane 16-byte gather (global_load_dwordx4) on gfx950 as a function of the number of
// distinct 128-byte lines the lanes touch (data resident in L2 / L1: a 8 MB table, lines chosen per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

// lanes_per_line lanes share one line; lines of one instruction are `spread` lines apart
template <int LPL>
__global__ void __launch_bounds__(256) k(const float4 *tab, float *out, int iters, int nlines_mask, int spread) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    float acc = 0.f;
    unsigned base = (unsigned)wave * 7919u;
    for (int it = 0; it < iters; it++) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned line = (base + (unsigned)(lane / LPL) * (unsigned)spread + (unsigned)u * 131u) & (unsigned)nlines_mask;
            v[u] = tab[(size_t)line * 8 + (lane % LPL) % 8];         // 8 float4 per 128-byte line
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x + v[u].w;
        base += 977u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int LPL>
void run(const float4 *tab, float *out, int mask, int spread, const char *name) {
    const int blocks = 256 * 4, iters = 400;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<LPL>), dim3(blocks), dim3(256), 0, 0, tab, out, 10, mask, spread);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); hipLaunchKernelGGL((k<LPL>), dim3(blocks), dim3(256), 0, 0, tab, out, iters, mask, spread); CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    double winstr = (double)blocks * 4 * iters * 4;
    printf("%-34s %7.3f ms  %.1f cycles per wave-load per CU (64/%d = %d lines)\n", name, ms, ms * 1e-3 * 2.4e9 * 256.0 / winstr, LPL, 64 / LPL);
}

int main() {
    const int nlines = 1 << 16;      // 8 MB
    float4 *tab; float *out;
    CHK(hipMalloc(&tab, (size_t)nlines * 128)); CHK(hipMemset(tab, 0, (size_t)nlines * 128));
    CHK(hipMalloc(&out, 1024 * 256 * 4));
    run<1>(tab, out, nlines - 1, 37, "64 lines, scattered");
    run<2>(tab, out, nlines - 1, 37, "32 lines, scattered");
    run<4>(tab, out, nlines - 1, 37, "16 lines, scattered");
    run<8>(tab, out, nlines - 1, 37, "8 lines, scattered");
    run<1>(tab, out, nlines - 1, 1, "64 lines, consecutive");
    run<4>(tab, out, nlines - 1, 1, "16 lines, consecutive");
    run<8>(tab, out, nlines - 1, 1, "8 lines, consecutive (1 KB contiguous)");
    run<1>(tab, out, 1023, 37, "64 lines, scattered, 128 KB table");
    run<4>(tab, out, 1023, 37, "16 lines, scattered, 128 KB table");
    return 0;
}
