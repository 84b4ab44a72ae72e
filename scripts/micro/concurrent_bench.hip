// Micro-benchmark: do two kernels from two streams share the CUs of gfx950?  A = vector-issue-bound, one 512-thread block per CU
// (128 KB of dynamic LDS, ~168 VGPRs worth of state is not modelled: launch bounds only), B = gather-bound, 256 threads and 16 KB of LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

__global__ void __launch_bounds__(512, 1) kA(float *out, int iters) {
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) { a = a * b + c; d = d * b + a; c = c * b + d; }
    }
    out[(blockIdx.x & 4095) * 512 + threadIdx.x] = a + c + d + lds[(threadIdx.x + 1) & 127];
}
__global__ void __launch_bounds__(256, 4) kB(const float4 *tab, float *out, int iters, unsigned mask) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 0.f;
    unsigned idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        float4 v0 = tab[idx & mask], v1 = tab[(idx >> 3) & mask], v2 = tab[(idx >> 5) & mask], v3 = tab[(idx >> 7) & mask];
        acc += v0.x + v1.y + v2.z + v3.w;
        idx = idx * 1664525u + 1013904223u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + lds[threadIdx.x];
}

int main(int argc, char **argv) {
    const int ldsA = (argc > 1 ? atoi(argv[1]) : 128) * 1024, thrA = argc > 2 ? atoi(argv[2]) : 512;
    const unsigned mask = (1u << 20) - 1;      // 16 MB table
    float4 *tab; float *oa, *ob;
    CHK(hipMalloc(&tab, (size_t)(mask + 1) * 16)); CHK(hipMemset(tab, 0, (size_t)(mask + 1) * 16));
    CHK(hipMalloc(&oa, 256 * 64 * 512 * 4)); CHK(hipMalloc(&ob, 256 * 256 * 256 * 4));
    CHK(hipFuncSetAttribute((const void *)kA, hipFuncAttributeMaxDynamicSharedMemorySize, ldsA));
    hipStream_t s1, s2, s3; int lo, hi;
    CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CHK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, hi));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int gA = 256 * 16 * 10, gB = 256 * 32, itA = 4000, itB = 200;
    auto time = [&](const char *name, int mode) {
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, s1));
            hipStream_t sb = mode == 2 ? s2 : (mode == 3 ? s3 : s1);
            if (mode != 1) hipLaunchKernelGGL(kA, dim3(gA * 512 / thrA), dim3(thrA), ldsA, s1, oa, itA);
            if (mode != 0) hipLaunchKernelGGL(kB, dim3(gB), dim3(256), 16 * 1024, sb, tab, ob, itB, mask);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e1, s1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-52s %8.2f ms\n", name, ms);
        }
    };
    printf("stream priorities: low %d, high %d; A: %d KB of LDS, %d threads\n", lo, hi, ldsA / 1024, thrA);
    time("A alone (issue-bound, 128 KB LDS, 1 block per CU)", 0);
    time("B alone (gathers, 16 KB LDS)", 1);
    time("A then B on ONE stream", 4);
    time("A and B on two streams", 2);
    time("A and B on two streams, B at high priority", 3);
    return 0;
}
