// Micro-benchmark: issue rates of plain / packed fp32 FMA and of the 4x4x1 fp32 MFMA on gfx950, and whether the two pipes
// overlap inside one wave and across the waves of a SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o ub_pipes ub_pipes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, float seed) {
    const int lane = threadIdx.x & 63;
    v2f a[8]; v4f c[8]; float s[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (v2f){ seed * i, seed + lane }; c[i] = (v4f){ seed, 0.f, 1.f, (float)lane }; s[i] = seed * lane + i; }
    const v2f m = { 1.0001f, 0.9999f }; const float ms = 1.0001f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0 || MODE == 3) a[i] = __builtin_elementwise_fma(a[i], m, m);                    // packed FMA
            if (MODE == 1) s[i] = fmaf(s[i], ms, ms);                                                    // plain FMA
            if (MODE == 2 || MODE == 3) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(ms, s[i], c[i], 0, 0, 0);
        }
        if (MODE == 4) {
            if (__builtin_amdgcn_readfirstlane(threadIdx.x) & 256) {
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = __builtin_elementwise_fma(a[i], m, m);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(ms, s[i], c[i], 0, 0, 0);
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y + c[i].x + c[i].y + c[i].z + c[i].w + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;
    k<MODE><<<blocks, 1024>>>(d, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 1024>>>(d, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks*16 waves / 1024 SIMDs, each ITER*8 "slots"
    double slots = (double)blocks * 16 / 1024 * ITER * 8;
    printf("%-28s %8.3f ms  %6.2f cycles per slot per SIMD (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / slots);
}

int main() {
    float *d; hipMalloc(&d, 256 * 4 * 1024 * 4);
    run<0>("packed fma", d); run<1>("plain fma", d); run<2>("mfma 4x4x1", d); run<3>("packed fma + mfma (1 wave)", d); run<4>("odd waves fma, even mfma", d);
    return 0;
}
