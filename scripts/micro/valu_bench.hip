// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (wave64) on gfx950, 16 waves per CU, 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
    float x[8]; v2f y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = (v2f){ x[i], x[i] + 0.5f }; }
    const v2f av = { a, a }, bv = { b, b };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else y[i] = __builtin_elementwise_fma(y[i], av, bv);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name) {
    float *out; CHK(hipMalloc(&out, 4096 * 256 * 4));
    const int blocks = 256 * 4, iters = 2000;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    double winstr = (double)blocks * 4 * iters * 64;        // wave-instructions
    double per_simd_cycle = winstr / (256.0 * 4) / (ms * 1e-3 * 2.4e9);
    double flops = winstr * 64 * 2 * (MODE ? 2 : 1) / (ms * 1e-3);
    printf("%-14s %7.3f ms  %.3f wave-instr/cycle/SIMD (at 2.4 GHz) = %.1f cycles per instr; %.1f TFLOP/s\n", name, ms, per_simd_cycle, 1.0 / per_simd_cycle, flops / 1e12);
    CHK(hipFree(out));
}

int main() { run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); return 0; }
