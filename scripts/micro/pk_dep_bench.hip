// Does a packed-f32 instruction that reads the result of the one just before it cost an extra issue slot?  One wave per SIMD,
// 2048 v_pk_fma_f32 per loop trip: one dependent chain against four interleaved chains against plain v_fma_f32.
//   hipcc -O3 --offload-arch=gfx950 -o pk_dep_bench pk_dep_bench.hip && ./pk_dep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int reps, float seed) {
    v2f a = { seed, seed + 1.f }, b = { seed + 2.f, seed }, c = { seed, -seed }, d = { 1.f, seed };
    const v2f w = { 0.999f, 0.001f };
    float fa = seed, fb = seed + 1.f, fc = seed + 2.f, fd = seed + 3.f;
    for (int r = 0; r < reps; r++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 512; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(w));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 512; i++) asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 512; i++) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(fa) : "v"(w.x));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 512; i++) asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : "v"(w.x));
        } else {
            // dependent pairs the way a complex product has them: mul then fma on its result, two such pairs interleaved or not
#pragma unroll
            for (int i = 0; i < 512; i++) asm volatile("v_pk_mul_f32 %0, %1, %4\n\tv_pk_fma_f32 %1, %1, %4, %0\n\tv_pk_mul_f32 %2, %3, %4\n\tv_pk_fma_f32 %3, %3, %4, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(w));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a.x + b.y + c.x + d.y + fa + fb + fc + fd;
}

template <int MODE> static int run(const char *what, int blocks_per_cu) {
    float *d; CK(hipMalloc(&d, 256 * 256 * 4 * blocks_per_cu));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, reps, 0.5f);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, reps, 0.5f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_per_instr = ms * 1e6 / (reps * 2048.0) / blocks_per_cu;
    printf("%-58s %d wave(s)/SIMD: %.3f ns per instruction and wave = %.2f cycles at 2.0 GHz\n", what, blocks_per_cu, ns_per_instr, ns_per_instr * 2.0);
    (void)hipFree(d);
    return 0;
}

int main() {
    for (int b = 1; b <= 2; b++) {
        if (run<0>("v_pk_fma_f32, one dependent chain", b)) return 1;
        if (run<1>("v_pk_fma_f32, four independent chains", b)) return 1;
        if (run<4>("v_pk_mul_f32 -> v_pk_fma_f32 pairs (a complex product)", b)) return 1;
        if (run<2>("v_fma_f32, one dependent chain", b)) return 1;
        if (run<3>("v_fma_f32, four independent chains", b)) return 1;
    }
    return 0;
}
