// LDS write bandwidth on gfx950 by store width: 4 waves per block (one per SIMD), one block per CU, every wave writes 16 KB per
// iteration (64 lanes x contiguous 4 / 8 / 16 bytes x rows), cycles per iteration by s_memtime.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W> __global__ void __launch_bounds__(256) k(unsigned long long *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(smem + wave * 16384 + lane * W);
    typedef float v4 __attribute__((ext_vector_type(4))); typedef float v2 __attribute__((ext_vector_type(2)));
    v4 v = { (float)tid, 1.f, 2.f, 3.f }; v2 v8 = { (float)tid, 1.f }; float v1 = (float)tid;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        constexpr int ROWS = 16384 / (64 * W);
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const unsigned a = base + r * 64 * W;
            if constexpr (W == 16) asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory");
            else if constexpr (W == 8) asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v8) : "memory");
            else asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(v1) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    unsigned long long *d, h;
    hipMalloc(&d, 8);
    const int iters = 2000;
    hipFuncSetAttribute((const void *)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 65536, 0, d, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("b128: %.0f memtime ticks per 64 KB (%.1f B/tick)\n", (double)h / iters, 65536.0 * iters / h);
        hipLaunchKernelGGL(k<8>, dim3(256), dim3(256), 65536, 0, d, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("b64:  %.0f memtime ticks per 64 KB (%.1f B/tick)\n", (double)h / iters, 65536.0 * iters / h);
        hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 65536, 0, d, iters); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("b32:  %.0f memtime ticks per 64 KB (%.1f B/tick)\n", (double)h / iters, 65536.0 * iters / h);
    }
    return 0;
}
