// Micro-benchmark: LDS atomic throughput on gfx950 (wave-instructions per cycle per CU) for f32 / u32 / u64 adds and
// plain read-modify-write, with lane-distinct addresses (stride 1 and stride 3 words) and with 2-way same-address pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1); } } while (0)

template <int MODE, int STRIDE, int DUP>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    __shared__ float buf[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) buf[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = ((lane / DUP) * STRIDE + wave * 1024) % 4000;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int a = base + u * 512;
            if (MODE == 0) atomicAdd(&buf[a], v);
            else if (MODE == 1) atomicAdd((unsigned *)&buf[a], (unsigned)lane);
            else if (MODE == 2) atomicAdd((unsigned long long *)&buf[(a & ~1)], (unsigned long long)lane);
            else if (MODE == 3) { buf[a] += v; }
            else if (MODE == 4) { float r = atomicAdd(&buf[a], v); v += r * 1e-30f; }
        }
        base = (base + 7) % 4000;
    }
    __syncthreads();
    float s = 0; for (int i = threadIdx.x; i < 12288; i += 256) s += buf[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + v;
}

template <int MODE, int STRIDE, int DUP>
void run(const char *name) {
    float *out; CHK(hipMalloc(&out, 4096 * 256 * 4));
    const int blocks = 256 * 3, iters = 2000;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, STRIDE, DUP>), dim3(blocks), dim3(256), 0, 0, out, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a)); hipLaunchKernelGGL((k<MODE, STRIDE, DUP>), dim3(blocks), dim3(256), 0, 0, out, iters); CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    double winstr = (double)blocks * 4 * iters * 8;
    double per_cu_per_cycle = winstr / 256.0 / (ms * 1e-3 * 2.4e9);
    printf("%-28s %8.3f ms  %.4f wave-instr/cycle/CU  = %.1f cycles per wave-instr per CU\n", name, ms, per_cu_per_cycle, 1.0 / per_cu_per_cycle);
    CHK(hipFree(out));
}

int main() {
    run<0, 1, 1>("ds_add_f32 stride1");
    run<0, 3, 1>("ds_add_f32 stride3");
    run<0, 1, 2>("ds_add_f32 stride1 dup2");
    run<0, 1, 4>("ds_add_f32 stride1 dup4");
    run<4, 1, 1>("ds_add_rtn_f32 stride1");
    run<1, 1, 1>("ds_add_u32 stride1");
    run<1, 3, 1>("ds_add_u32 stride3");
    run<1, 1, 2>("ds_add_u32 stride1 dup2");
    run<2, 2, 1>("ds_add_u64 stride2");
    run<3, 1, 1>("plain rmw stride1");
    return 0;
}
