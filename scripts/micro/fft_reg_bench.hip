// In-register transforms of ppm_fft_reg.h: results against a direct sum on the host, and the issue rate of a fully unrolled
// 64-point transform at one and at two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -I../../pyp_amd/csrc -o fft_reg_bench fft_reg_bench.hip && ./fft_reg_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <complex>
#include "ppm_fft_reg.h"
using namespace ppm::fr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int N>
__global__ void __launch_bounds__(256) k_fft(const float2 *in, float2 *out, const float2 *tw, int reps) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    v2f x[N];
#pragma unroll
    for (int i = 0; i < N; i++) { const float2 v = in[(size_t)t * N + i]; x[i] = (v2f){ v.x, v.y }; }
    for (int r = 0; r < reps; r++) {
        fft_inreg<N, 64>(x, (TwPtr)tw);
        if (r + 1 < reps) {
#pragma unroll
            for (int i = 0; i < N; i++) x[i] *= 0.125f;          // keep the numbers finite over the repetitions
        }
    }
    static_for<0, N>([&](auto mc) { constexpr int m = decltype(mc)::value; constexpr int p = pos_of(N, m); const v2f v = x[p]; out[(size_t)t * N + m] = make_float2(v.x, v.y); });
}

__global__ void k_helpers(const float2 *in, float2 *out) {
    const int t = threadIdx.x;
    const v2f a = { in[2 * t].x, in[2 * t].y }, b = { in[2 * t + 1].x, in[2 * t + 1].y }, acc = { 0.25f, -0.5f };
    v2f r[10];
    r[0] = add_i(a, b); r[1] = sub_i(a, b); r[2] = add_conj(a, b); r[3] = sub_conj(a, b); r[4] = conj_add_i_conj(a, b);
    r[5] = cmul_v(a, b); r[6] = cmac_v(acc, a, b); r[7] = cmac_conj_v(acc, a, b); r[8] = cmsub_v(acc, a, b); r[9] = cmsub_conj_v(acc, a, b);
    for (int i = 0; i < 10; i++) out[10 * t + i] = make_float2(r[i].x, r[i].y);
}

template <int N> static int check(const float2 *d_tw) {
    const int T = 256;
    std::vector<float2> h((size_t)T * N), o((size_t)T * N);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float2((float)std::sin(0.37 * i + 1.0), (float)std::cos(0.11 * i * i));
    float2 *d_in, *d_out;
    CK(hipMalloc(&d_in, h.size() * 8)); CK(hipMalloc(&d_out, h.size() * 8));
    CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fft<N>, dim3(T / 64), dim3(64), 0, 0, d_in, d_out, d_tw, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < T; t++) for (int m = 0; m < N; m++) {
        std::complex<double> s = 0;
        for (int n = 0; n < N; n++) s += std::complex<double>(h[(size_t)t * N + n].x, h[(size_t)t * N + n].y) * std::polar(1.0, 2.0 * M_PI * n * m / N);
        worst = std::max(worst, std::abs(s - std::complex<double>(o[(size_t)t * N + m].x, o[(size_t)t * N + m].y)));
    }
    printf("N = %2d: max |difference to the direct sum| = %.3g\n", N, worst);
    (void)hipFree(d_in); (void)hipFree(d_out);
    return worst < 1e-4 * N ? 0 : 1;
}

int main() {
    std::vector<float2> tw(64);
    for (int t = 0; t < 64; t++) tw[t] = make_float2((float)std::cos(2.0 * M_PI * t / 64), (float)std::sin(2.0 * M_PI * t / 64));
    float2 *d_tw; CK(hipMalloc(&d_tw, 64 * 8)); CK(hipMemcpy(d_tw, tw.data(), 64 * 8, hipMemcpyHostToDevice));
    int bad = 0;
    {   // helpers
        std::vector<float2> h(128), o(640);
        for (int i = 0; i < 128; i++) h[i] = make_float2((float)std::sin(0.7 * i + 0.3), (float)std::cos(1.3 * i));
        float2 *d_in, *d_out; CK(hipMalloc(&d_in, 128 * 8)); CK(hipMalloc(&d_out, 640 * 8));
        CK(hipMemcpy(d_in, h.data(), 128 * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_helpers, dim3(1), dim3(64), 0, 0, d_in, d_out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o.data(), d_out, 640 * 8, hipMemcpyDeviceToHost));
        typedef std::complex<double> C;
        const C I(0, 1), acc(0.25, -0.5);
        double worst[10] = { 0 };
        for (int t = 0; t < 64; t++) {
            const C a(h[2 * t].x, h[2 * t].y), b(h[2 * t + 1].x, h[2 * t + 1].y);
            const C want[10] = { a + I * b, a - I * b, a + std::conj(b), a - std::conj(b), std::conj(a) + I * std::conj(b), a * b, acc + a * b, acc + a * std::conj(b), acc - a * b, acc - a * std::conj(b) };
            for (int i = 0; i < 10; i++) worst[i] = std::max(worst[i], std::abs(want[i] - C(o[10 * t + i].x, o[10 * t + i].y)));
        }
        const char *names[10] = { "add_i", "sub_i", "add_conj", "sub_conj", "conj_add_i_conj", "cmul_v", "cmac_v", "cmac_conj_v", "cmsub_v", "cmsub_conj_v" };
        for (int i = 0; i < 10; i++) { printf("%-16s max error %.3g\n", names[i], worst[i]); if (worst[i] > 1e-5) bad++; }
    }
    bad += check<8>(d_tw); bad += check<16>(d_tw); bad += check<32>(d_tw); bad += check<64>(d_tw);
    // issue rate: 64-point transforms, 256-thread blocks, B blocks per CU (B = 1: one wave per SIMD)
    for (int bpc = 1; bpc <= 2; bpc++) {
        const int T = 256 * 256 * bpc, reps = 400;
        float2 *d_in, *d_out;
        CK(hipMalloc(&d_in, (size_t)T * 64 * 8)); CK(hipMalloc(&d_out, (size_t)T * 64 * 8));
        CK(hipMemset(d_in, 0, (size_t)T * 64 * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_fft<64>, dim3(T / 256), dim3(256), 0, 0, d_in, d_out, d_tw, reps);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_fft<64>, dim3(T / 256), dim3(256), 0, 0, d_in, d_out, d_tw, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_wave_us = ms * 1e3 / reps / bpc;        // one SIMD runs `bpc` waves one after the other or interleaved
        printf("%d wave(s) per SIMD: %.3f ms for %d transforms per thread -> %.3f us per 64-point transform and wave (%.0f cycles at 2.0 GHz)\n",
               bpc, ms, reps, per_wave_us, per_wave_us * 2000.0);
        (void)hipFree(d_in); (void)hipFree(d_out);
    }
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
