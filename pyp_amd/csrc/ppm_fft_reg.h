// In-register complex transforms for the full-window correlation kernel (ppm_gfft.h).
//
// A thread holds a whole line of N <= 64 complex values in VGPR pairs (re, im) and transforms it without touching LDS: every
// butterfly is a packed two-lane instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), so one instruction does the real and
// the imaginary part.  What the compiler does not find by itself is that a multiplication by +-i and the cross terms of a
// complex product are SOURCE MODIFIERS of those instructions (op_sel swaps the halves of a 64-bit operand, neg_lo / neg_hi negate
// one of them): written in C it emits a v_xor + v_mov pair per rotation.  The four helpers below pin the modifier forms.
// Twiddles are wave-uniform: they come out of a table e^{2 pi i t / TWN} in device memory through the constant address space
// (scalar loads into SGPR pairs) and enter the packed instructions as their one scalar operand.
//
// Sign convention: all transforms here are X[m] = sum_n x[n] e^{+2 pi i n m / N} (the correlation's inverse transforms).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace ppm {
namespace fr {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v2f *TwPtr;

// a + i b  = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f add_i(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a - i b  = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2f sub_i(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ v2f add_conj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f sub_conj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// conj(a) + i conj(b) = (a.x + b.y, b.x - a.y)
__device__ __forceinline__ v2f conj_add_i_conj(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * w, w = (cos, sin) in an SGPR pair
__device__ __forceinline__ v2f cmul_s(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d), "=&v"(t) : "v"(a), "s"(w));
    return d;
}
// the same with the twiddle in a VGPR pair (per-lane twiddles)
__device__ __forceinline__ v2f cmul_v(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d), "=&v"(t) : "v"(a), "v"(w));
    return d;
}
// acc + a * w and acc + a * conj(w), all three in VGPR pairs (the products W P and W conj(P) of the correlation)
__device__ __forceinline__ v2f cmac_v(v2f acc, v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d), "=&v"(t) : "v"(a), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ v2f cmac_conj_v(v2f acc, v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d), "=&v"(t) : "v"(a), "v"(w), "v"(acc));
    return d;
}
// acc - a * w, acc - a * conj(w)
__device__ __forceinline__ v2f cmsub_v(v2f acc, v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\tv_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(d), "=&v"(t) : "v"(a), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ v2f cmsub_conj_v(v2f acc, v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_fma_f32 %1, %2, %3, %4 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\tv_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(d), "=&v"(t) : "v"(a), "v"(w), "v"(acc));
    return d;
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n / 2); }

// Where frequency m of an N-point transform ends up after the decimation-in-frequency passes below (radix 4 while the block
// has at least four points, one radix-2 pass for what is left): pos_of(N, m), and its inverse freq_at(N, p).
constexpr int freq_at(int M, int p) {
    if (M == 1) return 0;
    if (M == 2) return p;
    const int q = M / 4;
    return 4 * freq_at(q, p % q) + p / q;
}
constexpr int pos_of(int M, int m) {
    for (int p = 0; p < M; p++) if (freq_at(M, p) == m) return p;
    return -1;
}
// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N - 1 (register arrays need constant indices)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// In-place transform of x[BASE .. BASE + M) of an N-element register array; tw = e^{2 pi i t / TWN}, TWN a multiple of M.
template <int N, int M, int BASE, int TWN>
__device__ __forceinline__ void fft_dif(v2f (&x)[N], TwPtr tw) {
    if constexpr (M == 2) {
        const v2f a = x[BASE], b = x[BASE + 1];
        x[BASE] = a + b; x[BASE + 1] = a - b;
    } else if constexpr (M >= 4) {
        constexpr int q = M / 4, ts = TWN / M;
#pragma unroll
        for (int k = 0; k < q; k++) {
            const v2f a = x[BASE + k], b = x[BASE + k + q], c = x[BASE + k + 2 * q], d = x[BASE + k + 3 * q];
            const v2f s0 = a + c, s1 = a - c, s2 = b + d, s3 = b - d;
            const v2f y0 = s0 + s2, y2 = s0 - s2, y1 = add_i(s1, s3), y3 = sub_i(s1, s3);
            x[BASE + k] = y0;
            if (k == 0) { x[BASE + q] = y1; x[BASE + 2 * q] = y2; x[BASE + 3 * q] = y3; }
            else {
                x[BASE + k + q] = cmul_s(y1, tw[k * ts]);
                x[BASE + k + 2 * q] = cmul_s(y2, tw[2 * k * ts]);
                x[BASE + k + 3 * q] = cmul_s(y3, tw[3 * k * ts]);
            }
        }
        fft_dif<N, q, BASE, TWN>(x, tw);
        fft_dif<N, q, BASE + q, TWN>(x, tw);
        fft_dif<N, q, BASE + 2 * q, TWN>(x, tw);
        fft_dif<N, q, BASE + 3 * q, TWN>(x, tw);
    }
}

// X[m] = sum_n x[n] e^{+2 pi i n m / N}; afterwards x[pos_of(N, m)] holds X[m].
template <int N, int TWN>
__device__ __forceinline__ void fft_inreg(v2f (&x)[N], TwPtr tw) { fft_dif<N, N, 0, TWN>(x, tw); }

}  // namespace fr
}  // namespace ppm
