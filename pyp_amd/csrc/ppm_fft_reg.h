// In-register complex transforms for the full-window correlation kernel (ppm_gfft.h).
//
// A thread holds a whole line of N <= 64 complex values in VGPR pairs (re, im) and transforms it without touching LDS: every
// butterfly step is a packed two-lane instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), so one instruction does the real
// and the imaginary part.  A multiplication by +-i and the cross terms of a complex product are SOURCE MODIFIERS of those
// instructions (op_sel swaps the halves of a 64-bit operand, neg_lo / neg_hi negate one of them).
//
// Why whole butterflies are single asm statements: (1) hipcc reaches op_sel from a two-element shuffle but not a negated lane;
// (2) ROCm 7.2's hazard recogniser takes op_sel_hi[0] of a packed instruction — set in the DEFAULT encoding of every v_pk_* — for
// "writes the high half of its destination" and puts an s_nop in front of any instruction that reads the result directly
// afterwards (scripts/micro/pk_dep_bench: a dependent v_pk_fma_f32 issues exactly as fast as an independent one, so the nop
// buys nothing), and it assumes the same of every value an asm statement defines.  At one wave per SIMD every instruction,
// s_nop included, costs a 4-cycle issue slot (2.35 ns per v_pk_fma_f32, dependent or not), so the kernel wants straight runs of
// packed instructions: a statement holds a complete radix-4 butterfly with its three twiddle products (14 instructions), two rows
// of the correlation's products (8), or four twiddle products (8); statements of one pass are independent of their neighbours.
// Twiddles are wave-uniform (cos, sin) pairs read from tables in LDS (all lanes one address: a broadcast read) two statements ahead.
//
// Sign convention: all transforms here are X[m] = sum_n x[n] e^{+2 pi i n m / N} (the correlation's inverse transforms).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace ppm {
namespace fr {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

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
// compile-time loop: f(std::integral_constant<int, I>) for I = I0 .. N - 1 (register arrays need constant indices)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// ---- twiddle tables (built on the host, copied into LDS by the kernel, read by broadcast ds_reads)
// Twiddles are (cos, sin) pairs in VGPRs: the scalar path was tried first (s_load_dwordx16 into SGPR operands) and lost — scalar loads
// return out of order, so the only wait is lgkmcnt(0), and every table entry cost a full exposed scalar-cache round trip (36 % of the
// kernel's wave cycles sat in s_waitcnt, profiles/r05_pmc_refine0_a.json).  LDS reads return in order and can be counted, so the
// entries travel two butterflies ahead of their use.
// Butterfly table of an N-point transform: for every radix-4 pass with block size M = N, N/4, ... >= 8 and every k = 1 .. M/4 - 1 one
// 8-float entry (w^k, w^2k, w^3k, padding), w = e^{2 pi i / M}.  bfly_entry(N, M, k) is its index.
constexpr int bfly_entries(int N) { int e = 0; for (int M = N; M >= 8; M /= 4) e += M / 4 - 1; return e; }
constexpr int bfly_entry(int N, int M, int k) { int e = 0; for (int m = N; m > M; m /= 4) e += m / 4 - 1; return e + k - 1; }
// Line table of a 2 L-point grid: w^n, n = 0 .. L - 1, w = e^{2 pi i / (2 L)} (the decimation twiddles of the column pass and the
// half-length trick of the row pass).
constexpr int tw_table_floats(int L) { return bfly_entries(L) * 8 + L * 2; }

// The radix-4 passes of an N-point transform, breadth first: butterfly i works on x[pos[i][0..3]]; entry[i] is its table entry (-1: k = 0,
// no twiddles) and tidx[i] its number among the twiddled ones, whose entries in order are seq[0 .. NT).
template <int N> struct FftPlan {
    static constexpr int NS4 = ilog2(N) / 2, NB = NS4 * (N / 4);
    int pos[NB > 0 ? NB : 1][4], entry[NB > 0 ? NB : 1], tidx[NB > 0 ? NB : 1], seq[NB > 0 ? NB : 1], NT;
    constexpr FftPlan() : pos{}, entry{}, tidx{}, seq{}, NT(0) {
        int i = 0;
        for (int M = N; M >= 4; M /= 4) {
            const int q = M / 4;
            for (int base = 0; base < N; base += M)
                for (int k = 0; k < q; k++, i++) {
                    for (int j = 0; j < 4; j++) pos[i][j] = base + k + j * q;
                    entry[i] = k > 0 ? bfly_entry(N, M, k) : -1;
                    tidx[i] = -1;
                    if (k > 0) { tidx[i] = NT; seq[NT++] = entry[i]; }
                }
        }
    }
};

// ---- one radix-4 decimation-in-frequency butterfly, in place; with TW the outputs 1 .. 3 are multiplied by w1, w2, w3 = (cos, sin)
template <bool TW>
__device__ __forceinline__ void bfly4(v2f &x0, v2f &x1, v2f &x2, v2f &x3, v2f w1, v2f w2, v2f w3) {
    v2f a = x0, b = x1, c = x2, d = x3, t0, t1;
    if constexpr (TW) {
        asm("v_pk_add_f32 %4, %0, %2\n\t"
            "v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %2, %1, %3\n\t"
            "v_pk_add_f32 %1, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %4, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %4, %4, %2\n\t"
            "v_pk_add_f32 %2, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            "v_pk_mul_f32 %5, %2, %6 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %2, %2, %6, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
            "v_pk_mul_f32 %5, %3, %7 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %3, %3, %7, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
            "v_pk_mul_f32 %5, %0, %8 op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %0, %8, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1)
            : "v"(w1), "v"(w2), "v"(w3));
    } else {
        asm("v_pk_add_f32 %4, %0, %2\n\t"
            "v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %2, %1, %3\n\t"
            "v_pk_add_f32 %1, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %4, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %4, %4, %2\n\t"
            "v_pk_add_f32 %2, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0) : );
    }
    x0 = t0; x1 = c; x2 = d; x3 = a;          // y0 = s0 + s2, y1 = s1 + i s3, y2 = s0 - s2, y3 = s1 - i s3 (renamings, no moves)
}
// two radix-2 butterflies in one statement (the last pass of 8- and 32-point transforms)
__device__ __forceinline__ void bfly2x2(v2f &x0, v2f &x1, v2f &x2, v2f &x3) {
    v2f a = x0, b = x1, c = x2, d = x3, t0, t1;
    asm("v_pk_add_f32 %4, %0, %1\n\t"
        "v_pk_add_f32 %5, %2, %3\n\t"
        "v_pk_add_f32 %1, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %2, %3 neg_lo:[0,1] neg_hi:[0,1]"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1) : );
    x0 = t0; x1 = b; x2 = t1; x3 = d;
}

// X[m] = sum_n x[n] e^{+2 pi i n m / N}, in place; afterwards x[pos_of(N, m)] holds X[m].  tb: the butterfly table of N in LDS.
template <int N>
__device__ __forceinline__ void fft_inreg(v2f (&x)[N], const float *tb) {
    constexpr FftPlan<N> P{};
    constexpr int NB = FftPlan<N>::NB, NT = P.NT, DQ = 3;
    v4f wa[DQ]; v2f wb[DQ];                         // (w1, w2) and w3 of the twiddled butterflies in flight
    auto fetch = [&](auto tc) { constexpr int t = decltype(tc)::value; wa[t % DQ] = *(const v4f *)(tb + P.seq[t] * 8); wb[t % DQ] = *(const v2f *)(tb + P.seq[t] * 8 + 4); };
    if constexpr (NT > 0) fetch(std::integral_constant<int, 0>{});
    if constexpr (NT > 1) fetch(std::integral_constant<int, 1>{});
    static_for<0, NB>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (P.entry[i] < 0) bfly4<false>(x[P.pos[i][0]], x[P.pos[i][1]], x[P.pos[i][2]], x[P.pos[i][3]], v2f{}, v2f{}, v2f{});
        else {
            constexpr int t = P.tidx[i];
            const v4f w12 = wa[t % DQ]; const v2f w3 = wb[t % DQ];
            if constexpr (t + 2 < NT) fetch(std::integral_constant<int, t + 2>{});
            __builtin_amdgcn_sched_barrier(0);
            bfly4<true>(x[P.pos[i][0]], x[P.pos[i][1]], x[P.pos[i][2]], x[P.pos[i][3]], (v2f){ w12.x, w12.y }, (v2f){ w12.z, w12.w }, w3);
        }
    });
    if constexpr (ilog2(N) % 2 == 1) static_for<0, N / 4>([&](auto ic) { constexpr int i = 4 * decltype(ic)::value; bfly2x2(x[i], x[i + 1], x[i + 2], x[i + 3]); });
}

// ---- four products with consecutive entries of the line table: x_i *= w_i
__device__ __forceinline__ void cmul4(v2f &x0, v2f &x1, v2f &x2, v2f &x3, v2f w0, v2f w1, v2f w2, v2f w3) {
    v2f a = x0, b = x1, c = x2, d = x3, t0, t1;
    asm("v_pk_mul_f32 %4, %0, %6 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %5, %1, %7 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %0, %6, %4 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %1, %7, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_mul_f32 %4, %2, %8 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %5, %3, %9 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %2, %2, %8, %4 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %3, %9, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1)
        : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
    x0 = a; x1 = b; x2 = c; x3 = d;
}

// ---- the product chains of the correlation's column pass, two rows per statement:
// d_r = wa_r p~a_r +- wb_r p~b_r with p~ = p (E = 1) or conj(p) (E = 0) and the sign - for H = 1; all operands per-lane VGPR pairs.
#define PPM_PROD2X2(N1, N2, N3) \
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[1,0]\n\t" \
        "v_pk_mul_f32 %1, %6, %7 op_sel_hi:[1,0]\n\t" \
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] " N1 "\n\t" \
        "v_pk_fma_f32 %1, %6, %7, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] " N1 "\n\t" \
        "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1] " N2 "\n\t" \
        "v_pk_fma_f32 %1, %8, %9, %1 op_sel_hi:[1,0,1] " N2 "\n\t" \
        "v_pk_fma_f32 %0, %4, %5, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] " N3 "\n\t" \
        "v_pk_fma_f32 %1, %8, %9, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] " N3 \
        : "=&v"(d0), "=&v"(d1) : "v"(wa0), "v"(pa0), "v"(wb0), "v"(pb0), "v"(wa1), "v"(pa1), "v"(wb1), "v"(pb1))
template <int E, int H>
__device__ __forceinline__ void prod2x2(v2f &d0, v2f &d1, v2f wa0, v2f pa0, v2f wb0, v2f pb0, v2f wa1, v2f pa1, v2f wb1, v2f pb1) {
    // cross term of w p: (-w.y p.y, w.x p.y) -> neg_lo on the swapped w; of w conj(p): (w.y p.y, -w.x p.y) -> neg_hi
    if constexpr (E == 1 && H == 0) PPM_PROD2X2("neg_lo:[1,0,0]", "", "neg_lo:[1,0,0]");
    else if constexpr (E == 0 && H == 0) PPM_PROD2X2("neg_hi:[1,0,0]", "", "neg_hi:[1,0,0]");
    else if constexpr (E == 1 && H == 1) PPM_PROD2X2("neg_lo:[1,0,0]", "neg_lo:[1,0,0] neg_hi:[1,0,0]", "neg_hi:[1,0,0]");
    else PPM_PROD2X2("neg_hi:[1,0,0]", "neg_lo:[1,0,0] neg_hi:[1,0,0]", "neg_lo:[1,0,0]");
}
#undef PPM_PROD2X2

// ---- the half-length trick of a real 2 L-point transform (row pass): with s = X[k] + conj X[L-k], t = w^k (X[k] - conj X[L-k]),
// Z[k] = s + i t and Z[L-k] = conj(s) + i conj(t).  Two pairs per statement; w = (cos, sin) of w^k.
__device__ __forceinline__ void halfpair2(v2f &xk0, v2f &xl0, v2f &xk1, v2f &xl1, v2f w0, v2f w1) {
    v2f a = xk0, b = xl0, c = xk1, d = xl1, t0, t1;
    asm("v_pk_add_f32 %4, %0, %1 neg_lo:[0,1]\n\t"                               // d = a - conj(b)
        "v_pk_add_f32 %5, %2, %3 neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1 neg_hi:[0,1]\n\t"                               // s = a + conj(b)
        "v_pk_add_f32 %2, %2, %3 neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %4, %6 op_sel_hi:[1,0]\n\t"                            // t = d w
        "v_pk_mul_f32 %3, %5, %7 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %4, %6, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %5, %7, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_add_f32 %4, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"  // Z[k] = s + i t
        "v_pk_add_f32 %5, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %1, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"  // Z[L-k] = (s.x + t.y, t.x - s.y)
        "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1)
        : "v"(w0), "v"(w1));
    xk0 = t0; xl0 = b; xk1 = t1; xl1 = d;
}
// one pair (the odd one out: k = 1)
__device__ __forceinline__ void halfpair1(v2f &xk, v2f &xl, v2f w) {
    v2f a = xk, b = xl, t0;
    asm("v_pk_add_f32 %2, %0, %1 neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %0, %0, %1 neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
        "v_pk_add_f32 %2, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %1, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]"
        : "+v"(a), "+v"(b), "=&v"(t0) : "v"(w));
    xk = t0; xl = b;
}

}  // namespace fr
}  // namespace ppm
