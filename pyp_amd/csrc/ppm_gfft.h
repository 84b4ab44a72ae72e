// ppm_gfft.h — global search over WIDE shift windows: the whole correlation image of every (particle, orientation) by a pruned
// two-dimensional transform in registers and LDS (gfx950 only).
//
// PYP's default call sends "search range X / Y" = 0 = the mask radius (frealign.py:3954-3957, config/pyp_config.toml:5338-5350):
// at a 256 box and a 4 A search band that is +-41 steps of the 128-point search grid — 83 x 83 shifts per orientation, where
// k_global's register-held window of <= 17 x 17 needs 25 passes over the slice bank.  Here a block owns one particle and, per
// stored slice (which serves psi and psi + 180 deg), computes
//
//     c(sx, sy) = Re sum_{kx >= 0, ky} W(k) conj(P(k)) e^{+2 pi i (kx sx + ky sy) / Ns},    Ns = 2 L,
//
// for all shifts of the window as a zero-filled Ns x Ns inverse transform would (the form the CPU checker states it in), in two
// passes that each keep a whole line in the registers of one thread (ppm_fft_reg.h):
//
//  * COLUMN pass, lane = kx: the transform over ky.  A thread forms x[n] = W conj(P) (or W P for psi + 180) for ky = n and
//    ky = n - L straight from the bank row (one 16-byte load: the bank stores the two rows side by side) and the particle's
//    W table in LDS (one 16-byte read), takes the first decimation-in-frequency step on the way — y[n] = x[n] + x[n + L] gives
//    the even output rows, (x[n] - x[n + L]) w^n the odd ones — and runs an L-point transform on its 64 registers.  The four
//    waves of a block are the four (orientation, row parity) combinations, so the sign pattern is wave-uniform code, not data.
//    The rows inside the window go to the LDS image T[orientation][sy][kx].
//  * ROW pass, thread = one row (orientation, sy): the real 2 L-point transform over kx through one L-point complex transform
//    (Z[k] = (X[k] + conj X[L-k]) + i w^k (X[k] - conj X[L-k]); its output is c(2n) + i c(2n+1)), then the maximum over the
//    window's columns.  Only the MAXIMUM is formed per orientation — the shift of an orientation matters only for the K best
//    of them, so after the top-K selection the K winning slices are transformed once more with the arg-max switched on
//    (K / n_orient = 0.5 % more work instead of two compare-selects per correlation value everywhere).
//
// LDS at L = 64: T 2 x (2 RSy + 1) x 66 x 8 B (86 KB at RSy = 41) + W 64 KB — one block per CU, one wave per SIMD; a single wave
// issues one instruction per four cycles, which a packed instruction fills (scripts/micro/fft_reg_bench: a 64-point transform
// costs 2 840 cycles at one wave per SIMD against 2 500 at two).  Smaller search grids put 64 / L slices into one pass.
#pragma once
#include "ppm_fft_reg.h"

namespace ppm {

struct GfftP {
    const float4 *bank4;     // [nslices][L][L]: (P(ky = n, kx), P(ky = n - L, kx)), zero outside the search band (k_bank4)
    unsigned bank4_bytes;    // its size (below 4 GB: the column pass reads it through a buffer descriptor)
    const float2 *Wp;        // [n][Hs][64] search tables of the chunk (k_prep), rows ky + Bs
    const float *nP, *nI;    // slice norms [n][nslices] (k_slice_norms), image norms [n]
    const float *tw;         // twiddle tables (ppm_fft_reg.h: tw_table_floats(L) floats — the butterfly table of the L-point transform, then
                             // the line table w^0 .. w^(L-1) of the Ns-point grid) followed by the column penalties of the row pass: pairs
                             // (0 or -3e38) for the columns sx(2 f), sx(2 f + 1), f = freq_at(L, position); copied into LDS by every block
    float *part;             // [n][n_orient][NPART] raw window maxima
    float *cc;               // [n][n_orient] scores for the top-K pass when they do not fit the LDS
    Hit *hits;               // [n][K]
    int n, Bs, Hs, RSx, RSy, n_dir, n_psi, npsi_store, n_orient, K, topk_lds;
    int RC, nchunk;          // rows of T held at a time, and how many such chunks cover the 2 RSy + 1 rows (1 unless LDS is short)
    int t_bytes;             // bytes of the T image (the small arrays follow it)
};

constexpr int gfft_row_stride(int L) { return L + 2; }       // float2 per T row: 16-byte row reads of 64 lanes hit distinct banks
constexpr int gfft_slices_per_pass(int L) { return 64 / L; }
constexpr size_t gfft_small_bytes(int L) { return 256 * 8 + PPM_MAX_TOP_HITS * 8 + 256 + (size_t)(fr::tw_table_floats(L) + 2 * L) * 4; }

// ---------------------------------------------------------------------------------- slice bank in the layout of the column pass
struct Bank4P { CubeView cv; const float *mats; float4 *bank4; int nslices, Bs, L; float r_s2; };

__global__ void __launch_bounds__(256) k_bank4(Bank4P P) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, per = (size_t)P.L * P.L;
    if (i >= per * P.nslices) return;
    const int sl = (int)(i / per), r = (int)(i - (size_t)sl * per), n = r / P.L, kx = r - n * P.L;
    const float *m = P.mats + (size_t)sl * 6;
    float2 v[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ky = h ? n - P.L : n;
        const float k2 = (float)(kx * kx + ky * ky);
        v[h] = make_float2(0.f, 0.f);
        if (kx <= P.Bs && ky >= -P.Bs && ky <= P.Bs && k2 < P.r_s2) {
            const float fx = (float)kx, fy = (float)ky;
            v[h] = sample_cube(P.cv, m[0] * fx + m[1] * fy, m[2] * fx + m[3] * fy, m[4] * fx + m[5] * fy);
        }
    }
    P.bank4[i] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
}

// A wave-uniform integer the compiler may not look through: the window tests of a pass compare against loop invariants, and hoisted
// out of the slice loop every one of them becomes a 64-bit mask parked in VGPR lanes (two v_readlane per test instead of one s_cmp)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+s"(v)); return v; }

// ---------------------------------------------------------------------------------- the column pass of one thread
// Every (orientation, parity) wave of a block needs the WHOLE bank slice for its columns.  Loaded by each wave for itself that is four
// times the slice through the CU's vector L1, whose 64 B per clock then set the pace of the pass (5 800 of a slice's 22 000 cycles,
// PPM_GFFT_STAMPS).  So the slice is STAGED: every wave fetches a quarter of its rows from the bank (a whole pass ahead, into
// registers), writes them into LDS — the region of the image T, which is idle between the row pass of one slice and the stores of
// the next — and after a barrier all four read their rows from there at the LDS's 256 B per clock.
constexpr int gfft_depth(int L) { return L / 4; }       // bank rows a wave stages per slice and lane

// bank row N of the rows a lane stages: lane offset + (N % 4) rows as the instruction's immediate + 4 (N / 4) rows in a scalar register
// (left to itself the compiler keeps one VGPR offset per row and parks them in AGPRs)
template <int L>
__device__ __forceinline__ void gfft_prefetch(fr::v4f (&pb)[gfft_depth(L)], __amdgpu_buffer_rsrc_t bank, unsigned voff) {
    constexpr int D = gfft_depth(L);
    int soff[D / 4 + 1];
    soff[0] = opaque(0);
    fr::static_for<1, D / 4 + 1>([&](auto ic) { constexpr int i = decltype(ic)::value; soff[i] = soff[i - 1] + 4 * L * 16; });
    fr::static_for<0, D>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        pb[i] = __builtin_amdgcn_raw_buffer_load_b128(bank, (int)(voff + (i % 4) * L * 16), soff[i / 4], 0);
    });
}

// y[n] = x[n] + x[n + L] (H = 0) or x[n] - x[n + L] (H = 1; the factor w^n follows in a pass of its own), x = W conj(P) (E = 0) or
// W P (E = 1).  pl / wl point at this lane's column of the staged slice / of the W table, both in LDS, rows L float4 apart; the reads
// travel DW rows ahead of their use and the scheduling barriers keep the compiler from sinking them back.
template <int L, int E, int H>
__device__ __forceinline__ void gfft_col_products(fr::v2f (&y)[L], const float4 *pl, const float4 *wl) {
    using namespace fr;
    constexpr int DW = L < 6 ? L : 6;
    float4 pb[DW], wb[DW];
    static_for<0, DW>([&](auto ic) { constexpr int i = decltype(ic)::value; pb[i] = pl[i * L]; wb[i] = wl[i * L]; });
    static_for<0, L / 2>([&](auto nc) {
        constexpr int n = 2 * decltype(nc)::value;
        const float4 p0 = pb[n % DW], p1 = pb[(n + 1) % DW], w0 = wb[n % DW], w1 = wb[(n + 1) % DW];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n + DW < L) { pb[n % DW] = pl[(n + DW) * L]; wb[n % DW] = wl[(n + DW) * L]; pb[(n + 1) % DW] = pl[(n + 1 + DW) * L]; wb[(n + 1) % DW] = wl[(n + 1 + DW) * L]; }
        prod2x2<E, H>(y[n], y[n + 1], (v2f){ w0.x, w0.y }, (v2f){ p0.x, p0.y }, (v2f){ w0.z, w0.w }, (v2f){ p0.z, p0.w },
                      (v2f){ w1.x, w1.y }, (v2f){ p1.x, p1.y }, (v2f){ w1.z, w1.w }, (v2f){ p1.z, p1.w });
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Four LDS stores of the column pass, each predicated on a bit of a wave-uniform mask by switching EXEC (a uniform test compiles to
// a branch, and the 60 odd taken branches of a pass cost more than its butterflies: scripts/.. stamps, CHANGELOG round 5).  All
// lanes are active on entry (wave-uniform control flow around the call); EXEC is all ones again on exit.
template <int B0, int B1, int B2, int B3, int O0, int O1, int O2, int O3>
__device__ __forceinline__ void lds_store4_masked(const void *addr, unsigned mask, fr::v2f v0, fr::v2f v1, fr::v2f v2, fr::v2f v3) {
    const unsigned a = (unsigned)(size_t)addr;
    asm volatile("s_bitcmp1_b32 %5, %6\n\ts_cselect_b64 exec, -1, 0\n\tds_write_b64 %4, %0 offset:%10\n\t"
                 "s_bitcmp1_b32 %5, %7\n\ts_cselect_b64 exec, -1, 0\n\tds_write_b64 %4, %1 offset:%11\n\t"
                 "s_bitcmp1_b32 %5, %8\n\ts_cselect_b64 exec, -1, 0\n\tds_write_b64 %4, %2 offset:%12\n\t"
                 "s_bitcmp1_b32 %5, %9\n\ts_cselect_b64 exec, -1, 0\n\tds_write_b64 %4, %3 offset:%13\n\t"
                 "s_mov_b64 exec, -1"
                 :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(a), "s"(mask), "n"(B0), "n"(B1), "n"(B2), "n"(B3), "n"(O0), "n"(O1), "n"(O2), "n"(O3) : "memory", "scc");
}

// v_max3_f32 as it is (fmaxf() adds a canonicalising v_max per operand); eight values per statement
__device__ __forceinline__ float max8_raw(float m, float a, float b, float c, float d, float e, float f, float g, float h) {
    asm("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %0, %0, %7, %8" : "+v"(m) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h));
    return m;
}
__device__ __forceinline__ float max3_raw(float m, float a, float b) { asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b)); return m; }

// maximum over groups of W consecutive lanes (W = 16, 32, 64), every lane gets its group's result; full EXEC required
template <int W> __device__ __forceinline__ float group_max(float v) {
    v = max_raw(v, dpp_mov<kDppXor1>(v)); v = max_raw(v, dpp_mov<kDppXor2>(v)); v = max_raw(v, dpp_mov<kDppHalfMirror>(v)); v = max_raw(v, dpp_mov<kDppMirror>(v));
    if constexpr (W >= 32) { float w = v; lane_swap<16>(v, w); v = max_raw(v, w); }
    if constexpr (W >= 64) { float w = v; lane_swap<32>(v, w); v = max_raw(v, w); }
    return v;
}

// CHUNKED: the window's rows pass through T in several chunks (the column pass then runs once per chunk and tests every output
// against the chunk); only search grids of 128 points with more than +-44 steps need it.
// Diagnostic build (-DPPM_GFFT_STAMPS): cycles per phase of the slice loop (s_memtime), summed per wave over the timed passes and
// written by blocks 0 .. 3 into the `cc` scratch (which the LDS top-K path leaves unused): [block][wave][phase] as floats.
#ifdef PPM_GFFT_STAMPS
#define GF_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define GF_STAMP(i) do { } while (0)
#endif

template <int LN, bool CHUNKED>
__global__ void __launch_bounds__(256) k_gfft(GfftP P) {
    using namespace fr;
    constexpr int Ns = 1 << LN, L = Ns / 2, G = gfft_slices_per_pass(L), TS = gfft_row_stride(L), NPART = L == 64 ? 2 : 1;
    constexpr int GW = 2 * L < 64 ? 2 * L : 64;                    // lanes of a wave that share one (slice, orientation) in the row pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *T = (float2 *)smem;                                     // [G][2][RC][TS], first: its rows are addressed base + immediate
    float4 *W4 = (float4 *)(smem + P.t_bytes);                      // [L][L]
    char *small = smem + (size_t)L * L * sizeof(float4) + P.t_bytes;
    float *red_v = (float *)small; int *red_k = (int *)(red_v + 256);      // per-thread (maximum, key) of an arg-max pass
    int *win_o = red_k + 256; float *win_c = (float *)(win_o + PPM_MAX_TOP_HITS);   // the K winning orientations and their scores
    float *rv = win_c + PPM_MAX_TOP_HITS; int *ri = (int *)(rv + 16);               // top-K scratch (4 waves)

    const int tid = threadIdx.x, lane = tid & 63, p = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Bs = P.Bs, RSy = P.RSy, RC = P.RC, NR = 2 * RSy + 1;
    const int nslices = P.n_dir * P.npsi_store;
    const bool half = P.npsi_store != P.n_psi;
    const float *twb = (const float *)(rv + 32), *twl = twb + bfly_entries(L) * 8;   // twiddle tables in LDS (16-byte aligned)
    const float *pen = twb + tw_table_floats(L);                                      // column penalties of the row pass, [L] pairs
    const __amdgpu_buffer_rsrc_t bank = __builtin_amdgcn_make_buffer_rsrc((void *)P.bank4, 0, (int)P.bank4_bytes, 0x00020000);

    // ---- the twiddle tables and the particle's W table, two ky rows side by side like the bank
    for (int i = tid; i < tw_table_floats(L) + Ns; i += 256) ((float *)twb)[i] = P.tw[i];
    {
        const float2 *src = P.Wp + (size_t)p * P.Hs * 64;
        for (int i = tid; i < L * L; i += 256) {
            const int n = i / L, kx = i - n * L;
            float2 a = make_float2(0.f, 0.f), b = a;
            if (kx <= Bs) {
                if (n <= Bs) a = src[(n + Bs) * 64 + kx];
                if (L - n <= Bs) b = src[(n - L + Bs) * 64 + kx];
            }
            W4[i] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
    __syncthreads();

    // column-pass role of this thread: wave = (orientation half e, output parity h), lane = (slice of the pass, kx)
    const int ce = wave >> 1, ch = wave & 1, cg = lane / L, ckx = lane - cg * L;
    const float4 *wl = W4 + ckx;
    float2 *Tc = T + (size_t)((cg * 2 + ce) * RC) * TS + ckx;
    // row-pass role: 4 L consecutive threads per slice, 2 L per orientation
    const int rg = tid / (4 * L), re = (tid / (2 * L)) & 1, rr = tid & (2 * L - 1);
    const float2 *Tr = T + (size_t)((rg * 2 + re) * RC + rr) * TS;

    const int npass0 = (nslices + G - 1) / G;
    int npass1 = 0;                                                  // arg-max passes over the winners, known after the top-K step
    float *partp = P.part + (size_t)p * P.n_orient * NPART;
    const float nIp = P.nI[p];
    const float *nPp = P.nP + (size_t)p * nslices;
    Hit *hitp = P.hits + (size_t)p * P.K;

    // window rows of this wave's outputs (column pass, not CHUNKED): bit f of mask_pos for f = 0 .. fmax, bit j - 1 of mask_neg for j = 1 .. jmax
    const int fmax_ = min((RSy - ch) >> 1, L / 2 - 1), jmax_ = min((RSy + ch) >> 1, L / 2);
    const unsigned mask_pos = fmax_ >= 31 ? 0xffffffffu : ((1u << (fmax_ + 1)) - 1u), mask_neg = jmax_ >= 32 ? 0xffffffffu : ((1u << jmax_) - 1u);
    // this wave's quarter of the bank rows of the slice(s) the next column pass works on (rows wave L/4 .. of every lane's slice),
    // requested during the current pass; `stage` is where the four quarters meet (LDS, the region of T)
    constexpr int DQ = gfft_depth(L);
    v4f pb[DQ];
    unsigned pf_voff = 0xffffffffu;
    auto col_voff = [&](int sl) { return (((unsigned)(sl < 0 ? 0 : sl) * (unsigned)L + (unsigned)(wave * DQ)) * (unsigned)L + (unsigned)ckx) * 16u; };
    float4 *stage = (float4 *)T;                                     // [G][L][L]
    float4 *st_w = stage + (size_t)(cg * L + wave * DQ) * L + ckx;   // where this lane puts its rows
    const float4 *st_r = stage + (size_t)cg * L * L + ckx;           // this lane's column of its slice

#ifdef PPM_GFFT_STAMPS
    unsigned long long st_acc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, st_t = __builtin_amdgcn_s_memtime();
#endif
    for (int it = 0;; it++) {
        if (it == npass0) {
#ifdef PPM_GFFT_STAMPS
            if (blockIdx.x < 4 && lane == 0) for (int i = 0; i < 12; i++) P.cc[(blockIdx.x * 4 + wave) * 12 + i] = (float)st_acc[i];
#endif
            // ---- scores of all orientations, top-K (ties -> lower orientation index, like the oracle)
            __threadfence_block();
            __syncthreads();
            float *sc = P.topk_lds ? (float *)T : P.cc + (size_t)p * P.n_orient;
            for (int o = tid; o < P.n_orient; o += 256) {
                const int dir = o / P.n_psi, k = o - dir * P.n_psi;
                const int e = (half && k >= P.npsi_store) ? 1 : 0, sl = dir * P.npsi_store + k - e * P.npsi_store;
                float v = partp[(size_t)o * NPART];
                if constexpr (NPART == 2) v = fmaxf(v, partp[(size_t)o * NPART + 1]);
                const float nP = nPp[sl];
                const float inv = (nP > 0.f && nIp > 0.f) ? rsqrtf(nP * nIp) : 0.f;
                sc[o] = 0.5f * v * inv;                              // the row transform yields twice the correlation
            }
            __threadfence_block();
            __syncthreads();
            for (int k = 0; k < P.K; k++) {
                float bv = -3.0e38f; int bi = 0x7fffffff;
                for (int o = tid; o < P.n_orient; o += 256) {
                    const float v = sc[o];
                    if (v > bv || (v == bv && o < bi)) { bv = v; bi = o; }
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    const float ov = __shfl_xor(bv, m, 64); const int oi = __shfl_xor(bi, m, 64);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
                __syncthreads();
                if (tid == 0) {
                    for (int w = 1; w < 4; w++) if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
                    if (bi >= P.n_orient) { bi = 0; bv = 0.f; }      // nothing comparable left (NaN scores): stay inside the tables
                    win_o[k] = bi; win_c[k] = bv;
                    sc[bi] = -__builtin_inff();
                }
                __threadfence_block();
                __syncthreads();
            }
            npass1 = (P.K + G - 1) / G;
        }
        if (it >= npass0 && it - npass0 >= npass1) break;
        const bool argpass = it >= npass0;

        // ---- which slice each role works on in this pass
        auto slice_of = [&](int g, int &e_hit, int &hit) {
            if (!argpass) { e_hit = -1; hit = -1; const int s = it * G + g; return s < nslices ? s : -1; }
            hit = (it - npass0) * G + g;
            if (hit >= P.K) { e_hit = -1; hit = -1; return -1; }
            const int o = win_o[hit], dir = o / P.n_psi, k = o - dir * P.n_psi;
            e_hit = (half && k >= P.npsi_store) ? 1 : 0;
            return dir * P.npsi_store + k - e_hit * P.npsi_store;
        };
        int c_eh, c_hit, r_eh, r_hit;
        const int c_sl = slice_of(cg, c_eh, c_hit), r_sl = slice_of(rg, r_eh, r_hit);

        for (int c = 0; c < P.nchunk; c++) {
            const int c0 = c * RC;                                   // first T row (slot) of this chunk
            // ================= column pass
            const bool col_on = half || ce == 0;
            v2f y[L];
            {   // the slice, staged: this wave's rows go to LDS (T is idle: the row pass of the last slice has finished)
                const unsigned voff = col_voff(c_sl);
                if (pf_voff != voff) gfft_prefetch<L>(pb, bank, voff);          // first pass, after the top-K step, or a lane whose slice changed
                static_for<0, DQ>([&](auto ic) { constexpr int i = decltype(ic)::value; st_w[i * L] = make_float4(pb[i].x, pb[i].y, pb[i].z, pb[i].w); });
            }
            lds_barrier();
            GF_STAMP(7);
            if (col_on) {
                if (ch == 0) { if (ce == 0) gfft_col_products<L, 0, 0>(y, st_r, wl); else gfft_col_products<L, 1, 0>(y, st_r, wl); }
                else { if (ce == 0) gfft_col_products<L, 0, 1>(y, st_r, wl); else gfft_col_products<L, 1, 1>(y, st_r, wl); }
            }
            GF_STAMP(0);
            lds_barrier();                                           // every wave has read the staged slice: T may be written again
            {   // the next column pass: the same slice again (next row chunk) or the next slices of the grid
                const int nsl = (c + 1 < P.nchunk) ? c_sl : (it + 1 < npass0 ? min((it + 1) * G + cg, nslices - 1) : c_sl);
                pf_voff = col_voff(nsl);
                gfft_prefetch<L>(pb, bank, pf_voff);
            }
            if (col_on) {
                if (ch != 0) {
                    // the decimation twiddles w^n of the odd rows; table entries two statements ahead
                    v4f wq[3][2];
                    auto fetch = [&](auto jc) { constexpr int j = decltype(jc)::value; wq[j % 3][0] = *(const v4f *)(twl + 8 * j); wq[j % 3][1] = *(const v4f *)(twl + 8 * j + 4); };
                    fetch(std::integral_constant<int, 0>{});
                    if constexpr (L / 4 > 1) fetch(std::integral_constant<int, 1>{});
                    static_for<0, L / 4>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const v4f wa = wq[j % 3][0], wb = wq[j % 3][1];
                        if constexpr (j + 2 < L / 4) fetch(std::integral_constant<int, j + 2>{});
                        __builtin_amdgcn_sched_barrier(0);
                        cmul4(y[4 * j], y[4 * j + 1], y[4 * j + 2], y[4 * j + 3], (v2f){ wa.x, wa.y }, (v2f){ wa.z, wa.w }, (v2f){ wb.x, wb.y }, (v2f){ wb.z, wb.w });
                    });
                }
                GF_STAMP(6);
                fft_inreg<L>(y, twb);
                GF_STAMP(1);
                // output f of the L-point transform is row sy = 2 f + h (mod Ns) of the image: rows 0 .. RSy sit in slots 0 .. RSy,
                // rows -1 .. -RSy in slots RSy + 1 .. 2 RSy
                if constexpr (!CHUNKED) {
                    // output f < L / 2 is inside the window when f <= fmax (row 2 f + h <= RSy), output f = L - j >= L / 2 when j <= jmax
                    // (row -(2 j - h) >= -RSy): bits of two wave-uniform masks that predicate the stores
                    const float2 *Tpos = Tc + ch * TS, *Tneg = Tc + (RSy - ch) * TS;
                    static_for<0, L / 8>([&](auto gc) {
                        constexpr int f0 = 4 * decltype(gc)::value, q0 = pos_of(L, f0), q1 = pos_of(L, f0 + 1), q2 = pos_of(L, f0 + 2), q3 = pos_of(L, f0 + 3);
                        lds_store4_masked<f0, f0 + 1, f0 + 2, f0 + 3, 2 * f0 * TS * 8, 2 * (f0 + 1) * TS * 8, 2 * (f0 + 2) * TS * 8, 2 * (f0 + 3) * TS * 8>(
                            Tpos, mask_pos, y[q0], y[q1], y[q2], y[q3]);
                    });
                    static_for<0, L / 8>([&](auto gc) {
                        constexpr int j0 = 4 * decltype(gc)::value + 1, q0 = pos_of(L, L - j0), q1 = pos_of(L, L - j0 - 1), q2 = pos_of(L, L - j0 - 2), q3 = pos_of(L, L - j0 - 3);
                        lds_store4_masked<j0 - 1, j0, j0 + 1, j0 + 2, 2 * j0 * TS * 8, 2 * (j0 + 1) * TS * 8, 2 * (j0 + 2) * TS * 8, 2 * (j0 + 3) * TS * 8>(
                            Tneg, mask_neg, y[q0], y[q1], y[q2], y[q3]);
                    });
                } else {
                    const int spos = ch - c0, sneg = RSy - ch - c0;
                    static_for<0, L>([&](auto pc) {
                        constexpr int pp = decltype(pc)::value, f = freq_at(L, pp);
                        if constexpr (2 * f < L) {
                            const int slot = 2 * f + spos;
                            if (2 * f + ch <= RSy && slot >= 0 && slot < RC) Tc[slot * TS] = make_float2(y[pp].x, y[pp].y);
                        } else {
                            const int slot = 2 * (L - f) + sneg;
                            if (2 * (L - f) - ch <= RSy && slot >= 0 && slot < RC) Tc[slot * TS] = make_float2(y[pp].x, y[pp].y);
                        }
                    });
                }
            }
            GF_STAMP(2);
            lds_barrier();
            GF_STAMP(3);
            // ================= row pass
            const int slot = rr + c0;
            const bool active = rr < RC && slot < NR && r_sl >= 0 && (half || re == 0) && (!argpass || re == r_eh);
            float best = -3.0e38f; int bkey = 0x7fffffff;
            if (active) {
                v2f z[L];
                // the row, 16 bytes per read.  (The empty asm statements keep the reads whole: left alone the compiler drops the unused
                // imaginary part of X[0], re-pairs the rest into ds_read2_b64 — and those hit the row stride's 2-way bank conflict.)
                {
                    v4f zr[L / 2];
                    static_for<0, L / 2>([&](auto ic) { constexpr int i = decltype(ic)::value; zr[i] = *(const v4f *)(Tr + 2 * i); });
                    static_for<0, L / 2>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        asm volatile("" : "+v"(zr[i]));
                        z[2 * i] = (v2f){ zr[i].x, zr[i].y }; z[2 * i + 1] = (v2f){ zr[i].z, zr[i].w };
                    });
                }
                {
                    // the real Ns-point transform of the row through one L-point complex transform, in place: pairs (k, L - k), k = 1 alone,
                    // then (2, 3), (4, 5), ...; their twiddles w^k four statements ahead.
                    // Z[k] = (X[k] + conj X[L-k]) + i w^k (X[k] - conj X[L-k]), Z[L-k] = conj(s) + i conj(t); Z[0] = 2 Re X[0] (1 + i), Z[L/2] = 2 conj X[L/2]
                    constexpr int DQ = 5, NP = L / 4;
                    v4f wq[DQ];
                    auto fetch = [&](auto qc) { constexpr int q = decltype(qc)::value; wq[q % DQ] = *(const v4f *)(twl + 4 * q); };      // w^(2q), w^(2q+1)
                    static_for<0, (NP < DQ - 1 ? NP : DQ - 1)>(fetch);
                    static_for<0, NP>([&](auto qc) {
                        constexpr int q = decltype(qc)::value, k = 2 * q;
                        const v4f w = wq[q % DQ];
                        if constexpr (q + DQ - 1 < NP) fetch(std::integral_constant<int, q + DQ - 1>{});
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (k == 0) halfpair1(z[1], z[L - 1], (v2f){ w.z, w.w });
                        else halfpair2(z[k], z[L - k], z[k + 1], z[L - k - 1], (v2f){ w.x, w.y }, (v2f){ w.z, w.w });
                    });
                    z[0] = (v2f){ 2.f * z[0].x, 2.f * z[0].x };
                    z[L / 2] = (v2f){ 2.f * z[L / 2].x, -2.f * z[L / 2].y };
                }
                GF_STAMP(8);
                fft_inreg<L>(z, twb);
                GF_STAMP(9);
                // z[position of f] = 2 (c(2 f), c(2 f + 1)), column j = the shift sx = j (j < L) or j - Ns.  Columns |sx| <= RSx are inside
                // the window: tested per group of four |sx|, per column only in the group the window's edge cuts
                // the window's columns: a penalty of 0 or -3e38 per column (table in LDS, in the order the transform leaves its outputs, read
                // four statements ahead), then the maximum over all of them — no test, no branch
                {
                    constexpr int DQ = 5, NG = L / 4;
                    v4f pq[DQ][2];
                    auto fetch = [&](auto gc) { constexpr int g = decltype(gc)::value; pq[g % DQ][0] = *(const v4f *)(pen + 8 * g); pq[g % DQ][1] = *(const v4f *)(pen + 8 * g + 4); };
                    static_for<0, (NG < DQ - 1 ? NG : DQ - 1)>(fetch);
                    static_for<0, NG>([&](auto gc) {
                        constexpr int g = decltype(gc)::value, p0 = 4 * g;
                        const v4f pa = pq[g % DQ][0], pb2 = pq[g % DQ][1];
                        if constexpr (g + DQ - 1 < NG) fetch(std::integral_constant<int, g + DQ - 1>{});
                        const v2f a0 = z[p0] + (v2f){ pa.x, pa.y }, a1 = z[p0 + 1] + (v2f){ pa.z, pa.w }, a2 = z[p0 + 2] + (v2f){ pb2.x, pb2.y }, a3 = z[p0 + 3] + (v2f){ pb2.z, pb2.w };
                        __builtin_amdgcn_sched_barrier(0);
                        best = max8_raw(best, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y);
                    });
                }
                GF_STAMP(10);
                if (argpass) {
                    // lowest column (scan order of the oracle: sx ascending) that holds the row's maximum
                    const int rsx = opaque(P.RSx);
                    int bsx = 0x7fff;
                    static_for<0, Ns - 1>([&](auto jc) {
                        constexpr int sx = L - 1 - decltype(jc)::value;                  // L - 1 down to -(L - 1): the last match that sticks is the lowest
                        constexpr int jj = (sx + Ns) % Ns, pp = pos_of(L, jj / 2);
                        const float v = (jj & 1) ? z[pp].y : z[pp].x;
                        if (v == best && sx >= -rsx && sx <= rsx) bsx = sx;
                    });
                    const int sy = slot <= RSy ? slot : RSy - slot;
                    bkey = (sy + RSy) * Ns + (bsx + L);
                }
            }
            if (!argpass) {
                if (c > 0) best = fmaxf(best, red_v[tid]);           // rows of the earlier chunks (same thread, same role)
                if (c + 1 < P.nchunk) red_v[tid] = best;
                else {
                    const float m = group_max<GW>(best);
                    if ((lane & (GW - 1)) == 0 && r_sl >= 0 && (half || re == 0)) {
                        const int dir = r_sl / P.npsi_store, ks = r_sl - dir * P.npsi_store;
                        const int o = dir * P.n_psi + ks + re * P.npsi_store;
                        partp[(size_t)o * NPART + (NPART == 2 ? (wave & 1) : 0)] = m;
                    }
                }
            } else {
                if (c > 0 && (red_v[tid] > best || (red_v[tid] == best && red_k[tid] < bkey))) { best = red_v[tid]; bkey = red_k[tid]; }
                red_v[tid] = best; red_k[tid] = bkey;
            }
            GF_STAMP(4);
            lds_barrier();                                           // T may be overwritten, red_* are visible
            GF_STAMP(5);
        }
        if (argpass && rr == 0 && r_hit >= 0 && re == r_eh) {
            // one thread per hit: best (value, key) over the rows of its orientation
            float bv = -3.0e38f; int bk = 0x7fffffff;
            for (int j = 0; j < 2 * L; j++) {
                const float v = red_v[tid + j]; const int k = red_k[tid + j];
                if (v > bv || (v == bv && k < bk)) { bv = v; bk = k; }
            }
            Hit h; h.cc = win_c[r_hit]; h.orient = win_o[r_hit]; h.sx = 0; h.sy = 0;
            if (bk != 0x7fffffff) { h.sy = bk / Ns - RSy; h.sx = bk % Ns - L; }
            hitp[r_hit] = h;
        }
        if (argpass) lds_barrier();                                  // red_* are reused by the next pass
    }
}

#undef GF_STAMP

}  // namespace ppm
