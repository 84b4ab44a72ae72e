// ppm_dev.h — device-side building blocks shared by the kernels of libpypmatch (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppm {

constexpr float kPiF = 3.14159265358979323846f;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }


// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it waits for every global load
// in flight, which defeats register prefetching across the barrier-heavy FFT stages.  Use it where the data exchanged
// between the threads lives in LDS; global writes read back by other threads of the block still need __syncthreads().
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- wave64 cross-lane steps without the LDS crossbar (ds_bpermute): DPP inside a row of 16 lanes, the gfx950
// v_permlane16_swap / v_permlane32_swap between rows.  Pairings: xor 1 and xor 2 (quad_perm), i <-> 7 - i (row_half_mirror;
// flips bit 2), xor 8 (row_ror:8), rows 0 <-> 1 and 2 <-> 3 (16-swap), halves (32-swap).
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppRor8 = 0x128;
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_mov(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false); }
// Inclusive scans over the 64 lanes with DPP only (no LDS round trips): row shifts by 1, 2, 4, 8, then the last lane of row 0 / 2 broadcast
// into the next row and that of the lower half into the upper half (row_bcast15 / row_bcast31, GFX9); lanes without a source keep 0.
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp_from(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int wave_scan_add(int v) {
    v += dpp_from<0x111, 0xf>(v); v += dpp_from<0x112, 0xf>(v); v += dpp_from<0x114, 0xf>(v); v += dpp_from<0x118, 0xf>(v);
    v += dpp_from<0x142, 0xa>(v); v += dpp_from<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ int wave_scan_max0(int v) {          // values >= 0
    v = max(v, dpp_from<0x111, 0xf>(v)); v = max(v, dpp_from<0x112, 0xf>(v)); v = max(v, dpp_from<0x114, 0xf>(v)); v = max(v, dpp_from<0x118, 0xf>(v));
    v = max(v, dpp_from<0x142, 0xa>(v)); v = max(v, dpp_from<0x143, 0xc>(v));
    return v;
}
// rows 1 and 3 of `a` trade places with rows 0 and 2 of `b` (M = 16), or the upper half of `a` with the lower half of `b`
// (M = 32); afterwards lanes with (lane & M) == 0 hold both halves of their `a` pair, the others those of their `b` pair
template <int M> __device__ __forceinline__ void lane_swap(float &a, float &b) {
    static_assert(M == 16 || M == 32, "row swaps only");
    if constexpr (M == 32) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r[0]); b = __uint_as_float(r[1]); }
    else { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false); a = __uint_as_float(r[0]); b = __uint_as_float(r[1]); }
}
template <int M> __device__ __forceinline__ void lane_swap(int &a, int &b) {
    if constexpr (M == 32) { auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false); a = (int)r[0]; b = (int)r[1]; }
    else { auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false); a = (int)r[0]; b = (int)r[1]; }
}

// wave64 all-lanes sum / max / min (every lane gets the result).  FULL EXEC REQUIRED: the DPP moves read 0 from an inactive
// source lane (old = 0, bound_ctrl off) and the permlane swaps exchange whole rows, so these must be called with all 64 lanes
// active, outside lane-divergent branches — a silently injected 0 would also win a max over negative scores.  Every caller
// in ppm_kernels*.h sits at wave-uniform control flow.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<kDppXor1>(v); v += dpp_mov<kDppXor2>(v); v += dpp_mov<kDppHalfMirror>(v); v += dpp_mov<kDppMirror>(v);
    float w = v; lane_swap<16>(v, w); v += w;
    w = v; lane_swap<32>(v, w); return v + w;
}
// v_max_f32 as it is (returns the other operand for a NaN): fmaxf() adds a canonicalising v_max per operand
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float wave_max(float v) {
    v = max_raw(v, dpp_mov<kDppXor1>(v)); v = max_raw(v, dpp_mov<kDppXor2>(v)); v = max_raw(v, dpp_mov<kDppHalfMirror>(v)); v = max_raw(v, dpp_mov<kDppMirror>(v));
    float w = v; lane_swap<16>(v, w); v = max_raw(v, w);
    w = v; lane_swap<32>(v, w); return max_raw(v, w);
}
__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }
// the same inside each half of the wave (lanes 0-31 / 32-63 separately)
__device__ __forceinline__ float half_max(float v) {
    v = max_raw(v, dpp_mov<kDppXor1>(v)); v = max_raw(v, dpp_mov<kDppXor2>(v)); v = max_raw(v, dpp_mov<kDppHalfMirror>(v)); v = max_raw(v, dpp_mov<kDppMirror>(v));
    float w = v; lane_swap<16>(v, w); return max_raw(v, w);
}
__device__ __forceinline__ int half_min(int v) {
    v = min(v, dpp_mov<kDppXor1>(v)); v = min(v, dpp_mov<kDppXor2>(v)); v = min(v, dpp_mov<kDppHalfMirror>(v)); v = min(v, dpp_mov<kDppMirror>(v));
    int w = v; lane_swap<16>(v, w); return min(v, w);
}
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, dpp_mov<kDppXor1>(v)); v = min(v, dpp_mov<kDppXor2>(v)); v = min(v, dpp_mov<kDppHalfMirror>(v)); v = min(v, dpp_mov<kDppMirror>(v));
    int w = v; lane_swap<16>(v, w); v = min(v, w);
    w = v; lane_swap<32>(v, w); return min(v, w);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// FFT plan of one length n = prod fac[] (factors 4, 2, 3, 5, 7): tw[k] = (cos 2 pi k/n, sin 2 pi k/n), k < n;
// perm[i] = LDS position of input sample i (digit-reversed staging), both in global memory.
struct FftPlan { int n, nfac; int fac[12]; const float2 *tw; const unsigned short *perm; };

// In-place mixed-radix decimation-in-time FFT of `nlines` lines of length n held in LDS.  Input must
// already sit at the positions plan.perm gives; output is in natural order.  Every stage combines r
// blocks of length Lp into one of length r Lp: twiddle w^(q j), then the r-point butterfly.  All threads of
// the block must call.
// i / d for 0 <= i < 2^22 with inv = 1.0f / d (float quotient, one fix-up step)
__device__ __forceinline__ int fast_div(int i, int d, float inv) {
    int q = (int)((float)i * inv);
    const int r = i - q * d;
    return r < 0 ? q - 1 : (r >= d ? q + 1 : q);
}

// `tw` = the plan's twiddle table, or a copy of it the caller keeps in LDS (the per-stage global loads are what the
// stages wait for otherwise).
__device__ inline void lds_fft(float2 *buf, const FftPlan &pl, int nlines, int lstride, bool inverse, int tid, int nthr, const float2 *tw) {
    const int n = pl.n;
    const float sgn = inverse ? 1.f : -1.f;       // sign of the exponent
    int Lp = 1;
    for (int st = 0; st < pl.nfac; st++) {
        const int r = pl.fac[st], L = Lp * r, m = n / r, tws = n / L;
        const float inv_m = 1.0f / (float)m, inv_Lp = 1.0f / (float)Lp;
        lds_barrier();
        for (int i = tid; i < nlines * m; i += nthr) {
            const int line = fast_div(i, m, inv_m), t = i - line * m, blk = fast_div(t, Lp, inv_Lp), j = t - blk * Lp;
            float2 *p = buf + line * lstride + blk * L + j;
            float2 x[7];
#pragma unroll
            for (int q = 0; q < 7; q++) {
                if (q < r) {
                    float2 v = p[q * Lp];
                    if (q > 0) { float2 w = tw[q * j * tws]; w.y *= sgn; v = cmul(v, w); }
                    x[q] = v;
                }
            }
            if (r == 2) {
                p[0] = cadd(x[0], x[1]); p[Lp] = csub(x[0], x[1]);
            } else if (r == 4) {
                float2 t0 = cadd(x[0], x[2]), t1 = csub(x[0], x[2]), t2 = cadd(x[1], x[3]), d = csub(x[1], x[3]);
                float2 jd = make_float2(-sgn * d.y, sgn * d.x);      // (sgn i) d
                p[0] = cadd(t0, t2); p[2 * Lp] = csub(t0, t2);
                p[Lp] = cadd(t1, jd); p[3 * Lp] = csub(t1, jd);
            } else if (r == 3) {
                float2 sm = cadd(x[1], x[2]), d = csub(x[1], x[2]);
                float2 h = make_float2(x[0].x - 0.5f * sm.x, x[0].y - 0.5f * sm.y);
                const float c = 0.8660254037844386f * sgn;           // (sgn i sqrt(3)/2) d
                float2 jd = make_float2(-c * d.y, c * d.x);
                p[0] = cadd(x[0], sm); p[Lp] = cadd(h, jd); p[2 * Lp] = csub(h, jd);
            } else if (r == 5) {
                const float c1 = 0.30901699437494745f, c2 = -0.8090169943749475f, s1 = 0.9510565162951535f, s2 = 0.5877852522924731f;
                float2 a1 = cadd(x[1], x[4]), a2 = cadd(x[2], x[3]), b1 = csub(x[1], x[4]), b2 = csub(x[2], x[3]);
                float2 t1 = make_float2(x[0].x + c1 * a1.x + c2 * a2.x, x[0].y + c1 * a1.y + c2 * a2.y);
                float2 t2 = make_float2(x[0].x + c2 * a1.x + c1 * a2.x, x[0].y + c2 * a1.y + c1 * a2.y);
                float2 u1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
                float2 u2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
                float2 j1 = make_float2(-sgn * u1.y, sgn * u1.x), j2 = make_float2(-sgn * u2.y, sgn * u2.x);   // (sgn i) u
                p[0] = make_float2(x[0].x + a1.x + a2.x, x[0].y + a1.y + a2.y);
                p[Lp] = cadd(t1, j1); p[4 * Lp] = csub(t1, j1);
                p[2 * Lp] = cadd(t2, j2); p[3 * Lp] = csub(t2, j2);
            } else {
                // r == 7: X[p] = t_p + (sgn i) u_p, X[7 - p] = t_p - (sgn i) u_p with t_p = x0 + sum_k cos(2 pi p k / 7) a_k,
                // u_p = sum_k sin(2 pi p k / 7) b_k over the pairs a_k = x[k] + x[7 - k], b_k = x[k] - x[7 - k]
                const float c1 = 0.6234898018587336f, c2 = -0.2225209339563144f, c3 = -0.9009688679024191f;
                const float s1 = 0.7818314824680298f, s2 = 0.9749279121818236f, s3 = 0.4338837391175581f;
                float2 a1 = cadd(x[1], x[6]), a2 = cadd(x[2], x[5]), a3 = cadd(x[3], x[4]);
                float2 b1 = csub(x[1], x[6]), b2 = csub(x[2], x[5]), b3 = csub(x[3], x[4]);
                float2 t1 = make_float2(x[0].x + c1 * a1.x + c2 * a2.x + c3 * a3.x, x[0].y + c1 * a1.y + c2 * a2.y + c3 * a3.y);
                float2 t2 = make_float2(x[0].x + c2 * a1.x + c3 * a2.x + c1 * a3.x, x[0].y + c2 * a1.y + c3 * a2.y + c1 * a3.y);
                float2 t3 = make_float2(x[0].x + c3 * a1.x + c1 * a2.x + c2 * a3.x, x[0].y + c3 * a1.y + c1 * a2.y + c2 * a3.y);
                float2 u1 = make_float2(s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y);
                float2 u2 = make_float2(s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y);
                float2 u3 = make_float2(s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y);
                float2 j1 = make_float2(-sgn * u1.y, sgn * u1.x), j2 = make_float2(-sgn * u2.y, sgn * u2.x), j3 = make_float2(-sgn * u3.y, sgn * u3.x);
                p[0] = make_float2(x[0].x + a1.x + a2.x + a3.x, x[0].y + a1.y + a2.y + a3.y);
                p[Lp] = cadd(t1, j1); p[6 * Lp] = csub(t1, j1);
                p[2 * Lp] = cadd(t2, j2); p[5 * Lp] = csub(t2, j2);
                p[3 * Lp] = cadd(t3, j3); p[4 * Lp] = csub(t3, j3);
            }
        }
        Lp = L;
    }
    lds_barrier();
}
__device__ inline void lds_fft(float2 *buf, const FftPlan &pl, int nlines, int lstride, bool inverse, int tid, int nthr) {
    lds_fft(buf, pl, nlines, lstride, inverse, tid, nthr, pl.tw);
}

// ---- N = 256 fast path: 16 x 16 four-step FFT with the 16-point transforms held in registers.
// forward radix-4 butterfly on (a, b, c, d) -> outputs 0..3
__device__ __forceinline__ void bfly4(float2 &a, float2 &b, float2 &c, float2 &d) {
    const float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), e = csub(b, d);
    const float2 t3 = make_float2(e.y, -e.x);                    // -i (b - d)
    a = cadd(t0, t2); c = csub(t0, t2); b = cadd(t1, t3); d = csub(t1, t3);
}
// in-place forward 16-point DFT: x[n] -> X[k] (natural order in and out)
__device__ __forceinline__ void dft16(float2 (&x)[16]) {
    // step 1: over n2 for each n1 (n = n1 + 4 n2): results Y[n1][k2] land in x[n1 + 4 k2]
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) bfly4(x[n1], x[n1 + 4], x[n1 + 8], x[n1 + 12]);
    // step 2: twiddles W16^(n1 k2) = (cos, -sin)(2 pi n1 k2 / 16)
    constexpr float C1 = 0.9238795325112867f, S1 = 0.3826834323650898f, C2 = 0.7071067811865476f;
    x[1 + 4] = cmul(x[1 + 4], make_float2(C1, -S1));             // n1 = 1, k2 = 1 : m = 1
    x[1 + 8] = cmul(x[1 + 8], make_float2(C2, -C2));             // m = 2
    x[1 + 12] = cmul(x[1 + 12], make_float2(S1, -C1));           // m = 3
    x[2 + 4] = cmul(x[2 + 4], make_float2(C2, -C2));             // n1 = 2: m = 2
    x[2 + 8] = make_float2(x[2 + 8].y, -x[2 + 8].x);             // m = 4 : -i
    x[2 + 12] = cmul(x[2 + 12], make_float2(-C2, -C2));          // m = 6
    x[3 + 4] = cmul(x[3 + 4], make_float2(S1, -C1));             // n1 = 3: m = 3
    x[3 + 8] = cmul(x[3 + 8], make_float2(-C2, -C2));            // m = 6
    x[3 + 12] = cmul(x[3 + 12], make_float2(-C1, S1));           // m = 9
    // step 3: over n1 for each k2: X[4 k1 + k2] from (Y[0][k2], Y[1][k2], Y[2][k2], Y[3][k2]) = x[4 k2 + 0..3]
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) bfly4(x[4 * k2], x[4 * k2 + 1], x[4 * k2 + 2], x[4 * k2 + 3]);
    // now x[4 k2 + k1] = X[4 k1 + k2]: transpose the 4 x 4 index
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i + 1; j < 4; j++) { const float2 t = x[4 * i + j]; x[4 * i + j] = x[4 * j + i]; x[4 * j + i] = t; }
}

// Forward FFT of `nlines` lines of 256 samples held in LDS in NATURAL order (line stride >= 272: the intermediate uses a
// padded [16][17] layout against bank conflicts); output in natural order.  16 threads per line.  All threads must call.
__device__ inline void lds_fft256(float2 *buf, int nlines, int lstride, int tid, int nthr, const float2 *tw) {
    const int work = nlines * 16;
    lds_barrier();
    for (int base = 0; base < work; base += nthr) {              // pass 1: over n2 for each n1, times W256^(n1 k2)
        const int w = base + tid, line = w >> 4, n1 = w & 15;
        float2 x[16];
        float2 *row = buf + line * lstride;
        if (w < work) {
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) x[n2] = row[n1 + 16 * n2];
        }
        lds_barrier();
        if (w < work) {
            dft16(x);
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                float2 v = x[k2];
                if (k2 > 0) { float2 t = tw[n1 * k2]; t.y = -t.y; v = cmul(v, t); }
                row[k2 * 17 + n1] = v;
            }
        }
    }
    lds_barrier();
    for (int base = 0; base < work; base += nthr) {              // pass 2: over n1 for each k2 -> X[16 k1 + k2]
        const int w = base + tid, line = w >> 4, k2 = w & 15;
        float2 y[16];
        float2 *row = buf + line * lstride;
        if (w < work) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) y[n1] = row[k2 * 17 + n1];
        }
        lds_barrier();
        if (w < work) {
            dft16(y);
#pragma unroll
            for (int k1 = 0; k1 < 16; k1++) row[16 * k1 + k2] = y[k1];
        }
    }
    lds_barrier();
}

// Band-limited reference cube, x in 0..B+1, y and z in -B-1..B+1 (stored index = coordinate + off), in a BLOCKED layout:
// one 128-byte line holds a 4 (x) x 2 (y) x 2 (z) brick of complex voxels, so that the 8 taps of a trilinear sample fall
// into 1-4 lines (on average 2.25) whatever the slice orientation, and neighbouring samples of a slice share them; with
// x fastest over the whole row every tap pair of a tilted slice sat in its own line.  The x-pair (x0, x0+1) is always
// read as one 16-byte load: a second copy of the cube whose bricks start at x = 2 serves the pairs that would straddle two
// bricks of the first copy (x0 % 4 == 3).
//   element(copy, x, y, z) = copy * LB + (((z >> 1) * NBY + (y >> 1)) * NBX + ((x - 2 copy) >> 2)) * 16
//                                      + ((z & 1) * 2 + (y & 1)) * 4 + ((x - 2 copy) & 3)
struct CubeView { const float2 *cube; int NBX, NBY, off; unsigned LB; float scale; };   // scale = padding factor: sample k sits at scale k

__host__ __device__ __forceinline__ size_t cube_element(int NBX, int NBY, unsigned LB, int copy, int x, int y, int z) {
    const int xs = x - 2 * copy;
    return (size_t)copy * LB + ((size_t)((z >> 1) * NBY + (y >> 1)) * NBX + (xs >> 2)) * 16 + ((z & 1) * 2 + (y & 1)) * 4 + (xs & 3);
}

// The 8 taps as four 16-byte x-pairs plus the interpolation fractions: fetch and interpolation are split so that the
// gathers of the NEXT evaluation can be in flight while the current one is interpolated and scored (k_local).
struct CubeTaps { float4 a, b, c, d; float fx, fy, fz; bool cj; };

typedef float cube_v4f __attribute__((ext_vector_type(4)));

// The four 16-byte x-pairs come through BUFFER loads: one 32-bit byte offset per load against a wave-uniform resource descriptor
// (the cube's base and size in SGPRs) instead of a 64-bit address per lane and load - the address arithmetic of a gather drops from
// ~20 vector instructions to ~8, and an offset beyond the cube reads zeros instead of faulting.
__device__ __forceinline__ CubeTaps cube_fetch(const CubeView &cv, float X, float Y, float Z) {
    CubeTaps t;
    X *= cv.scale; Y *= cv.scale; Z *= cv.scale;
    t.cj = X < 0.f;                                  // Friedel symmetry supplies x < 0
    if (t.cj) { X = -X; Y = -Y; Z = -Z; }
    const float xf = floorf(X), yf = floorf(Y), zf = floorf(Z);
    t.fx = X - xf; t.fy = Y - yf; t.fz = Z - zf;
    const int x0 = (int)xf, y0 = (int)yf + cv.off, z0 = (int)zf + cv.off;
    const bool second = (x0 & 3) == 3;
    const int xs = second ? x0 - 2 : x0;
    // byte offsets; brick counts and offsets stay below 2^24 elements: 24-bit multiplies (full rate; v_mul_lo_u32 issues at a quarter of it)
    const unsigned rowb = (unsigned)cv.NBX * 128u, planeb = __umul24(rowb, (unsigned)cv.NBY);
    const unsigned xl = (unsigned)(xs >> 2) * 128u + (unsigned)(xs & 3) * 8u + (second ? cv.LB * 8u : 0u);
    const unsigned yl0 = __umul24((unsigned)(y0 >> 1), rowb) + (unsigned)(y0 & 1) * 32u, dy = (y0 & 1) ? rowb - 32u : 32u;
    const unsigned zl0 = __umul24((unsigned)(z0 >> 1), planeb) + (unsigned)(z0 & 1) * 64u, dz = (z0 & 1) ? planeb - 64u : 64u;
    const unsigned oa = xl + yl0 + zl0, ob = oa + dy, oc = oa + dz, od = ob + dz;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)cv.cube, 0, (int)(2u * cv.LB * 8u), 0x00020000);
    const cube_v4f a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oa, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)ob, 0, 0);
    const cube_v4f c = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oc, 0, 0), d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)od, 0, 0);
    t.a = make_float4(a.x, a.y, a.z, a.w); t.b = make_float4(b.x, b.y, b.z, b.w);
    t.c = make_float4(c.x, c.y, c.z, c.w); t.d = make_float4(d.x, d.y, d.z, d.w);
    return t;
}

// ---- address tables (k_local / k_csp_eval): the byte offset of a tap is separable, o = ox(x0) + oy(y0) + oz(z0), and the steps to the
// y + 1 / z + 1 taps depend on y0 / z0 alone.  Three small LDS tables (filled once per block) replace ~25 vector instructions of
// shifts, masks, multiplies and selects per gather by three LDS reads and four adds.  Tables cover |coordinate| <= R (in cube
// voxels, from the band and the padding factor); entry layout: ty0[yi] = (offset of row y0 = yi + off, step to y0 + 1).
struct CubeTab { const uint2 *ty0, *tz0; const unsigned *tx0; };
__host__ __device__ inline int cube_tab_radius(int band_px, float scale) { return (int)ceilf((float)(band_px + 2) * scale) + 2; }
__host__ __device__ inline size_t cube_tab_bytes(int R) { return R > 0 ? (size_t)2 * (2 * R + 2) * 8 + (size_t)(R + 2) * 4 : 0; }
__device__ inline CubeTab cube_tab_fill(const CubeView &cv, void *mem, int R, int tid, int nthr) {
    uint2 *ty = (uint2 *)mem, *tz = ty + (2 * R + 2);
    unsigned *tx = (unsigned *)(tz + (2 * R + 2));
    const unsigned rowb = (unsigned)cv.NBX * 128u, planeb = rowb * (unsigned)cv.NBY;
    for (int i = tid; i < 2 * R + 2; i += nthr) {
        const int y0 = i - (R + 1) + cv.off;           // below 0 only where no sample of the band lands: the offset wraps and the load returns zeros
        ty[i] = make_uint2((unsigned)(y0 >> 1) * rowb + (unsigned)(y0 & 1) * 32u, (y0 & 1) ? rowb - 32u : 32u);
        tz[i] = make_uint2((unsigned)(y0 >> 1) * planeb + (unsigned)(y0 & 1) * 64u, (y0 & 1) ? planeb - 64u : 64u);
    }
    for (int x0 = tid; x0 < R + 2; x0 += nthr) {
        const bool second = (x0 & 3) == 3;
        const int xs = second ? x0 - 2 : x0;
        tx[x0] = (unsigned)(xs >> 2) * 128u + (unsigned)(xs & 3) * 8u + (second ? cv.LB * 8u : 0u);
    }
    CubeTab t; t.ty0 = ty + (R + 1); t.tz0 = tz + (R + 1); t.tx0 = tx;
    return t;
}

__device__ __forceinline__ int floor_to_int(float x) { int i; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x)); return i; }

// cube_fetch for coordinates that already carry the padding factor, through the address tables
__device__ __forceinline__ CubeTaps cube_fetch_tab(const CubeView &cv, const CubeTab &tb, float X, float Y, float Z) {
    CubeTaps t;
    t.cj = X < 0.f;
    if (t.cj) { X = -X; Y = -Y; Z = -Z; }
    t.fx = __builtin_amdgcn_fractf(X); t.fy = __builtin_amdgcn_fractf(Y); t.fz = __builtin_amdgcn_fractf(Z);
    const unsigned xl = tb.tx0[floor_to_int(X)];
    const uint2 ey = tb.ty0[floor_to_int(Y)], ez = tb.tz0[floor_to_int(Z)];
#ifdef PPM_DBG_SMALLCUBE
    const unsigned oa = (xl + ey.x + ez.x) & 0x3ff0u, ob = (oa + ey.y) & 0x3ff0u, oc = (oa + ez.y) & 0x3ff0u, od = (ob + ez.y) & 0x3ff0u;
#else
    const unsigned oa = xl + ey.x + ez.x, ob = oa + ey.y, oc = oa + ez.y, od = ob + ez.y;
#endif
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)cv.cube, 0, (int)(2u * cv.LB * 8u), 0x00020000);
    const cube_v4f a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oa, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)ob, 0, 0);
    const cube_v4f c = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oc, 0, 0), d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)od, 0, 0);
    t.a = make_float4(a.x, a.y, a.z, a.w); t.b = make_float4(b.x, b.y, b.z, b.w);
    t.c = make_float4(c.x, c.y, c.z, c.w); t.d = make_float4(d.x, d.y, d.z, d.w);
    return t;
}

__device__ __forceinline__ float2 cube_interp(const CubeTaps &t) {
    const float fx = t.fx, fy = t.fy, fz = t.fz;
    const float ar = t.a.x + fx * (t.a.z - t.a.x), ai = t.a.y + fx * (t.a.w - t.a.y);
    const float br = t.b.x + fx * (t.b.z - t.b.x), bi = t.b.y + fx * (t.b.w - t.b.y);
    const float cr = t.c.x + fx * (t.c.z - t.c.x), ci = t.c.y + fx * (t.c.w - t.c.y);
    const float dr = t.d.x + fx * (t.d.z - t.d.x), di = t.d.y + fx * (t.d.w - t.d.y);
    const float er = ar + fy * (br - ar), ei = ai + fy * (bi - ai);
    const float gr = cr + fy * (dr - cr), gi = ci + fy * (di - ci);
    const float rr = er + fz * (gr - er), ri = ei + fz * (gi - ei);
    return make_float2(rr, t.cj ? -ri : ri);
}

// Trilinear sample of the reference cube at Fourier coordinate (X, Y, Z)
__device__ __forceinline__ float2 sample_cube(const CubeView &cv, float X, float Y, float Z) { return cube_interp(cube_fetch(cv, X, Y, Z)); }

// CTF of one particle: -sin(pi lambda s^2 (df(phi) - Cs lambda^2 s^2 / 2) + phase + amp)
struct CtfP { float lambda, cs, dsum, ddif, c2a, s2a, extra, inv_na2; };

__host__ __device__ __forceinline__ CtfP ctf_from_row(const double *row, int N, double a) {
    CtfP c;
    double v = row[PPM_VOLTAGE] * 1000.0;
    double lam = 12.2639 / sqrt(v + 0.97845e-6 * v * v);
    double w = row[PPM_AMP], ast = row[PPM_ANGAST] * 3.14159265358979323846 / 180.0;
    c.lambda = (float)lam;
    c.cs = (float)(row[PPM_CS] * 1e7);
    c.dsum = (float)(row[PPM_DF1] + row[PPM_DF2]);
    c.ddif = (float)(row[PPM_DF1] - row[PPM_DF2]);
    c.c2a = (float)cos(2.0 * ast);
    c.s2a = (float)sin(2.0 * ast);
    c.extra = (float)(row[PPM_PSHIFT] + atan(w / sqrt(1.0 - w * w)));
    c.inv_na2 = (float)(1.0 / ((double)N * a * N * a));
    return c;
}

__device__ __forceinline__ float ctf_eval(const CtfP &c, int kx, int ky) {
    float k2 = (float)(kx * kx + ky * ky);
    if (k2 == 0.f) return -sinf(c.extra);
    float s2 = k2 * c.inv_na2;
    float ik2 = 1.0f / k2;
    float c2 = (float)(kx * kx - ky * ky) * ik2, s2p = (float)(2 * kx * ky) * ik2;
    float df = 0.5f * (c.dsum + c.ddif * (c2 * c.c2a + s2p * c.s2a));
    float chi = kPiF * c.lambda * s2 * (df - 0.5f * c.cs * c.lambda * c.lambda * s2) + c.extra;
    return -sinf(chi);
}

// same CTF with the phase reduced to one revolution in fp32 and the hardware sine (v_sin_f32): chi itself carries an fp32
// rounding of ~|chi| 6e-8, the hardware sine adds ~1e-6 absolute
__device__ __forceinline__ float ctf_eval_fast(const CtfP &c, int kx, int ky) {
    float k2 = (float)(kx * kx + ky * ky);
    float s2 = k2 * c.inv_na2;
    float ik2 = __frcp_rn(fmaxf(k2, 1.f));
    float c2 = (float)(kx * kx - ky * ky) * ik2, s2p = (float)(2 * kx * ky) * ik2;
    float df = 0.5f * (c.dsum + c.ddif * (c2 * c.c2a + s2p * c.s2a));
    float chi = kPiF * c.lambda * s2 * (df - 0.5f * c.cs * c.lambda * c.lambda * s2) + c.extra;
    float rev = chi * 0.15915494309189535f; rev -= floorf(rev);
    return -__sinf(6.283185307179586f * rev);
}

__device__ __forceinline__ void unpack_sample(uint32_t u, int &kx, int &ky, int &alpha, int &ring) {
    kx = (int)(u & 511u);
    ky = (int)((u >> 9) & 1023u) - 256;
    alpha = (int)((u >> 19) & 3u);
    ring = (int)(u >> 21);
}

}  // namespace ppm
