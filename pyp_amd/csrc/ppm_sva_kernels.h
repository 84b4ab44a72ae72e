// ppm_sva_kernels.h — sub-tomogram alignment kernels of libpypmatch (gfx950): pre-processing of a sub-volume into its
// band-limited half-space transform, and the wedge-weighted correlation of that transform with the rotated reference for a
// set of candidate poses (include/ppm.h, ppm_sva_cfg).
#pragma once
#include <type_traits>
#include "ppm_csp_kernels.h"

namespace ppm {

// sum and sum of squares of one sub-volume: grid (blocks, n_vol), stats[v] = {sum, sumsq} (double atomics, one per block)
__global__ void __launch_bounds__(256) k_sva_stats(const float *__restrict__ vols, size_t n3, double *stats) {
    __shared__ double r1[4], r2[4];
    const float *v = vols + (size_t)blockIdx.y * n3;
    double s1 = 0, s2 = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n3; i += (size_t)gridDim.x * 256) { const double x = v[i]; s1 += x; s2 += x * x; }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[2 * blockIdx.y], ((r1[0] + r1[1]) + r1[2]) + r1[3]);
        atomicAdd(&stats[2 * blockIdx.y + 1], ((r2[0] + r2[1]) + r2[2]) + r2[3]);
    }
}

struct SvaWin { float w[3], sigma; };

// Pruned 3-D transform of a sub-volume: only kx <= R and |ky|, |kz| <= R are ever sampled (the band of the protocol's low-pass
// filter), so (1) the x pass reads the REAL volume, applies (v - mean) / sigma and the real-space window on the way into LDS and
// writes only the first KX = R + 1 coefficients of every line into a compact [z][y][KX] array, (2) the y pass (k_fft_lines)
// works on that array, (3) the z pass only on the lines with |ky| <= R.  Traffic per sub-volume at 192^3, R = 60: 110 MB
// instead of 425 MB for three full complex passes behind a separate load kernel.
struct SvaXP { const float *vol; const double *stats; float2 *out; FftPlan plan; int n, L, KX; long nlines; SvaWin W; };

__global__ void __launch_bounds__(256) k_sva_xpass(SvaXP P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float wx[512], wyz[16];               // window along x; window of every line's (y, z) (P.L <= 16, n <= 512)
    __shared__ float2 tw_s[512];
    float2 *buf = (float2 *)smem;
    const int tid = threadIdx.x, n = P.n;
    const long l0 = (long)blockIdx.x * P.L;
    const int nl = (int)((P.nlines - l0) < P.L ? (P.nlines - l0) : P.L);
    if (nl <= 0) return;
    const double n3 = (double)n * n * n;
    // P.L divides n * n: the lines of a block belong to one sub-volume of the batch (P.vol / P.out / P.stats point at its first one)
    const double *st = P.stats + 2 * (l0 / ((long)n * n));
    const double mu = st[0] / n3, var = st[1] / n3 - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
    const float fmu = (float)mu, finv = (float)(1.0 / sd);
    auto win1 = [&](int c, int k) {
        if (!(P.W.w[k] > 0.f)) return 1.f;
        const float d = fabsf((float)c) - P.W.w[k];
        return d > 0.f ? (P.W.sigma > 0.f ? expf(-d * d / (2.f * P.W.sigma * P.W.sigma)) : 0.f) : 1.f;
    };
    for (int e = tid; e < n; e += 256) { wx[e] = win1(e - n / 2, 0); tw_s[e] = P.plan.tw[e]; }
    if (tid < nl) { const long l = l0 + tid; wyz[tid] = win1((int)(l % n) - n / 2, 1) * win1((int)((l / n) % n) - n / 2, 2); }
    __syncthreads();
    {   // (line, e) of element i = tid, tid + 256, ... without a division per element
        const int dl = 256 / n, de = 256 % n;
        int line = tid / n, e = tid % n;
        const float *src = P.vol + l0 * n;
        for (int i = tid; i < nl * n; i += 256) {
            buf[line * n + P.plan.perm[e]] = make_float2((src[i] - fmu) * finv * (wx[e] * wyz[line]), 0.f);
            line += dl; e += de;
            if (e >= n) { e -= n; line++; }
        }
    }
    lds_fft(buf, P.plan, nl, n, false, tid, 256, tw_s);
    {
        const int KX = P.KX, dl = 256 / KX, de = 256 % KX;
        int line = tid / KX, e = tid % KX;
        float2 *dst = P.out + l0 * KX;
        for (int i = tid; i < nl * KX; i += 256) {
            dst[i] = buf[line * n + e];
            line += dl; e += de;
            if (e >= KX) { e -= KX; line++; }
        }
    }
}

// packed sample: kx (10 bits) | ky + 512 (11 bits) << 10 | kz + 512 (11 bits) << 21
__host__ __device__ __forceinline__ uint32_t sva_pack(int kx, int ky, int kz) { return (uint32_t)kx | ((uint32_t)(ky + 512) << 10) | ((uint32_t)(kz + 512) << 21); }
__host__ __device__ __forceinline__ void sva_unpack(uint32_t u, int &kx, int &ky, int &kz) { kx = (int)(u & 1023u); ky = (int)((u >> 10) & 2047u) - 512; kz = (int)(u >> 21) - 512; }

// ---- box sizes that are multiples of 16 (192 = 16 x 12): two-step transforms, N = 16 M.  Step 1: thread (line, t) holds
// x[t + M j], j = 0 .. 15, in registers, transforms them (dft16), multiplies by W_N^(t k2) and puts Y[k2][t] back into the
// line; step 2: thread (k2, line) sums the M-point transform over t for the outputs k = 16 k1 + k2 that are kept — the band's
// pruning removes most of them, so the direct sum costs less than a staged transform and needs no bit reversal.  The line buffer
// is reused between the steps (all loads of step 1 precede its stores); odd line stride: step 2 walks 16 lines at one offset.
// The arrays are laid out so that every pass reads whole lines and writes runs of L values:
//   x pass: real volume [z][y][x] -> A[z][kx][y], kx < KX;   y pass: A -> B[kx][kyi][z], |ky| <= R (kyi = ky, or ky - N + KY below
//   zero);   z pass: B in place, |kz| <= R;   k_sva_gather16 picks the band's samples out of B.
// MT = M at compile time (the line of step 2 is read once into registers and the loops unroll), or 0 for any M.
template <int MT, typename Need, typename Emit>
__device__ __forceinline__ void fft16m(float2 *buf, int nl, int N, int LS, const float2 *tw, int tid, Need need, Emit emit) {
    const int M = MT > 0 ? MT : (MT < 0 ? -MT : N >> 4);
    {
        const int line = tid / M, t = tid - line * M;            // nl * M <= 256 tasks: one per thread
        const bool on = tid < nl * M;
        float2 x[16];
        if (on) {
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = buf[line * LS + t + M * j];
        }
        __syncthreads();
        if (on) {
            dft16(x);
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                float2 v = x[k2];
                if (k2 > 0) { const float2 w = tw[t * k2]; v = cmul(v, make_float2(w.x, -w.y)); }
                buf[line * LS + k2 * M + t] = v;
            }
        }
        __syncthreads();
    }
    for (int task = tid; task < nl * 16; task += 256) {
        const int k2 = task / nl, line = task - k2 * nl;          // line fastest: a wave's stores are runs of nl values (k2 fastest for the in-place z pass: no gain)
        const float2 *row = buf + line * LS + k2 * M;
        if constexpr (MT > 0) {
            float2 v[MT];
#pragma unroll
            for (int t = 0; t < MT; t++) v[t] = row[t];
#pragma unroll
            for (int k1 = 0; k1 < MT; k1++) {
                const int k = 16 * k1 + k2;
                if (!need(k)) continue;
                float ar = v[0].x, ai = v[0].y;
#pragma unroll
                for (int t = 1; t < MT; t++) {
                    const float2 w = tw[((t * k1) % MT) << 4];    // compile-time index: W_M^(t k1) = conj(tw[16 (t k1 mod M)])
                    ar += v[t].x * w.x + v[t].y * w.y; ai += v[t].y * w.x - v[t].x * w.y;
                }
                emit(line, k, make_float2(ar, ai));
            }
        } else if constexpr (MT < 0) {                            // the line in registers, the outputs in a loop (fewer registers than MT > 0)
            constexpr int MR = -MT;
            float2 v[MR];
#pragma unroll
            for (int t = 0; t < MR; t++) v[t] = row[t];
#pragma unroll 1
            for (int k1 = 0; k1 < MR; k1++) {
                const int k = 16 * k1 + k2;
                if (!need(k)) continue;
                float ar = v[0].x, ai = v[0].y;
                int idx = 0;
#pragma unroll
                for (int t = 1; t < MR; t++) {
                    idx += k1; if (idx >= MR) idx -= MR;
                    const float2 w = tw[idx << 4];
                    ar += v[t].x * w.x + v[t].y * w.y; ai += v[t].y * w.x - v[t].x * w.y;
                }
                emit(line, k, make_float2(ar, ai));
            }
        } else {
            for (int k1 = 0; k1 < M; k1++) {
                const int k = 16 * k1 + k2;
                if (!need(k)) continue;
                float ar = 0.f, ai = 0.f;
                int idx = 0;                                      // (t k1) mod M
                for (int t = 0; t < M; t++) {
                    const float2 v = row[t], w = tw[idx << 4];    // W_M^(t k1) = conj(tw[16 idx])
                    ar += v.x * w.x + v.y * w.y; ai += v.y * w.x - v.x * w.y;
                    idx += k1; if (idx >= M) idx -= M;
                }
                emit(line, k, make_float2(ar, ai));
            }
        }
    }
}
// (Instantiating M at compile time for the common boxes — the line of step 2 in registers, unrolled sums — ran 1.5 x SLOWER at
// 192^3: the kernels' register count is the largest of all instances.  The loop form is the one used.)
template <bool SPECIAL192, typename Need, typename Emit>
__device__ __forceinline__ void fft16m_any(float2 *buf, int nl, int N, int LS, const float2 *tw, int tid, Need need, Emit emit) {
    if (SPECIAL192 && N == 192) fft16m<12>(buf, nl, N, LS, tw, tid, need, emit);      // BASELINE config 5's box, x pass only (92 registers; the y / z passes would need 196)
    else if (N == 192) fft16m<-12>(buf, nl, N, LS, tw, tid, need, emit);             // y / z passes of that box: the line in registers only
    else fft16m<0>(buf, nl, N, LS, tw, tid, need, emit);
}

// mode 0: (v - mean) / sigma from `stats` (k_sva_stats ran before); mode 1: the raw windowed volume, and the block leaves its sum
// and sum of squares in `stats`[2 block .. ] for k_sva_stats_sum (the transform is linear: k_sva_gather16 subtracts mean x the window's transform and divides by
// sigma, so the volume is read once instead of twice); mode 2: the window itself (v = 1, nothing read)
struct SvaX16P { const float *vol; double *stats; float2 *A; const float2 *tw; int n, L, KX, mode; long nlines; SvaWin W; };

__global__ void __launch_bounds__(256) k_sva_x16(SvaX16P P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float wx[512], wyz[16];
    __shared__ float2 tw_s[512];
    float2 *buf = (float2 *)smem;
    const int tid = threadIdx.x, n = P.n, LS = n + 1;
    const long l0 = (long)blockIdx.x * P.L;
    const int nl = (int)((P.nlines - l0) < P.L ? (P.nlines - l0) : P.L);
    if (nl <= 0) return;
    const double n3 = (double)n * n * n;
    const long nn = (long)n * n, v = l0 / nn;                    // P.L divides n: a block's lines share the sub-volume and z
    float fmu = 0.f, finv = 1.f;
    if (P.mode == 0) {
        const double *st = P.stats + 2 * v;
        const double mu = st[0] / n3, var = st[1] / n3 - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
        fmu = (float)mu; finv = (float)(1.0 / sd);
    }
    auto win1 = [&](int c, int k) {
        if (!(P.W.w[k] > 0.f)) return 1.f;
        const float d = fabsf((float)c) - P.W.w[k];
        return d > 0.f ? (P.W.sigma > 0.f ? expf(-d * d / (2.f * P.W.sigma * P.W.sigma)) : 0.f) : 1.f;
    };
    for (int e = tid; e < n; e += 256) { wx[e] = win1(e - n / 2, 0); tw_s[e] = P.tw[e]; }
    const int y0 = (int)(l0 % n), z = (int)((l0 / n) % n);
    if (tid < nl) wyz[tid] = win1(y0 + tid - n / 2, 1) * win1(z - n / 2, 2);
    __syncthreads();
    {
        const float *src1 = P.vol + l0 * n;
        const bool vec = P.mode == 2 || (((size_t)src1) & 15) == 0;    // n is a multiple of 16: four voxels per load unless the caller's pointer is oddly aligned
        const float4 *src = (const float4 *)src1;
        double s1 = 0, s2 = 0;
        const int n4 = n >> 2;
        for (int i4 = tid; i4 < nl * n4; i4 += 256) {
            const int line = i4 / n4, e = (i4 - line * n4) << 2;
            float4 x4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (P.mode != 2) x4 = vec ? src[i4] : make_float4(src1[4 * i4], src1[4 * i4 + 1], src1[4 * i4 + 2], src1[4 * i4 + 3]);
            const float xs[4] = { x4.x, x4.y, x4.z, x4.w };
            const float wl = wyz[line];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float x = xs[c];
                if (P.mode == 1) { s1 += (double)x; s2 += (double)x * (double)x; }
                buf[line * LS + e + c] = make_float2((x - fmu) * finv * (wx[e + c] * wl), 0.f);
            }
        }
        if (P.mode == 1) {
            __shared__ double r1[4], r2[4];
            s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
            if ((tid & 63) == 0) { r1[tid >> 6] = s1; r2[tid >> 6] = s2; }
            __syncthreads();
            if (tid == 0) {                                      // per-block partial sums; k_sva_stats_sum adds them in block order
                P.stats[2 * (size_t)blockIdx.x] = ((r1[0] + r1[1]) + r1[2]) + r1[3];
                P.stats[2 * (size_t)blockIdx.x + 1] = ((r2[0] + r2[1]) + r2[2]) + r2[3];
            }
        }
    }
    __syncthreads();
    const int KX = P.KX;
    float2 *dst = P.A + ((v * n + z) * KX) * (long)n + y0;
    fft16m_any<true>(buf, nl, n, LS, tw_s, tid, [&](int k) { return k < KX; }, [&](int line, int k, float2 val) { dst[(long)k * n + line] = val; });
}

// sum and sum of squares of every sub-volume from the partial sums of its `per_vol` k_sva_x16 blocks, added in block order
// (one wave per sub-volume; lane l adds blocks l, l + 64, ...)
__global__ void __launch_bounds__(64) k_sva_stats_sum(const double *__restrict__ part, int per_vol, double *__restrict__ stats) {
    const double *p = part + 2 * (size_t)blockIdx.x * per_vol;
    double s1 = 0, s2 = 0;
    for (int b = threadIdx.x; b < per_vol; b += 64) { s1 += p[2 * b]; s2 += p[2 * b + 1]; }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = s1; stats[2 * blockIdx.x + 1] = s2; }
}

// y pass: block = L lines z0 .. z0 + L - 1 of one (sub-volume, kx); z pass (in_place): block = L consecutive lines of B
// z pass with `pos` set (round 5): the band's samples leave the pass directly — pos[(kx KY + kyi) KY + kzi] is the sample's place in the
// list (bit 31: the sign (-1)^(kx+ky+kz) of the origin shift; 0x7fffffff: not in the band), F the sub-volumes' sample arrays — instead of
// going back into B for k_sva_gather16 to pick them out (33 MB of scattered reads per 192^3 sub-volume)
struct SvaYZ16P { const float2 *A; float2 *B; const float2 *tw; int n, L, KX, KY, R, in_place; long nlines;
                  const unsigned *pos; float2 *F; int S; const double *stats; const float2 *Fw; };

__global__ void __launch_bounds__(256) k_sva_yz16(SvaYZ16P P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float2 tw_s[512];
    float2 *buf = (float2 *)smem;
    const int tid = threadIdx.x, n = P.n, LS = n + 1, KX = P.KX, KY = P.KY, R = P.R;
    for (int e = tid; e < n; e += 256) tw_s[e] = P.tw[e];
    const bool prune = KY < n;
    auto need = [&](int k) { return !prune || k <= R || k >= n - R; };
    if (P.in_place) {
        const long l0 = (long)blockIdx.x * P.L;
        const int nl = (int)((P.nlines - l0) < P.L ? (P.nlines - l0) : P.L);
        if (nl <= 0) return;
        float2 *base = P.B + l0 * n;
        {   // two complex values per load (n is even; B is 16-byte aligned and so is every line of it)
            const float4 *b4 = (const float4 *)base;
            const int n2 = n >> 1;
            for (int i2 = tid; i2 < nl * n2; i2 += 256) {
                const int line = i2 / n2, e = (i2 - line * n2) << 1;
                const float4 v = b4[i2];
                buf[line * LS + e] = make_float2(v.x, v.y); buf[line * LS + e + 1] = make_float2(v.z, v.w);
            }
        }
        __syncthreads();
        if (P.pos) {
            const long per = (long)KX * KY;
            const double n3 = (double)n * n * n;
            fft16m_any<false>(buf, nl, n, LS, tw_s, tid, need, [&](int line, int k, float2 val) {
                const long gl = l0 + line, v = gl / per, r = gl - v * per;
                const int kzi = (!prune || k <= R) ? k : k - n + KY;
                const unsigned u = P.pos[r * KY + kzi];
                if (u == 0x7fffffffu) return;
                const unsigned i = u & 0x7fffffffu;
                const float sg = (u >> 31) ? -1.f : 1.f;
                float2 o = make_float2(val.x * sg, val.y * sg);
                if (P.stats) {
                    const double mu = P.stats[2 * v] / n3, var = P.stats[2 * v + 1] / n3 - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
                    const float fmu = (float)mu, finv = (float)(1.0 / sd);
                    const float2 w = P.Fw[i];
                    o = make_float2((o.x - fmu * w.x) * finv, (o.y - fmu * w.y) * finv);
                }
                P.F[(size_t)v * P.S + i] = o;
            });
        } else
            fft16m_any<false>(buf, nl, n, LS, tw_s, tid, need, [&](int line, int k, float2 val) { base[(long)line * n + k] = val; });
    } else {
        const int zblocks = n / P.L;
        const int zb = blockIdx.x % zblocks, kx = (blockIdx.x / zblocks) % KX;
        const long v = blockIdx.x / ((long)zblocks * KX);
        const int z0 = zb * P.L, nl = P.L;
        {
            const int n2 = n >> 1;
            for (int i2 = tid; i2 < nl * n2; i2 += 256) {
                const int line = i2 / n2, e = (i2 - line * n2) << 1;
                const float4 v4 = *(const float4 *)(P.A + ((v * n + z0 + line) * KX + kx) * (long)n + e);
                buf[line * LS + e] = make_float2(v4.x, v4.y); buf[line * LS + e + 1] = make_float2(v4.z, v4.w);
            }
        }
        __syncthreads();
        float2 *dst = P.B + ((v * KX + kx) * KY) * (long)n + z0;
        fft16m_any<false>(buf, nl, n, LS, tw_s, tid, need, [&](int line, int k, float2 val) {
            const int kyi = (!prune || k <= R) ? k : k - n + KY;
            dst[(long)kyi * n + line] = val;
        });
    }
}

// band-limited half-space transform of one sub-volume out of B[kx][kyi][kz] (grid.y = sub-volume of the batch)
// stats / Fw (may be null): the sub-volume's sum and sum of squares and the window's own transform at the samples — the
// transform in B is that of the raw windowed volume (k_sva_x16 mode 1), the normalised one is (B - mean Fw) / sigma
__global__ void k_sva_gather16(const float2 *__restrict__ B, const uint32_t *__restrict__ samples, int S, int N, int KX, int KY, float2 *__restrict__ F,
                               const double *__restrict__ stats, const float2 *__restrict__ Fw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    B += (size_t)blockIdx.y * KX * KY * N; F += (size_t)blockIdx.y * S;
    int kx, ky, kz; sva_unpack(samples[i], kx, ky, kz);
    const int kyi = ky >= 0 ? ky : ky + KY;
    const float2 v = B[((size_t)kx * KY + kyi) * N + ((kz + N) % N)];
    const float sg = ((kx + ky + kz) & 1) ? -1.f : 1.f;
    float2 o = make_float2(v.x * sg, v.y * sg);
    if (stats) {
        const double n3 = (double)N * N * N, mu = stats[2 * blockIdx.y] / n3, var = stats[2 * blockIdx.y + 1] / n3 - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
        const float fmu = (float)mu, finv = (float)(1.0 / sd);
        const float2 w = Fw[i];
        o = make_float2((o.x - fmu * w.x) * finv, (o.y - fmu * w.y) * finv);
    }
    F[i] = o;
}

// band-limited half-space transform of one sub-volume out of the compact [z][y][KX] array, origin moved to the box centre
// (grid.y = sub-volume of the batch)
__global__ void k_sva_gather(const float2 *__restrict__ f, const uint32_t *__restrict__ samples, int S, int N, int KX, float2 *__restrict__ F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    f += (size_t)blockIdx.y * N * N * KX; F += (size_t)blockIdx.y * S;
    int kx, ky, kz; sva_unpack(samples[i], kx, ky, kz);
    const float2 v = f[((size_t)((kz + N) % N) * N + ((ky + N) % N)) * KX + kx];
    const float sg = ((kx + ky + kz) & 1) ? -1.f : 1.f;
    F[i] = make_float2(v.x * sg, v.y * sg);
}

struct SvaEvalP {
    CubeView cv; const uint32_t *samples; const float *bandw; const float2 *F; int S, N;
    int S_used; float rmax2;
    int ncand, nrot, use_wedge;   // candidates per volume; 1 .. nrot are the rotated ones (nrot = the kernel's NROT)
    int tabR;                     // radius of the LDS address tables (ppm_dev.h)
    const float *wedges;          // [n_vol][2]
    const double *poses;          // [n_vol][12] N row-major + shift
    const double *delta;          // [n_vol][ncand][6]
    double *out;                  // [n_vol][ncand]
    const int *vmap;              // null, or [n_states]: the sub-volume (transform, wedge) a state belongs to; poses / delta / out are per STATE
    double *partial;              // [n_states][kSvaParts][2 kMaxCand + 1]: every block's sums (k_sva_finish combines them in part order)
};
// A state's samples are dealt out over kSvaParts blocks (grid.y): one block per sub-volume left the chip at two waves per SIMD
// with every gather's latency exposed.  The number is a constant, so a sub-volume's result does not depend on its batch.
constexpr int kSvaParts = 4;

// Block = one sub-volume: thread q < ncand derives candidate q's pose in double precision (rotations about the specimen
// axes, then the shift, exactly like a particle unit of k_csp_eval).  Candidate layout (host): 0 = the unit's own pose (or the
// trial pose), 1 .. NROT = rotated candidates (one gather each), the rest keep candidate 0's rotation (shift variants: they
// share its gather).  Every thread accumulates its samples' sums in registers (static indices); the block combines them
// through LDS in a fixed order.
// Round 4, after the sweep of k_local (ppm_kernels2.h): NROT is a template parameter (0: translation-only and trial sweeps, 6: a
// compass sweep), so the gathers of rotation c + 1 are issued into a second register set before rotation c is interpolated, with no
// condition around a fetch; tap addresses come from the LDS tables of ppm_dev.h; the sub-volume's sample is turned by the conjugate
// phase of the shift once (the rotated candidates keep the unit's shift: checked per block) so that a score term is two instructions
// on the raw interpolated value.  128 registers instead of 208: four waves per SIMD instead of two.
#ifdef PPM_SVA_SERIAL        // A/B probe: one rotation's taps at a time (the next rotation finds the lines of this one in L1 if few waves share the CU)
constexpr bool kSvaSerial = true;
#else
constexpr bool kSvaSerial = false;
#endif
#ifndef PPM_SVA_EVAL_MINW
#define PPM_SVA_EVAL_MINW 4      // blocks of 256 threads per CU the register allocation leaves room for
#endif
template <int NROT>
__global__ void __launch_bounds__(256, PPM_SVA_EVAL_MINW) k_sva_eval(SvaEvalP P) {
    static_assert(NROT >= 0 && NROT < kMaxGroup, "rotated candidates");
    __shared__ float cm[kMaxCand][9], csh[kMaxCand][3];
    __shared__ float red[4][2 * kMaxCand + 1];
    __shared__ int same_shift;
    extern __shared__ __attribute__((aligned(8))) char tabmem[];
    const int st = blockIdx.x, part = blockIdx.y, v = P.vmap ? P.vmap[st] : st, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ncand = P.ncand;
    if (tid < ncand) {
        const double *d = P.delta + ((size_t)st * ncand + tid) * 6, *pose = P.poses + (size_t)st * 12;
        double Nm[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Nm[k] = pose[k];
        for (int k = 0; k < 3; k++)
            if (d[k] != 0.0) { double R[9], T[9]; d_rot_xyz(k, d[k], R); d_mat_mul3(Nm, R, T); for (int q = 0; q < 9; q++) Nm[q] = T[q]; }
#pragma unroll
        for (int k = 0; k < 9; k++) cm[tid][k] = (float)Nm[k];
        for (int k = 0; k < 3; k++) csh[tid][k] = (float)(pose[9 + k] + d[3 + k]);
    }
    const CubeTab tab = cube_tab_fill(P.cv, tabmem, P.tabR, tid, 256);
    __syncthreads();
    if (tid == 0) {
        int same = 1;
        for (int c = 1; c <= NROT; c++) same &= (csh[c][0] == csh[0][0] && csh[c][1] == csh[0][1] && csh[c][2] == csh[0][2]) ? 1 : 0;
        same_shift = same;
    }
    __syncthreads();
    const bool same = same_shift != 0;
    const float lw = P.wedges[2 * v], uw = P.wedges[2 * v + 1], invN = 1.0f / (float)P.N;
    const float2 *F = P.F + (size_t)v * P.S;
    float A[kMaxCand], B[kMaxGroup], Csum = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxCand; c++) A[c] = 0.f;
#pragma unroll
    for (int g = 0; g < kMaxGroup; g++) B[g] = 0.f;
    // The sample loop, once for the usual case (the rotated candidates keep candidate 0's shift: SAME) and once for the general one: a
    // branch INSIDE the sequence of groups splits it into basic blocks, and the compiler then sinks every interpolation below all
    // fetches (seven tap sets live, spills) - the sequence has to be straight-line code.
    auto sample_loop = [&](auto same_tag) {
    constexpr bool SAME = decltype(same_tag)::value;
    constexpr int C0 = 0, NC = NROT + 1;
    constexpr bool LEAD = true;
    const int s_first = part * 256 + tid, s_step = 256 * kSvaParts;
    for (int s = s_first; s < P.S_used; s += s_step) {
        int kx, ky, kz; sva_unpack(P.samples[s], kx, ky, kz);
        float w = P.bandw[s];
        if (!((float)(kx * kx + ky * ky + kz * kz) < P.rmax2)) w = 0.f;
        if (P.use_wedge && !(kx == 0 && kz == 0)) {
            float a = atan2f((float)kz, (float)kx) * 57.29577951308232f;
            if (a > 90.f) a -= 180.f;
            if (a <= -90.f) a += 180.f;
            if (!(a >= lw && a <= uw)) w = 0.f;
        }
        if (w == 0.f) continue;         // adds exact zeros; a wave's 64 consecutive samples are a compact patch of a shell (Z-order), so whole waves skip the missing wedge
        const float2 iv = F[s];
        const float fkx = (float)kx, fky = (float)ky, fkz = (float)kz;
        const float wx = w * iv.x, wy = w * iv.y;
        if constexpr (LEAD) Csum += wx * iv.x + wy * iv.y;
        int opq = 0;
        asm volatile("" : "+v"(opq));           // (see `opaque` below: LDS reads that must not be hoisted out of the sample loop)
        auto turned = [&](int c, float &bx, float &by) {        // conj of (weighted sample x e^{-i phase of candidate c's shift}): score term = bx p.x + by p.y
            const float *sh = csh[c] + opq;
            float rev = (fkx * sh[0] + fky * sh[1] + fkz * sh[2]) * invN;
            rev -= floorf(rev);
            const float sn = __sinf(6.283185307179586f * rev), cs = __cosf(6.283185307179586f * rev);
            bx = wx * cs + wy * sn; by = wy * cs - wx * sn;
        };
        float b0x, b0y;
        turned(0, b0x, b0y);
        // the candidates' matrices stay in LDS: read through an offset the compiler cannot see through, or it keeps all 7 x 9 (+ 13 x 3 shifts)
        // loop-invariant values in registers for the whole sample loop (that, not the taps, made the round-3 kernel a 208-register one)
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        auto fetch = [&](int c) {
            const float *m = cm[c] + opaque;
            return cube_fetch_tab(P.cv, tab, m[0] * fkx + m[1] * fky + m[2] * fkz, m[3] * fkx + m[4] * fky + m[5] * fkz, m[6] * fkx + m[7] * fky + m[8] * fkz);
        };
        float2 p0 = make_float2(0.f, 0.f);
        CubeTaps T0 = fetch(C0), T1;
        // rotation C0 + K is scored while the taps of rotation C0 + K + 1 are in flight (T0 / T1 alternate; all indices are compile-time constants)
#define PPM_SVA_GROUP(K, CUR, NXT)                                                                                          \
        if constexpr ((K) < NC) {                                                                                           \
            constexpr int C = C0 + (K);                                                                                     \
            if constexpr (!kSvaSerial && (K) + 1 < NC) { NXT = fetch(C + 1); __builtin_amdgcn_sched_barrier(0); }           \
            if constexpr (kSvaSerial && (K) > 0) { CUR = fetch(C); __builtin_amdgcn_sched_barrier(0); }                     \
            float bx = b0x, by = b0y;                                                                                       \
            if constexpr (C > 0 && !SAME) turned(C, bx, by);                                                                \
            const float2 p = cube_interp(CUR);                                                                              \
            if constexpr (C == 0) p0 = p;                                                                                   \
            B[C] += w * (p.x * p.x + p.y * p.y);                                                                            \
            A[C] += bx * p.x + by * p.y;                                                                                    \
            asm volatile("" : "+v"(A[C]), "+v"(B[C]));      /* the sums exist HERE: the compiler otherwise sinks every interpolation below the */ \
                                                            /* branches at the end of the body, i.e. below all fetches (all tap sets live) */ \
            __builtin_amdgcn_sched_barrier(0);      /* the fetch after next stays behind this group: two tap sets live, not three */ \
        }
        PPM_SVA_GROUP(0, T0, T1) PPM_SVA_GROUP(1, T1, T0) PPM_SVA_GROUP(2, T0, T1) PPM_SVA_GROUP(3, T1, T0)
        PPM_SVA_GROUP(4, T0, T1) PPM_SVA_GROUP(5, T1, T0) PPM_SVA_GROUP(6, T0, T1)
#undef PPM_SVA_GROUP
        if constexpr (LEAD) {
#pragma unroll
            for (int c = NROT + 1; c < kMaxCand; c++)
                if (c < ncand) { float bx, by; turned(c, bx, by); A[c] += bx * p0.x + by * p0.y; }
        }
    }
    };
    if (same) sample_loop(std::true_type{}); else sample_loop(std::false_type{});
#pragma unroll
    for (int c = 0; c < kMaxCand; c++) { const float t = wave_sum(A[c]); if (lane == 0) red[wave][c] = t; }
#pragma unroll
    for (int g = 0; g < kMaxGroup; g++) { const float t = wave_sum(B[g]); if (lane == 0) red[wave][kMaxCand + g] = t; }
    { const float t = wave_sum(Csum); if (lane == 0) red[wave][2 * kMaxCand] = t; }
    __syncthreads();
    if (tid < 2 * kMaxCand + 1)
        P.partial[((size_t)st * kSvaParts + part) * (2 * kMaxCand + 1) + tid] = (((double)red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// scores of k_sva_eval's candidates from the kSvaParts partial sums of every state, added in part order
__global__ void k_sva_finish(const double *__restrict__ partial, int n_states, int ncand, int nrot, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states * ncand) return;
    const int st = i / ncand, q = i - st * ncand, g = (q >= 1 && q <= nrot) ? q : 0;
    const double *p = partial + (size_t)st * kSvaParts * (2 * kMaxCand + 1);
    double a = 0, b = 0, c = 0;
    for (int k = 0; k < kSvaParts; k++, p += 2 * kMaxCand + 1) { a += p[q]; b += p[kMaxCand + g]; c += p[2 * kMaxCand]; }
    out[i] = (b > 0 && c > 0) ? a / sqrt(b * c) : 0.0;
}

// Global rotational grid (ppm_sva_cfg.search_mode 1, include/ppm.h): block = (sub-volume, run of RC grid rotations).  A rotation's
// rank is the correlation of the AMPLITUDES |F(k)| and |Ref(N k)| over the coarse band (a shift only moves phases): one gather
// per sample and rotation, three sums.
struct SvaGlobalP {
    CubeView cv; const uint32_t *samples; const float *bandw; const float2 *F; int S, N, S_used; float rmax2; int use_wedge;
    const float *wedges; const double *poses;     // [n_vol][2], [n_vol][12]
    const float *grid;                            // [n_grid][9] grid rotations G (row-major)
    int n_grid, RC;
    float *score;                                 // [n_vol][n_grid]
};

__global__ void __launch_bounds__(256) k_sva_global(SvaGlobalP P) {
    __shared__ float Nq[9], red[4][3];
    const int v = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float lw = P.wedges[2 * v], uw = P.wedges[2 * v + 1];
    const float2 *F = P.F + (size_t)v * P.S;
    const int q0 = blockIdx.x * P.RC, q1 = min(q0 + P.RC, P.n_grid);
    for (int q = q0; q < q1; q++) {
        __syncthreads();
        if (tid < 9) {                       // Nq = N0 G in double, like the oracle
            const int i = tid / 3, j = tid % 3;
            double a = 0;
            for (int k = 0; k < 3; k++) a += P.poses[(size_t)v * 12 + i * 3 + k] * (double)P.grid[(size_t)q * 9 + k * 3 + j];
            Nq[tid] = (float)a;
        }
        __syncthreads();
        float A = 0.f, B = 0.f, Csum = 0.f;
        for (int s = tid; s < P.S_used; s += 256) {
            int kx, ky, kz; sva_unpack(P.samples[s], kx, ky, kz);
            float w = P.bandw[s];
            if (!((float)(kx * kx + ky * ky + kz * kz) < P.rmax2)) w = 0.f;
            if (P.use_wedge && !(kx == 0 && kz == 0)) {
                float a = atan2f((float)kz, (float)kx) * 57.29577951308232f;
                if (a > 90.f) a -= 180.f;
                if (a <= -90.f) a += 180.f;
                if (!(a >= lw && a <= uw)) w = 0.f;
            }
            if (w == 0.f) continue;
            const float2 iv = F[s];
            const float fkx = (float)kx, fky = (float)ky, fkz = (float)kz;
            const float2 p = sample_cube(P.cv, Nq[0] * fkx + Nq[1] * fky + Nq[2] * fkz, Nq[3] * fkx + Nq[4] * fky + Nq[5] * fkz, Nq[6] * fkx + Nq[7] * fky + Nq[8] * fkz);
            const float m2 = p.x * p.x + p.y * p.y, f2 = iv.x * iv.x + iv.y * iv.y;
            A += w * sqrtf(m2 * f2); B += w * m2; Csum += w * f2;
        }
        { float t = wave_sum(A); if (lane == 0) red[wave][0] = t; t = wave_sum(B); if (lane == 0) red[wave][1] = t; t = wave_sum(Csum); if (lane == 0) red[wave][2] = t; }
        __syncthreads();
        if (tid == 0) {
            const double a = (((double)red[0][0] + red[1][0]) + red[2][0]) + red[3][0], b = (((double)red[0][1] + red[1][1]) + red[2][1]) + red[3][1],
                         c = (((double)red[0][2] + red[1][2]) + red[2][2]) + red[3][2];
            P.score[(size_t)v * P.n_grid + q] = (b > 0 && c > 0) ? (float)(a / sqrt(b * c)) : 0.f;
        }
    }
}


// ---------------------------------------------------------------------------------- sub-tomogram average (ppm_sva_insert)
// The 3-D analogue of K7 as a GATHER: one thread per voxel q of the accumulator's half space (|q| < N/2 - 1; on the kx = 0 plane
// only the canonical half: its Friedel mates are filled in by the fold of ppm_finalize) walks the sub-volumes of the batch, samples
// each one's transform at k = N^T q by trilinear interpolation (Hermitian mates for kx < 0), removes the shift's phase and adds
// value and weight (the sub-volume's missing-wedge mask at k) to its own sums - no atomics, the same sums in the same order whatever
// the launch shape.  The voxel's cell of the half-map the sub-volume belongs to (parity of its index) is read and written once.
// T: the batch's full transforms, layout 1 = B[v][kx][kyi][kz] of the two-step passes (KX = N/2 + 1, KY = N), layout 0 = f[v][z][y][KX].
struct SvaInsP {
    const float2 *T; int layout, N, KX, KY, nv;
    const double *poses;      // [nv][12] N row-major + shift (pixels)
    const float *wedges;      // [nv][2]
    const double *stats;      // null (the transform is that of the normalised volume), or [nv][2] sum and sum of squares of the raw volume
    const int *half;          // [nv] 0 / 1
    int use_wedge; float scale;
    float *acc;
};
constexpr int kSvaInsBatch = 32;

__global__ void __launch_bounds__(256) k_sva_insert(SvaInsP P) {
    __shared__ float sNt[kSvaInsBatch][9], ssh[kSvaInsBatch][3], swd[kSvaInsBatch][2], sinv[kSvaInsBatch];
    __shared__ int shalf[kSvaInsBatch];
    const int tid = threadIdx.x, N = P.N, NX = N / 2 + 1;
    if (tid < P.nv) {
        const double *pose = P.poses + (size_t)tid * 12;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) sNt[tid][i * 3 + j] = (float)pose[j * 3 + i];          // N^T
        for (int k = 0; k < 3; k++) ssh[tid][k] = (float)pose[9 + k];
        swd[tid][0] = P.wedges[2 * tid]; swd[tid][1] = P.wedges[2 * tid + 1];
        float inv = P.scale;
        if (P.stats) {
            const double n3 = (double)N * N * N, mu = P.stats[2 * tid] / n3, var = P.stats[2 * tid + 1] / n3 - mu * mu;
            inv *= (float)(1.0 / (var > 0 ? sqrt(var) : 1.0));
        }
        sinv[tid] = inv; shalf[tid] = P.half[tid];
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + tid, tot = (size_t)N * N * NX;
    if (i >= tot) return;
    const int qx = (int)(i % NX), qy = (int)((i / NX) % N) - N / 2, qz = (int)(i / ((size_t)NX * N)) - N / 2;
    const float rmax = (float)(N / 2 - 1);
    const float q2 = (float)(qx * qx + qy * qy + qz * qz);
    if (!(q2 < rmax * rmax) || q2 == 0.f) return;
    if (qx == 0 && (qy < 0 || (qy == 0 && qz < 0))) return;
    const float fqx = (float)qx, fqy = (float)qy, fqz = (float)qz, invN = 1.0f / (float)N;
    float ar0 = 0.f, ai0 = 0.f, aw0 = 0.f, ar1 = 0.f, ai1 = 0.f, aw1 = 0.f;      // the two half-maps' sums (scalars: run-time indexed arrays would live in scratch)
    auto fetch = [&](const float2 *Tv, int x, int y, int z) {            // F(x, y, z), origin at the box centre
        const bool mate = x < 0;
        if (mate) { x = -x; y = -y; z = -z; }
        if ((x | y | z) == 0) return make_float2(0.f, 0.f);      // the transform is that of the RAW volume when P.stats is set: its DC term
                                                                 // (mean x N^3) is the one coefficient the normalisation (v - mean) / sigma changes - to zero
        const int yi = y < 0 ? y + N : y, zi = z < 0 ? z + N : z;
        const float2 v = P.layout ? Tv[((size_t)x * P.KY + yi) * N + zi] : Tv[((size_t)zi * N + yi) * P.KX + x];
        const float sg = ((x + y + z) & 1) ? -1.f : 1.f;
        return make_float2(v.x * sg, mate ? -v.y * sg : v.y * sg);
    };
    const size_t per = P.layout ? (size_t)P.KX * P.KY * N : (size_t)N * N * P.KX;
    for (int v = 0; v < P.nv; v++) {
        const float *m = sNt[v];
        const float kx = m[0] * fqx + m[1] * fqy + m[2] * fqz, ky = m[3] * fqx + m[4] * fqy + m[5] * fqz, kz = m[6] * fqx + m[7] * fqy + m[8] * fqz;
        if (P.use_wedge) {
            float a = atan2f(kz, kx) * 57.29577951308232f;
            if (a > 90.f) a -= 180.f;
            if (a <= -90.f) a += 180.f;
            if (!(a >= swd[v][0] && a <= swd[v][1])) continue;
        }
        const float xf = floorf(kx), yf = floorf(ky), zf = floorf(kz);
        const int x0 = (int)xf, y0 = (int)yf, z0 = (int)zf;
        const float fx = kx - xf, fy = ky - yf, fz = kz - zf;
        const float2 *Tv = P.T + (size_t)v * per;
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int dz = 0; dz < 2; dz++)
#pragma unroll
            for (int dy = 0; dy < 2; dy++)
#pragma unroll
                for (int dx = 0; dx < 2; dx++) {
                    const float wt = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy) * (dz ? fz : 1.f - fz);
                    const float2 t = fetch(Tv, x0 + dx, y0 + dy, z0 + dz);
                    sr += wt * t.x; si += wt * t.y;
                }
        float rev = -(kx * ssh[v][0] + ky * ssh[v][1] + kz * ssh[v][2]) * invN;       // F(k) = Ref(N k) e^{+2 pi i k.p / N}: take the shift out
        rev -= floorf(rev);
        const float sn = __sinf(6.283185307179586f * rev), cs = __cosf(6.283185307179586f * rev);
        const float s = sinv[v], vr = s * (sr * cs - si * sn), vi = s * (sr * sn + si * cs);
        if (shalf[v]) { ar1 += vr; ai1 += vi; aw1 += 1.f; } else { ar0 += vr; ai0 += vi; aw0 += 1.f; }      // block-uniform branch
    }
    const size_t half_sz = tot * 3;
    if (aw0 > 0.f) { float *o = P.acc + i * 3; o[0] += ar0; o[1] += ai0; o[2] += aw0; }
    if (aw1 > 0.f) { float *o = P.acc + half_sz + i * 3; o[0] += ar1; o[1] += ai1; o[2] += aw1; }
}

}  // namespace ppm
