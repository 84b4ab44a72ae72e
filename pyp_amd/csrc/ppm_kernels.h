// ppm_kernels.h — the HIP kernels of libpypmatch (gfx950 / CDNA4, wave64).
//
//   k_fft_lines      batched strided 1-D complex FFT through LDS (3-D transforms of volumes)
//   k_ref_load/crop  reference preparation (sinc^2 pre-compensation, centring, band-limited cube)
//   k_prep           per-particle pre-processing: normalise, mask, 2-D real FFT in LDS (two rows
//                    packed per complex transform, band-limited columns), ring whitening, CTF tables
//   k_bank           central slices of the global-search grid (shared by all particles)
//   k_global         pruned correlation image of every (particle, orientation): lane = kx, rows
//                    stream over ky, wavefront-shuffle reduction of the shift window, top-K hits
//   k_local          ring-wise weighted correlation at arbitrary poses + compass refinement
//   k_insert_bricks  CTF-weighted trilinear insertion into the half-map accumulators, brick by brick in LDS (ppm_kernels2.h)
//   finalise         shell statistics, Wiener division, mask and gridding correction
#pragma once
#include "ppm_dev.h"

namespace ppm {

// ---------------------------------------------------------------------------------- 3-D FFT passes
struct FftLinesP {
    float2 *data; FftPlan plan;
    int n, inverse, line_major, L;
    long nlines, inner, inner_stride, outer_stride, elem_stride;
};

__global__ void __launch_bounds__(256) k_fft_lines(FftLinesP P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ long lbase[16];                       // first element of every line of the block (P.L <= 16)
    __shared__ float2 tw_s[512];
    float2 *buf = (float2 *)smem;
    const int tid = threadIdx.x, n = P.n, LS = n + 1;      // odd line stride: line-major passes touch 16 lines at the same element
    long l0 = (long)blockIdx.x * P.L;
    int nl = (int)((P.nlines - l0) < P.L ? (P.nlines - l0) : P.L);
    if (nl <= 0) return;
    if (tid < nl) { const long l = l0 + tid; lbase[tid] = (l / P.inner) * P.outer_stride + (l % P.inner) * P.inner_stride; }
    for (int i = tid; i < n; i += 256) tw_s[i] = P.plan.tw[i];       // the stages' twiddles from LDS instead of dependent global loads
    __syncthreads();
    // (line, e) of element i = tid, tid + 256, ... without a division per element
    const int dl = P.line_major ? 256 % nl : 256 / n, de = P.line_major ? 256 / nl : 256 % n;
    const int line0 = P.line_major ? tid % nl : tid / n, e0 = P.line_major ? tid / nl : tid % n;
    auto next = [&](int &line, int &e) {
        line += dl; e += de;
        if (P.line_major) { if (line >= nl) { line -= nl; e++; } } else if (e >= n) { e -= n; line++; }
    };
    {
        int line = line0, e = e0;
        for (int i = tid; i < nl * n; i += 256, next(line, e)) buf[line * LS + P.plan.perm[e]] = P.data[lbase[line] + e * P.elem_stride];
    }
    lds_fft(buf, P.plan, nl, LS, P.inverse != 0, tid, 256, tw_s);
    {
        int line = line0, e = e0;
        for (int i = tid; i < nl * n; i += 256, next(line, e)) P.data[lbase[line] + e * P.elem_stride] = buf[line * LS + e];
    }
}

// ---------------------------------------------------------------------------------- reference
// volume / sinc^2 (trilinear pre-compensation at the padded sampling), embedded in the centre of a zeroed (np)^3 box
__global__ void k_ref_load(const float *__restrict__ vol, float2 *__restrict__ f, int n, int np) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n3 = (size_t)n * n * n;
    if (i >= n3) return;
    int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((size_t)n * n));
    float g = 1.f;
    int c[3] = { x, y, z };
#pragma unroll
    for (int q = 0; q < 3; q++) {
        float u = kPiF * (float)(c[q] - n / 2) / (float)np;
        float sv = fabsf(u) < 1e-6f ? 1.f : sinf(u) / u;
        g *= 1.f / (sv * sv);
    }
    const int o0 = (np - n) / 2;
    f[((size_t)(z + o0) * np + (y + o0)) * np + (x + o0)] = make_float2(vol[i] * g, 0.f);
}

// ringw (may be null): radial weight per Fourier pixel of the UNPADDED transform, [nw], linearly interpolated ("use statistics")
__global__ void k_ref_crop(const float2 *__restrict__ f, float2 *__restrict__ cube, int n, int n_orig, int B, int CX, int CY, int NBX, int NBY, unsigned LB,
                           const float *__restrict__ ringw, int nw) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, tot = (size_t)CX * CY * CY;
    if (i >= tot) return;
    const int x = (int)(i % CX), ys = (int)((i / CX) % CY), zs = (int)(i / ((size_t)CX * CY));      // stored indices
    const int y = ys - (B + 1), z = zs - (B + 1);
    int iz = ((z % n) + n) % n, iy = ((y % n) + n) % n, ix = x % n;
    float2 v = f[((size_t)iz * n + iy) * n + ix];
    float sg = (((x + y + z) & 1) ? -1.f : 1.f) / (float)n_orig;
    if (ringw) {
        const float kk = sqrtf((float)(x * x + y * y + z * z)) * (float)n_orig / (float)n;      // radius in unpadded Fourier pixels
        const int k0 = (int)kk;
        const float fr = kk - (float)k0;
        sg *= k0 + 1 < nw ? ringw[k0] + fr * (ringw[k0 + 1] - ringw[k0]) : ringw[nw - 1];
    }
    v = make_float2(v.x * sg, v.y * sg);
    cube[cube_element(NBX, NBY, LB, 0, x, ys, zs)] = v;                     // both copies of the blocked layout (ppm_dev.h)
    if (x >= 2) cube[cube_element(NBX, NBY, LB, 1, x, ys, zs)] = v;
}

// ---------------------------------------------------------------------------------- pre-processing
struct PrepP {
    const float *images; const double *rows; FftPlan plan;
    int N, B, W, H;
    float r_hi2, Rm, wfall, a;
    float focus[4];   // focus mask: sphere centre (pixels from the box centre) and radius in the reference; radius <= 0: centred mask of radius Rm
    int normalize, invert, do_mask, whiten;
    int nc, nchunks, L;
    float2 *band;  // [n][H*W] unscaled band spectrum (scratch; the final result for insertion)
    int TS, WS;    // line strides of the column buffer T and of the row work buffer Wk (>= 272 on the 256 fast path)
    int fast256;   // N == 256: register-level 16 x 16 FFT (lds_fft256), natural-order staging
    int inreg;     // N == 256, k_prep<512, 2> only: the half spectrum stays in registers between the two phases (no global scratch)
    unsigned *band_max; // may be null: bits of max |re|, |im| over the band images of the launch (atomicMax; floats >= 0)
    float2 *spill; // [n][N][W] row-transformed half spectrum (global scratch between the row and the column phase)
    float *wring;  // [n][B+2] ring weights 1/sqrt(mean power), may be null
    // ring-ordered list outputs (may be null)
    const uint32_t *samples; int S_pad; float2 *Il; float *cw;
    // search-layout outputs (may be null): [n][Hs*64]
    float2 *Wp; float *C2; float *nI; int Bs, Hs; float r_s2, r_lo2;
};

// PT = 512 threads and <= 80 KB of LDS: two blocks per CU, so that one block's global-memory waits (image reads, the
// spilled half spectrum) overlap the other's FFT work; PT = 1024 / 160 KB is kept for comparison (PPM_PREP_PT).
// MINW = waves per SIMD the register allocation has to leave room for: 4 with PT = 512 makes two blocks per CU resident
// (128 VGPRs, some spills), 1 lets the compiler keep everything in registers (one block per CU)
template <int PT, int MINW>
__global__ void __launch_bounds__(PT, MINW) k_prep(PrepP P) {
    constexpr int PW = PT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, N = P.N, B = P.B, W = P.W, H = P.H;
    const int p = blockIdx.x;
    const int TS = P.TS, WS = P.WS;                   // padded line strides (bank spread)
    float2 *T = (float2 *)smem;                       // [nc][TS] column chunk, followed by
    constexpr bool kInreg = (PT == 512 && MINW == 2) || (PT == 1024 && MINW == 1);   // the instantiations that carry the scratch-free path
    constexpr int HR = PT / 128, NIT = 64 / (HR > 0 ? HR : 1);   // scratch-free path: thread t holds column t & 127 of the rows (t >> 7) + HR it, it < NIT, of every row pass
    const bool inreg = kInreg && P.inreg;
    float2 *Wk = inreg ? T : T + (size_t)P.nc * TS;   // [L][WS]  the row work buffer (shares T's storage on the scratch-free path)
    // ring power sums in 64-bit fixed point and integer counts: a ds_add_f32 costs ~190 LDS cycles per wave-instruction on
    // gfx950, ds_add_u64 / ds_add_u32 ~8 / ~5 (scripts/micro/lds_atomic_bench.hip)
    const size_t regionA = inreg ? ((size_t)P.nc * TS > (size_t)P.L * WS ? (size_t)P.nc * TS : (size_t)P.L * WS) : (size_t)P.nc * TS + (size_t)P.L * WS;
    unsigned long long *ringq = (unsigned long long *)(T + regionA);  // [B+2]
    unsigned *ringc = (unsigned *)(ringq + (B + 2));  // [B+2]
    float *ringpw = (float *)(ringc + (B + 2));       // [B+2] ring weights
    double *red = (double *)(((uintptr_t)(ringpw + (B + 2)) + 15) & ~(uintptr_t)15);  // [PW*4 + PW]
    float *stat = (float *)(red + PW * 5);            // mu, scale, fixed-point scale, nI partials
    float *fmask = stat + 4 + PW;                     // mask disc of this particle: centre (pixels from the box centre) and radius; [4], [5]: beam-tilt phase coefficients
    float2 *tw_s = (float2 *)(stat + 12 + PW);        // [N] twiddles and [N] staging positions of the FFT plan, kept in LDS
    unsigned short *perm_s = (unsigned short *)(tw_s + N);
    unsigned short *iperm_s = perm_s + N;             // inverse: the sample that is staged at LDS position d
    for (int i = tid; i < N; i += PT) {
        tw_s[i] = P.plan.tw[i];
        const unsigned short q = P.fast256 ? (unsigned short)i : P.plan.perm[i];
        perm_s[i] = q; iperm_s[q] = (unsigned short)i;
    }
    __syncthreads();
    const float *img = P.images + (size_t)p * N * N;

    // ---- statistics of the background (outside the mask radius); whole image if that is empty
    double s1 = 0, s2 = 0, cnt = 0, t1 = 0, t2 = 0;
    const float Rm2 = P.Rm * P.Rm;
    // scratch-free path without a mask: (x - mu) sc only changes the DC term (never used) and the scale, so the statistics are
    // gathered while the pixels are staged for the row transforms and the scale is applied to the band output: one image read
    const bool fold = inreg && !P.do_mask;
    if (!fold) {
        const float invNf = 1.0f / (float)N;
        const float4 *img4 = (const float4 *)img;          // N even: N^2 is a multiple of 4, every image starts 16-byte aligned
        constexpr int UR = 8;                              // independent 16-byte loads in flight per thread
        const bool rows4 = (N & 3) == 0;                   // then the 4 pixels of a 16-byte load share their image row
        for (int b4 = tid; b4 < N * N / 4; b4 += PT * UR) {
            float4 qv[UR];
#pragma unroll
            for (int u = 0; u < UR; u++) { const int i4 = b4 + u * PT; if (i4 < N * N / 4) qv[u] = img4[i4]; }
            // the <= 32 pixels of one trip are summed in single precision, the trips in double (fp64 adds issue at a fraction
            // of the fp32 rate: five of them per pixel made this pass a fifth of the kernel's vector instructions)
            float f1 = 0.f, f2 = 0.f, fc = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int u = 0; u < UR; u++) {
                const int i4 = b4 + u * PT;
                if (i4 >= N * N / 4) continue;
                const float vv[4] = { qv[u].x, qv[u].y, qv[u].z, qv[u].w };
                const int i0 = 4 * i4, y0 = fast_div(i0, N, invNf), x0 = i0 - y0 * N;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int y = y0, x = x0 + j;
                    if (!rows4 && x >= N) { x -= N; y += 1; }
                    const float dx = (float)(x - N / 2), dy = (float)(y - N / 2), v = vv[j];
                    g1 += v; g2 = fmaf(v, v, g2);
                    if (dx * dx + dy * dy > Rm2) { f1 += v; f2 = fmaf(v, v, f2); fc += 1.f; }
                }
            }
            s1 += (double)f1; s2 += (double)f2; cnt += (double)fc; t1 += (double)g1; t2 += (double)g2;
        }
    }
    auto finish_stats = [&]() {     // block reduction of the partial sums -> stat[0..2] (mean, scale, fixed-point scale) and the mask / beam-tilt constants
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); cnt = wave_sum_d(cnt); t1 = wave_sum_d(t1); t2 = wave_sum_d(t2);
    if ((tid & 63) == 0) { int w = tid >> 6; red[w * 4] = s1; red[w * 4 + 1] = s2; red[w * 4 + 2] = cnt; red[w * 4 + 3] = t1; }
    __syncthreads();
    if (tid == 0) {
        double a1 = 0, a2 = 0, ac = 0, b1 = 0;
        for (int w = 0; w < PW; w++) { a1 += red[w * 4]; a2 += red[w * 4 + 1]; ac += red[w * 4 + 2]; b1 += red[w * 4 + 3]; }
        red[0] = a1; red[1] = a2; red[2] = ac; red[3] = b1;
    }
    __syncthreads();
    {
        double a1 = red[0], a2 = red[1], ac = red[2], b1 = red[3];
        __syncthreads();
        if ((tid & 63) == 0) red[PW * 4 + (tid >> 6)] = t2;
        __syncthreads();
        if (tid == 0) {
            double b2 = 0;
            for (int w = 0; w < PW; w++) b2 += red[PW * 4 + w];
            if (ac < 16) { a1 = b1; a2 = b2; ac = (double)N * N; }
            double mu = a1 / ac, var = a2 / ac - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
            stat[0] = (float)mu;
            stat[1] = (float)((P.normalize ? 1.0 / sd : 1.0) * (P.invert ? -1.0 : 1.0));
            // Parseval: sum_k |F_k / N|^2 = sum_x v_x^2 <= E, so no ring sum (weights 1 or 2) exceeds 2 E; scale = 2^(60-e), 4E < 2^e
            const double sc2 = (double)stat[1] * (double)stat[1];
            const double E = sc2 * (b2 - 2.0 * mu * b1 + (double)N * N * mu * mu);
            int e2 = 0; (void)frexp(E > 1e-30 ? 4.0 * E : 1.0, &e2);
            int ex = 60 - e2; ex = ex > 120 ? 120 : (ex < -120 ? -120 : ex);
            stat[2] = ldexpf(1.f, ex);
            {   // beam tilt (milliradians): phi(k) = k^2 (kx c_x + ky c_y) with c = 2 pi Cs lambda^2 b 1e-3 / (N a)^3
                const double *row = P.rows + (size_t)p * PPM_NCOL;
                const double v = row[PPM_VOLTAGE] * 1000.0, lam = 12.2639 / sqrt(v + 0.97845e-6 * v * v), na = (double)N * (double)P.a;
                const double cc = 2.0 * 3.14159265358979323846 * row[PPM_CS] * 1e7 * lam * lam * 1e-3 / (na * na * na);
                fmask[4] = (float)(cc * row[PPM_BTX]); fmask[5] = (float)(cc * row[PPM_BTY]);
            }
            // mask disc: centred with radius Rm, or around the projection of the focus sphere at the row's pose
            fmask[0] = 0.f; fmask[1] = 0.f; fmask[2] = P.Rm;
            if (P.focus[3] > 0.f) {
                const double *row = P.rows + (size_t)p * PPM_NCOL;
                const double d2r = 3.14159265358979323846 / 180.0;
                double sps, cps, sth, cth, sph, cph;
                sincos(row[PPM_PSI] * d2r, &sps, &cps); sincos(row[PPM_THETA] * d2r, &sth, &cth); sincos(row[PPM_PHI] * d2r, &sph, &cph);
                const double c0 = P.focus[0], c1 = P.focus[1], c2 = P.focus[2];
                // first two rows of M^T, M = Rz(phi) Ry(theta) Rz(psi)
                fmask[0] = (float)((cph * cth * cps - sph * sps) * c0 + (sph * cth * cps + cph * sps) * c1 - sth * cps * c2 + row[PPM_XSHIFT] / (double)P.a);
                fmask[1] = (float)((-cph * cth * sps - sph * cps) * c0 + (-sph * cth * sps + cph * cps) * c1 + sth * sps * c2 + row[PPM_YSHIFT] / (double)P.a);
                fmask[2] = P.focus[3];
            }
        }
        __syncthreads();
    }
    };
    if (!fold) finish_stats();
    float mu = fold ? 0.f : stat[0], sc = fold ? 1.f : stat[1], qscale = fold ? 1.f : stat[2], oscale = 1.f;    // fold: set after the row passes
    float mcx = fold ? 0.f : fmask[0], mcy = fold ? 0.f : fmask[1], mrad = fold ? 0.f : fmask[2];
    float btx = fold ? 0.f : fmask[4], bty = fold ? 0.f : fmask[5];
    bool beam_tilt = btx != 0.f || bty != 0.f;
    for (int i = tid; i < B + 2; i += PT) { ringq[i] = 0ull; ringc[i] = 0u; }

    const float wf = P.wfall < 1e-3f ? 1e-3f : P.wfall;
    float2 *bandp = P.band + (size_t)p * H * W;
    float2 *sp = P.spill + (size_t)p * N * W;         // [N][W] half spectrum after the row pass (global scratch, L2-resident)
    float omax = 0.f;
    // band output of `ncol` transformed columns starting at c0 (in T): origin to the box centre, 1 / N, beam-tilt phase, ring power sums
    auto emit = [&](int c0, int ncol) {
        const float inv_ncol = 1.0f / (float)ncol, invN = 1.f / (float)N;
        for (int i = tid; i < ncol * H; i += PT) {
            const int row = fast_div(i, ncol, inv_ncol), c = i - row * ncol, ky = row - B, kx = c0 + c;
            float k2 = (float)(kx * kx + ky * ky);
            float2 o = make_float2(0.f, 0.f);
            if (k2 < P.r_hi2 && k2 > 0.f) {
                float2 v = T[c * TS + (ky < 0 ? ky + N : ky)];
                float sg = ((kx + ky) & 1) ? -invN * oscale : invN * oscale;
                o = make_float2(v.x * sg, v.y * sg);
                if (beam_tilt) {                                 // remove the beam-tilt phase error: x exp(-i phi)
                    float sn, cs;
                    sincosf(k2 * ((float)kx * btx + (float)ky * bty), &sn, &cs);
                    o = make_float2(o.x * cs + o.y * sn, o.y * cs - o.x * sn);
                }
                omax = fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y)));
                if (P.whiten) {                                  // ring power sums are only read by the whitening weights
                    int b = (int)floorf(sqrtf(k2));
                    float al = kx == 0 ? 1.f : 2.f;
                    atomicAdd(&ringq[b], (unsigned long long)__double2ll_rn((double)(al * (o.x * o.x + o.y * o.y)) * (double)qscale));
                    atomicAdd(&ringc[b], kx == 0 ? 1u : 2u);
                }
            }
            bandp[row * W + kx] = o;
        }
    };
    auto mask_px = [&](int x, int y) {
        const float dx = (float)(x - N / 2) - mcx, dy = (float)(y - N / 2) - mcy;
        const float r = sqrtf(dx * dx + dy * dy);
        return r >= mrad + 0.5f * wf ? 0.f : (r > mrad - 0.5f * wf ? 0.5f * (1.f + cosf(kPiF * (r - mrad + 0.5f * wf) / wf)) : 1.f);
    };
    bool done = false;
    if constexpr (kInreg) {
        if (inreg) {
            // ---- scratch-free path (N = 256, 512 threads): two row passes of 64 row pairs through LDS; thread t then HOLDS column
            // t & 127 of the half spectrum for the rows 128 pass + 2 ((t >> 7) + 4 it) + {0, 1}: 64 complex values in registers.
            // The column pass assembles 64 columns at a time in LDS from those registers.  kx = 128 (and everything beyond the
            // band) is zero in the output, so 128 columns are all that is ever held.
            float2 holdA[NIT][2], holdB[NIT][2];     // pass 0 / pass 1 (two arrays: each small enough to be promoted to registers)
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int y0 = 128 * pass;
                lds_barrier();                                   // the previous pass has been read out
#pragma unroll 2
                for (int k = 0; k < 4096 / PT; k++) {            // 64 row pairs x 64 float4 = 4096 pairs of loads over the block's threads
                    const int i4 = tid + k * PT, l = i4 >> 6, x = 4 * (i4 & 63), ya = y0 + 2 * l, yb = ya + 1;
                    const float4 qa = *(const float4 *)(img + ya * N + x), qb = *(const float4 *)(img + yb * N + x);
                    const float ra[4] = { qa.x, qa.y, qa.z, qa.w }, rb[4] = { qb.x, qb.y, qb.z, qb.w };
                    float f1 = 0.f, f2 = 0.f, fc = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float va = (ra[j] - mu) * sc, vb = (rb[j] - mu) * sc;
                        if (P.do_mask) { va *= mask_px(x + j, ya); vb *= mask_px(x + j, yb); }
                        Wk[l * WS + x + j] = make_float2(va, vb);
                        if (fold) {
                            const float dx = (float)(x + j - N / 2), dya = (float)(ya - N / 2), dyb = dya + 1.f;
                            g1 += va + vb; g2 = fmaf(va, va, fmaf(vb, vb, g2));
                            if (dx * dx + dya * dya > Rm2) { f1 += va; f2 = fmaf(va, va, f2); fc += 1.f; }
                            if (dx * dx + dyb * dyb > Rm2) { f1 += vb; f2 = fmaf(vb, vb, f2); fc += 1.f; }
                        }
                    }
                    if (fold) { s1 += (double)f1; s2 += (double)f2; cnt += (double)fc; t1 += (double)g1; t2 += (double)g2; }
                }
                lds_fft256(Wk, 64, WS, tid, PT, tw_s);
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int l = (tid >> 7) + HR * it, kx = tid & 127;
                    const float2 z = Wk[l * WS + kx], zc = Wk[l * WS + (kx ? N - kx : 0)];
                    const float2 d = make_float2(z.x - zc.x, z.y + zc.y);
                    const float2 xa = make_float2(0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y)), xb = make_float2(0.5f * d.y, -0.5f * d.x);
                    if (pass == 0) { holdA[it][0] = xa; holdA[it][1] = xb; } else { holdB[it][0] = xa; holdB[it][1] = xb; }
                }
            }
            if (fold) {                                          // the statistics are complete: scales for the output
                lds_barrier();
                finish_stats();
                oscale = stat[1]; qscale = stat[2];
                btx = fmask[4]; bty = fmask[5]; beam_tilt = btx != 0.f || bty != 0.f;
            }
            const int ncols = W < 128 ? W : 128;
            for (int c0 = 0; c0 < ncols; c0 += 64) {
                const int ncol = ncols - c0 < 64 ? ncols - c0 : 64;
                lds_barrier();
                const int kx = tid & 127;
                if (kx >= c0 && kx < c0 + ncol) {
#pragma unroll
                    for (int pass = 0; pass < 2; pass++)
#pragma unroll
                        for (int it = 0; it < NIT; it++) {
                            const int y = 128 * pass + 2 * ((tid >> 7) + HR * it);
                            T[(kx - c0) * TS + y] = pass == 0 ? holdA[it][0] : holdB[it][0];
                            T[(kx - c0) * TS + y + 1] = pass == 0 ? holdA[it][1] : holdB[it][1];
                        }
                }
                lds_fft256(T, ncol, TS, tid, PT, tw_s);
                emit(c0, ncol);
            }
            for (int i = tid; i < (W - ncols) * H; i += PT)       // columns the band never reaches (kx = 128)
                bandp[(i / (W - ncols)) * W + ncols + i % (W - ncols)] = make_float2(0.f, 0.f);
            done = true;
        }
    }
    if (!done) {
    // ---- row pass: two real rows per complex transform; every row of the half spectrum goes to the scratch
    {
        {
            // the pixels of the NEXT pass are fetched into registers while this pass runs its FFT (host: L N <= 8 PT).
            // N % 4 == 0: 16-byte loads of 4 consecutive pixels of both rows, staged at their plan positions; otherwise
            // 4-byte loads gathered so that consecutive lanes write consecutive LDS positions.
            constexpr int MAXI = 8;
            float2 pre[MAXI];
            float4 pa[MAXI / 4], pb[MAXI / 4];
            const float invNf = 1.0f / (float)N, invN4 = 4.0f / (float)N;
            const bool wide = (N & 3) == 0;
            auto mask_at = [&](int x, int y) {
                const float dx = (float)(x - N / 2) - mcx, dy = (float)(y - N / 2) - mcy;
                const float r = sqrtf(dx * dx + dy * dy);
                return r >= mrad + 0.5f * wf ? 0.f : (r > mrad - 0.5f * wf ? 0.5f * (1.f + cosf(kPiF * (r - mrad + 0.5f * wf) / wf)) : 1.f);
            };
            auto fetch = [&](int y0) {
                if (wide) {
#pragma unroll
                    for (int k = 0; k < MAXI / 4; k++) {
                        const int i4 = tid + k * PT;
                        if (i4 < P.L * N / 4) {
                            const int l = fast_div(i4, N / 4, invN4), x = 4 * (i4 - l * (N / 4)), ya = y0 + 2 * l;
                            pa[k] = *(const float4 *)(img + ya * N + x); pb[k] = *(const float4 *)(img + (ya + 1) * N + x);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < MAXI; k++) {
                        const int i = tid + k * PT;
                        if (i < P.L * N) {
                            const int l = fast_div(i, N, invNf), d = i - l * N, x = iperm_s[d], ya = y0 + 2 * l;
                            pre[k] = make_float2(img[ya * N + x], img[(ya + 1) * N + x]);
                        }
                    }
                }
            };
            fetch(0);
            for (int y0 = 0; y0 < N; y0 += 2 * P.L) {
                lds_barrier();
                if (wide) {
#pragma unroll
                    for (int k = 0; k < MAXI / 4; k++) {
                        const int i4 = tid + k * PT;
                        if (i4 < P.L * N / 4) {
                            const int l = fast_div(i4, N / 4, invN4), x = 4 * (i4 - l * (N / 4));
                            const int ya = y0 + 2 * l, yb = ya + 1;
                            const float ra[4] = { pa[k].x, pa[k].y, pa[k].z, pa[k].w }, rb[4] = { pb[k].x, pb[k].y, pb[k].z, pb[k].w };
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                float va = (ra[j] - mu) * sc, vb = (rb[j] - mu) * sc;
                                if (P.do_mask) { va *= mask_at(x + j, ya); vb *= mask_at(x + j, yb); }
                                Wk[l * WS + perm_s[x + j]] = make_float2(va, vb);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < MAXI; k++) {
                        const int i = tid + k * PT;
                        if (i < P.L * N) {
                            const int l = fast_div(i, N, invNf), d = i - l * N, x = iperm_s[d];
                            const int ya = y0 + 2 * l, yb = ya + 1;
                            float va = (pre[k].x - mu) * sc, vb = (pre[k].y - mu) * sc;
                            if (P.do_mask) { va *= mask_at(x, ya); vb *= mask_at(x, yb); }
                            Wk[l * WS + d] = make_float2(va, vb);
                        }
                    }
                }
                if (y0 + 2 * P.L < N) fetch(y0 + 2 * P.L);
                if (P.fast256) lds_fft256(Wk, P.L, WS, tid, PT, tw_s);
                else lds_fft(Wk, P.plan, P.L, WS, false, tid, PT, tw_s);
                const float invW = 1.0f / (float)W;
                for (int i = tid; i < P.L * W; i += PT) {
                    const int l = fast_div(i, W, invW), kx = i - l * W;
                    float2 z = Wk[l * WS + kx], zc = Wk[l * WS + (kx ? N - kx : 0)];
                    float2 xa = make_float2(0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y));
                    float2 d = make_float2(z.x - zc.x, z.y + zc.y);
                    float2 xb = make_float2(0.5f * d.y, -0.5f * d.x);
                    const int ya = y0 + 2 * l;
                    sp[(size_t)ya * W + kx] = xa;
                    sp[(size_t)(ya + 1) * W + kx] = xb;
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();                                  // the scratch rows written above are read by other threads below
    // the NEXT chunk's columns are fetched into registers while this chunk is transformed (host: N nc <= MAXC threads)
    constexpr int MAXC = 12;
    float2 nx[MAXC];
    auto fetch_chunk = [&](int ch) {
        const int c0 = ch * P.nc, ncol = (W - c0) < P.nc ? (W - c0) : P.nc;
        const float inv_ncol = 1.0f / (float)ncol;
#pragma unroll
        for (int k = 0; k < MAXC; k++) {
            const int i = tid + k * PT;
            if (i < N * ncol) {
                const int y = fast_div(i, ncol, inv_ncol), c = i - y * ncol;
                nx[k] = sp[(size_t)y * W + c0 + c];
            }
        }
    };
    fetch_chunk(0);
    for (int ch = 0; ch < P.nchunks; ch++) {
        const int c0 = ch * P.nc, ncol = (W - c0) < P.nc ? (W - c0) : P.nc;
        const float inv_ncol = 1.0f / (float)ncol;
        lds_barrier();
#pragma unroll
        for (int k = 0; k < MAXC; k++) {                 // this chunk's columns, staged at their plan positions
            const int i = tid + k * PT;
            if (i < N * ncol) {
                const int y = fast_div(i, ncol, inv_ncol), c = i - y * ncol;
                T[c * TS + perm_s[y]] = nx[k];
            }
        }
        if (ch + 1 < P.nchunks) fetch_chunk(ch + 1);
        // ---- column pass
        if (P.fast256) lds_fft256(T, ncol, TS, tid, PT, tw_s);
        else lds_fft(T, P.plan, ncol, TS, false, tid, PT, tw_s);
        emit(c0, ncol);
    }
    }
    if (P.band_max) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o, 64));
        if ((tid & 63) == 0 && omax > 0.f && omax < 3.0e38f) atomicMax(P.band_max, __float_as_uint(omax));
    }
    __threadfence_block();
    __syncthreads();
    // ---- ring weights (re-using ringpw as the weight table)
    for (int b = tid; b < B + 2; b += PT) {
        float pw = ringc[b] > 0u ? (float)((double)ringq[b] / (double)qscale / (double)ringc[b]) : 0.f;
        float wgt = P.whiten ? (pw > 0.f ? rsqrtf(pw) : 0.f) : 1.f;
        ringpw[b] = wgt;
        if (P.wring) P.wring[(size_t)p * (B + 2) + b] = wgt;
    }
    __syncthreads();
    CtfP ctf = ctf_from_row(P.rows + (size_t)p * PPM_NCOL, N, (double)P.a);
    if (P.Il) {
        float2 *Ilp = P.Il + (size_t)p * P.S_pad;
        float *cwp = P.cw + (size_t)p * P.S_pad;
        for (int s = tid; s < P.S_pad; s += PT) {
            int kx, ky, al, ring;
            unpack_sample(P.samples[s], kx, ky, al, ring);
            float2 v = make_float2(0.f, 0.f); float c = 0.f;
            if (al) {
                float wgt = ringpw[ring];
                float2 u = bandp[(ky + B) * W + kx];
                v = make_float2(u.x * wgt, u.y * wgt);
                c = ctf_eval(ctf, kx, ky) * wgt;
            }
            Ilp[s] = v; cwp[s] = c;
        }
    }
    if (P.Wp) {
        float2 *Wpp = P.Wp + (size_t)p * P.Hs * 64;
        float *C2p = P.C2 + (size_t)p * P.Hs * 64;
        float ni = 0.f;
        for (int i = tid; i < P.Hs * 64; i += PT) {
            int kx = i & 63, ky = (i >> 6) - P.Bs;
            float k2 = (float)(kx * kx + ky * ky);
            float2 wv = make_float2(0.f, 0.f); float c2 = 0.f;
            if (kx <= P.Bs && k2 < P.r_s2 && k2 >= P.r_lo2 && k2 > 0.f) {
                int b = (int)floorf(sqrtf(k2));
                float wgt = ringpw[b], al = kx == 0 ? 1.f : 2.f;
                float2 u = bandp[(ky + B) * W + kx];
                float2 v = make_float2(u.x * wgt, u.y * wgt);
                float c = ctf_eval(ctf, kx, ky) * wgt;
                wv = make_float2(al * c * v.x, al * c * v.y);
                c2 = al * c * c;
                ni += al * (v.x * v.x + v.y * v.y);
            }
            Wpp[i] = wv; C2p[i] = c2;
        }
        ni = wave_sum(ni);
        __syncthreads();
        if ((tid & 63) == 0) stat[3 + (tid >> 6)] = ni;
        __syncthreads();
        if (tid == 0) { float t = 0.f; for (int w = 0; w < PW; w++) t += stat[3 + w]; P.nI[p] = t; }
    }
}

// ---------------------------------------------------------------------------------- slice bank
struct BankP { CubeView cv; const float *mats; float2 *bank; int nslices, Bs, Hs; float r_s2; };

__global__ void __launch_bounds__(256) k_bank(BankP P) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, per = (size_t)P.Hs * 64;
    if (i >= per * P.nslices) return;
    int sl = (int)(i / per), r = (int)(i - (size_t)sl * per);
    // rows are stored in the paired order of k_global: row 0 = ky 0, row 1 = empty, row 2t = +t, row 2t+1 = -t
    int kx = r & 63, rr = r >> 6, t = rr >> 1;
    int ky = (rr & 1) ? -t : t;
    float k2 = (float)(kx * kx + ky * ky);
    float2 v = make_float2(0.f, 0.f);
    if (rr != 1 && t <= P.Bs && kx <= P.Bs && k2 < P.r_s2) {
        const float *m = P.mats + (size_t)sl * 6;
        float fx = (float)kx, fy = (float)ky;
        v = sample_cube(P.cv, m[0] * fx + m[1] * fy, m[2] * fx + m[3] * fy, m[4] * fx + m[5] * fy);
    }
    P.bank[i] = v;
}

// ---------------------------------------------------------------------------------- slice norms (MFMA)
// nP[p][s] = sum_k C2_p(k) |P_s(k)|^2, the reference-side norm of every (particle, slice) pair: a dense product of the
// particles' CTF^2 tables (n x samples) with the squared slice bank (samples x slices) — the one GEMM-shaped piece of the
// search, so it runs on the matrix cores (v_mfma_f32_32x32x2_f32, fp32 in and out) instead of costing k_global six vector
// instructions per row pair and lane.  Block = 4 waves, tile 128 particles x 128 slices, K tile = half a paired bank row
// (32 kx); each wave owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers).  LDS rows are padded to 33 floats: the
// operand reads (lane = row, two k per instruction) then touch 32 distinct banks.
struct NormP { const float *C2; const float2 *bank; float *nP; int n, nslices, Bs, Hs, HsP; };
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256, 2) k_slice_norms(NormP P) {
    constexpr int TM = 128, TK = 32, LD = TK + 1;
    __shared__ float As[TM * LD], Bq[TM * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int p0 = blockIdx.x * TM, s0 = blockIdx.y * TM;
    const size_t HS = (size_t)P.Hs * 64, HSP = (size_t)P.HsP * 64;
    const int nhalf = P.Bs >= 32 ? 2 : 1;                         // kx > Bs holds zeros only
    v16f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int v = 0; v < 16; v++) acc[a][b][v] = 0.f;
    // this thread's share of a tile load: A 4 x float4 (row = idx / 8), B 8 x float4 = 16 complex (row = idx / 16);
    // rows beyond the tables are clamped (their products are never stored)
    for (int rr = 0; rr < 2 * P.Bs + 2; rr++) {
        if (rr == 1) continue;                                     // paired order: row 1 is empty
        const int t = rr >> 1, ky = (rr & 1) ? -t : t;
        const size_t aoff = (size_t)(ky + P.Bs) * 64, boff = (size_t)rr * 64;
        for (int h = 0; h < nhalf; h++) {
            float4 av[4], bv[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = tid + 256 * q, row = idx >> 3, c4 = idx & 7;
                const int pr = min(p0 + row, P.n - 1);
                av[q] = *(const float4 *)(P.C2 + (size_t)pr * HS + aoff + h * TK + c4 * 4);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int idx = tid + 256 * q, row = idx >> 4, c = idx & 15;
                const int sr = min(s0 + row, P.nslices - 1);
                bv[q] = *(const float4 *)(P.bank + (size_t)sr * HSP + boff + h * TK + c * 2);
            }
            __syncthreads();                                       // the previous tile has been consumed
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = tid + 256 * q, row = idx >> 3, c4 = idx & 7;
                float *d = As + row * LD + c4 * 4;
                d[0] = av[q].x; d[1] = av[q].y; d[2] = av[q].z; d[3] = av[q].w;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int idx = tid + 256 * q, row = idx >> 4, c = idx & 15;
                float *d = Bq + row * LD + c * 2;
                d[0] = fmaf(bv[q].x, bv[q].x, bv[q].y * bv[q].y); d[1] = fmaf(bv[q].z, bv[q].z, bv[q].w * bv[q].w);
            }
            __syncthreads();
            const float *ap = As + (64 * wm + (lane & 31)) * LD + (lane >> 5);
            const float *bp = Bq + (64 * wn + (lane & 31)) * LD + (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < TK; kk += 2) {
                const float a0 = ap[kk], a1 = ap[32 * LD + kk], b0 = bp[kk], b1 = bp[32 * LD + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }
    // accumulator register v of lane l: row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32 of the 32 x 32 tile
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int sc = s0 + 64 * wn + 32 * b + (lane & 31);
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int pr = p0 + 64 * wm + 32 * a + 8 * (v >> 2) + 4 * (lane >> 5) + (v & 3);
                if (pr < P.n && sc < P.nslices) P.nP[(size_t)pr * P.nslices + sc] = acc[a][b][v];
            }
        }
}

// ---------------------------------------------------------------------------------- global search
struct Hit { float cc; int orient; int sx, sy; };

// Shift windows wider than the kernel's (tiles, ppm_refine_batch): the search tables of a chunk are multiplied by the phase ramp
// e^{+2 pi i (kx dcx + ky dcy) / Ns}, which moves the window's centre by (dcx, dcy) search-grid steps.  Wp: [n][Hs][64].
__global__ void __launch_bounds__(256) k_wp_ramp(float2 *Wp, size_t total, int Bs, int Ns, int dcx, int dcy, const float2 *twN) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int kx = (int)(i & 63), ky = (int)((i >> 6) % (size_t)(2 * Bs + 1)) - Bs;
    const int t = (((kx * dcx + ky * dcy) % Ns) + Ns) % Ns;
    const float2 w = twN[t], v = Wp[i];
    Wp[i] = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
}

// Top-K of the union of the tiles' top-K lists (tiles: [ntiles][n][K], shifts relative to the tile centres cx / cy): an
// orientation's score is its best over the tiles; ties -> lower orientation index, like the single-window selection.  The true
// top-K of the whole window is contained in the union (an orientation that beats fewer than K others overall beats fewer than K
// in the tile that holds its maximum).  One thread per particle.
__global__ void k_merge_hits(Hit *tiles, Hit *out, int n, int K, int ntiles, const int *cx, const int *cy) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    for (int k = 0; k < K; k++) {
        int bt = -1, bj = -1; float bc = -3.0e38f; int bo = 0x7fffffff;
        for (int t = 0; t < ntiles; t++) {
            const Hit *h = tiles + ((size_t)t * n + p) * K;
            for (int j = 0; j < K; j++) {
                if (h[j].orient < 0) continue;
                if (h[j].cc > bc || (h[j].cc == bc && h[j].orient < bo)) { bc = h[j].cc; bo = h[j].orient; bt = t; bj = j; }
            }
        }
        Hit o; o.cc = 0.f; o.orient = 0; o.sx = 0; o.sy = 0;
        if (bt >= 0) {
            o = tiles[((size_t)bt * n + p) * K + bj];
            o.sx += cx[bt]; o.sy += cy[bt];
            for (int t = 0; t < ntiles; t++) {          // strike the orientation out everywhere
                Hit *h = tiles + ((size_t)t * n + p) * K;
                for (int j = 0; j < K; j++) if (h[j].orient == bo) h[j].orient = -1;
            }
        }
        out[(size_t)p * K + k] = o;
    }
}

// Pair twiddles e^{+2 pi i t j / Ns}, [t][j-1] for the row pair ky = +-t: wave-uniform, fetched with scalar loads.
// Stored as {cos, cos, sin, sin} so that a scalar load delivers the two operand pairs of the packed FMAs as they are.
// The table belongs to the reference handle (two references with different search grids may be live at once); it is read
// through the constant address space so that the wave-uniform loads stay scalar (s_load) although the kernel also stores
// to global memory inside the slice loop.
constexpr int kRowTwRows = 64;
// build-time knobs of k_global's default (R <= 3) shape, for A/B runs (scripts/ab_global.sh)
#ifndef PPM_GLOBAL_PARTICLES
#define PPM_GLOBAL_PARTICLES 2
#endif
#ifndef PPM_GLOBAL_PAIR_MAXR
#define PPM_GLOBAL_PAIR_MAXR 8
#endif
#ifndef PPM_GLOBAL_THREADS
#define PPM_GLOBAL_THREADS 512
#endif
#ifndef PPM_GLOBAL_UNROLL
#define PPM_GLOBAL_UNROLL 4
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f *RowTwPtr;

struct GlobP {
    const float2 *bank; const float2 *Wp; const float *nP; const float *nI; const float2 *twN;  // nP: [n][n_dir * npsi_store] slice norms (k_slice_norms); twN: Ns-entry table e^{2 pi i t/Ns}
    const float4 *rowtw;  // [kRowTwRows][PPM_MAX_SHIFT_STEPS] row-pair twiddles (device memory owned by the reference)
    float *cc; int *sh;   // [n][n_orient] scratch scores and packed shifts
    Hit *hits;            // [n][K]
    int n, Bs, Hs, HsP, Ns, RSx, RSy, n_dir, n_psi, npsi_store, n_orient, K;   // n: particles of this launch
    int topk_lds;         // 1: the top-K pass works on an LDS copy of the particle's scores (they fit), 0: on the global scratch
};

// The same for pairs of values held in packed registers (both orientations of a stored slice): the swaps and DPP steps work
// on 32-bit registers, the adds between rows on the pair.
template <int M> __device__ __forceinline__ v2f halve_pair2(v2f a, v2f b, int lane) {
    float a0 = a.x, a1 = a.y, b0 = b.x, b1 = b.y;
    if constexpr (M >= 16) {
        lane_swap<M>(a0, b0); lane_swap<M>(a1, b1);
        return (v2f){ a0, a1 } + (v2f){ b0, b1 };
    } else {
        constexpr int ctrl = M == 8 ? kDppRor8 : M == 4 ? kDppHalfMirror : M == 2 ? kDppXor2 : kDppXor1;
        const bool hi = (lane & M) != 0;
        const float t0 = a0 + dpp_mov<ctrl>(a0), t1 = a1 + dpp_mov<ctrl>(a1), u0 = b0 + dpp_mov<ctrl>(b0), u1 = b1 + dpp_mov<ctrl>(b1);
        return (v2f){ hi ? u0 : t0, hi ? u1 : t1 };
    }
}
template <int NV>
__device__ __forceinline__ v2f reduce_halving2(v2f (&v)[NV], int lane) {
    static_assert(NV <= 64, "one output register pair");
    constexpr int n1 = (NV + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2, n5 = (n4 + 1) / 2;
    const v2f z = { 0.f, 0.f };
    v2f a1[n1], a2[n2], a3[n3], a4[n4], a5[n5];
#pragma unroll
    for (int i = 0; i < n1; i++) a1[i] = halve_pair2<32>(v[2 * i], (2 * i + 1 < NV) ? v[2 * i + 1] : z, lane);
#pragma unroll
    for (int i = 0; i < n2; i++) a2[i] = halve_pair2<16>(a1[2 * i], (2 * i + 1 < n1) ? a1[2 * i + 1] : z, lane);
#pragma unroll
    for (int i = 0; i < n3; i++) a3[i] = halve_pair2<8>(a2[2 * i], (2 * i + 1 < n2) ? a2[2 * i + 1] : z, lane);
#pragma unroll
    for (int i = 0; i < n4; i++) a4[i] = halve_pair2<4>(a3[2 * i], (2 * i + 1 < n3) ? a3[2 * i + 1] : z, lane);
#pragma unroll
    for (int i = 0; i < n5; i++) a5[i] = halve_pair2<2>(a4[2 * i], (2 * i + 1 < n4) ? a4[2 * i + 1] : z, lane);
    return halve_pair2<1>(a5[0], (n5 > 1) ? a5[1] : z, lane);
}

// The same inside each half of the wave (no stage across the halves): NV <= 32 values, lane l of a half ends up with the total of
// value index bitrev5(l & 31) over the half's 32 lanes.
template <int NV>
__device__ __forceinline__ v2f reduce_halving2h(v2f (&v)[NV], int lane) {
    static_assert(NV <= 32, "one output register pair per half");
    constexpr int n2 = (NV + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2, n5 = (n4 + 1) / 2;
    const v2f z = { 0.f, 0.f };
    v2f a2[n2], a3[n3], a4[n4], a5[n5];
#pragma unroll
    for (int i = 0; i < n2; i++) a2[i] = halve_pair2<16>(v[2 * i], (2 * i + 1 < NV) ? v[2 * i + 1] : z, lane);
#pragma unroll
    for (int i = 0; i < n3; i++) a3[i] = halve_pair2<8>(a2[2 * i], (2 * i + 1 < n2) ? a2[2 * i + 1] : z, lane);
#pragma unroll
    for (int i = 0; i < n4; i++) a4[i] = halve_pair2<4>(a3[2 * i], (2 * i + 1 < n3) ? a3[2 * i + 1] : z, lane);
#pragma unroll
    for (int i = 0; i < n5; i++) a5[i] = halve_pair2<2>(a4[2 * i], (2 * i + 1 < n4) ? a4[2 * i + 1] : z, lane);
    return halve_pair2<1>(a5[0], (n5 > 1) ? a5[1] : z, lane);
}

// Particles per block: every slice row a wave streams from the bank is used for NQ particles from registers (the bank is
// re-read by every block, 141 MB per block at the default grid: with one particle per block the L2 -> L1 path, not the vector
// unit, set the pace).  Two particles keep W tables of 2 x 64 KB in LDS; 512 threads (two waves per SIMD, 139 registers)
// measured faster than 768 or 1024.
constexpr int global_particles(int R) { return R <= PPM_GLOBAL_PAIR_MAXR ? PPM_GLOBAL_PARTICLES : 1; }
constexpr int global_threads(int R) { return R <= 3 ? PPM_GLOBAL_THREADS : 512; }   // wider windows need > 128 VGPRs
constexpr int global_unroll(int R) { return R <= 3 ? PPM_GLOBAL_UNROLL : 4; }         // rows in flight per wave (prefetch depth), even

// pairwise halving step of a cross-lane reduction: afterwards lanes with (lane & M) == 0 carry the partial sum of `a`, the
// others that of `b`.  Between rows the two registers trade halves (one swap, one add); inside a row both are folded
// with a DPP operand and the lane keeps the one it owns.
template <int M> __device__ __forceinline__ float halve_pair(float a, float b, int lane) {
    if constexpr (M >= 16) { lane_swap<M>(a, b); return a + b; }
    else {
        constexpr int ctrl = M == 8 ? kDppRor8 : M == 4 ? kDppHalfMirror : M == 2 ? kDppXor2 : kDppXor1;
        const float t = a + dpp_mov<ctrl>(a), u = b + dpp_mov<ctrl>(b);
        return (lane & M) ? u : t;
    }
}

// Sums NV per-lane values over the 64 lanes in 6 halving stages (NV + NV/2 + ... cross-lane steps instead of
// 6 NV).  Returns one register: lane l holds the total of value index bitrev6(l) (if that is < NV <= 64).
template <int NV>
__device__ __forceinline__ float reduce_halving(float (&v)[NV], int lane) {
    static_assert(NV <= 64, "one output register");
    constexpr int n1 = (NV + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2, n5 = (n4 + 1) / 2;
    float a1[n1], a2[n2], a3[n3], a4[n4], a5[n5];
#pragma unroll
    for (int i = 0; i < n1; i++) a1[i] = halve_pair<32>(v[2 * i], (2 * i + 1 < NV) ? v[2 * i + 1] : 0.f, lane);
#pragma unroll
    for (int i = 0; i < n2; i++) a2[i] = halve_pair<16>(a1[2 * i], (2 * i + 1 < n1) ? a1[2 * i + 1] : 0.f, lane);
#pragma unroll
    for (int i = 0; i < n3; i++) a3[i] = halve_pair<8>(a2[2 * i], (2 * i + 1 < n2) ? a2[2 * i + 1] : 0.f, lane);
#pragma unroll
    for (int i = 0; i < n4; i++) a4[i] = halve_pair<4>(a3[2 * i], (2 * i + 1 < n3) ? a3[2 * i + 1] : 0.f, lane);
#pragma unroll
    for (int i = 0; i < n5; i++) a5[i] = halve_pair<2>(a4[2 * i], (2 * i + 1 < n4) ? a4[2 * i + 1] : 0.f, lane);
    return halve_pair<1>(a5[0], (n5 > 1) ? a5[1] : 0.f, lane);
}

// Block = one particle.  The particle's CTF-weighted spectrum W sits in LDS for the whole orientation loop (the slice
// norms sum C2 |P|^2 come from k_slice_norms); each wave streams whole slices of the bank: lane = kx, rows = ky in +-t pairs, U rows
// prefetched ahead.  With P the slice sample, A = Re(P) W and Bq = Im(P) (Wy, -Wx): W conj(P) = A + Bq
// (orientation psi) and W P = A - Bq (psi + 180 deg).  The transforms over ky use the even / odd parts of a
// row pair: sum_ky Q e^{i 2 pi ky j/Ns} = sum_t (Q(+t) + Q(-t)) cos + i (Q(+t) - Q(-t)) sin, accumulated in the
// (A, Bq) basis — 8 FMA per shift row j and row PAIR for both orientations — and recombined once per slice;
// the sum over kx of the shift window is a wavefront halving reduction.
// TWO (search bands of at most 32 pixels, e.g. PYP's default 10 A limit on a 256 box): a wave takes TWO slices at once, lanes 0-31 one,
// lanes 32-63 the other (kx = lane & 31), so that no lane sits beyond the band; every cross-lane step then stays inside a half.
template <int R, bool HALF, bool TWO>
__global__ void __launch_bounds__(global_threads(R)) k_global(GlobP P) {
    constexpr int NT = global_threads(R), NW = NT / 64, U = global_unroll(R), NS = 2 * R + 1, NQ = global_particles(R);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, p0 = blockIdx.x * NQ;
    const int sub = TWO ? lane >> 5 : 0, kxl = TWO ? lane & 31 : lane;       // which of the wave's two slices, and the column, this lane works on
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: slice addresses stay in SGPRs
    const int Hs = P.Hs, HsP = P.HsP, Bs = P.Bs, Ns = P.Ns, nsampP = HsP * 64;
    const int nslices = P.n_dir * P.npsi_store;
    float2 *Wl = (float2 *)smem;                                    // [NQ][nsampP]
    int pq[NQ];                                                     // a short last block computes its last particle twice
#pragma unroll
    for (int q = 0; q < NQ; q++) pq[q] = min(p0 + q, P.n - 1);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const float2 *src = P.Wp + (size_t)pq[q] * Hs * 64;
        for (int i = tid; i < nsampP; i += NT) {
            const int rr = i >> 6, t = rr >> 1, ky = (rr & 1) ? -t : t;
            const bool ok = rr != 1 && t <= Bs;
            const int srow = ky + Bs;
            Wl[q * nsampP + i] = ok ? src[srow * 64 + (i & 63)] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    float nI[NQ];
    const __attribute__((address_space(4))) float *c_nP[NQ];        // wave-uniform reads: scalar loads
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        nI[q] = P.nI[pq[q]];
        c_nP[q] = (const __attribute__((address_space(4))) float *)P.nP + (size_t)pq[q] * nslices;
    }
    float txc[R + 1], txs[R + 1];      // per-lane x twiddles e^{+2 pi i kx j / Ns}
#pragma unroll
    for (int j = 0; j <= R; j++) {
        float2 t = P.twN[(kxl * j) & (Ns - 1)];
        txc[j] = t.x; txs[j] = t.y;
    }
    const RowTwPtr c_rowtw = (RowTwPtr)P.rowtw;
    // The first rows of a wave's NEXT slice are requested by the last step of the current one (and those of its first slice
    // here): no load stall at a slice start, and the prefetch of every step is unconditional (a conditional one made the
    // compiler copy the 8 row registers of the untouched set on every step).
    float2 pv[U], pn[U];
    const int nsl = TWO ? (nslices + 1) / 2 : nslices;             // trips of the slice loop over all waves
    auto my_slice = [&](int s) { return TWO ? min(2 * s + sub, nslices - 1) : s; };     // the slice this lane reads on trip s (an odd last one is read twice)
    if (wave < nsl) {
        const float2 *P0 = P.bank + (size_t)my_slice(wave) * nsampP;
#pragma unroll
        for (int u = 0; u < U; u++) pv[u] = P0[u * 64 + kxl];
    }
    for (int sl = wave; sl < nsl; sl += NW) {
        const int msl = my_slice(sl);
        const bool live = !TWO || 2 * sl + sub < nslices;
        const float2 *Pp = P.bank + (size_t)msl * nsampP;         // wave-uniform base (per half when TWO), column added as a 32-bit offset
        const float2 *Pnext = P.bank + (size_t)my_slice(sl + NW < nsl ? sl + NW : sl) * nsampP;
        // accumulators (packed re/im pairs) per particle: s* = sum over rows (shift row 0); per j: even part x cos (ua, ub),
        // odd part x sin (va, vb)
        v2f sa[NQ], sb[NQ], ua[NQ][R], ub[NQ][R], va[NQ][R], vb[NQ][R];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            sa[q] = sb[q] = (v2f){ 0.f, 0.f };
#pragma unroll
            for (int j = 0; j < R; j++) { ua[q][j] = ub[q][j] = va[q][j] = vb[q][j] = (v2f){ 0.f, 0.f }; }
        }
        // U rows per step; the rows of the next step are prefetched into the other register set (ping-pong: no copies).
        // B = Im(P) (Wx, Wy) is accumulated instead of Bq = Im(P) (Wy, -Wx) (no swizzled copy of W per row): component swap
        // and sign commute with the row sums and are applied once per slice below (Bq.x = B.y, Bq.y = -B.x).
        auto step = [&](int row0, const float2 (&cur)[U], float2 (&nxt)[U]) {
            {
                const float2 *np = (row0 + U < HsP) ? Pp + (row0 + U) * 64 : Pnext;     // scalar base; the row offsets below fit the 12-bit immediate
#pragma unroll
                for (int u = 0; u < U; u++) nxt[u] = np[u * 64 + kxl];
            }
            // the (wave-uniform) twiddles of the NEXT row pair are requested before this pair is consumed
            v4f tw[R], twn[R];
#pragma unroll
            for (int j = 0; j < R; j++) tw[j] = c_rowtw[(row0 >> 1) * PPM_MAX_SHIFT_STEPS + j];
            // ... and so are the particles' W rows (LDS) of the next pair
            float2 wa[NQ], wb[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) { wa[q] = Wl[q * nsampP + row0 * 64 + kxl]; wb[q] = Wl[q * nsampP + (row0 + 1) * 64 + kxl]; }
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                const int ra = row0 + u, tp = ra >> 1;
                float2 wan[NQ], wbn[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) { wan[q] = wa[q]; wbn[q] = wb[q]; }
                if (u + 2 < U) {
#pragma unroll
                    for (int j = 0; j < R; j++) twn[j] = c_rowtw[(tp + 1) * PPM_MAX_SHIFT_STEPS + j];
#pragma unroll
                    for (int q = 0; q < NQ; q++) { wan[q] = Wl[q * nsampP + (ra + 2) * 64 + kxl]; wbn[q] = Wl[q * nsampP + (ra + 3) * 64 + kxl]; }
                }
                const float pax = cur[u].x, pay = cur[u].y, pbx = cur[u + 1].x, pby = cur[u + 1].y;
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    // A = Re(P) W, B = Im(P) W for both rows; even (+) and odd (-) parts of the pair
                    const v2f wav = { wa[q].x, wa[q].y }, wbv = { wb[q].x, wb[q].y };
                    const v2f aa = wav * pax, ab = wbv * pbx, ba = wav * pay, bb = wbv * pby;
                    const v2f as2 = aa + ab, ad2 = aa - ab, bs2 = ba + bb, bd2 = ba - bb;
                    sa[q] += as2; sb[q] += bs2;
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        const v4f t = tw[j];                                          // wave-uniform -> SGPRs {c, c, s, s}
                        const v2f tc = { t.x, t.y }, ts = { t.z, t.w };
                        ua[q][j] += as2 * tc; ub[q][j] += bs2 * tc;
                        va[q][j] += ad2 * ts; vb[q][j] += bd2 * ts;
                    }
                }
#pragma unroll
                for (int j = 0; j < R; j++) tw[j] = twn[j];
#pragma unroll
                for (int q = 0; q < NQ; q++) { wa[q] = wan[q]; wb[q] = wbn[q]; }
            }
        };
        __builtin_amdgcn_s_setprio(3);
        for (int row0 = 0; row0 < HsP; row0 += 2 * U) { step(row0, pv, pn); step(row0 + U, pn, pv); }      // HsP is a multiple of 2 U
        __builtin_amdgcn_s_setprio(0);          // the reduction tail is a chain of dependent cross-lane steps: let it issue first
        const int dir = msl / P.npsi_store, ks = msl - dir * P.npsi_store;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float nP = c_nP[q][msl];
            const float sax = sa[q].x, say = sa[q].y, sbx = sb[q].y, sby = -sb[q].x;
            float uax[R], uay[R], ubx[R], uby[R], vax[R], vay[R], vbx[R], vby[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                uax[j] = ua[q][j].x; uay[j] = ua[q][j].y; ubx[j] = ub[q][j].y; uby[j] = -ub[q][j].x;
                vax[j] = va[q][j].x; vay[j] = va[q][j].y; vbx[j] = vb[q][j].y; vby[j] = -vb[q][j].x;
            }
            const float inv = (nP > 0.f && nI[q] > 0.f) ? rsqrtf(nP * nI[q]) : 0.f;
            float *ccp = P.cc + (size_t)pq[q] * P.n_orient; int *shp = P.sh + (size_t)pq[q] * P.n_orient;
            if constexpr (!TWO && HALF && NS * NS <= 64) {
                // both orientations of the stored slice at once, in the halves of packed registers (.x: psi, sg = +1; .y: psi + 180 deg,
                // sg = -1): Q = A + sg Bq, U = ua + sg ub, V = va + sg vb; G(+j) = U + iV, G(-j) = U - iV;
                // value(iy, ix) = Re(G[iy] e^{+2 pi i kx (ix-R)/Ns})
                const v2f sg2 = { 1.f, -1.f };
                v2f val[NS * NS];
#pragma unroll
                for (int iy = 0; iy < NS; iy++) {
                    const int jy = iy - R, ja = jy < 0 ? -jy : jy;
                    v2f gx, gy;
                    if (jy == 0) { gx = sax + sg2 * sbx; gy = say + sg2 * sby; }
                    else {
                        const v2f ux = uax[ja - 1] + sg2 * ubx[ja - 1], uy = uay[ja - 1] + sg2 * uby[ja - 1];
                        const v2f vx = vax[ja - 1] + sg2 * vbx[ja - 1], vy = vay[ja - 1] + sg2 * vby[ja - 1];
                        gx = jy > 0 ? ux - vy : ux + vy; gy = jy > 0 ? uy + vx : uy - vx;
                    }
                    val[iy * NS + R] = gx;
#pragma unroll
                    for (int j = 1; j <= R; j++) {
                        const v2f pc = gx * txc[j], qs = gy * txs[j];
                        val[iy * NS + R + j] = pc - qs;
                        val[iy * NS + R - j] = pc + qs;
                    }
                }
                const v2f tot = reduce_halving2<NS * NS>(val, lane);
                const int vi = (int)(__brev((unsigned)lane) >> 26);            // the value index this lane ended up with
                const int iy = vi / NS, ix = vi - iy * NS;
                const int ay = iy - R < 0 ? R - iy : iy - R, ax = ix - R < 0 ? R - ix : ix - R;
                const bool inwin = vi < NS * NS && ax <= P.RSx && ay <= P.RSy;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float cand = inwin ? (e ? tot.y : tot.x) : -3.0e38f;
                    // arg-max over lanes; ties -> lower (sy, sx) index like the oracle's scan order
                    const float best = wave_max(cand);
                    int ci = wave_min(cand == best ? vi : 64);
                    if (ci > 63) ci = 0;                                         // no comparable value (NaN scores)
                    const int bsy_ = ci / NS - R, bsx_ = ci - (ci / NS) * NS - R;
                    if (lane == 0 && (q == 0 || p0 + q < P.n)) {
                        int o = dir * P.n_psi + ks + e * P.npsi_store;
                        ccp[o] = best * inv;
                        shp[o] = (bsx_ & 0xffff) | (bsy_ << 16);
                    }
                }
            } else if constexpr (TWO || NS * NS > 64) {
                // wide windows (R > 3, up to 17 x 17 shifts): the window is reduced ROW BY ROW — the NS values of a window row (both
                // orientations in the halves of packed registers) go through one halving reduction, lane bitrev6(ix) ends up with the
                // row's total for shift column ix and keeps a running best over the rows.  (A wave-wide sum per shift — 289 of them per
                // orientation at R = 8 — kept so many values alive that the kernel spilled 1.8 KB per thread.)
                const v2f sg2 = { 1.f, -1.f };
                const int vi = TWO ? (int)(__brev((unsigned)kxl) >> 27) : (int)(__brev((unsigned)lane) >> 26), ix = vi;
                const int ax = ix - R < 0 ? R - ix : ix - R;
                const bool incol = vi < NS && ax <= P.RSx;
                float bestv[2] = { -3.0e38f, -3.0e38f }; int besty[2] = { 0, 0 };
#pragma unroll
                for (int iy = 0; iy < NS; iy++) {
                    const int jy = iy - R, ja = jy < 0 ? -jy : jy;
                    v2f gx, gy;
                    if (jy == 0) { gx = sax + sg2 * sbx; gy = say + sg2 * sby; }
                    else {
                        const v2f ux = uax[ja - 1] + sg2 * ubx[ja - 1], uy = uay[ja - 1] + sg2 * uby[ja - 1];
                        const v2f vx = vax[ja - 1] + sg2 * vbx[ja - 1], vy = vay[ja - 1] + sg2 * vby[ja - 1];
                        gx = jy > 0 ? ux - vy : ux + vy; gy = jy > 0 ? uy + vx : uy - vx;
                    }
                    v2f val[NS];
                    val[R] = gx;
#pragma unroll
                    for (int j = 1; j <= R; j++) {
                        const v2f pc = gx * txc[j], qs = gy * txs[j];
                        val[R + j] = pc - qs;
                        val[R - j] = pc + qs;
                    }
                    v2f tot;
                    if constexpr (TWO) tot = reduce_halving2h<NS>(val, lane); else tot = reduce_halving2<NS>(val, lane);
                    const bool ok = incol && ja <= P.RSy;
#pragma unroll
                    for (int e = 0; e < (HALF ? 2 : 1); e++) {
                        const float cand = ok ? (e ? tot.y : tot.x) : -3.0e38f;
                        if (cand > bestv[e]) { bestv[e] = cand; besty[e] = iy; }       // strict: the first row wins a tie, like the oracle's scan order
                    }
                }
#pragma unroll
                for (int e = 0; e < (HALF ? 2 : 1); e++) {
                    const float best = TWO ? half_max(bestv[e]) : wave_max(bestv[e]);
                    const int key = bestv[e] == best ? besty[e] * NS + vi : 1 << 20;
                    int ci = TWO ? half_min(key) : wave_min(key);                              // ties -> lower (sy, sx) index
                    if (ci >= NS * NS) ci = R * NS + R;                                       // no comparable value (NaN scores): the centre
                    const int bsy_ = ci / NS - R, bsx_ = ci - (ci / NS) * NS - R;
                    if (kxl == 0 && live && (q == 0 || p0 + q < P.n)) {
                        int o = dir * P.n_psi + ks + e * P.npsi_store;
                        ccp[o] = best * inv;
                        shp[o] = (bsx_ & 0xffff) | (bsy_ << 16);
                    }
                }
            } else
#pragma unroll
            for (int e = 0; e < (HALF ? 2 : 1); e++) {
                // this orientation's Q = A + sg Bq: U = ua + sg ub, V = va + sg vb; G(+j) = U + iV, G(-j) = U - iV;
                // value(iy, ix) = Re(G[iy] e^{+2 pi i kx (ix-R)/Ns})
                const float sg = e ? -1.f : 1.f;
                float best = -3.0e38f; int bsx_ = 0, bsy_ = 0;
                if constexpr (NS * NS <= 64) {
                    float val[NS * NS];
#pragma unroll
                    for (int iy = 0; iy < NS; iy++) {
                        const int jy = iy - R, ja = jy < 0 ? -jy : jy;
                        float gx, gy;
                        if (jy == 0) { gx = sax + sg * sbx; gy = say + sg * sby; }
                        else {
                            const float ux = uax[ja - 1] + sg * ubx[ja - 1], uy = uay[ja - 1] + sg * uby[ja - 1];
                            const float vx = vax[ja - 1] + sg * vbx[ja - 1], vy = vay[ja - 1] + sg * vby[ja - 1];
                            gx = jy > 0 ? ux - vy : ux + vy; gy = jy > 0 ? uy + vx : uy - vx;
                        }
                        val[iy * NS + R] = gx;
#pragma unroll
                        for (int j = 1; j <= R; j++) {
                            const float pc = gx * txc[j], qs = gy * txs[j];
                            val[iy * NS + R + j] = pc - qs;
                            val[iy * NS + R - j] = pc + qs;
                        }
                    }
                    const float tot = reduce_halving<NS * NS>(val, lane);
                    const int vi = (int)(__brev((unsigned)lane) >> 26);            // the value index this lane ended up with
                    const int iy = vi / NS, ix = vi - iy * NS;
                    const int ay = iy - R < 0 ? R - iy : iy - R, ax = ix - R < 0 ? R - ix : ix - R;
                    const float cand = (vi < NS * NS && ax <= P.RSx && ay <= P.RSy) ? tot : -3.0e38f;
                    // arg-max over lanes; ties -> lower (sy, sx) index like the oracle's scan order
                    best = wave_max(cand);
                    int ci = wave_min(cand == best ? vi : 64);
                    if (ci > 63) ci = 0;                                         // no comparable value (NaN scores)
                    bsy_ = ci / NS - R; bsx_ = ci - (ci / NS) * NS - R;
                } else {
#pragma unroll
                    for (int iy = 0; iy < NS; iy++) {
                        const int jy = iy - R, ja = jy < 0 ? -jy : jy;
                        float gx, gy;
                        if (jy == 0) { gx = sax + sg * sbx; gy = say + sg * sby; }
                        else {
                            const float ux = uax[ja - 1] + sg * ubx[ja - 1], uy = uay[ja - 1] + sg * uby[ja - 1];
                            const float vx = vax[ja - 1] + sg * vbx[ja - 1], vy = vay[ja - 1] + sg * vby[ja - 1];
                            gx = jy > 0 ? ux - vy : ux + vy; gy = jy > 0 ? uy + vx : uy - vx;
                        }
#pragma unroll
                        for (int ix = 0; ix < NS; ix++) {
                            const int j = ix - R, jx = j < 0 ? -j : j;
                            float v = j >= 0 ? (gx * txc[jx] - gy * txs[jx]) : (gx * txc[jx] + gy * txs[jx]);
                            v = wave_sum(v);
                            bool ok = (jx <= P.RSx) && (ja <= P.RSy);
                            if (ok && v > best) { best = v; bsx_ = j; bsy_ = jy; }
                        }
                    }
                }
                if (lane == 0 && (q == 0 || p0 + q < P.n)) {
                    int o = dir * P.n_psi + ks + e * P.npsi_store;
                    ccp[o] = best * inv;
                    shp[o] = (bsx_ & 0xffff) | (bsy_ << 16);
                }
            }
        }
    }
    // ---- top-K of each particle's scores (ties -> lower orientation index): the scores are copied into LDS once (the W
    // tables are no longer needed) and every winner is struck out there
    __threadfence_block();
    __syncthreads();
    float *rv = (float *)smem; int *ri = (int *)(rv + 16);
    for (int q = 0; q < NQ && p0 + q < P.n; q++) {
        const int p = p0 + q;
        float *ccp = P.cc + (size_t)p * P.n_orient; int *shp = P.sh + (size_t)p * P.n_orient;
        float *sc = P.topk_lds ? rv + 32 : ccp;           // host: the LDS allocation covers 32 + n_orient floats when topk_lds is set
        if (P.topk_lds) for (int o = tid; o < P.n_orient; o += NT) sc[o] = ccp[o];
        __syncthreads();
        for (int k = 0; k < P.K; k++) {
            float bv = -3.0e38f; int bi = 0x7fffffff;
            for (int o = tid; o < P.n_orient; o += NT) {
                const float v = sc[o];                          // struck-out entries are -inf: never above the start value
                if (v > bv || (v == bv && o < bi)) { bv = v; bi = o; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                float ov = __shfl_xor(bv, m, 64); int oi = __shfl_xor(bi, m, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NW; w++) if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
                if (bi >= P.n_orient) { bi = 0; bv = 0.f; }     // nothing comparable left (NaN scores): stay inside the tables
                Hit h; h.cc = bv; h.orient = bi;
                int sv = shp[bi];
                h.sx = (int)(short)(sv & 0xffff); h.sy = sv >> 16;
                P.hits[(size_t)p * P.K + k] = h;
                if (bi < P.n_orient) sc[bi] = -__builtin_inff();
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

}  // namespace ppm
