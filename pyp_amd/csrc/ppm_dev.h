// ppm_dev.h — device-side building blocks shared by the kernels of libpypmatch (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppm {

constexpr float kPiF = 3.14159265358979323846f;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }


// wave64 all-lanes sum (every lane gets the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// FFT plan of one length n = prod fac[] (factors 4, 2, 3, 5): tw[k] = (cos 2 pi k/n, sin 2 pi k/n), k < n;
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
        __syncthreads();
        for (int i = tid; i < nlines * m; i += nthr) {
            const int line = fast_div(i, m, inv_m), t = i - line * m, blk = fast_div(t, Lp, inv_Lp), j = t - blk * Lp;
            float2 *p = buf + line * lstride + blk * L + j;
            float2 x[5];
#pragma unroll
            for (int q = 0; q < 5; q++) {
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
            } else {
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
            }
        }
        Lp = L;
    }
    __syncthreads();
}
__device__ inline void lds_fft(float2 *buf, const FftPlan &pl, int nlines, int lstride, bool inverse, int tid, int nthr) {
    lds_fft(buf, pl, nlines, lstride, inverse, tid, nthr, pl.tw);
}

// Trilinear sample of the band-limited reference cube at Fourier coordinate (X,Y,Z).
// cube: float2 [CY][CY][CX], x fastest, index ((z+off)*CY + (y+off))*CX + x, x in 0..B+1.
// Friedel symmetry supplies x < 0.  The two x-neighbours are fetched as one 16-byte load.
struct CubeView { const float2 *cube; int CX, CY, off; };

__device__ __forceinline__ float2 sample_cube(const CubeView &cv, float X, float Y, float Z) {
    bool cj = X < 0.f;
    if (cj) { X = -X; Y = -Y; Z = -Z; }
    float xf = floorf(X), yf = floorf(Y), zf = floorf(Z);
    float fx = X - xf, fy = Y - yf, fz = Z - zf;
    int x0 = (int)xf, y0 = (int)yf + cv.off, z0 = (int)zf + cv.off;
    const float2 *p = cv.cube + ((size_t)z0 * cv.CY + y0) * cv.CX + x0;
    const size_t sy = cv.CX, sz = (size_t)cv.CX * cv.CY;
    // (re0, im0, re1, im1) of the two x taps; 8-byte aligned addresses
    float2 a0 = p[0], a1 = p[1];
    float2 b0 = p[sy], b1 = p[sy + 1];
    float2 c0 = p[sz], c1 = p[sz + 1];
    float2 d0 = p[sz + sy], d1 = p[sz + sy + 1];
    float ar = a0.x + fx * (a1.x - a0.x), ai = a0.y + fx * (a1.y - a0.y);
    float br = b0.x + fx * (b1.x - b0.x), bi = b0.y + fx * (b1.y - b0.y);
    float cr = c0.x + fx * (c1.x - c0.x), ci = c0.y + fx * (c1.y - c0.y);
    float dr = d0.x + fx * (d1.x - d0.x), di = d0.y + fx * (d1.y - d0.y);
    float er = ar + fy * (br - ar), ei = ai + fy * (bi - ai);
    float gr = cr + fy * (dr - cr), gi = ci + fy * (di - ci);
    float rr = er + fz * (gr - er), ri = ei + fz * (gi - ei);
    return make_float2(rr, cj ? -ri : ri);
}

// CTF of one particle: -sin(pi lambda s^2 (df(phi) - Cs lambda^2 s^2 / 2) + phase + amp)
struct CtfP { float lambda, cs, dsum, ddif, c2a, s2a, extra, inv_na2; };

__device__ __forceinline__ CtfP ctf_from_row(const double *row, int N, double a) {
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
