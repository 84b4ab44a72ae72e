// ppm_dev.h — device-side building blocks shared by the kernels of libpypmatch (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppm {

constexpr float kPiF = 3.14159265358979323846f;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

__device__ __forceinline__ unsigned bitrev(unsigned x, int logn) { return __brev(x) >> (32 - logn); }

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
// sum within aligned groups of 16 lanes
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// In-place radix-2 decimation-in-time FFT of `nlines` lines of length n = 2^logn held in LDS.
// Input must already sit in bit-reversed order; output is in natural order.  tw[k] =
// (cos 2 pi k/n, sin 2 pi k/n), k < n/2 (global memory).  All threads of the block must call.
__device__ inline void lds_fft(float2 *buf, int n, int logn, int nlines, int lstride, bool inverse,
                               const float2 *__restrict__ tw, int tid, int nthr) {
    const int half = n >> 1;
    for (int s = 0; s < logn; s++) {
        const int h = 1 << s, q = half >> s;
        __syncthreads();
        for (int i = tid; i < nlines * half; i += nthr) {
            int line = i >> (logn - 1), j = i & (half - 1);
            int k = j & (h - 1);
            int base = ((j >> s) << (s + 1)) + k;
            float2 w = tw[k * q];
            if (!inverse) w.y = -w.y;
            float2 *p = buf + line * lstride;
            float2 a = p[base], b = p[base + h];
            float2 t = cmul(b, w);
            p[base] = cadd(a, t);
            p[base + h] = csub(a, t);
        }
    }
    __syncthreads();
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

__device__ __forceinline__ void unpack_sample(uint32_t u, int &kx, int &ky, int &alpha, int &ring) {
    kx = (int)(u & 511u);
    ky = (int)((u >> 9) & 1023u) - 256;
    alpha = (int)((u >> 19) & 3u);
    ring = (int)(u >> 21);
}

}  // namespace ppm
