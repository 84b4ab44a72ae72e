// ppm_geom.h — host-side derived geometry of a refinement call (band limits, shift grid,
// orientation grid, ring-ordered sample list).  Plain C++, no device code.
//
// The quantities restate the numeric answers of the refine3d prompt script
// (src/pyp/refine/frealign/frealign.py:3918-3994) in Fourier-pixel units; the grid is the
// build-defined global grid of SURVEY.md §8a K6 (theta = 0..180 step D, n_phi = round(360 sin(theta)/D)).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ppm.h"

namespace ppm {

constexpr double kPi = 3.14159265358979323846;

struct Geom {
    int N = 0;
    double a = 0;
    double r_hi = 0, r_lo = 0, r_s = 0, ring_signed = 0;
    double r_cls = 0;             // band of LOGP / SIGMA (answer 22, ppm_refine_cfg.res_classification); = r_hi when unset
    int B = 0, W = 0, H = 0;      // full band: half-width, row width B+1, rows 2B+1
    int Bs = 0, Hs = 0;           // search band
    int Ns = 0, RSx = 0, RSy = 0;
    double step = 0;              // global-search shift grid: Ns points over the box, step = N/Ns pixels
    int n_theta = 0, n_psi = 0, n_dir = 0, n_orient = 0, npsi_store = 0, half = 0;
    double dpsi = 0, dstep = 0, phi_max = 360, theta_max = 180;
    double range_asked_px = 0;    // largest shift search range the caller asked for, pixels (0: 'mask radius' / unlimited)
    bool range_capped = false;    // the window of the grid search is narrower than that
    double r_s_asked = 0;         // search band the caller asked for (> r_s when the 64-pixel cap of the grid search applied)
};

// asymmetric unit of the global grid (include/ppm.h, field `symmetry`)
inline void sym_limits(const char *sym, double &phi_max, double &theta_max) {
    phi_max = 360.0; theta_max = 180.0;
    if (!sym || !sym[0]) return;
    char t = sym[0] >= 'a' ? sym[0] - 32 : sym[0];
    int n = std::atoi(sym + 1);
    if (t == 'C' && n >= 1) phi_max = 360.0 / n;
    else if (t == 'D' && n >= 1) { phi_max = 360.0 / n; theta_max = 90.0; }
    else if (t == 'T' || t == 'I') { phi_max = 180.0; theta_max = 90.0; }
    else if (t == 'O') { phi_max = 90.0; theta_max = 90.0; }
}

// box sizes the FFTs handle: even, 32..512, prime factors 2, 3, 5, 7 only
inline bool box_ok(int n) {
    if (n < 32 || n > 512 || n % 2) return false;
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    while (n % 5 == 0) n /= 5;
    while (n % 7 == 0) n /= 7;
    return n == 1;
}
// factors (4s first, then 2, 3, 5, 7) and the digit-reversal staging permutation of an FFT length
inline void fft_factors(int n, std::vector<int> &fac, std::vector<unsigned short> &perm) {
    fac.clear();
    int m = n;
    while (m % 4 == 0) { fac.push_back(4); m /= 4; }
    while (m % 2 == 0) { fac.push_back(2); m /= 2; }
    while (m % 3 == 0) { fac.push_back(3); m /= 3; }
    while (m % 5 == 0) { fac.push_back(5); m /= 5; }
    while (m % 7 == 0) { fac.push_back(7); m /= 7; }
    perm.resize(n);
    for (int i = 0; i < n; i++) {
        int pos = 0, rem = i, L = n;
        for (int st = (int)fac.size() - 1; st >= 0; st--) { int r = fac[st]; L /= r; pos += (rem % r) * L; rem /= r; }
        perm[i] = (unsigned short)pos;
    }
}

inline int n_phi_at(double theta_deg, double dstep, double phi_max = 360.0) {
    int np = (int)std::floor(phi_max * std::sin(theta_deg * kPi / 180.0) / dstep + 0.5);
    return np < 1 ? 1 : np;
}

inline bool geom_init(Geom &g, const ppm_refine_cfg &c, std::string &err) {
    g = Geom();
    g.N = c.box; g.a = c.pixel_size;
    if (!box_ok(g.N)) { err = "box size must be even, 32..512, with prime factors 2, 3, 5 and 7 only"; return false; }
    if (!(g.a > 0) || !(c.res_high > 0)) { err = "pixel size and high-resolution limit must be positive"; return false; }
    double na = g.N * g.a;
    g.r_hi = na / c.res_high; if (g.r_hi > g.N / 2) g.r_hi = g.N / 2;
    g.r_lo = c.res_low > 0 ? na / c.res_low : 0.0;
    g.r_s = c.res_search > 0 ? na / c.res_search : g.r_hi; if (g.r_s > g.r_hi) g.r_s = g.r_hi;
    g.r_s_asked = g.r_s;
    if (c.global_search && g.r_s > 64.0) g.r_s = 64.0;   // the grid-search kernel covers 64 Fourier pixels (lane = kx); finer
                                                         // detail only enters through the refinement of the hits
    g.ring_signed = c.res_signed_cc > 0 ? na / c.res_signed_cc : 1e30;
    // answer 22 (frealign.py:3945): 0, beyond res_high, or a band of less than one Fourier pixel above r_lo -> the full band
    g.r_cls = c.res_classification > 0 ? na / c.res_classification : g.r_hi;
    if (g.r_cls > g.r_hi || g.r_cls < g.r_lo + 1.0) g.r_cls = g.r_hi;
    g.B = (int)std::ceil(g.r_hi) - 1; g.W = g.B + 1; g.H = 2 * g.B + 1;
    g.Bs = (int)std::ceil(g.r_s) - 1; g.Hs = 2 * g.Bs + 1;
    if (g.B < 2) { err = "resolution limits leave fewer than 3 Fourier pixels"; return false; }
    g.Ns = 2; while (g.Ns < 2 * (g.Bs + 1)) g.Ns <<= 1;
    g.step = (double)g.N / g.Ns;
    // answers 27 / 28: 0 means the mask radius ("0.0 = mask radius", config/pyp_config.toml:5338-5343); the window is limited
    // only by the search grid itself (shifts beyond Ns / 2 - 1 steps alias).  Windows wider than PPM_MAX_SHIFT_STEPS steps
    // either side are searched as overlapping tiles of that half-width (ppm_refine_batch).
    double rx = (c.search_range_x > 0 ? c.search_range_x : c.mask_radius) / g.a, ry = (c.search_range_y > 0 ? c.search_range_y : c.mask_radius) / g.a;
    g.RSx = (int)std::ceil(rx / g.step); g.RSy = (int)std::ceil(ry / g.step);
    if (g.RSx < 1) g.RSx = 1;
    if (g.RSy < 1) g.RSy = 1;
    g.range_asked_px = std::max(rx, ry);
    const int rs_max = g.Ns / 2 - 1;
    g.range_capped = g.RSx > rs_max || g.RSy > rs_max;
    if (g.RSx > rs_max) g.RSx = rs_max;
    if (g.RSy > rs_max) g.RSy = rs_max;
    g.dstep = c.angular_step > 0 ? c.angular_step : 15.0;
    char symbuf[9]; std::memcpy(symbuf, c.symmetry, 8); symbuf[8] = 0;
    sym_limits(symbuf, g.phi_max, g.theta_max);
    g.n_theta = (int)std::floor(g.theta_max / g.dstep + 0.5) + 1;
    if (g.n_theta < 2) g.n_theta = 2;
    g.n_psi = (int)std::floor(360.0 / g.dstep + 0.5); if (g.n_psi < 1) g.n_psi = 1;
    g.dpsi = 360.0 / g.n_psi;
    g.n_dir = 0;
    for (int i = 0; i < g.n_theta; i++) g.n_dir += n_phi_at(g.theta_max * i / (g.n_theta - 1), g.dstep, g.phi_max);
    g.n_orient = g.n_dir * g.n_psi;
    g.half = (g.n_psi % 2 == 0);
    g.npsi_store = g.half ? g.n_psi / 2 : g.n_psi;
    return true;
}

inline void grid_direction(const Geom &g, int dir, double &theta, double &phi) {
    int acc = 0;
    for (int i = 0; i < g.n_theta; i++) {
        double th = g.theta_max * i / (g.n_theta - 1);
        int np = n_phi_at(th, g.dstep, g.phi_max);
        if (dir < acc + np) { theta = th; phi = g.phi_max * (dir - acc) / np; return; }
        acc += np;
    }
    theta = phi = 0;
}

// M = Rz(phi) Ry(theta) Rz(psi), row-major 3x3 ("rotates the reference by PHI -> THETA -> PSI",
// src/pyp/analysis/geometry/core.py:1186-1187)
inline void euler_matrix(double psi, double theta, double phi, double M[9]) {
    double ps = psi * kPi / 180, th = theta * kPi / 180, ph = phi * kPi / 180;
    double cps = std::cos(ps), sps = std::sin(ps), cth = std::cos(th), sth = std::sin(th), cph = std::cos(ph), sph = std::sin(ph);
    M[0] = cph * cth * cps - sph * sps; M[1] = -cph * cth * sps - sph * cps; M[2] = cph * sth;
    M[3] = sph * cth * cps + cph * sps; M[4] = -sph * cth * sps + cph * cps; M[5] = sph * sth;
    M[6] = -sth * cps;                  M[7] = sth * sps;                    M[8] = cth;
}

// (psi, theta, phi) in degrees of M = Rz(phi) Ry(theta) Rz(psi); at theta = 0 / 180 everything goes into psi
inline void angles_from_matrix(const double M[9], double &psi, double &theta, double &phi) {
    const double r2d = 180.0 / kPi;
    double ct = M[8] > 1 ? 1 : (M[8] < -1 ? -1 : M[8]);
    double st = std::sqrt(M[2] * M[2] + M[5] * M[5]);
    if (st > 1e-7) { theta = std::atan2(st, ct) * r2d; phi = std::atan2(M[5], M[2]) * r2d; psi = std::atan2(M[7], -M[6]) * r2d; }
    else { theta = ct > 0 ? 0.0 : 180.0; phi = 0.0; psi = (ct > 0 ? std::atan2(M[3], M[0]) : std::atan2(-M[3], -M[0])) * r2d; }
    if (psi < 0) psi += 360;
    if (phi < 0) phi += 360;
}

inline void mat_mul3h(const double *a, const double *b, double *c) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double v = 0; for (int k = 0; k < 3; k++) v += a[i * 3 + k] * b[k * 3 + j]; t[i * 3 + j] = v; }
    std::memcpy(c, t, sizeof(t));
}
inline void rot_xyz(int k, double deg, double R[9]) {      // right-handed rotation about x (0), y (1), z (2)
    double t = deg * kPi / 180, c = std::cos(t), s = std::sin(t);
    double rx[9] = { 1, 0, 0, 0, c, -s, 0, s, c }, ry[9] = { c, 0, s, 0, 1, 0, -s, 0, c }, rz[9] = { c, -s, 0, s, c, 0, 0, 0, 1 };
    std::memcpy(R, k == 0 ? rx : (k == 1 ? ry : rz), sizeof(rx));
}
// Row pose of the constrained geometry (include/ppm.h, ppm_csp_cfg): M_row = N Ry(-tilt) Rz(axis),
// g = [Rz(-axis) Ry(tilt) (-p)]_xy + tilt shift (pixels)
// the four rotations a tilt contributes to its rows' poses (the trigonometry of csp_row_pose, shared by all rows of the tilt)
struct TiltRot { double a[9], b[9], ai[9], bi[9]; };     // Ry(-tilt), Rz(axis), Rz(-axis), Ry(tilt)
inline void tilt_rotations(double tilt, double axis, TiltRot &r) { rot_xyz(1, -tilt, r.a); rot_xyz(2, axis, r.b); rot_xyz(2, -axis, r.ai); rot_xyz(1, tilt, r.bi); }
inline void csp_row_pose(const double N[9], const double p[3], const TiltRot &r, double tsx, double tsy, double M[9], double g[2]) {
    double t[9];
    mat_mul3h(N, r.a, t); mat_mul3h(t, r.b, M);
    double q[3] = { -p[0], -p[1], -p[2] }, u[3], v[3];
    for (int i = 0; i < 3; i++) u[i] = r.bi[i * 3] * q[0] + r.bi[i * 3 + 1] * q[1] + r.bi[i * 3 + 2] * q[2];
    for (int i = 0; i < 3; i++) v[i] = r.ai[i * 3] * u[0] + r.ai[i * 3 + 1] * u[1] + r.ai[i * 3 + 2] * u[2];
    g[0] = v[0] + tsx; g[1] = v[1] + tsy;
}
inline void csp_row_pose(const double N[9], const double p[3], double tilt, double axis, double tsx, double tsy, double M[9], double g[2]) {
    TiltRot r; tilt_rotations(tilt, axis, r);
    csp_row_pose(N, p, r, tsx, tsy, M, g);
}

// Ring-ordered sample list of the half plane kx >= 0, 0 < k^2 < r_hi^2, ring = floor(|k|); every
// ring padded to a multiple of 16 samples with weightless dummies so that a 16-lane group never
// straddles two rings.  Packed: kx (9 bits) | ky+256 (10 bits) << 9 | alpha (2 bits) << 19 | ring << 21.
struct SampleList {
    std::vector<uint32_t> packed;
    std::vector<int> ring_off;   // ring_off[b] = length of the list prefix holding all samples of rings < b; size B+3
};

inline uint32_t pack_sample(int kx, int ky, int alpha, int ring) {
    return (uint32_t)kx | ((uint32_t)(ky + 256) << 9) | ((uint32_t)alpha << 19) | ((uint32_t)ring << 21);
}

inline void build_samples(const Geom &g, SampleList &sl) {
    int B = g.B;
    std::vector<std::vector<uint32_t>> rings(B + 2);
    double r2 = g.r_hi * g.r_hi;
    for (int ky = -B; ky <= B; ky++) for (int kx = 0; kx <= B; kx++) {
        double k2 = (double)kx * kx + (double)ky * ky;
        if (k2 >= r2 || k2 == 0) continue;
        int b = (int)std::floor(std::sqrt(k2));
        rings[b].push_back(pack_sample(kx, ky, kx == 0 ? 1 : 2, b));
    }
    // List order: bands of kRingBand consecutive rings; inside a band the 16-sample groups (one ring each, ky-ordered = along
    // the arc) of all its rings are sorted by their angular position, so that the four groups a wavefront takes (64
    // consecutive entries) form a compact 4 x 16 patch of the slice rather than a 64-sample arc: fewer distinct cache lines
    // of the reference cube per gather.  ring_off[b] = length of the list prefix that holds every sample of the rings < b
    // (the end of the band of ring b - 1; samples beyond a band limit are masked individually by the kernels).
    constexpr int kRingBand = 4;
    sl.packed.clear(); sl.ring_off.assign(B + 3, 0);
    for (int b0 = 0; b0 <= B + 1; b0 += kRingBand) {
        struct Grp { double key; int ring; int first; };
        std::vector<Grp> grps;
        const int b1 = std::min(b0 + kRingBand - 1, B + 1);
        for (int b = b0; b <= b1; b++) {
            while (rings[b].size() % 16) rings[b].push_back(pack_sample(0, 0, 0, b));
            const int G = (int)rings[b].size() / 16;
            for (int gidx = 0; gidx < G; gidx++) grps.push_back({ (gidx + 0.5) / G, b, gidx * 16 });
        }
        std::stable_sort(grps.begin(), grps.end(), [](const Grp &x, const Grp &y) { return x.key < y.key; });
        for (const Grp &gr : grps) for (int i = 0; i < 16; i++) sl.packed.push_back(rings[gr.ring][gr.first + i]);
        for (int b = b0; b <= b1; b++) sl.ring_off[b + 1] = (int)sl.packed.size();
    }
    sl.ring_off[B + 2] = (int)sl.packed.size();
}

// Point-group operators (row-major 3x3 each); "C1","Cn","Dn","T","O","I"
inline void mat_mul3(const double *a, const double *b, double *c) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double v = 0; for (int k = 0; k < 3; k++) v += a[i * 3 + k] * b[k * 3 + j];
        t[i * 3 + j] = v;
    }
    std::memcpy(c, t, sizeof(t));
}
inline void rot_axis(const double ax[3], double deg, double *m) {
    double n = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    double x = ax[0] / n, y = ax[1] / n, z = ax[2] / n, t = deg * kPi / 180, c = std::cos(t), s = std::sin(t), C = 1 - c;
    double r[9] = { c + x * x * C, x * y * C - z * s, x * z * C + y * s, y * x * C + z * s, c + y * y * C, y * z * C - x * s,
                    z * x * C - y * s, z * y * C + x * s, c + z * z * C };
    std::memcpy(m, r, sizeof(r));
}
inline int symmetry_ops(const char *sym, std::vector<double> &ops) {
    double gens[27]; int ng = 0;
    double z[3] = { 0, 0, 1 }, x[3] = { 1, 0, 0 }, d111[3] = { 1, 1, 1 };
    if (!sym || !sym[0]) sym = "C1";
    char t = sym[0] >= 'a' ? sym[0] - 32 : sym[0];
    int n = std::atoi(sym + 1);
    if (t == 'C' && n >= 1) { rot_axis(z, 360.0 / n, gens); ng = 1; }
    else if (t == 'D' && n >= 1) { rot_axis(z, 360.0 / n, gens); rot_axis(x, 180, gens + 9); ng = 2; }
    else if (t == 'T') { rot_axis(z, 180, gens); rot_axis(d111, 120, gens + 9); ng = 2; }
    else if (t == 'O') { rot_axis(z, 90, gens); rot_axis(d111, 120, gens + 9); ng = 2; }
    else if (t == 'I') {
        double phi = (1 + std::sqrt(5.0)) / 2, a5[3] = { 0, 1, phi };
        rot_axis(z, 180, gens); rot_axis(d111, 120, gens + 9); rot_axis(a5, 72, gens + 18); ng = 3;
    } else return -1;
    ops.assign(9, 0.0); ops[0] = ops[4] = ops[8] = 1.0;
    int cnt = 1;
    for (bool grew = true; grew;) {
        grew = false;
        for (int i = 0; i < cnt && cnt < 60; i++) for (int j = 0; j < ng && cnt < 60; j++) {
            double c[9]; mat_mul3(&ops[i * 9], gens + j * 9, c);
            bool found = false;
            for (int k = 0; k < cnt && !found; k++) {
                double d = 0; for (int q = 0; q < 9; q++) d += std::fabs(ops[k * 9 + q] - c[q]);
                if (d < 1e-6) found = true;
            }
            if (!found) { ops.insert(ops.end(), c, c + 9); cnt++; grew = true; }
        }
    }
    return cnt;
}

}  // namespace ppm
