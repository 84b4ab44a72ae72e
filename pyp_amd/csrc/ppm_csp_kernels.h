// ppm_csp_kernels.h — constrained (tilt-series) scoring kernel of libpypmatch (gfx950).
//
// A projection row's pose follows from its particle's 3-D pose and its tilt's geometry (include/ppm.h, ppm_csp_cfg; the
// relation restates csp_euler_angles, src/pyp/analysis/geometry/core.py:1081-1213).  The optimiser's state lives on the device
// (round 5): a compass iteration is a fixed sequence of launches — score the candidates (k_csp_eval), average per unit
// (k_csp_unit_means), parabolic trial step (k_csp_step_trial), score it, average, accept and lay out the next iteration's candidates
// (k_csp_step_accept) — and the host (ppm_csp_refine in ppm_lib.hip) enqueues all iterations without waiting in between; the rows of a
// unit are averaged in a fixed order, so results do not depend on the launch shape.
#pragma once
#include "ppm_kernels2.h"

namespace ppm {

struct CspEvalP {
    CubeView cv; const uint32_t *samples; const float2 *Il; const float *cw;
    int S_pad, N, nr; float rlo2, ring_signed;
    int tabR;                             // radius of the LDS address tables (k_csp_eval<true>)
    int S_used; float rmax2;              // band of this sweep (frequency marching)
    int kind, ncand;                      // PPM_CSP_PARTICLES / PPM_CSP_MICROGRAPHS; candidates per unit (<= kMaxCand)
    const int *eval_rows;                 // [grid] row evaluated by each block
    const int *row_part, *row_tilt;       // [n_proj] unit indices of a row
    const int *unit_slot;                 // [n_part] or [n_tilt]: position of the unit in `delta` (-1: not refined)
    const double *Nmat;                   // [n_part][9] particle orientation E(-ppsi, -ptheta, -pphi)
    const double *pshift;                 // [n_part][3] particle shift, pixels
    const double *tl;                     // [n_tilt][4] tilt angle, tilt-axis angle, shift x, shift y
    const double *delta;                  // [n_units][ncand][6] displacement of every candidate
    const double *s0, *g0;                // [n_proj][2] row shift (pixels) and geometric shift at the start
    double *out;                          // [grid][ncand] scores
};

__device__ __forceinline__ void d_rot_xyz(int k, double deg, double *R) {      // right-handed rotation about x, y, z
    double s, c;
    sincos(deg * 3.14159265358979323846 / 180.0, &s, &c);
    if (k == 0) { R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = c; R[5] = -s; R[6] = 0; R[7] = s; R[8] = c; }
    else if (k == 1) { R[0] = c; R[1] = 0; R[2] = s; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = -s; R[7] = 0; R[8] = c; }
    else { R[0] = c; R[1] = -s; R[2] = 0; R[3] = s; R[4] = c; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1; }
}

// M_row = N Ry(-tilt) Rz(axis); g = [Rz(-axis) Ry(tilt) (-p)]_xy + tilt shift
__device__ inline void d_csp_row_pose(const double *N, const double *p, double tilt, double axis, double tsx, double tsy, double *M, double *g) {
    double a[9], b[9], t[9];
    d_rot_xyz(1, -tilt, a); d_rot_xyz(2, axis, b);
    d_mat_mul3(N, a, t); d_mat_mul3(t, b, M);
    d_rot_xyz(2, -axis, a); d_rot_xyz(1, tilt, b);
    const double q0 = -p[0], q1 = -p[1], q2 = -p[2];
    double u[3], v[2];
#pragma unroll
    for (int i = 0; i < 3; i++) u[i] = b[i * 3] * q0 + b[i * 3 + 1] * q1 + b[i * 3 + 2] * q2;
#pragma unroll
    for (int i = 0; i < 2; i++) v[i] = a[i * 3] * u[0] + a[i * 3 + 1] * u[1] + a[i * 3 + 2] * u[2];
    g[0] = v[0] + tsx; g[1] = v[1] + tsy;
}

// Block = one projection row, 256 threads: thread q < ncand derives candidate q's row pose in double precision; the
// candidates that keep the unit's rotation share one gather group (shift variants), every rotated candidate is a group
// of its own; one sweep (sweep_plan) scores them all.
template <bool TAB>
__global__ void __launch_bounds__(256, 4) k_csp_eval(CspEvalP P) {
    __shared__ SweepPlan plan;
    extern __shared__ float lsm[];
    __shared__ double score[kMaxCand];
    __shared__ float cm[kMaxCand][6], csh[kMaxCand][2];
    __shared__ int csame[kMaxCand], cslot[kMaxCand];
    const int tid = threadIdx.x, nthr = blockDim.x, nw = nthr >> 6, nr = P.nr;
    const int j = P.eval_rows[blockIdx.x], ip = P.row_part[j], it = P.row_tilt[j];
    const int ncand = P.ncand;
    if (tid < ncand) {
        const int unit = P.kind == PPM_CSP_PARTICLES ? ip : it;
        const double *d = P.delta + ((size_t)P.unit_slot[unit] * ncand + tid) * 6;
        double N[9], p[3], tl[4], M[9], g[2];
#pragma unroll
        for (int k = 0; k < 9; k++) N[k] = P.Nmat[(size_t)ip * 9 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = P.pshift[(size_t)ip * 3 + k];
#pragma unroll
        for (int k = 0; k < 4; k++) tl[k] = P.tl[(size_t)it * 4 + k];
        int same;
        if (P.kind == PPM_CSP_PARTICLES) {
            same = d[0] == 0.0 && d[1] == 0.0 && d[2] == 0.0;
            for (int k = 0; k < 3; k++)
                if (d[k] != 0.0) { double R[9], T[9]; d_rot_xyz(k, d[k], R); d_mat_mul3(N, R, T); for (int q = 0; q < 9; q++) N[q] = T[q]; }
            p[0] += d[3]; p[1] += d[4]; p[2] += d[5];
        } else {
            same = d[0] == 0.0 && d[1] == 0.0;
            tl[0] += d[0]; tl[1] += d[1]; tl[2] += d[3]; tl[3] += d[4];
        }
        d_csp_row_pose(N, p, tl[0], tl[1], tl[2], tl[3], M, g);
        cm[tid][0] = (float)M[0]; cm[tid][1] = (float)M[1]; cm[tid][2] = (float)M[3]; cm[tid][3] = (float)M[4]; cm[tid][4] = (float)M[6]; cm[tid][5] = (float)M[7];
        csh[tid][0] = (float)(P.s0[2 * j] + g[0] - P.g0[2 * j]); csh[tid][1] = (float)(P.s0[2 * j + 1] + g[1] - P.g0[2 * j + 1]);
        csame[tid] = same;
    }
    __syncthreads();
    if (tid == 0) {
        int ng = 0, q = 0, first_same = -1;
        for (int c = 0; c < ncand; c++) if (csame[c]) { first_same = c; break; }
        if (first_same >= 0) {                     // group 0: the unit's own rotation with all its shift variants
            for (int k = 0; k < 6; k++) plan.m[0][k] = cm[first_same][k];
            plan.slot0[0] = 0;
            for (int c = 0; c < ncand; c++) if (csame[c]) { plan.sh[q][0] = csh[c][0]; plan.sh[q][1] = csh[c][1]; cslot[c] = q++; }
            plan.nv[0] = q; ng = 1;
        }
        for (int c = 0; c < ncand; c++) {
            if (csame[c]) continue;
            for (int k = 0; k < 6; k++) plan.m[ng][k] = cm[c][k];
            plan.slot0[ng] = q; plan.nv[ng] = 1; plan.sh[q][0] = csh[c][0]; plan.sh[q][1] = csh[c][1]; cslot[c] = q++; ng++;
        }
        plan.ng = ng; plan.nslots = q; plan.q_same = 0; plan.S_used = P.S_used; plan.rmax2 = P.rmax2;
    }
    __syncthreads();
    SweepCtx SC;
    SC.cv = P.cv; SC.samples = P.samples; SC.Il = P.Il + (size_t)j * P.S_pad; SC.cw = P.cw + (size_t)j * P.S_pad;
    SC.invN = 1.0f / (float)P.N; SC.rlo2 = P.rlo2; SC.ring_signed = P.ring_signed; SC.nr = nr; SC.nw = nw;
    SC.ringA = lsm; SC.sumB = lsm + kMaxCand * nw * nr; SC.sumC = SC.sumB + kMaxCand * nw; SC.score = score;
    if constexpr (TAB) SC.tab = cube_tab_fill(P.cv, (char *)lsm + ring_lds_bytes8(nw, kMaxCand, nr), P.tabR, tid, nthr);
    sweep_plan<TAB>(plan, SC, tid, nthr);
    if (tid < ncand) P.out[(size_t)blockIdx.x * ncand + tid] = score[cslot[tid]];
}

// Mean score of every (active unit, candidate) over the unit's evaluated rows: the rows of a unit are consecutive in the evaluation
// list (uoff[a] .. uoff[a + 1]) and are added in that order, like the host loop this replaces (2 MB of per-row scores per sweep
// stay on the device; 50 KB of means go back).
__global__ void k_csp_unit_means(const double *__restrict__ out, const int *__restrict__ uoff, int n_units, int ncand, double *__restrict__ mean) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_units * ncand) return;
    const int a = i / ncand, c = i - a * ncand, lo = uoff[a], hi = uoff[a + 1];
    double s = 0;
    for (int r = lo; r < hi; r++) s += out[(size_t)r * ncand + c];
    mean[i] = s / (double)(hi - lo);
}

// ---- the compass search's decisions, one thread per active unit (double precision, the same rule as the CPU checker's loop and k_local's
// compass iteration).  Candidate 0 is the unit as it stands; then +h and -h for every enabled parameter in order.
struct CspStepP {
    int kind, n_active, ncand;
    const int *active;        // [n_active] index of the unit in the particle / tilt tables (null: the identity)
    const int *unit_slot;     // [n_units] row of the unit in the displacement tables (null: the identity)
    int en[6]; double tol[6];
    double ha, hs;            // steps of THIS iteration (degrees, pixels)
    double ha_next, hs_next;  // steps of the next one (k_csp_step_accept lays out its candidates)
    const double *mean;       // [n_active][ncand] unit means of the compass sweep
    const double *tmean;      // [n_active] unit means of the trial sweep
    double *acc;              // [n_active][6] displacement accumulated so far (bounded by +-tol)
    double *dtrial;           // [n_active][6] trial step
    double *fpm;              // [n_active][12] f(+h), f(-h) per parameter (-1e300: outside the bounds)
    double *delta_c;          // [n_slots][ncand][6] candidates of the compass sweep
    double *delta_t;          // [n_slots][6] the trial step as k_csp_eval reads it
    double *Nmat, *pshift, *tl;   // unit state (k_csp_eval's tables; the sub-volume search keeps N and p in one row of 12)
    int nstride, pstride;         // doubles between two units' N / p (9 and 3 in k_csp_eval's tables, 12 and 12 in k_sva_eval's)
};
__device__ __forceinline__ int d_csp_unit(const CspStepP &P, int a) { return P.active ? P.active[a] : a; }
__device__ __forceinline__ int d_csp_slot(const CspStepP &P, int u) { return P.unit_slot ? P.unit_slot[u] : u; }

__device__ __forceinline__ void d_csp_layout_candidates(const CspStepP &P, int slot, double ha, double hs) {
    double *d = P.delta_c + (size_t)slot * P.ncand * 6;
    for (int k = 0; k < P.ncand * 6; k++) d[k] = 0.0;
    int c = 1;
    for (int i = 0; i < 6; i++) {
        if (!P.en[i]) continue;
        const double h = i < 3 ? ha : hs;
        d[(size_t)c * 6 + i] = h; d[(size_t)(c + 1) * 6 + i] = -h;
        c += 2;
    }
}

// candidates of the first iteration
__global__ void k_csp_step_init(CspStepP P) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= P.n_active) return;
    d_csp_layout_candidates(P, d_csp_slot(P, d_csp_unit(P, a)), P.ha_next, P.hs_next);
}

// after the compass sweep: the parabolic trial step of every unit
__global__ void k_csp_step_trial(CspStepP P) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= P.n_active) return;
    const double *mean = P.mean + (size_t)a * P.ncand, *acc = P.acc + (size_t)a * 6;
    double *d = P.dtrial + (size_t)a * 6, *fpm = P.fpm + (size_t)a * 12;
    const double f0 = mean[0];
    for (int i = 0, c = 1; i < 6; i++) {
        d[i] = 0; fpm[2 * i] = fpm[2 * i + 1] = -1e300;
        if (!P.en[i]) continue;
        const double h = i < 3 ? P.ha : P.hs, tol = P.tol[i];
        const bool okp = fabs(acc[i] + h) <= tol + 1e-9, okm = fabs(acc[i] - h) <= tol + 1e-9;
        const double fp = okp ? mean[c] : -1e300, fm = okm ? mean[c + 1] : -1e300;
        c += 2;
        fpm[2 * i] = fp; fpm[2 * i + 1] = fm;
        if (okp && okm) {
            const double den = 2.0 * f0 - fp - fm;
            if (den > 1e-12) { const double t = 0.5 * h * (fp - fm) / den; d[i] = t > h ? h : (t < -h ? -h : t); }
            else { const double best = fp > fm ? fp : fm; d[i] = best > f0 ? (fp > fm ? h : -h) : 0.0; }
        } else if (okp) d[i] = fp > f0 ? h : 0.0;
        else if (okm) d[i] = fm > f0 ? -h : 0.0;
        if (acc[i] + d[i] > tol) d[i] = tol - acc[i];
        if (acc[i] + d[i] < -tol) d[i] = -tol - acc[i];
    }
    double *dt = P.delta_t + (size_t)d_csp_slot(P, d_csp_unit(P, a)) * 6;
    for (int i = 0; i < 6; i++) dt[i] = d[i];
}

// after the trial sweep: keep the trial step, the best single probe, or nothing; move the unit; lay out the next candidates
__global__ void k_csp_step_accept(CspStepP P) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= P.n_active) return;
    const int u = d_csp_unit(P, a);
    const double f0 = P.mean[(size_t)a * P.ncand], ft = P.tmean[a];
    const double *fpm = P.fpm + (size_t)a * 12, *dtr = P.dtrial + (size_t)a * 6;
    int bi = -1, bs = 0; double fb = f0;
    for (int i = 0; i < 6; i++) {
        if (!P.en[i]) continue;
        if (fpm[2 * i] > fb) { fb = fpm[2 * i]; bi = i; bs = 1; }
        if (fpm[2 * i + 1] > fb) { fb = fpm[2 * i + 1]; bi = i; bs = -1; }
    }
    double d[6] = { 0, 0, 0, 0, 0, 0 };
    bool move = false;
    if (ft > f0 && ft >= fb) { for (int i = 0; i < 6; i++) d[i] = dtr[i]; move = true; }
    else if (bi >= 0) { d[bi] = bs * (bi < 3 ? P.ha : P.hs); move = true; }
    if (move) {
        if (P.kind == PPM_CSP_PARTICLES) {
            double *N = P.Nmat + (size_t)u * P.nstride, *p = P.pshift + (size_t)u * P.pstride;
            for (int k = 0; k < 3; k++)
                if (d[k] != 0.0) { double R[9], T[9]; d_rot_xyz(k, d[k], R); d_mat_mul3(N, R, T); for (int q = 0; q < 9; q++) N[q] = T[q]; }
            for (int k = 0; k < 3; k++) p[k] += d[3 + k];
        } else {
            double *tl = P.tl + (size_t)u * 4;
            tl[0] += d[0]; tl[1] += d[1]; tl[2] += d[3]; tl[3] += d[4];
        }
        for (int k = 0; k < 6; k++) P.acc[(size_t)a * 6 + k] += d[k];
    }
    d_csp_layout_candidates(P, d_csp_slot(P, u), P.ha_next, P.hs_next);
}

}  // namespace ppm
