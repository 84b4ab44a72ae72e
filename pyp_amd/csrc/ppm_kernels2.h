// ppm_kernels2.h — local refinement, Fourier insertion and finalisation kernels (gfx950).
#pragma once
#include "ppm_kernels.h"

namespace ppm {

// ---------------------------------------------------------------------------------- local refinement
// One refinement trajectory.  M = Rz(phi) Ry(theta) Rz(psi) row-major; shifts in pixels.
struct LState { double M[9]; double sh[2]; double f, ha, hs; int particle; int pad; double fc; };   // fc: score over the classification band (answer 22), set by the final launch

__device__ inline void d_mat_mul3(const double *a, const double *b, double *c) {
    double t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double v = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) v += a[i * 3 + k] * b[k * 3 + j];
            t[i * 3 + j] = v;
        }
#pragma unroll
    for (int i = 0; i < 9; i++) c[i] = t[i];
}

__device__ inline void d_euler(double psi, double theta, double phi, double *M) {
    const double d2r = 3.14159265358979323846 / 180.0;
    double cps = cos(psi * d2r), sps = sin(psi * d2r), cth = cos(theta * d2r), sth = sin(theta * d2r), cph = cos(phi * d2r), sph = sin(phi * d2r);
    M[0] = cph * cth * cps - sph * sps; M[1] = -cph * cth * sps - sph * cps; M[2] = cph * sth;
    M[3] = sph * cth * cps + cph * sps; M[4] = -sph * cth * sps + cph * cps; M[5] = sph * sth;
    M[6] = -sth * cps;                  M[7] = sth * sps;                    M[8] = cth;
}

__device__ inline void d_angles(const double *M, double &psi, double &theta, double &phi) {
    const double r2d = 180.0 / 3.14159265358979323846;
    double ct = M[8] > 1 ? 1 : (M[8] < -1 ? -1 : M[8]);
    double st = sqrt(M[2] * M[2] + M[5] * M[5]);
    if (st > 1e-7) {
        theta = atan2(st, ct) * r2d; phi = atan2(M[5], M[2]) * r2d; psi = atan2(M[7], -M[6]) * r2d;
    } else {
        theta = ct > 0 ? 0.0 : 180.0; phi = 0.0;
        psi = (ct > 0 ? atan2(M[3], M[0]) : atan2(-M[3], -M[0])) * r2d;
    }
    if (psi < 0) psi += 360;
    if (phi < 0) phi += 360;
}

// which: 0 = in-plane (psi), 1 / 2 = tilt about image x / y when tilt_frame, else theta / phi Euler steps.
// The three image-frame steps are right-multiplications by Rz / Rx / Ry: plain column mixes.
// (column indices are compile-time constants: with run-time indices the 3 x 3 temporaries lived in scratch memory, ~1.2 MB
// of scratch traffic per particle from the serial set-up sections)
template <int A, int B, int K>
__device__ __forceinline__ void d_col_mix(const double *M, double s, double c, double *out) {
    // M R with R rotating the (A, B) coordinate pair: out[:,A] = c M[:,A] + s M[:,B], out[:,B] = -s M[:,A] + c M[:,B]
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double ma = M[r * 3 + A], mb = M[r * 3 + B];
        out[r * 3 + A] = ma * c + mb * s;
        out[r * 3 + B] = mb * c - ma * s;
        out[r * 3 + K] = M[r * 3 + K];
    }
}
__device__ inline void d_rot_step(const double *M, int which, int tilt_frame, double hdeg, double *out) {
    double s, c;
    sincos(hdeg * 3.14159265358979323846 / 180.0, &s, &c);
    if (which == 0) { d_col_mix<0, 1, 2>(M, s, c, out); return; }
    if (tilt_frame) {
        if (which == 1) d_col_mix<1, 2, 0>(M, s, c, out); else d_col_mix<2, 0, 1>(M, s, c, out);
        return;
    }
    if (which == 2) { double r[9] = { c, -s, 0, s, c, 0, 0, 0, 1 }; d_mat_mul3(r, M, out); return; }
    double psi, th, ph; d_angles(M, psi, th, ph);
    double cp = cos(ph * 3.14159265358979323846 / 180.0), sp = sin(ph * 3.14159265358979323846 / 180.0);
    double rz[9] = { cp, -sp, 0, sp, cp, 0, 0, 0, 1 }, rzt[9] = { cp, sp, 0, -sp, cp, 0, 0, 0, 1 }, ry[9] = { c, 0, s, 0, 1, 0, -s, 0, c };
    double T[9], L[9];
    d_mat_mul3(rz, ry, T); d_mat_mul3(T, rzt, L); d_mat_mul3(L, M, out);
}

constexpr int kMaxIters = 24;
struct LocalP {
    CubeView cv; const uint32_t *samples; const float2 *Il; const float *cw;
    int S_pad, nrings, N;
    int nr;                // rings this launch can touch (<= nrings): sizes the per-wave ring sums in dynamic LDS
    int tabR;              // radius of the LDS address tables (k_local<true>; ppm_dev.h)
    float rlo2, ring_signed;
    LState *states; int T, final_rescore; int en[5];
    // frequency marching: band (squared) and sample-list prefix of every iteration, and of the final score
    float rmax2_it[kMaxIters]; int S_it[kMaxIters];
    float rmax2_final; int S_final;
    float rmax2_class; int S_class;     // answer 22: band of LOGP / SIGMA (S_class = 0: the full band, no extra sweep)
    // answer 7 "use priors" (include/ppm.h): Gaussian restraint on the refined parameters; w = 1 / (2 var n_s), shifts in pixels
    int use_priors; double pmean[5], pw[5];
};

// restraint of one pose (k_local's compass; same expression as the oracle's prior_pen)
__device__ inline double d_prior_pen(const LocalP &P, const double *M, double shx, double shy) {
    double v[5]; d_angles(M, v[0], v[1], v[2]); v[3] = shx; v[4] = shy;
    double pen = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        if (!(P.pw[i] > 0)) continue;
        double d = v[i] - P.pmean[i];
        if (i < 3) { d = fmod(d, 360.0); if (d > 180.0) d -= 360.0; if (d < -180.0) d += 360.0; }
        pen += P.pw[i] * d * d;
    }
    return pen;
}

constexpr int kMaxCand = 13;   // scores per sweep: 2 per free parameter + the centre (k_local: 11; constrained particle search: 13), or 1
constexpr int kMaxGroup = 7;   // gathers per sweep: 6 angular neighbours + the centre (shared with the 4 shift neighbours)

// sum over aligned groups of 16 lanes with DPP only (quad swaps, then half-row and row mirrors);
// every lane of the group ends up with the total
__device__ __forceinline__ float group16_sum_dpp(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// A sweep evaluates `ng` gather groups; group g = one rotation (6 floats) with nv[g] shift variants, the
// scores of which go to consecutive slots starting at slot0[g].  ms / bsl are filled by sweep_plan itself.
struct SweepPlan {
    float m[kMaxGroup][6]; float sh[kMaxCand][2]; int nv[kMaxGroup]; int slot0[kMaxGroup]; int ng, nslots, S_used, q_same; float rmax2;   // slots 0 .. q_same use the shift sh[0]
    float ms[kMaxGroup][6];      // m times the cube's padding factor
    int bsl[kMaxCand];           // slot whose model-power sum serves slot q (the first slot of q's group: |model|^2 does not depend on the shift)
};

// sum over aligned groups of 8 lanes (double), every lane of the group gets the total; fixed combination order
__device__ __forceinline__ double group8_sum_d(double v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}

// dynamic LDS of k_local / k_defocus for `nw` waves, `nq` score slots and `nr` rings
__host__ __device__ inline size_t ring_lds_bytes(int nw, int nq, int nr) { return ((size_t)nq * nw * nr + (size_t)nq * nw + nw) * sizeof(float); }
__host__ __device__ inline size_t ring_lds_bytes8(int nw, int nq, int nr) { return (ring_lds_bytes(nw, nq, nr) + 7) & ~(size_t)7; }     // where the address tables start

// Context of a sweep that does not change between sweeps of one block
struct SweepCtx {
    CubeView cv; const uint32_t *samples; const float2 *Il; const float *cw;
    float invN, rlo2, ring_signed; int nr, nw;
    float *ringA, *sumB, *sumC; double *score;     // LDS: per-wave ring tables (ring_lds_bytes) and the slot scores
    CubeTab tab;                                   // LDS address tables (TAB = true)
};

// One sweep over the ring-ordered sample list for the poses of `plan` (in LDS): scores of all slots -> C.score[].
// All threads of the block call.  Shared by k_local (single-image compass search) and k_csp_eval (constrained search).
//
// Per sample and rotation the kernel issues ~75 vector instructions (round 3: ~150; the ISA was counted, CHANGELOG.md round 4):
//  * the image value is turned by the conjugate phase of the shift ONCE per sample (and slot, for shifted slots) and carries the
//    CTF weight, so a score term is Re(conj(b) v) = two instructions on the RAW interpolated value v instead of a complex rotation;
//  * the gathers of the next rotation are issued into a second register set (A / B ping-pong, the loop is unrolled by two): no
//    register copies between "next" and "current";
//  * slot numbers and the LDS cells of a group are wave-uniform and kept in scalar registers;
//  * tap addresses come from the LDS tables of ppm_dev.h (TAB), coordinates are pre-multiplied by the padding factor;
//  * the model power |c v|^2 goes to ONE LDS cell per group (bsl), not to one per slot.
template <bool TAB>
__device__ __forceinline__ void sweep_plan(SweepPlan &plan, const SweepCtx &C, const int tid, const int nthr) {
    const int lane = tid & 63, wave = tid >> 6, nw = C.nw, nr = C.nr;
    float *const ringA = C.ringA, *const sumB = C.sumB, *const sumC = C.sumC; double *const score = C.score;
    const float2 *const Il = C.Il; const float *const cw = C.cw; const float invN = C.invN;
#ifdef PPM_DBG_NOSWEEP
    const int nslots = plan.nslots, ng = plan.ng, S_used = 0;
#else
    const int nslots = plan.nslots, ng = plan.ng, S_used = plan.S_used;
#endif
    const float rmax2 = plan.rmax2;
    for (int i = tid; i < nslots * nw * nr; i += nthr) ringA[i] = 0.f;
    if (tid < kMaxCand * nw) sumB[tid] = 0.f;
    if (tid < ng * 6) (&plan.ms[0][0])[tid] = (&plan.m[0][0])[tid] * C.cv.scale;
    { const int g = nthr > 64 ? tid - 64 : tid; if (g >= 0 && g < ng) { const int q0 = plan.slot0[g]; for (int v = 0; v < plan.nv[g]; v++) plan.bsl[q0 + v] = q0; } }
    __syncthreads();
    float *const myA = ringA + wave * nr, *const myB = sumB + wave;
    const int q_same = plan.q_same, strideA = nw * nr;
    const bool head = (lane & 15) == 0;
    float accC = 0.f;
    for (int s0 = 0; s0 < S_used; s0 += nthr) {
        const int s = s0 + tid;
        int kx = 0, ky = 0, al = 0, ring = 0;
        float2 iv = make_float2(0.f, 0.f); float c = 0.f;
        if (s < S_used) {
            unpack_sample(C.samples[s], kx, ky, al, ring);
            float k2 = (float)(kx * kx + ky * ky);
            if (!(k2 < rmax2 && k2 >= C.rlo2)) al = 0;
            iv = Il[s]; c = cw[s];
        }
        const float fal = (float)al, fkx = (float)kx, fky = (float)ky;
        accC += fal * (iv.x * iv.x + iv.y * iv.y);
        const float ac = fal * c, ax = ac * iv.x, ay = ac * iv.y, w = ac * c;      // weighted image value (with the CTF weight) and the weight of |v|^2
        auto turned = [&](int q, float &bx, float &by) {              // conj(image value x e^{-i phase of slot q's shift}): score term = bx v.x + by v.y
            float rev = -(fkx * plan.sh[q][0] + fky * plan.sh[q][1]) * invN;       // phase in revolutions
            rev -= floorf(rev);
            const float sn = __sinf(6.283185307179586f * rev), cs = __cosf(6.283185307179586f * rev);
            bx = ax * cs + ay * sn; by = ay * cs - ax * sn;
        };
        float b0x, b0y;
        turned(0, b0x, b0y);           // the angular neighbours and the centre share one shift (slots 0 .. q_same)
        auto fetch = [&](int g) {
            const float *m = plan.ms[g];
            const float X = m[0] * fkx + m[1] * fky, Y = m[2] * fkx + m[3] * fky, Z = m[4] * fkx + m[5] * fky;
            if constexpr (TAB) return cube_fetch_tab(C.cv, C.tab, X, Y, Z);
            else { CubeView one = C.cv; one.scale = 1.f; return cube_fetch(one, X, Y, Z); }
        };
        const bool mine = head && ring < nr;
        // timing probe (results are wrong; scripts/ab_local.sh): PPM_DBG_NOATOM keeps the sums alive without LDS traffic
#if defined(PPM_DBG_NOATOM)
        auto ring_add = [&](float *cell, float v) { accC += 1e-30f * v; };
#else
        auto ring_add = [&](float *cell, float v) { atomicAdd(cell, v); };
#endif
        float *const cellA = myA + ring;
        auto score_group = [&](int g, const CubeTaps &t) {
            const float2 v = cube_interp(t);
            const int nv = __builtin_amdgcn_readfirstlane(plan.nv[g]), q0 = __builtin_amdgcn_readfirstlane(plan.slot0[g]);
            const float n2 = w * (v.x * v.x + v.y * v.y);
            if (nv == 1 && q0 <= q_same) {                           // an angular neighbour (or a single pose): two sums side by side
                float av = b0x * v.x + b0y * v.y, bv = n2;
                av += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(av), 0xB1, 0xF, 0xF, true));
                bv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(bv), 0xB1, 0xF, 0xF, true));
                av += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(av), 0x4E, 0xF, 0xF, true));
                bv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(bv), 0x4E, 0xF, 0xF, true));
                av += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(av), 0x141, 0xF, 0xF, true));
                bv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(bv), 0x141, 0xF, 0xF, true));
                av += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(av), 0x140, 0xF, 0xF, true));
                bv += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(bv), 0x140, 0xF, 0xF, true));
                if (mine) { ring_add(cellA + q0 * strideA, av); ring_add(myB + q0 * nw, bv); }
                return;
            }
            const float bv = group16_sum_dpp(n2);
            if (mine) ring_add(myB + q0 * nw, bv);
            for (int k = 0; k < nv; k++) {
                const int q = q0 + k;
                float bx = b0x, by = b0y;
                if (q > q_same) turned(q, bx, by);                   // a shifted probe (wave-uniform branch)
                const float av = group16_sum_dpp(bx * v.x + by * v.y);
                if (mine) ring_add(cellA + q * strideA, av);
            }
        };
        // the next group's gathers fly while this group is scored; A and B alternate.  Inside the loop both fetches are unconditional,
        // so the wait before a group's interpolation is for ITS four loads only (a fetch under a condition forces the wait for the
        // oldest loads of either path, i.e. for the prefetch as well)
        CubeTaps A = fetch(0), B;
        int g = 0;
        for (; g + 2 < ng; g += 2) {          // sched_barrier: the loads are issued BEFORE the scheduler may start on the other set's interpolation
            B = fetch(g + 1); __builtin_amdgcn_sched_barrier(0); score_group(g, A);
            A = fetch(g + 2); __builtin_amdgcn_sched_barrier(0); score_group(g + 1, B);
        }
        if (g + 1 < ng) { B = fetch(g + 1); __builtin_amdgcn_sched_barrier(0); score_group(g, A); score_group(g + 1, B); }
        else score_group(g, A);
    }
    accC = wave_sum(accC);
    if (lane == 0) sumC[wave] = accC;
    __syncthreads();
    {   // 8 lanes per slot: lane j takes the rings j, j + 8, ...; waves and lanes are combined in a fixed order
        const int j = tid & 7;
        for (int slot0_ = 0; slot0_ < nslots; slot0_ += nthr >> 3) {          // one trip with 128 or 256 threads (<= 13 slots), two with 64
        const int slot = slot0_ + (tid >> 3);
        double sa = 0;
        if (slot < nslots)
            for (int b = j; b < nr; b += 8) {
                const float *cell = ringA + (size_t)slot * nw * nr + b;
                float a = cell[0];
                for (int w = 1; w < nw; w++) a += cell[w * nr];
                sa += ((float)b <= C.ring_signed) ? (double)a : fabs((double)a);
            }
        sa = group8_sum_d(sa);
        if (slot < nslots && j == 0) {
            const int bs = plan.bsl[slot];
            float fb = sumB[bs * nw], fc = sumC[0];
            for (int w = 1; w < nw; w++) { fb += sumB[bs * nw + w]; fc += sumC[w]; }
            const double sb = fb, sc = fc;
            score[slot] = (sb > 0 && sc > 0) ? sa / sqrt(sb * sc) : 0.0;
        }
        }
    }
    __syncthreads();
}

// Block = one trajectory, 128 or 256 threads.  A compass iteration scores the centre and the neighbouring poses in
// one sweep over the ring-ordered samples (image value and CTF weight loaded once per sample; the centre and
// the four shift neighbours share one interpolated slice value), then one trial pose; all at the
// iteration's band (frequency marching: a prefix of the ring-ordered list).  Ring sums: 16-lane DPP
// reduction (a 16-lane group never straddles a ring), then one LDS add per group into the WAVE'S OWN ring table:
// a wave's adds happen in program order (and the lanes of one ds_add in lane order), the tables of the waves are
// combined in a fixed order afterwards, so a score does not depend on how the waves of the block interleave
// (bit-identical results from run to run; with one table shared by the waves a late compass decision could flip).
#ifndef PPM_LOCAL_MINW
#define PPM_LOCAL_MINW 4
#endif
template <bool TAB>
__global__ void __launch_bounds__(256, PPM_LOCAL_MINW) k_local(LocalP P) {
    __shared__ SweepPlan plan;
    extern __shared__ float lsm[];                    // ringA[slot][wave][nr], sumB[slot][wave], sumC[wave]; then the address tables (TAB)
    __shared__ double score[kMaxCand];
    __shared__ LState st;
    __shared__ double sfp[5], sfm[5], sd[5], sMt[9], sshq[2], sf0;
    __shared__ double spen[kMaxCand];                 // restraint of every slot's pose (use_priors)
    const int tid = threadIdx.x, nthr = blockDim.x;      // 128 or 256 threads (few samples per sweep: smaller blocks)
    const int nw = nthr >> 6, nr = P.nr;
    float *const ringA = lsm, *const sumB = lsm + kMaxCand * nw * nr, *const sumC = sumB + kMaxCand * nw;
    if (tid == 0) st = P.states[blockIdx.x];
    __syncthreads();
    const int part = st.particle;
    const float2 *Il = P.Il + (size_t)part * P.S_pad;
    const float *cw = P.cw + (size_t)part * P.S_pad;
    const float invN = 1.0f / (float)P.N;
    const int tilt = P.en[1] && P.en[2];

    SweepCtx SC;
    SC.cv = P.cv; SC.samples = P.samples; SC.Il = Il; SC.cw = cw; SC.invN = invN; SC.rlo2 = P.rlo2; SC.ring_signed = P.ring_signed;
    SC.nr = nr; SC.nw = nw; SC.ringA = ringA; SC.sumB = sumB; SC.sumC = sumC; SC.score = score;
    if constexpr (TAB) SC.tab = cube_tab_fill(P.cv, (char *)lsm + ring_lds_bytes8(nw, kMaxCand, nr), P.tabR, tid, nthr);    // the first sweep's barrier publishes them
    auto set_rot = [&](int g, const double *M) {
        plan.m[g][0] = (float)M[0]; plan.m[g][1] = (float)M[1]; plan.m[g][2] = (float)M[3];
        plan.m[g][3] = (float)M[4]; plan.m[g][4] = (float)M[6]; plan.m[g][5] = (float)M[7];
    };
    auto single = [&](const double *M, const double *sh) {      // plan for one pose (band fields are set by the caller)
        set_rot(0, M); plan.sh[0][0] = (float)sh[0]; plan.sh[0][1] = (float)sh[1];
        plan.nv[0] = 1; plan.slot0[0] = 0; plan.ng = 1; plan.nslots = 1; plan.q_same = 0;
    };

    int nfree = 0;
    for (int i = 0; i < 5; i++) nfree += P.en[i] ? 1 : 0;
    // One sweep call site: the kernel is a sequence of (build a plan, sweep, use the scores) steps -
    // per iteration the compass (centre + neighbours) and the trial pose, then the final score and the classification band.
    enum { PH_COMPASS, PH_TRIAL, PH_FINAL, PH_CLASS, PH_DONE };
    int it = 0, phase = PH_COMPASS;
    if (nfree == 0 || P.T <= 0) {
        if (tid == 0) for (int t = 0; t < P.T; t++) { st.ha *= 0.5; st.hs *= 0.5; }
        phase = P.final_rescore ? PH_FINAL : PH_DONE;
        __syncthreads();
    }
    while (phase != PH_DONE) {
        if (phase == PH_COMPASS) {
            // ---- slots: 2 per free angle (+h, -h), then the centre, then 2 per free shift.  Lanes 0..5 build one angular
            // neighbour each (the double-precision trig is the serial part of an iteration), lane 6 the centre group.
            int nang = 0;
            for (int i = 0; i < 3; i++) nang += P.en[i] ? 2 : 0;
            if (tid < 6) {
                const int i = tid >> 1, sg = tid & 1;
                if (P.en[i]) {
                    int g = 0;
                    for (int k = 0; k < i; k++) g += P.en[k] ? 2 : 0;
                    g += sg;
                    double Mq[9];
                    d_rot_step(st.M, i, tilt, sg ? -st.ha : st.ha, Mq);
                    set_rot(g, Mq); plan.nv[g] = 1; plan.slot0[g] = g;
                    plan.sh[g][0] = (float)st.sh[0]; plan.sh[g][1] = (float)st.sh[1];
                    if (P.use_priors) spen[g] = d_prior_pen(P, Mq, st.sh[0], st.sh[1]);
                }
            } else if (tid == 6) {
                int q = nang, g = nang;
                set_rot(g, st.M); plan.slot0[g] = q;
                plan.sh[q][0] = (float)st.sh[0]; plan.sh[q][1] = (float)st.sh[1];
                if (P.use_priors) spen[q] = d_prior_pen(P, st.M, st.sh[0], st.sh[1]);
                q++;
                int nv = 1;
                for (int i = 3; i < 5; i++) {
                    if (!P.en[i]) continue;
                    for (int sg = 0; sg < 2; sg++) {
                        double shq[2] = { st.sh[0], st.sh[1] };
                        shq[i - 3] += sg ? -st.hs : st.hs;
                        plan.sh[q][0] = (float)shq[0]; plan.sh[q][1] = (float)shq[1];
                        if (P.use_priors) spen[q] = d_prior_pen(P, st.M, shq[0], shq[1]);
                        q++; nv++;
                    }
                }
                plan.nv[g] = nv;
                plan.ng = g + 1; plan.nslots = q; plan.q_same = nang; plan.S_used = P.S_it[it]; plan.rmax2 = P.rmax2_it[it];
            }
        } else if (phase == PH_FINAL) {
            if (tid == 0) { single(st.M, st.sh); plan.S_used = P.S_final; plan.rmax2 = P.rmax2_final; }
        } else if (phase == PH_CLASS) {                  // the same pose over the classification band: LOGP / SIGMA of the row
            if (tid == 0) { plan.S_used = P.S_class; plan.rmax2 = P.rmax2_class; }
        }                                                // PH_TRIAL: the plan was written when the compass scores were read
        __syncthreads();
        sweep_plan<TAB>(plan, SC, tid, nthr);
        if (tid == 0) {
            if (phase == PH_COMPASS) {
                int q = 0, qc = 0;
                for (int i = 0; i < 3; i++) qc += P.en[i] ? 2 : 0;
                if (P.use_priors) for (int k = 0; k < plan.nslots; k++) score[k] -= spen[k];
                const double f0 = score[qc];
                sf0 = f0;
                for (int i = 0; i < 5; i++) {
                    sd[i] = 0; sfp[i] = sfm[i] = -1e300;
                    if (i == 3) q = qc + 1;
                    if (!P.en[i]) continue;
                    double h = i < 3 ? st.ha : st.hs;
                    sfp[i] = score[q]; sfm[i] = score[q + 1]; q += 2;
                    double den = 2.0 * f0 - sfp[i] - sfm[i];
                    if (den > 1e-12) {
                        double t = 0.5 * h * (sfp[i] - sfm[i]) / den;
                        sd[i] = t > h ? h : (t < -h ? -h : t);
                    } else {
                        double best = sfp[i] > sfm[i] ? sfp[i] : sfm[i];
                        sd[i] = best > f0 ? (sfp[i] > sfm[i] ? h : -h) : 0.0;
                    }
                }
                double T9[9];
                for (int k = 0; k < 9; k++) sMt[k] = st.M[k];
                for (int i = 0; i < 3; i++) if (P.en[i] && sd[i] != 0) { d_rot_step(sMt, i, tilt, sd[i], T9); for (int k = 0; k < 9; k++) sMt[k] = T9[k]; }
                sshq[0] = st.sh[0] + sd[3]; sshq[1] = st.sh[1] + sd[4];
                single(sMt, sshq);                       // band fields stay those of the iteration
            } else if (phase == PH_TRIAL) {
                const double ft = score[0] - (P.use_priors ? d_prior_pen(P, sMt, sshq[0], sshq[1]) : 0.0), f0 = sf0;
                int bi = -1, bs = 0; double fb = f0;
                for (int i = 0; i < 5; i++) {
                    if (!P.en[i]) continue;
                    if (sfp[i] > fb) { fb = sfp[i]; bi = i; bs = 1; }
                    if (sfm[i] > fb) { fb = sfm[i]; bi = i; bs = -1; }
                }
                st.f = f0;
                if (ft > f0 && ft >= fb) {
                    for (int k = 0; k < 9; k++) st.M[k] = sMt[k];
                    st.sh[0] = sshq[0]; st.sh[1] = sshq[1]; st.f = ft;
                } else if (bi >= 0) {
                    if (bi < 3) { double T9[9]; d_rot_step(st.M, bi, tilt, bs * st.ha, T9); for (int k = 0; k < 9; k++) st.M[k] = T9[k]; }
                    else st.sh[bi - 3] += bs * st.hs;
                    st.f = fb;
                }
                st.ha *= 0.5; st.hs *= 0.5;
            } else if (phase == PH_FINAL) st.f = st.fc = score[0];
            else st.fc = score[0];
        }
        if (phase == PH_COMPASS) phase = PH_TRIAL;
        else if (phase == PH_TRIAL) { it++; phase = it < P.T ? PH_COMPASS : (P.final_rescore ? PH_FINAL : PH_DONE); }
        else if (phase == PH_FINAL) phase = P.S_class > 0 ? PH_CLASS : PH_DONE;
        else phase = PH_DONE;
        __syncthreads();
    }
    if (tid == 0) P.states[blockIdx.x] = st;
}

// states from global-search hits: one thread per (particle, hit)
__global__ void k_states_from_hits(const Hit *hits, LState *states, int n, int K, const double *dir_theta,
                                   const double *dir_phi, int n_psi, double dpsi, double step, double ha0, double hs0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    Hit h = hits[i];
    LState s;
    int dir = h.orient / n_psi, k = h.orient - dir * n_psi;
    d_euler(k * dpsi, dir_theta[dir], dir_phi[dir], s.M);
    s.sh[0] = h.sx * step; s.sh[1] = h.sy * step;
    s.f = h.cc; s.fc = h.cc; s.ha = ha0; s.hs = hs0; s.particle = i / K; s.pad = 0;
    states[i] = s;
}

__global__ void k_states_from_rows(const double *rows, LState *states, int n, double a, double ha0, double hs0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *r = rows + (size_t)i * PPM_NCOL;
    LState s;
    d_euler(r[PPM_PSI], r[PPM_THETA], r[PPM_PHI], s.M);
    s.sh[0] = r[PPM_XSHIFT] / a; s.sh[1] = r[PPM_YSHIFT] / a;
    s.f = 0; s.fc = 0; s.ha = ha0; s.hs = hs0; s.particle = i; s.pad = 0;
    states[i] = s;
}

// best of K trajectories per particle (first wins ties)
__global__ void k_select_best(const LState *in, LState *out, int n, int K) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = 0;
    for (int k = 1; k < K; k++) if (in[(size_t)i * K + k].f > in[(size_t)i * K + b].f) b = k;
    out[i] = in[(size_t)i * K + b];
}

// ---------------------------------------------------------------------------------- defocus refinement
// One block per particle: the projection at the final pose is gathered once per sample; every defocus offset only changes
// the CTF factor, so all 2 nt + 1 scores come out of one sweep (ring-wise sums per offset in LDS).  Best offset: highest
// score, the unshifted one on ties, then the lower index (the oracle's scan order).
struct DefocusP {
    CubeView cv; const uint32_t *samples; const float2 *Il; const float *wring; int S_pad, nrings, N, B;
    float rlo2, rmax2, ring_signed, a;
    const double *rows; LState *states; float *ddef;   // ddef[n]: offset chosen (Angstrom)
    int nt; float step;
    int tchunk;            // offsets scored per pass over the samples (sizes the dynamic LDS)
    double *all_scores;    // null, or [n][2 nt + 1]: every offset's score is written out and nothing is chosen (constrained search)
    float rcls2;           // > 0: the chosen offset is scored once more over the classification band (answer 22) -> states[p].fc
};

__global__ void __launch_bounds__(256) k_defocus(DefocusP P) {
    constexpr int MAXT = 2 * PPM_MAX_DEFOCUS_STEPS + 1, NW = 4;
    extern __shared__ float dsm[];                    // ringA[TC][NW][nrings], sumB[TC][NW], sumC[NW]: per-wave tables as in k_local
    __shared__ double score[MAXT];
    __shared__ CtfP ctf0;
    __shared__ float m_s[6], sh_s[2];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = 2 * P.nt + 1, TC = P.tchunk, nr = P.nrings;
    float *const ringA = dsm, *const sumB = dsm + (size_t)TC * NW * nr, *const sumC = sumB + TC * NW;
    if (tid == 0) {
        const LState &st = P.states[p];
        ctf0 = ctf_from_row(P.rows + (size_t)p * PPM_NCOL, P.N, (double)P.a);
        m_s[0] = (float)st.M[0]; m_s[1] = (float)st.M[1]; m_s[2] = (float)st.M[3]; m_s[3] = (float)st.M[4]; m_s[4] = (float)st.M[6]; m_s[5] = (float)st.M[7];
        sh_s[0] = (float)st.sh[0]; sh_s[1] = (float)st.sh[1];
    }
    __syncthreads();
    const float2 *Il = P.Il + (size_t)p * P.S_pad;
    const float *wr = P.wring + (size_t)p * (P.B + 2);
    const float invN = 1.0f / (float)P.N;
    // offsets are scored TC at a time (the per-wave ring tables of all 2 nt + 1 offsets do not fit the LDS at wide bands);
    // every chunk gathers the projection again
    auto score_offsets = [&](const int t0, const int tn, const float rmax2) {      // offsets t0 .. t0 + tn - 1 over k^2 < rmax2 -> score[t0 ..]
        for (int i = tid; i < tn * NW * nr + TC * NW; i += 256) { if (i < tn * NW * nr) ringA[i] = 0.f; else sumB[i - tn * NW * nr] = 0.f; }
        __syncthreads();
        float *const myA = ringA + wave * nr, *const myB = sumB + wave;
        float accC = 0.f;
        for (int s0 = 0; s0 < P.S_pad; s0 += 256) {
            const int s = s0 + tid;
            int kx = 0, ky = 0, al = 0, ring = 0;
            float2 iv = make_float2(0.f, 0.f);
            if (s < P.S_pad) {
                unpack_sample(P.samples[s], kx, ky, al, ring);
                const float k2 = (float)(kx * kx + ky * ky);
                if (!(k2 < rmax2 && k2 >= P.rlo2)) al = 0;
                iv = Il[s];
            }
            const float fal = (float)al, fkx = (float)kx, fky = (float)ky;
            accC += fal * (iv.x * iv.x + iv.y * iv.y);
            float2 pv = sample_cube(P.cv, m_s[0] * fkx + m_s[1] * fky, m_s[2] * fkx + m_s[3] * fky, m_s[4] * fkx + m_s[5] * fky);
            float rev = -(fkx * sh_s[0] + fky * sh_s[1]) * invN;
            rev -= floorf(rev);
            const float sn = __sinf(6.283185307179586f * rev), cs = __cosf(6.283185307179586f * rev);
            const float mr = pv.x * cs - pv.y * sn, mi = pv.x * sn + pv.y * cs;
            const float a0 = fal * (iv.x * mr + iv.y * mi), b0 = fal * (pv.x * pv.x + pv.y * pv.y), wgt = al ? wr[ring] : 0.f;
            for (int t = 0; t < tn; t++) {
                CtfP c = ctf0;
                c.dsum += 2.f * (float)(t0 + t - P.nt) * P.step;          // both defocus values move by the offset
                const float ct = ctf_eval(c, kx, ky) * wgt;
                const float av = group16_sum_dpp(a0 * ct), bv = group16_sum_dpp(b0 * ct * ct);
                if ((lane & 15) == 0 && ring < nr) { atomicAdd(&myA[(size_t)t * NW * nr + ring], av); atomicAdd(&myB[t * NW], bv); }
            }
        }
        accC = wave_sum(accC);
        if (lane == 0) sumC[wave] = accC;
        __syncthreads();
        for (int tb = 0; tb < tn; tb += 32) {          // 8 lanes per offset, 32 offsets per pass
            const int t = tb + (tid >> 3), j = tid & 7;
            double sa = 0;
            if (t < tn)
                for (int b = j; b < nr; b += 8) {
                    const float *cell = ringA + (size_t)t * NW * nr + b;
                    const float a = ((cell[0] + cell[nr]) + cell[2 * nr]) + cell[3 * nr];
                    sa += ((float)b <= P.ring_signed) ? (double)a : fabs((double)a);
                }
            sa = group8_sum_d(sa);
            if (t < tn && j == 0) {
                const double sb = ((sumB[t * NW] + sumB[t * NW + 1]) + sumB[t * NW + 2]) + sumB[t * NW + 3];
                const double sc = ((sumC[0] + sumC[1]) + sumC[2]) + sumC[3];
                score[t0 + t] = (sb > 0 && sc > 0) ? sa / sqrt(sb * sc) : 0.0;
            }
        }
        __syncthreads();
    };
    for (int t0 = 0; t0 < T; t0 += TC) score_offsets(t0, min(TC, T - t0), P.rmax2);
    if (P.all_scores) {
        if (tid < T) P.all_scores[(size_t)p * T + tid] = score[tid];
        return;
    }
    __shared__ int s_bt;
    if (tid == 0) {
        int bt = P.nt; double bf = score[P.nt];                  // the unshifted CTF is the incumbent
        for (int t = 0; t < T; t++) if (t != P.nt && score[t] > bf) { bf = score[t]; bt = t; }
        P.states[p].f = P.states[p].fc = bf;
        P.ddef[p] = (float)(bt - P.nt) * P.step;
        s_bt = bt;
    }
    if (P.rcls2 > 0.f) {                                         // LOGP / SIGMA: the chosen CTF over the classification band
        __syncthreads();
        const int bt = s_bt;
        score_offsets(bt, 1, P.rcls2);
        if (tid == 0) P.states[p].fc = score[bt];
    }
}

__global__ void k_rows_out(const LState *states, const double *rows_in, double *rows_out, int n, double a,
                           double r_cls, double r_lo, const float *ddef) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const LState &s = states[i];
    double *o = rows_out + (size_t)i * PPM_NCOL;
    const double *r = rows_in + (size_t)i * PPM_NCOL;
    for (int c = 0; c < PPM_NCOL; c++) o[c] = r[c];
    double psi, th, ph; d_angles(s.M, psi, th, ph);
    o[PPM_PSI] = psi; o[PPM_THETA] = th; o[PPM_PHI] = ph;
    o[PPM_XSHIFT] = s.sh[0] * a; o[PPM_YSHIFT] = s.sh[1] * a;
    if (ddef) { o[PPM_DF1] = r[PPM_DF1] + (double)ddef[i]; o[PPM_DF2] = r[PPM_DF2] + (double)ddef[i]; }
    o[PPM_SCORE] = 100.0 * s.f;
    // LOGP / SIGMA over r_lo .. r_cls (answer 22; r_cls = r_hi and fc = f when no classification limit applies)
    double cc = s.fc, res = 1.0 - cc * cc; if (res < 1e-6) res = 1e-6;
    o[PPM_SIGMA] = sqrt(res);
    o[PPM_LOGP] = -0.5 * (3.14159265358979323846 * (r_cls * r_cls - r_lo * r_lo)) * (log(2.0 * 3.14159265358979323846 * res) + 1.0);
}

// ---------------------------------------------------------------------------------- brick-binned insertion
// No per-sample global atomics: the accumulator is cut into bricks of BE^3 voxels; one block owns one brick of one half-map
// (and one slice of the chunk's particles), keeps it in LDS in 64-bit fixed point, finds the slice samples whose 8-tap
// footprint touches the brick and adds them with integer LDS atomics; the brick is written back once (plain
// read-modify-write if the block is the brick's only owner in this launch, float atomics per brick cell otherwise).
struct PartIns {       // per-particle constants, written by k_insert_params
    float m[6];        // X = m0 kx + m1 ky, Y = m2 kx + m3 ky, Z = m4 kx + m5 ky
    CtfP ctf;
    float sx, sy, w0, wexp;   // shifts (px), occupancy weight, exponent coefficient of the score weighting (per k^2)
    float dexp, dcap2;        // dose weighting: weight x exp(dexp min(k^2, dcap2)) (dexp <= 0; 0 = off)
    int half, valid;
};

// One work item = one brick x one slice of the batch's particles (heavy bricks near the origin are cut into more slices,
// so that items carry about equal work; the host sorts them heavy-first).
struct BrickItem { unsigned short bx, by, bz; unsigned char s, S; };

// one entry per (particle, symmetry operator): unit normal of the inserted slice plane; flag = half-map, or -1 if skipped
struct CullEnt { float nx, ny, nz; int flag; };

struct InsertBrickP {
    const float2 *band; const PartIns *pp; const CullEnt *cull; const float *symops; int nsym;
    float *acc; int N, B, W, H, n_img;
    const BrickItem *items;
    const unsigned *maxima;   // [0] bits of max |band| component (k_prep), [1] bits of max particle weight (k_insert_params); floats >= 0, atomicMax
    float r2;
};

__global__ void k_insert_params(const double *rows, PartIns *pp, CullEnt *cull, const float *symops, int nsym, int n, int N, double a, double bfac,
                                double score_avg, double score_thr, int split_by_pind, double r2, unsigned long long *counts, unsigned *maxima,
                                const float *dose_q, int n_dose, float dose_exponent, float dose_cap2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *row = rows + (size_t)i * PPM_NCOL;
    PartIns q;
    const double occ = row[PPM_OCC], scr = row[PPM_SCORE];
    q.valid = (occ > 0 && !(scr < score_thr)) ? 1 : 0;
    long key = split_by_pind ? (long)row[PPM_PIND] : (long)row[PPM_POS];
    q.half = (int)(((key % 2) + 2) % 2);
    double M[9]; d_euler(row[PPM_PSI], row[PPM_THETA], row[PPM_PHI], M);
    q.m[0] = (float)M[0]; q.m[1] = (float)M[1]; q.m[2] = (float)M[3]; q.m[3] = (float)M[4]; q.m[4] = (float)M[6]; q.m[5] = (float)M[7];
    q.ctf = ctf_from_row(row, N, a);
    q.sx = (float)(row[PPM_XSHIFT] / a); q.sy = (float)(row[PPM_YSHIFT] / a);
    q.w0 = (float)(occ / 100.0);
    q.wexp = (float)(-0.25 * bfac * (score_avg - scr)) * q.ctf.inv_na2;
    q.dexp = 0.f; q.dcap2 = dose_cap2;
    if (dose_q) {
        const long t = (long)row[PPM_TIND];
        const float dq = (t >= 0 && t < n_dose) ? dose_q[t] : 0.f;
        if (dq > 0.f && dq < 1.f) q.dexp = dose_exponent * logf(dq) / dose_cap2;
    }
    pp[i] = q;
    for (int so = 0; so < nsym; so++) {
        const float *S = symops + so * 9;
        const float m2 = (float)M[2], m5 = (float)M[5], m8 = (float)M[8];
        CullEnt c;
        c.nx = S[0] * m2 + S[1] * m5 + S[2] * m8; c.ny = S[3] * m2 + S[4] * m5 + S[5] * m8; c.nz = S[6] * m2 + S[7] * m5 + S[8] * m8;
        c.flag = q.valid ? q.half : -1;
        cull[(size_t)i * nsym + so] = c;
    }
    if (q.valid) {
        atomicAdd(&counts[q.half], 1ull);
        const float wmax = q.w0 * fmaxf(1.f, expf(q.wexp * (float)r2));
        if (wmax > 0.f && wmax < 3.0e38f) atomicMax(maxima + 1, __float_as_uint(wmax));
    }
}

// grid: (items, 2 halves), NW waves.  A brick OWNS the samples whose base voxel floor(Q) lies in its BE^3 box and keeps a
// one-voxel halo on the + faces for their upper taps ((BE+1)^3 cells in LDS): every sample is evaluated exactly once, none
// of its 8 taps needs a bounds test, and the halo cells are added to the neighbours' voxels when the brick is written back.
// Per round of up to CULL_CAP (particle, operator) entries of the item's slice:
//   CULL  every thread tests entries (slice-plane normal vs the brick's box) and the cutting ones are collected in
//         a block-wide LDS list;
//   WORK  waves pull cuts from that list (dynamic balance); for one cut, lane = slice row: the kx interval that can reach
//         the brick is solved per row (three slabs + the band), the candidates of all rows are dealt out densely over the
//         lanes (prefix sum + search), TESTED exactly (position only), the hits compacted into a per-wave LDS queue, and
//         full groups of 64 hits EVALUATED (CTF, weights, phase, 8 taps).
// The brick is accumulated in 64-bit FIXED POINT: on gfx950 a ds_add_f32 wave-instruction occupies the LDS for ~190
// cycles (lanes are serialised), a ds_add_u64 for ~8 (scripts/micro/lds_atomic_bench.hip).  Every tap is rounded to a
// 31-bit integer relative to the largest possible value of the chunk (max |band| x max weight, found on the device
// beforehand), then summed exactly: the LDS part of the sum does not depend on the order of the adds.
constexpr int CULL_CAP = 4096;
// Diagnostic build (-DPPM_INS_STAMPS): cycles per phase (s_memtime), summed over all waves of a launch, and event counts
#ifdef PPM_INS_STAMPS
__device__ unsigned long long g_ins_stamps[24];
#define INS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#define INS_COUNT(i, v) do { st_acc[i] += (unsigned long long)(v); } while (0)
#else
#define INS_STAMP(i) do { } while (0)
#define INS_COUNT(i, v) do { } while (0)
#endif
template <int BE, int NW>
__global__ void __launch_bounds__(NW * 64) k_insert_bricks(InsertBrickP P) {
    extern __shared__ long long brick[];           // [BE+1][BE+1][BE+1][3] with padded row / plane strides SY, SZ (in 8-byte cells):
    constexpr int BH = BE + 1, SY = BH * 3 + 1, SZ = BH * SY + 3;   // odd strides spread the 8 taps of neighbouring samples over the banks
                                                       // (unpadded, 3/4 of the LDS atomic cycles were bank conflicts)
    __shared__ unsigned queue_s[NW][128];
    __shared__ unsigned deal_s[NW][64];
    __shared__ int cut_list[CULL_CAP];
    __shared__ int n_cut, cut_head;
    // value scale 2^(30-e) with bound < 2^e, so |tap| < 2^30; likewise for the weight channel
    float sv, sw;
    {
        int ev, ew;
        const float bv = __uint_as_float(P.maxima[0]) * __uint_as_float(P.maxima[1]), bw = __uint_as_float(P.maxima[1]);
        (void)frexpf(bv > 0.f ? bv : 1.f, &ev); (void)frexpf(bw > 0.f ? bw : 1.f, &ew);
        sv = ldexpf(1.f, min(30 - ev, 120)); sw = ldexpf(1.f, min(30 - ew, 120));
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = P.N, B = P.B, W = P.W;
    const float invN = 1.0f / (float)N;
    const int h = blockIdx.y;
    const BrickItem it = P.items[blockIdx.x];
    // voxel COORDINATES covered by this brick: x in [x_lo, x_lo+BE), y, z likewise (stored index = coordinate + N/2)
    const int x_lo = it.bx * BE, y_lo = it.by * BE - N / 2, z_lo = it.bz * BE - N / 2;
    const int p_lo = (int)((long)P.n_img * it.s / it.S), p_hi = (int)((long)P.n_img * (it.s + 1) / it.S);
#ifdef PPM_INS_STAMPS
    unsigned long long st_acc[24] = { 0 }, st_t = __builtin_amdgcn_s_memtime();
#endif
    for (int i = tid; i < BH * SZ; i += NW * 64) brick[i] = 0ll;
    __syncthreads();
    INS_STAMP(0);
    // sample positions Q whose floor() lies in [lo, lo+BE-1] belong to the brick:  lo <= Q < lo+BE
    const float cx = x_lo + 0.5f * BE, cy = y_lo + 0.5f * BE, cz = z_lo + 0.5f * BE, hh = 0.5f * BE;
    unsigned *queue = queue_s[wave];
    volatile unsigned *deal = deal_s[wave];
    int qn = 0;
    bool touched = false;
    const int e_hi = p_hi * P.nsym;
    for (int r0 = p_lo * P.nsym; r0 < e_hi; r0 += CULL_CAP) {
    const int r1 = min(r0 + CULL_CAP, e_hi);
    if (tid == 0) { n_cut = 0; cut_head = 0; }
    __syncthreads();
    for (int eb = r0; eb < r1; eb += NW * 64) {          // CULL
        const int e = eb + tid;
        bool cut = false;
        if (e < r1) {
            const CullEnt c = P.cull[e];
            cut = c.flag == h && fabsf(c.nx * cx + c.ny * cy + c.nz * cz) <= (fabsf(c.nx) + fabsf(c.ny) + fabsf(c.nz)) * hh + 1e-3f;
        }
        const unsigned long long m = __ballot(cut);
        if (m != 0ull) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&n_cut, (int)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (cut) cut_list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = e;
        }
    }
    INS_STAMP(1);
    __syncthreads();
    INS_STAMP(2);
    const int nc = n_cut;
    for (;;) {                                           // WORK
        int ci = 0;
        if (lane == 0) ci = atomicAdd(&cut_head, 1);
        ci = __builtin_amdgcn_readfirstlane(ci);
        if (ci >= nc) break;
        const int idx = __builtin_amdgcn_readfirstlane(cut_list[ci]);
        const int p = idx / P.nsym, so = idx - p * P.nsym;
        const PartIns &q = P.pp[p];
        const float *S = P.symops + so * 9;
        // first two columns of S M
        const float a0 = S[0] * q.m[0] + S[1] * q.m[2] + S[2] * q.m[4], a1 = S[0] * q.m[1] + S[1] * q.m[3] + S[2] * q.m[5];
        const float b0 = S[3] * q.m[0] + S[4] * q.m[2] + S[5] * q.m[4], b1 = S[3] * q.m[1] + S[4] * q.m[3] + S[5] * q.m[5];
        const float c0 = S[6] * q.m[0] + S[7] * q.m[2] + S[8] * q.m[4], c1 = S[6] * q.m[1] + S[7] * q.m[3] + S[8] * q.m[5];
        // sample position along one axis: explicit rounding steps, so that the hit test and the evaluation — two places the compiler
        // is free to contract differently — agree to the last bit on which brick owns a sample
        auto pos = [](float u, float v, int kx, int ky) { return __fmaf_rn(u, (float)kx, __fmul_rn(v, (float)ky)); };
        // EVALUATE one queued sample per lane
        auto evaluate = [&](unsigned e, bool on) {
            if (!on) return;
            const int kx = (int)(e & 0xffffu), ky = (int)(e >> 16) - 512;
            const float k2 = (float)(kx * kx + ky * ky);
            float X = pos(a0, a1, kx, ky), Y = pos(b0, b1, kx, ky), Z = pos(c0, c1, kx, ky);       // bit-identical to the hit test's position
            const bool refl = X < 0.f;
            if (refl) { X = -X; Y = -Y; Z = -Z; }
            const float xf = floorf(X), yf = floorf(Y), zf = floorf(Z);
            const int x0 = (int)xf - x_lo, y0 = (int)yf - y_lo, z0 = (int)zf - z_lo;      // brick-local base tap
            if ((unsigned)x0 >= (unsigned)BE || (unsigned)y0 >= (unsigned)BE || (unsigned)z0 >= (unsigned)BE) return;   // cannot happen (same arithmetic as the test); never write outside the brick
            const float fx = X - xf, fy = Y - yf, fz = Z - zf;
            const float cv = ctf_eval_fast(q.ctf, kx, ky);
            float w = q.w0 * (q.wexp != 0.f ? expf(q.wexp * k2) : 1.f);
            if (q.dexp != 0.f) w *= expf(q.dexp * fminf(k2, q.dcap2));
            float rev = (kx * q.sx + ky * q.sy) * invN; rev -= floorf(rev);
            const float sn = __sinf(6.283185307179586f * rev), cs = __cosf(6.283185307179586f * rev);
            const float2 iv = P.band[((size_t)p * P.H + (ky + B)) * W + kx];
            const float vr = sv * w * cv * (iv.x * cs - iv.y * sn), vw = sw * w * cv * cv;
            float vi = sv * w * cv * (iv.x * sn + iv.y * cs);
            if (refl) vi = -vi;
            unsigned long long *const v0 = (unsigned long long *)brick + z0 * SZ + y0 * SY + x0 * 3;      // base voxel inside the box: all taps inside the haloed brick
#pragma unroll
            for (int dz = 0; dz < 2; dz++)
#pragma unroll
                for (int dy = 0; dy < 2; dy++)
#pragma unroll
                    for (int dx = 0; dx < 2; dx++) {
                        const float wt = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy) * (dz ? fz : 1.f - fz);
                        unsigned long long *v = v0 + dz * SZ + dy * SY + dx * 3;
                        atomicAdd(v, (unsigned long long)(long long)__float2int_rn(wt * vr));
                        atomicAdd(v + 1, (unsigned long long)(long long)__float2int_rn(wt * vi));
                        atomicAdd(v + 2, (unsigned long long)(long long)__float2int_rn(wt * vw));
                    }
        };
        // candidate rectangle of (kx, ky) = (col0 . Q, col1 . Q) over the box, for Q = +P (sgn +1) and Q = -P (sgn -1)
        const float ea = (fabsf(a0) + fabsf(b0) + fabsf(c0)) * hh, eb = (fabsf(a1) + fabsf(b1) + fabsf(c1)) * hh;
        const float ka = a0 * cx + b0 * cy + c0 * cz, kb = a1 * cx + b1 * cy + c1 * cz;
        INS_STAMP(3); INS_COUNT(12, 1);
#pragma unroll 1
        for (int sgn = 1; sgn >= -1; sgn -= 2) {
            int kx0 = (int)floorf(sgn * ka - ea) - 1, kx1 = (int)ceilf(sgn * ka + ea) + 1;
            int ky0 = (int)floorf(sgn * kb - eb) - 1, ky1 = (int)ceilf(sgn * kb + eb) + 1;
            kx0 = kx0 < 0 ? 0 : kx0; kx1 = kx1 > B ? B : kx1; ky0 = ky0 < -B ? -B : ky0; ky1 = ky1 > B ? B : ky1;
            if (ky1 > ky0 + 63) ky1 = ky0 + 63;               // cannot happen for BE <= 16 (at most 2 sqrt(3) (BE+1)/2 + 4 rows)
            if (kx1 < kx0 || ky1 < ky0) continue;
            // lane = slice row ky0 + lane: the kx interval on which s (kx A + ky B) can fall into the brick's expanded box and
            // the band (conservative by one sample on both sides; the exact test follows)
            const int kyr = ky0 + lane;
            int lo = kx0, hi = kyr <= ky1 ? kx1 : kx0 - 1;
            {
                const float rem = P.r2 - (float)(kyr * kyr);
                if (rem <= 0.f) hi = lo - 1; else { const int m = (int)sqrtf(rem) + 1; hi = hi < m ? hi : m; }
                const float fs = (float)sgn, fky = fs * (float)kyr;
                const float aa[3] = { fs * a0, fs * b0, fs * c0 }, tt[3] = { fky * a1, fky * b1, fky * c1 };
                const float LL[3] = { (float)x_lo, (float)y_lo, (float)z_lo };
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    const float a = aa[ax], t = tt[ax], L = LL[ax], U = LL[ax] + (float)BE;
                    if (fabsf(a) > 1e-3f) {
                        const float ra = __frcp_rn(a), e0 = (L - t) * ra, e1 = (U - t) * ra;
                        const float el = fminf(fmaxf(fminf(e0, e1), -1024.f), 1024.f), eh = fminf(fmaxf(fmaxf(e0, e1), -1024.f), 1024.f);
                        const int l2 = (int)ceilf(el) - 1, h2 = (int)floorf(eh) + 1;
                        lo = lo > l2 ? lo : l2; hi = hi < h2 ? hi : h2;
                    } else if (t < L - 0.5f || t > U + 0.5f) hi = lo - 1;
                }
            }
            const int cnt = hi >= lo ? hi - lo + 1 : 0;
            const int incl = wave_scan_add(cnt);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            const int start = incl - cnt, base = lo - start;      // kx of candidate j in this row = base + j
            // rows are dealt out over the lanes through a 64-entry LDS line: every row with candidates marks the lane of its first one
            // with (row + 1, base), a running maximum along the lanes (DPP) carries the mark to the row's other candidates
            const unsigned mark = ((unsigned)(lane + 1) << 16) | (unsigned)((base + 32768) & 0xffff);
            int carry = 0;
            INS_STAMP(4); INS_COUNT(13, total);
#pragma unroll 1
            for (int j0 = 0; j0 < total; j0 += 64) {
                const int j = j0 + lane;
                deal[lane] = 0u;
                if (cnt > 0 && (unsigned)(start - j0) < 64u) deal[start - j0] = mark;
                __builtin_amdgcn_wave_barrier();
                int mk = (int)deal[lane];
                if (lane == 0 && mk == 0) mk = carry;             // the row that began in an earlier group of 64
                mk = wave_scan_max0(mk);
                carry = __builtin_amdgcn_readlane(mk, 63);
                __builtin_amdgcn_wave_barrier();
                const int r = (mk >> 16) - 1;
                const int kx = (mk & 0xffff) - 32768 + j, ky = ky0 + (r < 0 ? 0 : r);
                const float k2 = (float)(kx * kx + ky * ky);
                float X = pos(a0, a1, kx, ky), Y = pos(b0, b1, kx, ky), Z = pos(c0, c1, kx, ky);
                const bool refl = X < 0.f;
                if (refl) { X = -X; Y = -Y; Z = -Z; }
                const int x0 = (int)floorf(X) - x_lo, y0 = (int)floorf(Y) - y_lo, z0 = (int)floorf(Z) - z_lo;
                const bool hit = j < total && k2 < P.r2 && k2 != 0.f && refl == (sgn < 0) &&
                                 (unsigned)x0 < (unsigned)BE && (unsigned)y0 < (unsigned)BE && (unsigned)z0 < (unsigned)BE;
                const unsigned long long m = __ballot(hit);
                if (m == 0ull) continue;
                if (hit) queue[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned)kx | ((unsigned)(ky + 512) << 16);
                qn += __popcll(m);
                INS_COUNT(14, __popcll(m));
                __builtin_amdgcn_wave_barrier();
                INS_STAMP(5);
                if (qn >= 64) {
                    qn -= 64;
                    evaluate(queue[qn + lane], true);
                    __builtin_amdgcn_wave_barrier();
                    INS_STAMP(6); INS_COUNT(15, 1);
                }
                touched = true;
            }
            INS_STAMP(5);
        }
        if (qn > 0) {              // the queue never crosses a (particle, operator): the rotation above is wave-uniform
            evaluate(queue[lane], lane < qn);
            __builtin_amdgcn_wave_barrier();
            qn = 0;
            INS_STAMP(7); INS_COUNT(16, 1);
        }
    }
    INS_STAMP(3);
    __syncthreads();
    INS_STAMP(8);
    }
    const int any = __syncthreads_or(touched ? 1 : 0);
#ifdef PPM_INS_STAMPS
    if (!any) { if (lane == 0) for (int i = 0; i < 24; i++) if (st_acc[i]) atomicAdd(&g_ins_stamps[i], st_acc[i]); return; }
#endif
    if (!any) return;
    const size_t NX = N / 2 + 1;
    float *A = P.acc + (size_t)h * N * N * NX * 3;
    for (int i = tid; i < BH * BH * BH * 3; i += NW * 64) {       // BH*3 consecutive floats per (y, z) row of the brick
        const int x3 = i % (BH * 3), yi = (i / (BH * 3)) % BH, zi = i / (BH * BH * 3);
        const long long vq = brick[zi * SZ + yi * SY + x3];
        if (vq == 0ll) continue;
        const int gy = y_lo + yi + N / 2, gz = z_lo + zi + N / 2;
        if (x_lo * 3 + x3 >= (int)NX * 3 || gy >= N || gz >= N) continue;
        float *o = A + (((size_t)gz * N + gy) * NX + x_lo) * 3 + x3;
        const float v = (float)((double)vq / (double)(i % 3 == 2 ? sw : sv));
        atomicAdd(o, v);                 // halo cells belong to the neighbours' boxes: no voxel has a sole owner
    }
    INS_STAMP(9);
#ifdef PPM_INS_STAMPS
    if (lane == 0) for (int i = 0; i < 24; i++) if (st_acc[i]) atomicAdd(&g_ins_stamps[i], st_acc[i]);
#endif
}
#undef INS_STAMP
#undef INS_COUNT

// ---------------------------------------------------------------------------------- matching projections
// refine3d answers 8 / 43 (frealign.py:3929-3931, refine_fmatch): the reference projected at a row's pose, times the row's CTF,
// moved to the particle's position — the noise-free model of the stored particle image.  One thread per entry of the full
// N x N spectrum (FFT order; kx < 0 from the Hermitian mate); the inverse transforms are two k_fft_lines passes.
struct MatchRow { float m[6]; float sx, sy; CtfP ctf; };
struct MatchP { CubeView cv; const MatchRow *rows; float2 *f; int N, B; float r_hi2; int n; };

__global__ void __launch_bounds__(256) k_match_fill(MatchP P) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, NN = (size_t)P.N * P.N;
    if (i >= NN * P.n) return;
    const int im = (int)(i / NN), r = (int)(i - (size_t)im * NN), y = r / P.N, x = r - y * P.N;
    int kx = x < P.N / 2 ? x : x - P.N, ky = y < P.N / 2 ? y : y - P.N;
    const bool mate = kx < 0;
    if (mate) { kx = -kx; ky = -ky; }
    float2 v = make_float2(0.f, 0.f);
    const float k2 = (float)(kx * kx + ky * ky);
    if (kx <= P.B && ky >= -P.B && ky <= P.B && k2 < P.r_hi2) {
        const MatchRow &q = P.rows[im];
        const float fx = (float)kx, fy = (float)ky;
        const float2 s = sample_cube(P.cv, q.m[0] * fx + q.m[1] * fy, q.m[2] * fx + q.m[3] * fy, q.m[4] * fx + q.m[5] * fy);
        float c = ctf_eval(q.ctf, kx, ky);
        if ((kx + ky) & 1) c = -c;                                   // projection centred on pixel (N/2, N/2)
        float rev = -(fx * q.sx + fy * q.sy) / (float)P.N; rev -= floorf(rev);
        float sn, cs; __sincosf(6.283185307179586f * rev, &sn, &cs);
        v = make_float2(c * (s.x * cs - s.y * sn), c * (s.x * sn + s.y * cs));
        if (mate) v.y = -v.y;
    }
    P.f[i] = v;
}

__global__ void k_match_real(const float2 *__restrict__ f, float *__restrict__ out, size_t n, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = f[i].x * scale;
}

// ---------------------------------------------------------------------------------- finalisation
// kx = 0 plane: fold Friedel mates together (reads `src`, writes `dst`)
__global__ void k_fold_plane(const float *__restrict__ src, float *__restrict__ dst, int N) {
    size_t NX = N / 2 + 1, half_sz = (size_t)N * N * NX * 3;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)2 * N * N) return;
    int h = (int)(i / ((size_t)N * N)); int r = (int)(i % ((size_t)N * N));
    int yi = r % N, zi = r / N, y = yi - N / 2, z = zi - N / 2;
    if (y == -N / 2 || z == -N / 2) return;
    const float *m = src + h * half_sz + (((size_t)(-z + N / 2) * N + (-y + N / 2)) * NX) * 3;
    const float *v = src + h * half_sz + (((size_t)zi * N + yi) * NX) * 3;
    float *o = dst + h * half_sz + (((size_t)zi * N + yi) * NX) * 3;
    o[0] = v[0] + m[0]; o[1] = v[1] - m[1]; o[2] = v[2] + m[2];
}

// pass 1: shell sums of the weights.  sums: [4][ns] = den1, den2, count, den1+den2
// (both shell passes first add up per block in LDS and then issue one global atomic per shell the block touched: a global
// double atomic per voxel into <= 256 addresses serialised the whole pass)
__global__ void __launch_bounds__(256) k_shell_den(const float *__restrict__ acc, double *sums, int N) {
    __shared__ double part[4 * 256];
    const int ns = N / 2, tid = threadIdx.x;
    for (int k = tid; k < 4 * ns; k += 256) part[k] = 0.0;
    __syncthreads();
    size_t NX = N / 2 + 1, tot = (size_t)N * N * NX, half_sz = tot * 3;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tot) {
        int x = (int)(i % NX), y = (int)((i / NX) % N) - N / 2, z = (int)(i / (NX * N)) - N / 2;
        int b = (int)floorf(sqrtf((float)(x * x + y * y + z * z)) + 0.5f);
        if (b < ns) {
            double al = x == 0 ? 1.0 : 2.0;
            double d1 = acc[i * 3 + 2], d2 = acc[half_sz + i * 3 + 2];
            atomicAdd(&part[b], al * d1); atomicAdd(&part[ns + b], al * d2); atomicAdd(&part[2 * ns + b], al); atomicAdd(&part[3 * ns + b], al * (d1 + d2));
        }
    }
    __syncthreads();
    for (int k = tid; k < 4 * ns; k += 256) if (part[k] != 0.0) atomicAdd(&sums[k], part[k]);
}

// pass 2: FSC sums between the two halves.  fsc: [3][ns] = c12, c11, c22
__global__ void __launch_bounds__(256) k_shell_fsc(const float *__restrict__ acc, const double *sums, double *fsc, int N) {
    __shared__ double part[3 * 256];
    const int ns = N / 2, tid = threadIdx.x;
    for (int k = tid; k < 3 * ns; k += 256) part[k] = 0.0;
    __syncthreads();
    size_t NX = N / 2 + 1, tot = (size_t)N * N * NX, half_sz = tot * 3;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tot) {
        int x = (int)(i % NX), y = (int)((i / NX) % N) - N / 2, z = (int)(i / (NX * N)) - N / 2;
        int b = (int)floorf(sqrtf((float)(x * x + y * y + z * z)) + 0.5f);
        if (b < ns) {
            double cnt = sums[2 * ns + b];
            double e1 = 1e-3 * sums[b] / cnt + 1e-20, e2 = 1e-3 * sums[ns + b] / cnt + 1e-20;
            double d1 = acc[i * 3 + 2] + e1, d2 = acc[half_sz + i * 3 + 2] + e2;
            double ar = acc[i * 3] / d1, ai = acc[i * 3 + 1] / d1, br = acc[half_sz + i * 3] / d2, bi = acc[half_sz + i * 3 + 1] / d2;
            double al = x == 0 ? 1.0 : 2.0;
            atomicAdd(&part[b], al * (ar * br + ai * bi)); atomicAdd(&part[ns + b], al * (ar * ar + ai * ai)); atomicAdd(&part[2 * ns + b], al * (br * br + bi * bi));
        }
    }
    __syncthreads();
    for (int k = tid; k < 3 * ns; k += 256) if (part[k] != 0.0) atomicAdd(&fsc[k], part[k]);
}

// Wiener division into a full N^3 complex spectrum in FFT order (which: 0/1 = half maps, 2 = sum)
__global__ void k_wiener(const float *__restrict__ acc, const double *kappa, float2 *f, int N, int which) {
    size_t NX = N / 2 + 1, tot = (size_t)N * N * NX, half_sz = tot * 3;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= tot) return;
    int x = (int)(i % NX), y = (int)((i / NX) % N) - N / 2, z = (int)(i / (NX * N)) - N / 2;
    int ns = N / 2, b = (int)floorf(sqrtf((float)(x * x + y * y + z * z)) + 0.5f);
    if (b >= ns) return;
    double nr, ni, dn;
    if (which < 2) { nr = acc[which * half_sz + i * 3]; ni = acc[which * half_sz + i * 3 + 1]; dn = acc[which * half_sz + i * 3 + 2]; }
    else { nr = (double)acc[i * 3] + acc[half_sz + i * 3]; ni = (double)acc[i * 3 + 1] + acc[half_sz + i * 3 + 1]; dn = (double)acc[i * 3 + 2] + acc[half_sz + i * 3 + 2]; }
    double d = dn + kappa[b], sg = ((x + y + z) & 1) ? -1.0 : 1.0;
    float vr = (float)(sg * nr / d), vi = (float)(sg * ni / d);
    int ix = x % N, iy = (y + N) % N, iz = (z + N) % N;
    f[((size_t)iz * N + iy) * N + ix] = make_float2(vr, vi);
    if (x > 0 && x < N / 2) f[((size_t)((N - iz) % N) * N + ((N - iy) % N)) * N + (N - ix)] = make_float2(vr, -vi);
}

__global__ void k_map_post(const float2 *__restrict__ f, float *__restrict__ out, int N, float rout, float rin, float fo) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n3 = (size_t)N * N * N;
    if (i >= n3) return;
    int x = (int)(i % N), y = (int)((i / N) % N), z = (int)(i / ((size_t)N * N));
    float dx = (float)(x - N / 2), dy = (float)(y - N / 2), dz = (float)(z - N / 2);
    float g3 = 1.f, t[3] = { dx / N, dy / N, dz / N };
#pragma unroll
    for (int q = 0; q < 3; q++) { float u = kPiF * t[q], sv = fabsf(u) < 1e-6f ? 1.f : sinf(u) / u; g3 *= sv * sv; }
    float rho = sqrtf(dx * dx + dy * dy + dz * dz), m = 1.f;
    if (rout > 0.f) {
        if (rho >= rout + 0.5f * fo) m = 0.f;
        else if (rho > rout - 0.5f * fo) m = 0.5f * (1.f + cosf(kPiF * (rho - rout + 0.5f * fo) / fo));
    }
    if (rin > 0.f && rho < rin) m = 0.f;
    out[i] = f[i].x / ((float)N * (float)N) / g3 * m;
}

// ---------------------------------------------------------------------------------- particle extraction
struct ExtractP {
    const float *image; int rows, cols;
    const double *coords; int box; double cbin; float radius2; int normalize, fix_empty;
    float *out;
};

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// Block = one particle.  Window bounds exactly as the reference computes them (including its habit of dropping
// the last row / column when the window touches the far edge); pass 1 = mean of the inside part, pass 2 =
// emptiness probes + background statistics, pass 3 = write.  The micrograph window is re-read from L2 / Infinity Cache.
__global__ void __launch_bounds__(256) k_extract(ExtractP P) {
    __shared__ double red[4][5];
    __shared__ double s_fill, s_mu, s_sd;
    __shared__ int s_empty;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, box = P.box;
    const double bx = P.coords[2 * p], by = P.coords[2 * p + 1];
    int minx = 0, miny = 0, maxx = box, maxy = box;
    int minX = (int)floor(by / P.cbin - floor(box / 2.0)), maxX = minX + box;
    int minY = (int)floor(bx / P.cbin - floor(box / 2.0)), maxY = minY + box;
    if (minX < 0) { minx = -minX; minX = 0; } else if (maxX >= P.rows) { maxx = box - (maxX - P.rows + 1); maxX = P.rows - 1; }
    if (minY < 0) { miny = -minY; minY = 0; } else if (maxY >= P.cols) { maxy = box - (maxY - P.cols + 1); maxY = P.cols - 1; }
    // python slices clamp: inside = image[minX:maxX, minY:maxY]
    const int iX0 = minX < P.rows ? minX : P.rows, iX1 = maxX < P.rows ? (maxX > iX0 ? maxX : iX0) : P.rows;
    const int iY0 = minY < P.cols ? minY : P.cols, iY1 = maxY < P.cols ? (maxY > iY0 ? maxY : iY0) : P.cols;
    const int nX = iX1 - iX0, nY = iY1 - iY0;
    const bool has_inside = nX > 0 && nY > 0;
    auto raw_at = [&](int r, int c, float fill) -> float {       // raw[r][c]: the window pixel or the fill value
        int ir = r - minx, ic = c - miny;
        bool in = has_inside && r >= minx && r < maxx && c >= miny && c < maxy && ir < nX && ic < nY;
        return in ? P.image[(size_t)(iX0 + ir) * P.cols + (iY0 + ic)] : fill;
    };
    auto block_sum = [&](double v, int slot) {
        v = wave_sum_d(v);
        if (lane == 0) red[wave][slot] = v;
    };
    // ---- pass 1: mean of the inside part
    // All passes walk the window row by row (a wave covers 64 consecutive pixels of one row, no index division) and keep
    // UR independent loads in flight per thread: with one load per loop trip every pass ran at memory latency.
    constexpr int UR = 8;
    auto sweep = [&](int nr, int ncl, auto load, auto use) {
        for (int r0 = wave; r0 < nr; r0 += 4 * UR)
            for (int c = lane; c < ncl; c += 64) {
                float v[UR];
#pragma unroll
                for (int u = 0; u < UR; u++) { const int r = r0 + 4 * u; v[u] = r < nr ? load(r, c) : 0.f; }
#pragma unroll
                for (int u = 0; u < UR; u++) { const int r = r0 + 4 * u; if (r < nr) use(r, c, v[u]); }
            }
    };
    double s = 0;
    if (has_inside)
        sweep(nX, nY, [&](int r, int c) { return P.image[(size_t)(iX0 + r) * P.cols + iY0 + c]; }, [&](int, int, float v) { s += v; });
    block_sum(s, 0);
    __syncthreads();
    if (tid == 0) s_fill = has_inside ? (red[0][0] + red[1][0] + red[2][0] + red[3][0]) / ((double)nX * nY) : 0.0;
    __syncthreads();
    const float fill = (float)s_fill;
    // ---- pass 2: emptiness probes (three candidate "constant" values, min/max) together with the background statistics
    // (outside radius_px of the box centre) of the window as it is; an "empty" window (rare) is replaced by noise and its
    // statistics are taken again.
    const int npx = box * box;
    double b1 = 0, b2 = 0, bc = 0;
    auto stat_use = [&](int r, int c, float vf) {
        const int dr = r - box / 2, dc = c - box / 2;
        if ((float)(dr * dr + dc * dc) > P.radius2) { const double v = vf; b1 += v; b2 += v * v; bc += 1; }
    };
    if (P.fix_empty) {
        const float c0 = raw_at(0, 0, fill), c1 = raw_at(box / 2, box / 2, fill), c2 = fill;
        double n0 = 0, n1 = 0, n2 = 0; float mn = 3e38f, mx = -3e38f;
        sweep(box, box, [&](int r, int c) { return raw_at(r, c, fill); }, [&](int r, int c, float v) {
            n0 += v == c0; n1 += v == c1; n2 += v == c2; mn = fminf(mn, v); mx = fmaxf(mx, v);
            if (P.normalize) stat_use(r, c, v);
        });
        for (int m = 32; m >= 1; m >>= 1) { mn = fminf(mn, __shfl_xor(mn, m, 64)); mx = fmaxf(mx, __shfl_xor(mx, m, 64)); }
        __syncthreads();
        block_sum(n0, 0); block_sum(n1, 1); block_sum(n2, 2); block_sum((double)mn, 3); block_sum((double)mx, 4);
        __syncthreads();
        if (tid == 0) {
            double t0 = 0, t1 = 0, t2 = 0, gmn = 3e38, gmx = -3e38;
            for (int w = 0; w < 4; w++) { t0 += red[w][0]; t1 += red[w][1]; t2 += red[w][2]; gmn = fmin(gmn, red[w][3]); gmx = fmax(gmx, red[w][4]); }
            double most = fmax(t0, fmax(t1, t2));
            s_empty = (gmn == gmx) || ((double)npx - most < 0.01 * npx);
        }
        __syncthreads();
    } else {
        if (tid == 0) s_empty = 0;
        if (P.normalize) sweep(box, box, [&](int r, int c) { return raw_at(r, c, fill); }, stat_use);
        __syncthreads();
    }
    const bool empty = s_empty != 0;
    auto value_at = [&](int r, int c) -> float {       // the frame that gets normalised
        if (!empty) return raw_at(r, c, fill);
        const int i = r * box + c;
        unsigned h1 = hash32((unsigned)p * 2654435761u + (unsigned)i * 2u + 1u), h2 = hash32(h1 ^ 0x9e3779b9u);
        float u1 = ((h1 >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
        return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);       // unit white noise instead of numpy's
    };
    if (empty && P.normalize) {
        b1 = b2 = bc = 0;
        sweep(box, box, [&](int r, int c) { return value_at(r, c); }, stat_use);
    }
    __syncthreads();
    block_sum(b1, 0); block_sum(b2, 1); block_sum(bc, 2);
    __syncthreads();
    if (tid == 0) {
        double a1 = 0, a2 = 0, ac = 0;
        for (int w = 0; w < 4; w++) { a1 += red[w][0]; a2 += red[w][1]; ac += red[w][2]; }
        double mu = ac > 0 ? a1 / ac : 0.0, var = ac > 0 ? a2 / ac - mu * mu : 0.0;
        s_mu = P.normalize ? mu : 0.0; s_sd = (P.normalize && var > 0) ? sqrt(var) : 1.0;
    }
    __syncthreads();
    const double mu = s_mu, isd = 1.0 / s_sd;
    float *o = P.out + (size_t)p * npx;
    sweep(box, box, [&](int r, int c) { return value_at(r, c); }, [&](int r, int c, float v) { o[r * box + c] = (float)(((double)v - mu) * isd); });
}

__global__ void k_axpy(float *__restrict__ y, const float *__restrict__ x, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}

}  // namespace ppm
