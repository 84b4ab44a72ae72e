/* ppm_oracle.c — CPU restatement of the projection-matching + Fourier-insertion path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pyp_amd/ may import, link or execute this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED.  The arithmetic PYP runs on this path lives in binaries whose source is not in
 * the reference checkout (git-LFS pointers: external/cistem2/{refine3d,reconstruct3d,merge3d},
 * external/frealign_v9.11/bin/refine3d) and the reference's regression
 * fixtures (tests/test_pyp.py:445-491) are absent, so no golden vector pins these numbers.
 * What is restated here is the published FREALIGN / cisTEM method (Grigorieff 2007, 2016;
 * Lyumkis et al. 2013; Grant, Rohou & Grigorieff 2018) on the call surface the reference does
 * show: the refine3d answers (src/pyp/refine/frealign/frealign.py:3918-3994), the reconstruct3d
 * answers (:1780-1824), merge3d (:2075-2093, table parsed at :2557-2567), the 32-column rows
 * (src/pyp/inout/metadata/cistem_star_file.py:596-628), angles in degrees / shifts in Angstrom
 * (src/pyp/analysis/scores.py:693) and the ZYZ Euler order phi -> theta -> psi
 * (src/pyp/analysis/geometry/core.py:1186-1197).  The oracle is pinned instead by synthetic
 * ground truth (tests/test_oracle.py): poses recovered from projections of a known volume.
 *
 * Style: plain loops, double accumulators, no tricks.  The global-search correlation image is
 * computed BOTH ways: by a zero-filled inverse 2-D FFT (mode 0, the textbook statement) and by
 * the pruned direct transform over the shift window (mode 1, what the HIP kernel does).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ppm.h"

#define ORC_PI 3.14159265358979323846

typedef struct { float re, im; } cpx;

/* ------------------------------------------------------------------ FFT (mixed radix 2, 3, 4, 5) */
/* Decimation in time over the factors of n (4s first, then 2, 3, 5): the input is permuted so that the
 * r_m interleaved sub-sequences sit in consecutive blocks, then every stage combines r blocks of length L
 * into one of length r L with the textbook r-point DFT and twiddles w^(q j), w = e^(-+2 pi i/(r L)).
 * Twiddle and permutation tables are cached per length (built before any parallel region). */
#define ORC_MAXN 4096
typedef struct { int n, nfac, fac[16]; double *c, *s; int *perm; } fplan_t;
static fplan_t *PLANS[ORC_MAXN + 1];

static int fft_size_ok(int n) {
    if (n < 2 || n > ORC_MAXN) return 0;
    while (n % 2 == 0) n /= 2;
    while (n % 3 == 0) n /= 3;
    while (n % 5 == 0) n /= 5;
    while (n % 7 == 0) n /= 7;
    return n == 1;
}

static fplan_t *fft_plan(int n) {
    if (PLANS[n]) return PLANS[n];
    fplan_t *p = (fplan_t *)calloc(1, sizeof(fplan_t));
    p->n = n;
    int m = n;
    while (m % 4 == 0) { p->fac[p->nfac++] = 4; m /= 4; }
    while (m % 2 == 0) { p->fac[p->nfac++] = 2; m /= 2; }
    while (m % 3 == 0) { p->fac[p->nfac++] = 3; m /= 3; }
    while (m % 5 == 0) { p->fac[p->nfac++] = 5; m /= 5; }
    while (m % 7 == 0) { p->fac[p->nfac++] = 7; m /= 7; }
    p->c = (double *)malloc(sizeof(double) * n); p->s = (double *)malloc(sizeof(double) * n);
    for (int k = 0; k < n; k++) { p->c[k] = cos(2.0 * ORC_PI * k / n); p->s[k] = sin(2.0 * ORC_PI * k / n); }
    /* position of input sample i: P_m(i) = (i mod r_m) L_{m-1} + P_{m-1}(i div r_m) */
    p->perm = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) {
        int pos = 0, rem = i, L = n;
        for (int st = p->nfac - 1; st >= 0; st--) { int r = p->fac[st]; L /= r; pos += (rem % r) * L; rem /= r; }
        p->perm[i] = pos;
    }
    PLANS[n] = p;
    return p;
}

static void fft_tables(void) {
    /* plans for every size the library accepts, built once (call before any parallel region) */
    static int done = 0;
    if (done) return;
    for (int n = 2; n <= 1024; n++) if (fft_size_ok(n)) fft_plan(n);
    done = 1;
}

static void fft1d(cpx *x, int n, int stride, int inverse) {
    const fplan_t *p = PLANS[n];
    double br[ORC_MAXN > 1024 ? 1024 : ORC_MAXN], bi[1024];
    for (int i = 0; i < n; i++) { br[p->perm[i]] = x[i * stride].re; bi[p->perm[i]] = x[i * stride].im; }
    int Lp = 1;
    for (int st = 0; st < p->nfac; st++) {
        const int r = p->fac[st], L = Lp * r, tws = n / L, rs = n / r;
        for (int blk = 0; blk < n; blk += L) for (int j = 0; j < Lp; j++) {
            double xr[7], xi[7];
            for (int q = 0; q < r; q++) {
                double vr = br[blk + q * Lp + j], vi = bi[blk + q * Lp + j];
                int t = q * j * tws;                         /* < n */
                double wr = p->c[t], wi = inverse ? p->s[t] : -p->s[t];
                xr[q] = vr * wr - vi * wi; xi[q] = vr * wi + vi * wr;
            }
            for (int pp = 0; pp < r; pp++) {
                double sr = 0, si = 0;
                for (int q = 0; q < r; q++) {
                    int t = ((pp * q) % r) * rs;
                    double wr = p->c[t], wi = inverse ? p->s[t] : -p->s[t];
                    sr += xr[q] * wr - xi[q] * wi; si += xr[q] * wi + xi[q] * wr;
                }
                br[blk + pp * Lp + j] = sr; bi[blk + pp * Lp + j] = si;
            }
        }
        Lp = L;
    }
    for (int i = 0; i < n; i++) { x[i * stride].re = (float)br[i]; x[i * stride].im = (float)bi[i]; }
}

/* exported for the FFT known-answer test (numpy.fft) */
int orc_fft1d(float *data /* n interleaved complex */, int n, int inverse) {
    fft_tables();
    if (!fft_size_ok(n) || n > 1024) return -22;
    fft1d((cpx *)data, n, 1, inverse);
    return 0;
}

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
static int box_ok(int n) { return n >= 16 && n <= 512 && n % 2 == 0 && fft_size_ok(n); }

/* ------------------------------------------------------------------ derived geometry */
typedef struct {
    int N; double a;
    double r_hi, r_lo, r_s, ring_signed;
    double r_cls;           /* band of LOGP / SIGMA (answer 22, ppm_refine_cfg.res_classification); = r_hi when unset */
    int B, W, H;            /* band half-width, row width B+1, rows 2B+1 */
    int Ns, RSx, RSy; double step; /* global-search shift grid: Ns points over the box, step = N/Ns pixels */
    int n_theta, n_psi, n_dir, n_orient;
    double dpsi, phi_max, theta_max;
} geom_t;

/* asymmetric unit of the global grid: phi < 360/n for Cn and Dn, theta <= 90 for Dn; T and I use the D2 unit they
 * contain, O the D4 unit */
static void sym_limits(const char *sym, double *phi_max, double *theta_max) {
    *phi_max = 360.0; *theta_max = 180.0;
    if (!sym || !sym[0]) return;
    char t = sym[0] >= 'a' ? sym[0] - 32 : sym[0];
    int n = atoi(sym + 1);
    if (t == 'C' && n >= 1) { *phi_max = 360.0 / n; }
    else if (t == 'D' && n >= 1) { *phi_max = 360.0 / n; *theta_max = 90.0; }
    else if (t == 'T' || t == 'I') { *phi_max = 180.0; *theta_max = 90.0; }
    else if (t == 'O') { *phi_max = 90.0; *theta_max = 90.0; }
}

static int geom_init(geom_t *g, const ppm_refine_cfg *c) {
    memset(g, 0, sizeof(*g));
    g->N = c->box; g->a = c->pixel_size;
    if (!box_ok(g->N) || g->a <= 0 || c->res_high <= 0) return -1;
    double na = g->N * g->a;
    g->r_hi = na / c->res_high; if (g->r_hi > g->N / 2) g->r_hi = g->N / 2;
    g->r_lo = c->res_low > 0 ? na / c->res_low : 0.0;
    g->r_s = c->res_search > 0 ? na / c->res_search : g->r_hi; if (g->r_s > g->r_hi) g->r_s = g->r_hi;
    if (c->global_search && g->r_s > 64.0) g->r_s = 64.0;   /* the grid search never uses more than 64 Fourier pixels (ppm.h) */
    g->ring_signed = c->res_signed_cc > 0 ? na / c->res_signed_cc : 1e30;
    /* answer 22 (frealign.py:3945): 0, beyond res_high, or a band of less than one Fourier pixel above r_lo -> the full band */
    g->r_cls = c->res_classification > 0 ? na / c->res_classification : g->r_hi;
    if (g->r_cls > g->r_hi || g->r_cls < g->r_lo + 1.0) g->r_cls = g->r_hi;
    g->B = (int)ceil(g->r_hi) - 1; g->W = g->B + 1; g->H = 2 * g->B + 1;
    int Bs = (int)ceil(g->r_s) - 1;
    g->Ns = 2; while (g->Ns < 2 * (Bs + 1)) g->Ns <<= 1;
    g->step = (double)g->N / g->Ns;
    /* answers 27 / 28: 0 = the mask radius (config/pyp_config.toml:5338-5343); limited only by the search grid (Ns / 2 - 1 steps) */
    double rx = (c->search_range_x > 0 ? c->search_range_x : c->mask_radius) / g->a, ry = (c->search_range_y > 0 ? c->search_range_y : c->mask_radius) / g->a;
    g->RSx = (int)ceil(rx / g->step); g->RSy = (int)ceil(ry / g->step);
    if (g->RSx < 1) g->RSx = 1;
    if (g->RSy < 1) g->RSy = 1;
    if (g->RSx > g->Ns / 2 - 1) g->RSx = g->Ns / 2 - 1;
    if (g->RSy > g->Ns / 2 - 1) g->RSy = g->Ns / 2 - 1;
    double d = c->angular_step > 0 ? c->angular_step : 15.0;
    sym_limits(c->symmetry, &g->phi_max, &g->theta_max);
    g->n_theta = (int)floor(g->theta_max / d + 0.5) + 1;
    if (g->n_theta < 2) g->n_theta = 2;
    g->n_psi = (int)floor(360.0 / d + 0.5); if (g->n_psi < 1) g->n_psi = 1;
    g->dpsi = 360.0 / g->n_psi;
    g->n_dir = 0;
    for (int i = 0; i < g->n_theta; i++) {
        double th = g->theta_max * i / (g->n_theta - 1);
        int np = (int)floor(g->phi_max * sin(th * ORC_PI / 180.0) / d + 0.5); if (np < 1) np = 1;
        g->n_dir += np;
    }
    g->n_orient = g->n_dir * g->n_psi;
    return 0;
}

/* direction list of the global grid: theta_i = 180 i/(n_theta-1), n_phi = max(1, round(360 sin(theta)/step)) */
static void grid_direction(const geom_t *g, double dstep, int dir, double *theta, double *phi) {
    int acc = 0;
    for (int i = 0; i < g->n_theta; i++) {
        double th = g->theta_max * i / (g->n_theta - 1);
        int np = (int)floor(g->phi_max * sin(th * ORC_PI / 180.0) / dstep + 0.5); if (np < 1) np = 1;
        if (dir < acc + np) { *theta = th; *phi = g->phi_max * (dir - acc) / np; return; }
        acc += np;
    }
    *theta = 0; *phi = 0;
}

static void mat_mul3(const double *a, const double *b, double *c) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double v = 0; for (int k = 0; k < 3; k++) v += a[i * 3 + k] * b[k * 3 + j];
        t[i * 3 + j] = v;
    }
    memcpy(c, t, sizeof(t));
}

/* M = Rz(phi) Ry(theta) Rz(psi); first two columns (the image plane) */
static void euler_cols(double psi, double theta, double phi, double m[6]) {
    double ps = psi * ORC_PI / 180, th = theta * ORC_PI / 180, ph = phi * ORC_PI / 180;
    double cps = cos(ps), sps = sin(ps), cth = cos(th), sth = sin(th), cph = cos(ph), sph = sin(ph);
    m[0] = cph * cth * cps - sph * sps;  m[1] = -cph * cth * sps - sph * cps;  /* row x */
    m[2] = sph * cth * cps + cph * sps;  m[3] = -sph * cth * sps + cph * cps;  /* row y */
    m[4] = -sth * cps;                   m[5] = sth * sps;                     /* row z */
}

/* ------------------------------------------------------------------ reference cube */
/* K1: what refine3d does with answer 4 "input reconstruction" (frealign.py:3923) at padding 1 (answer 35, :3962);
 * the sinc^2 pre-compensation is the real-space counterpart of trilinear interpolation in Fourier space. */
typedef struct { int N, B, CX, CY, pad; cpx *cube; } oref_t;   /* B, CX, CY count samples of the PADDED transform */

/* Reference preparation (answers "input reconstruction" and "padding factor" refine_iblow,
 * frealign.py:3923, :3962): the volume, divided by the sinc^2 envelope of trilinear interpolation, is embedded in the centre
 * of a (pad n)^3 box of zeros and transformed; the transform is then sampled pad times finer, a slice sample at image
 * frequency k sits at pad k.  pad = 1, 2 or 4 with pad n <= 512. */
void *orc_reference_create_padded(const float *vol, int n, float max_band_px, int pad) {
    fft_tables();
    if (!box_ok(n) || max_band_px <= 0 || (pad != 1 && pad != 2 && pad != 4) || n * pad > 512) return NULL;
    if (max_band_px > n / 2) max_band_px = n / 2;
    const int np = n * pad, o0 = (np - n) / 2;
    int B = (int)ceil((double)max_band_px * pad) - 1;
    if (B > np / 2 - 1) B = np / 2 - 1;
    size_t n3 = (size_t)np * np * np;
    cpx *f = (cpx *)calloc(n3, sizeof(cpx));
    if (!f) return NULL;
    /* pre-compensate the trilinear interpolation kernel: divide by sinc^2 along each axis
     * (the real-space envelope that linear interpolation of the padded transform imposes) */
    double *sc1 = (double *)malloc(n * sizeof(double));
    for (int i = 0; i < n; i++) {
        double u = ORC_PI * (double)(i - n / 2) / np, sv = fabs(u) < 1e-12 ? 1.0 : sin(u) / u;
        sc1[i] = 1.0 / (sv * sv);
    }
    for (int z = 0; z < n; z++) for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        size_t i = ((size_t)z * n + y) * n + x, o = ((size_t)(z + o0) * np + (y + o0)) * np + (x + o0);
        f[o].re = (float)(vol[i] * sc1[x] * sc1[y] * sc1[z]); f[o].im = 0;
    }
    free(sc1);
    for (int z = 0; z < np; z++) for (int y = 0; y < np; y++) fft1d(f + ((size_t)z * np + y) * np, np, 1, 0);
    for (int z = 0; z < np; z++) for (int x = 0; x < np; x++) fft1d(f + (size_t)z * np * np + x, np, np, 0);
    for (int y = 0; y < np; y++) for (int x = 0; x < np; x++) fft1d(f + (size_t)y * np + x, np, np * np, 0);
    oref_t *r = (oref_t *)calloc(1, sizeof(oref_t));
    r->N = n; r->pad = pad; r->B = B; r->CX = B + 2; r->CY = 2 * B + 3;
    r->cube = (cpx *)calloc((size_t)r->CX * r->CY * r->CY, sizeof(cpx));
    double sc = 1.0 / n;
    for (int z = -B - 1; z <= B + 1; z++) for (int y = -B - 1; y <= B + 1; y++) for (int x = 0; x <= B + 1; x++) {
        int iz = ((z % np) + np) % np, iy = ((y % np) + np) % np, ix = x % np;
        cpx v = f[((size_t)iz * np + iy) * np + ix];
        double sg = ((x + y + z) & 1) ? -sc : sc;      /* origin at the (padded) box centre */
        cpx *o = &r->cube[((size_t)(z + B + 1) * r->CY + (y + B + 1)) * r->CX + x];
        o->re = (float)(v.re * sg); o->im = (float)(v.im * sg);
    }
    free(f);
    return r;
}

void *orc_reference_create(const float *vol, int n, float max_band_px) { return orc_reference_create_padded(vol, n, max_band_px, 1); }

void orc_reference_destroy(void *p) { oref_t *r = (oref_t *)p; if (r) { free(r->cube); free(r); } }

/* trilinear sample of the cube at Fourier coordinate (X,Y,Z) */
static void sample_cube(const oref_t *r, double X, double Y, double Z, double *ore, double *oim) {
    int conj = 0;
    X *= r->pad; Y *= r->pad; Z *= r->pad;             /* the padded transform is sampled pad times finer */
    if (X < 0) { X = -X; Y = -Y; Z = -Z; conj = 1; }
    int x0 = (int)floor(X), y0 = (int)floor(Y), z0 = (int)floor(Z);
    double fx = X - x0, fy = Y - y0, fz = Z - z0;
    int off = r->B + 1;
    double sr = 0, si = 0;
    for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++) {
        double w = (dx ? fx : 1 - fx) * (dy ? fy : 1 - fy) * (dz ? fz : 1 - fz);
        const cpx *v = &r->cube[((size_t)(z0 + dz + off) * r->CY + (y0 + dy + off)) * r->CX + (x0 + dx)];
        sr += w * v->re; si += w * v->im;
    }
    *ore = sr; *oim = conj ? -si : si;
}

/* slice for one orientation into band layout [ky+B][kx], zero outside k^2 < rmax^2 */
static void extract_slice(const oref_t *r, const geom_t *g, const double m[6], double rmax, cpx *out) {
    double r2 = rmax * rmax;
    for (int ky = -g->B; ky <= g->B; ky++) for (int kx = 0; kx <= g->B; kx++) {
        cpx *o = &out[(size_t)(ky + g->B) * g->W + kx];
        if ((double)kx * kx + (double)ky * ky >= r2) { o->re = o->im = 0; continue; }
        double re, im;
        sample_cube(r, m[0] * kx + m[1] * ky, m[2] * kx + m[3] * ky, m[4] * kx + m[5] * ky, &re, &im);
        o->re = (float)re; o->im = (float)im;
    }
}

/* ------------------------------------------------------------------ CTF */
/* K3: from the row's DEFOCUS_1/2, DEFOCUS_ANGLE, PHASE_SHIFT, MICROSCOPE_VOLTAGE, MICROSCOPE_CS, AMPLITUDE_CONTRAST
 * (cistem_star_file.py:596-628); the .par surface passes kV / Cs / contrast as answers (wrapper_functions.py:526-528). */
typedef struct { double lambda, cs, df1, df2, ast, extra, inv_na2; } ctf_t;

static void ctf_init(ctf_t *c, const double *row, int N, double a) {
    double v = row[PPM_VOLTAGE] * 1000.0;
    c->lambda = 12.2639 / sqrt(v + 0.97845e-6 * v * v);
    c->cs = row[PPM_CS] * 1e7;
    c->df1 = row[PPM_DF1]; c->df2 = row[PPM_DF2];
    c->ast = row[PPM_ANGAST] * ORC_PI / 180.0;
    double w = row[PPM_AMP];
    c->extra = row[PPM_PSHIFT] + atan(w / sqrt(1.0 - w * w));
    c->inv_na2 = 1.0 / ((double)N * a * N * a);
}

static double ctf_eval(const ctf_t *c, int kx, int ky) {
    double k2 = (double)kx * kx + (double)ky * ky;
    if (k2 == 0) return -sin(c->extra);
    double s2 = k2 * c->inv_na2;
    double c2 = ((double)kx * kx - (double)ky * ky) / k2, s2a = 2.0 * kx * ky / k2;   /* cos 2phi, sin 2phi */
    double df = 0.5 * (c->df1 + c->df2 + (c->df1 - c->df2) * (c2 * cos(2 * c->ast) + s2a * sin(2 * c->ast)));
    double chi = ORC_PI * c->lambda * s2 * (df - 0.5 * c->cs * c->lambda * c->lambda * s2) + c->extra;
    return -sin(chi);
}

/* ------------------------------------------------------------------ particle preprocessing */
/* K2: answers 46 "normalize particles", 47 "invert contrast", 18 outer mask radius (frealign.py:3937, :3984-3988);
 * the background statistics are those of the stack's own normalisation (analysis/image.py:406-417). */
/* out: band layout [ky+B][kx] (zero outside k^2 < r_hi^2), whitened when `whiten`. */
/* `disc` (may be NULL): centre (pixels from the box centre) and radius (pixels) of the mask disc when it is not the centred one
 * of radius mask_radius_A — the focus mask of answers 29-32 / 44 (frealign.py:3846-3849, :3958); the background statistics
 * keep using mask_radius_A. */
/* `row` (may be NULL): BEAM_TILT_X / Y (mrad), voltage and Cs of the particle: its spectrum is multiplied by exp(-i phi),
 * phi(s) = 2 pi Cs lambda^2 |s|^2 (s . b) (include/ppm.h). */
static void preprocess_row(const float *img, const geom_t *g, double mask_radius_A, double falloff_A,
                       int normalize, int invert, int do_mask, int whiten, double rband, cpx *out,
                       double *wring /* B+2 ring weights 1/sqrt(P_b), or NULL */, const double *disc, const double *row) {
    int N = g->N;
    double Rm = mask_radius_A / g->a, w = falloff_A / g->a;
    const double mcx = disc ? disc[0] : 0.0, mcy = disc ? disc[1] : 0.0, mrad = disc ? disc[2] : Rm;
    double btx = 0, bty = 0;
    if (row) {
        const double v = row[PPM_VOLTAGE] * 1000.0, lam = 12.2639 / sqrt(v + 0.97845e-6 * v * v), na = (double)N * g->a;
        const double cc = 2.0 * ORC_PI * row[PPM_CS] * 1e7 * lam * lam * 1e-3 / (na * na * na);
        btx = cc * row[PPM_BTX]; bty = cc * row[PPM_BTY];
    }
    if (w < 1e-3) w = 1e-3;
    double s1 = 0, s2 = 0; long cnt = 0;
    for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) {
        double dx = x - N / 2, dy = y - N / 2;
        if (dx * dx + dy * dy > Rm * Rm) { double v = img[y * N + x]; s1 += v; s2 += v * v; cnt++; }
    }
    if (cnt < 16) {
        s1 = s2 = 0; cnt = (long)N * N;
        for (int i = 0; i < N * N; i++) { s1 += img[i]; s2 += (double)img[i] * img[i]; }
    }
    double mu = s1 / cnt, var = s2 / cnt - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
    double sgn = invert ? -1.0 : 1.0, sc = normalize ? 1.0 / sd : 1.0;
    cpx *f = (cpx *)malloc((size_t)N * N * sizeof(cpx));
    for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) {
        double dx = x - N / 2 - mcx, dy = y - N / 2 - mcy, rho = sqrt(dx * dx + dy * dy), m = 1.0;
        if (do_mask) {
            if (rho >= mrad + 0.5 * w) m = 0.0;
            else if (rho > mrad - 0.5 * w) m = 0.5 * (1.0 + cos(ORC_PI * (rho - mrad + 0.5 * w) / w));
        }
        f[y * N + x].re = (float)((img[y * N + x] - mu) * sc * sgn * m);
        f[y * N + x].im = 0;
    }
    for (int y = 0; y < N; y++) fft1d(f + (size_t)y * N, N, 1, 0);
    for (int x = 0; x < N; x++) fft1d(f + x, N, N, 0);
    double r2 = rband * rband, inv = 1.0 / N;
    int nr = g->B + 2;
    double *pw = (double *)calloc(nr, sizeof(double)), *pc = (double *)calloc(nr, sizeof(double));
    for (int ky = -g->B; ky <= g->B; ky++) for (int kx = 0; kx <= g->B; kx++) {
        cpx *o = &out[(size_t)(ky + g->B) * g->W + kx];
        double k2 = (double)kx * kx + (double)ky * ky;
        if (k2 >= r2 || k2 == 0) { o->re = o->im = 0; continue; }   /* DC dropped */
        cpx v = f[(size_t)((ky + N) % N) * N + kx];
        if (btx != 0 || bty != 0) {
            const double ph = k2 * (kx * btx + ky * bty), cr = cos(ph), ci = sin(ph);
            const double vr = v.re * cr + v.im * ci, vi = v.im * cr - v.re * ci;
            v.re = (float)vr; v.im = (float)vi;
        }
        double sg = ((kx + ky) & 1) ? -inv : inv;
        o->re = (float)(v.re * sg); o->im = (float)(v.im * sg);
        int b = (int)floor(sqrt(k2));
        double al = kx == 0 ? 1.0 : 2.0;
        pw[b] += al * ((double)o->re * o->re + (double)o->im * o->im); pc[b] += al;
    }
    if (wring) for (int b = 0; b < nr; b++) {
        double p = pc[b] > 0 ? pw[b] / pc[b] : 0;
        wring[b] = (whiten && p > 0) ? 1.0 / sqrt(p) : (whiten ? 0.0 : 1.0);
    }
    if (whiten) {
        for (int ky = -g->B; ky <= g->B; ky++) for (int kx = 0; kx <= g->B; kx++) {
            cpx *o = &out[(size_t)(ky + g->B) * g->W + kx];
            double k2 = (double)kx * kx + (double)ky * ky;
            if (k2 >= r2 || k2 == 0) continue;
            int b = (int)floor(sqrt(k2));
            double p = pc[b] > 0 ? pw[b] / pc[b] : 0;
            double s = p > 0 ? 1.0 / sqrt(p) : 0.0;
            o->re = (float)(o->re * s); o->im = (float)(o->im * s);
        }
    }
    free(pw); free(pc); free(f);
}

static void preprocess_disc(const float *img, const geom_t *g, double mask_radius_A, double falloff_A,
                       int normalize, int invert, int do_mask, int whiten, double rband, cpx *out, double *wring, const double *disc) {
    preprocess_row(img, g, mask_radius_A, falloff_A, normalize, invert, do_mask, whiten, rband, out, wring, disc, NULL);
}
static void preprocess(const float *img, const geom_t *g, double mask_radius_A, double falloff_A,
                       int normalize, int invert, int do_mask, int whiten, double rband, cpx *out, double *wring) {
    preprocess_row(img, g, mask_radius_A, falloff_A, normalize, invert, do_mask, whiten, rband, out, wring, NULL, NULL);
}

/* ------------------------------------------------------------------ local score */
/* K5: answers 19/20 low / high resolution limit and 21 "resolution limit for signed CC" (frealign.py:3939-3943);
 * SCORE is reported x100 and clipped to 0..100 by the caller (align/core.py:1467). */
/* ring-wise weighted correlation of image I against CTF * slice * shift; signed below
 * ring_signed, absolute above.  shifts in pixels. */
static double score_local(const oref_t *r, const geom_t *g, const ctf_t *c, const cpx *I,
                          const double *wr, double rmax, const double M[9], const double sh[2]) {
    double m[6] = { M[0], M[1], M[3], M[4], M[6], M[7] };
    int nr = g->B + 2;
    double *A = (double *)calloc(nr, sizeof(double));
    double sb = 0, sc = 0, rl2 = g->r_lo * g->r_lo, rh2 = rmax * rmax;
    for (int ky = -g->B; ky <= g->B; ky++) for (int kx = 0; kx <= g->B; kx++) {
        double k2 = (double)kx * kx + (double)ky * ky;
        if (k2 >= rh2 || k2 < rl2 || k2 == 0) continue;
        double pr, pi;
        sample_cube(r, m[0] * kx + m[1] * ky, m[2] * kx + m[3] * ky, m[4] * kx + m[5] * ky, &pr, &pi);
        int b = (int)floor(sqrt(k2));
        double cv = ctf_eval(c, kx, ky) * wr[b];     /* the model gets the image's whitening filter too */
        double ph = -2.0 * ORC_PI * (kx * sh[0] + ky * sh[1]) / g->N;
        double cr = cos(ph), ci = sin(ph);
        double mr = cv * (pr * cr - pi * ci), mi = cv * (pr * ci + pi * cr);
        const cpx *iv = &I[(size_t)(ky + g->B) * g->W + kx];
        double al = kx == 0 ? 1.0 : 2.0;
        A[b] += al * (iv->re * mr + iv->im * mi);
        sb += al * (mr * mr + mi * mi);
        sc += al * ((double)iv->re * iv->re + (double)iv->im * iv->im);
    }
    double sa = 0;
    for (int b = 0; b < nr; b++) sa += ((double)b <= g->ring_signed) ? A[b] : fabs(A[b]);
    free(A);
    return (sb > 0 && sc > 0) ? sa / sqrt(sb * sc) : 0.0;
}

/* ------------------------------------------------------------------ compass refinement */
/* State: rotation matrix M = Rz(phi) Ry(theta) Rz(psi) (row-major 3x3), shifts in pixels, score f,
 * current angular / shift steps.  The three rotational parameters are steps in the IMAGE frame
 * (right-multiplication): 0 = in-plane (about image z; this IS psi), 1 / 2 = tilts about the image
 * x / y axes.  They are decoupled at every theta, unlike (theta, phi).  When only one of theta / phi
 * is free, that Euler angle itself is stepped (left-multiplication forms below). */
typedef struct { double M[9], sh[2], f, ha, hs; } cstate_t;

static void euler_full(double psi, double theta, double phi, double M[9]) {
    double m[6]; euler_cols(psi, theta, phi, m);
    double th = theta * ORC_PI / 180, ph = phi * ORC_PI / 180;
    M[0] = m[0]; M[1] = m[1]; M[2] = cos(ph) * sin(th);
    M[3] = m[2]; M[4] = m[3]; M[5] = sin(ph) * sin(th);
    M[6] = m[4]; M[7] = m[5]; M[8] = cos(th);
}

/* Focus mask (answers 29-32 / 44): disc = centre (pixels from the box centre) and radius (pixels) around the projection of the
 * focus sphere at the row's input pose: centre = (M^T c)_xy + shift, M = Rz(phi) Ry(theta) Rz(psi).  Returns 0 when off. */
static int focus_disc(const ppm_refine_cfg *cfg, const geom_t *g, const double *row, double disc[3]) {
    if (!(cfg->focus[3] > 0.f)) return 0;
    double M[9]; euler_full(row[PPM_PSI], row[PPM_THETA], row[PPM_PHI], M);
    const double c0 = cfg->focus[0] / g->a, c1 = cfg->focus[1] / g->a, c2 = cfg->focus[2] / g->a;
    disc[0] = M[0] * c0 + M[3] * c1 + M[6] * c2 + row[PPM_XSHIFT] / g->a;
    disc[1] = M[1] * c0 + M[4] * c1 + M[7] * c2 + row[PPM_YSHIFT] / g->a;
    disc[2] = cfg->focus[3] / g->a;
    return 1;
}

static void angles_from_matrix(const double M[9], double *psi, double *theta, double *phi) {
    double ct = M[8] > 1 ? 1 : (M[8] < -1 ? -1 : M[8]);
    double st = sqrt(M[2] * M[2] + M[5] * M[5]);
    if (st > 1e-7) {
        *theta = atan2(st, ct) * 180 / ORC_PI;
        *phi = atan2(M[5], M[2]) * 180 / ORC_PI;
        *psi = atan2(M[7], -M[6]) * 180 / ORC_PI;
    } else {               /* theta = 0 or 180: only psi +- phi is defined; put it all in psi */
        *theta = ct > 0 ? 0.0 : 180.0; *phi = 0.0;
        *psi = (ct > 0 ? atan2(M[3], M[0]) : atan2(-M[3], -M[0])) * 180 / ORC_PI;
    }
    if (*psi < 0) *psi += 360; if (*phi < 0) *phi += 360;
}

static void rot_step(const double M[9], int which, int tilt_frame, double hdeg, double out[9]) {
    double h = hdeg * ORC_PI / 180, c = cos(h), s = sin(h), R[9], L[9], T[9];
    if (which == 0) { double r[9] = { c, -s, 0, s, c, 0, 0, 0, 1 }; mat_mul3(M, r, out); return; }
    if (tilt_frame) {
        if (which == 1) { double r[9] = { 1, 0, 0, 0, c, -s, 0, s, c }; mat_mul3(M, r, out); }
        else { double r[9] = { c, 0, s, 0, 1, 0, -s, 0, c }; mat_mul3(M, r, out); }
        return;
    }
    if (which == 2) { double r[9] = { c, -s, 0, s, c, 0, 0, 0, 1 }; mat_mul3(r, M, out); return; }   /* phi += h */
    /* theta += h: L = Rz(phi) Ry(h) Rz(-phi) */
    double psi, th, ph; angles_from_matrix(M, &psi, &th, &ph);
    double cp = cos(ph * ORC_PI / 180), sp = sin(ph * ORC_PI / 180);
    double rz[9] = { cp, -sp, 0, sp, cp, 0, 0, 0, 1 }, rzt[9] = { cp, sp, 0, -sp, cp, 0, 0, 0, 1 }, ry[9] = { c, 0, s, 0, 1, 0, -s, 0, c };
    mat_mul3(rz, ry, T); mat_mul3(T, rzt, L); mat_mul3(L, M, R); memcpy(out, R, sizeof(R));
}

/* band of one compass iteration (frequency marching): rings whose phase moves by more than about
 * band_factor radians under the iteration's largest probe displacement carry no usable gradient */
static double iter_band(const geom_t *g, double rm_px, double bf, const int en[5], double ha, double hs, double rcap) {
    if (bf < 0) return rcap;
    double d = 0;
    if (en[0] || en[1] || en[2]) d = rm_px * ha * ORC_PI / 180.0;
    if ((en[3] || en[4]) && hs > d) d = hs;
    if (!(d > 0)) return rcap;
    double rit = bf * g->N / (2.0 * ORC_PI * d);
    if (rit < 4.0) rit = 4.0;
    return rit < rcap ? rit : rcap;
}

/* answer 7 "use priors" (include/ppm.h, ppm_refine_cfg.use_priors): Gaussian restraint on the refined parameters */
typedef struct { int on; double mean[5], w[5]; } prior_t;       /* w = 1 / (2 var n_s), shifts in pixels; 0 = unrestrained */
static void prior_init(prior_t *p, const ppm_refine_cfg *cfg, const geom_t *g, const int en[5]) {
    memset(p, 0, sizeof(*p));
    if (!cfg->use_priors) return;
    const double ns = ORC_PI * (g->r_hi * g->r_hi - g->r_lo * g->r_lo);
    for (int i = 0; i < 5; i++) {
        double var = cfg->prior_var[i], mean = cfg->prior_mean[i];
        if (i >= 3) { mean /= g->a; var /= g->a * g->a; }
        p->mean[i] = mean;
        if (en[i] && var > 0 && ns > 0) { p->w[i] = 1.0 / (2.0 * var * ns); p->on = 1; }
    }
}
static double prior_pen(const prior_t *p, const double M[9], const double sh[2]) {
    if (!p || !p->on) return 0.0;
    double v[5]; angles_from_matrix(M, &v[0], &v[1], &v[2]); v[3] = sh[0]; v[4] = sh[1];
    double pen = 0;
    for (int i = 0; i < 5; i++) {
        if (!(p->w[i] > 0)) continue;
        double d = v[i] - p->mean[i];
        if (i < 3) { d = fmod(d, 360.0); if (d > 180.0) d -= 360.0; if (d < -180.0) d += 360.0; }
        pen += p->w[i] * d * d;
    }
    return pen;
}

static void compass_iter(const oref_t *r, const geom_t *g, const ctf_t *c, const cpx *I, const double *wr,
                         double rcap, double rm_px, double bf, const int en[5], cstate_t *s, long *nevals, double *sevals, const prior_t *pr) {
    /* en[]: psi, theta, phi, x, y.  rotational slots: 0 <- psi; 1,2 <- tilts if both theta and phi are
     * free, else slot 1 <- theta, slot 2 <- phi as Euler steps.  Every score of one iteration (centre,
     * 2 per free parameter, trial) is taken at the iteration's band. */
    int tilt = en[1] && en[2];
    int on[5] = { en[0], en[1], en[2], en[3], en[4] };
    double fp[5], fm[5], d[5], Mq[9], shq[2];
    int any = 0, nfree = 0;
    for (int i = 0; i < 5; i++) nfree += on[i] ? 1 : 0;
    if (nfree) {
        double rmax = iter_band(g, rm_px, bf, en, s->ha, s->hs, rcap);
        double sper = floor(ORC_PI * rmax * rmax / 2);
        double f0 = score_local(r, g, c, I, wr, rmax, s->M, s->sh) - prior_pen(pr, s->M, s->sh); *nevals += 1; *sevals += sper;
        for (int i = 0; i < 5; i++) {
            d[i] = 0; fp[i] = fm[i] = -1e300;
            if (!on[i]) continue;
            double h = i < 3 ? s->ha : s->hs;
            for (int sg = 0; sg < 2; sg++) {
                double hh = sg ? -h : h;
                memcpy(Mq, s->M, sizeof(Mq)); shq[0] = s->sh[0]; shq[1] = s->sh[1];
                if (i < 3) rot_step(s->M, i, tilt, hh, Mq); else shq[i - 3] += hh;
                double v = score_local(r, g, c, I, wr, rmax, Mq, shq) - prior_pen(pr, Mq, shq);
                if (sg) fm[i] = v; else fp[i] = v;
            }
            *nevals += 2; *sevals += 2 * sper; any = 1;
            double den = 2.0 * f0 - fp[i] - fm[i];
            if (den > 1e-12) {
                double t = 0.5 * h * (fp[i] - fm[i]) / den;
                d[i] = t > h ? h : (t < -h ? -h : t);
            } else {
                double best = fp[i] > fm[i] ? fp[i] : fm[i];
                d[i] = best > f0 ? (fp[i] > fm[i] ? h : -h) : 0.0;
            }
        }
        if (any) {
            double Mt[9], T[9];
            memcpy(Mt, s->M, sizeof(Mt));
            for (int i = 0; i < 3; i++) if (on[i] && d[i] != 0) { rot_step(Mt, i, tilt, d[i], T); memcpy(Mt, T, sizeof(T)); }
            shq[0] = s->sh[0] + d[3]; shq[1] = s->sh[1] + d[4];
            double ft = score_local(r, g, c, I, wr, rmax, Mt, shq) - prior_pen(pr, Mt, shq); *nevals += 1; *sevals += sper;
            int bi = -1, bs = 0; double fb = f0;
            for (int i = 0; i < 5; i++) {
                if (!on[i]) continue;
                if (fp[i] > fb) { fb = fp[i]; bi = i; bs = 1; }
                if (fm[i] > fb) { fb = fm[i]; bi = i; bs = -1; }
            }
            s->f = f0;
            if (ft > f0 && ft >= fb) { memcpy(s->M, Mt, sizeof(Mt)); s->sh[0] = shq[0]; s->sh[1] = shq[1]; s->f = ft; }
            else if (bi >= 0) {
                if (bi < 3) { rot_step(s->M, bi, tilt, bs * s->ha, T); memcpy(s->M, T, sizeof(T)); }
                else s->sh[bi - 3] += bs * s->hs;
                s->f = fb;
            }
        }
    }
    s->ha *= 0.5; s->hs *= 0.5;
}

/* ------------------------------------------------------------------ global search */
/* K5/K6: answers 24 search resolution, 25 angular step, 26 top hits, 27/28 search range X/Y, 36/37 global / local
 * (frealign.py:3949-3957, :3866-3871). */
typedef struct { double cc; int orient, sx, sy; } hit_t;

/* correlation window of one orientation.  W = alpha * ctf * I (band r_lo..r_s), C2 = alpha * ctf^2,
 * P = slice.  Returns max over the shift window of c(s)/sqrt(nP*nI).  mode 0: zero-filled Ns x Ns
 * inverse FFT; mode 1: pruned direct transform. `conjP`: use conj(P) (orientation psi+180). */
static double ccf_peak(const geom_t *g, const cpx *Wp, const float *C2, const cpx *P, int conjP,
                       double nI, int mode, cpx *work, int *bsx, int *bsy) {
    int B = g->B, Wd = g->W, Ns = g->Ns;
    double nP = 0;
    for (int i = 0; i < g->H * Wd; i++) nP += C2[i] * ((double)P[i].re * P[i].re + (double)P[i].im * P[i].im);
    double best = -1e300; *bsx = *bsy = 0;
    if (!(nP > 0 && nI > 0)) return 0.0;
    if (mode == 0) {
        memset(work, 0, (size_t)Ns * Ns * sizeof(cpx));
        for (int ky = -B; ky <= B; ky++) for (int kx = 0; kx <= B; kx++) {
            if (kx >= Ns / 2 || ky >= Ns / 2 || ky < -Ns / 2) continue;
            size_t i = (size_t)(ky + B) * Wd + kx;
            double pr = P[i].re, pi = conjP ? -P[i].im : P[i].im;
            /* Q = W conj(P); W carries alpha (1 on kx = 0, 2 elsewhere): full plane = Q/alpha at k and conj at -k */
            double qr = Wp[i].re * pr + Wp[i].im * pi, qi = Wp[i].im * pr - Wp[i].re * pi;
            double al = kx == 0 ? 1.0 : 2.0;
            qr /= al; qi /= al;
            cpx *a = &work[(size_t)((ky + Ns) % Ns) * Ns + kx];
            a->re += (float)qr; a->im += (float)qi;
            if (kx > 0) { cpx *b = &work[(size_t)((-ky + Ns) % Ns) * Ns + (Ns - kx)]; b->re += (float)qr; b->im -= (float)qi; }
        }
        for (int y = 0; y < Ns; y++) fft1d(work + (size_t)y * Ns, Ns, 1, 1);
        for (int x = 0; x < Ns; x++) fft1d(work + x, Ns, Ns, 1);
        for (int sy = -g->RSy; sy <= g->RSy; sy++) for (int sx = -g->RSx; sx <= g->RSx; sx++) {
            double v = work[(size_t)((sy + Ns) % Ns) * Ns + ((sx + Ns) % Ns)].re;
            if (v > best) { best = v; *bsx = sx; *bsy = sy; }
        }
    } else {
        /* separable pruned transform: G[kx][sy] = sum_ky Q e^{+i 2pi ky sy/Ns}, then the sum over kx */
        const double *tc = PLANS[Ns]->c, *ts = PLANS[Ns]->s;
        int nsy = 2 * g->RSy + 1;
        double *G = (double *)calloc((size_t)2 * Wd * nsy, sizeof(double));
        for (int ky = -B; ky <= B; ky++) for (int kx = 0; kx <= B; kx++) {
            size_t i = (size_t)(ky + B) * Wd + kx;
            if (C2[i] == 0) continue;
            double pr = P[i].re, pi = conjP ? -P[i].im : P[i].im;
            double qr = Wp[i].re * pr + Wp[i].im * pi, qi = Wp[i].im * pr - Wp[i].re * pi;
            for (int sy = -g->RSy; sy <= g->RSy; sy++) {
                int t = (((ky * sy) % Ns) + Ns) % Ns;
                double c = tc[t], s2 = ts[t];
                double *o = &G[((size_t)kx * nsy + (sy + g->RSy)) * 2];
                o[0] += qr * c - qi * s2; o[1] += qr * s2 + qi * c;
            }
        }
        for (int sy = -g->RSy; sy <= g->RSy; sy++) for (int sx = -g->RSx; sx <= g->RSx; sx++) {
            double acc = 0;
            for (int kx = 0; kx <= B; kx++) {
                int t = (((kx * sx) % Ns) + Ns) % Ns;
                double c = tc[t], s2 = ts[t];
                const double *o = &G[((size_t)kx * nsy + (sy + g->RSy)) * 2];
                acc += o[0] * c - o[1] * s2;
            }
            if (acc > best) { best = acc; *bsx = sx; *bsy = sy; }
        }
        free(G);
    }
    return best / sqrt(nP * nI);
}

/* ------------------------------------------------------------------ refine_batch */
/* bank: optional caller-supplied slice bank (n_dir * n_psi/2 (or n_psi) slices at band r_s); built if NULL */
int orc_refine_batch(void *refp, const ppm_refine_cfg *cfg, const float *images, int n_img,
                     const double *rows_in, double *rows_out, int ccf_mode, long *eval_counts) {
    fft_tables();
    oref_t *r = (oref_t *)refp;
    geom_t g;
    if (!r || geom_init(&g, cfg)) return -22;
    if (g.B > (r->B + 1) / r->pad - 1 || r->N != g.N) return -22;
    int K = cfg->top_hits > 0 ? cfg->top_hits : 20; if (K > PPM_MAX_TOP_HITS) K = PPM_MAX_TOP_HITS;
    /* answers 36 / 37 (frealign.py:3866-3871): PYP's default is global = yes, local = no together with "top hits to refine" = 20
     * (:3953), so a global search always refines its top hits; answer 37 only decides whether the best one continues at the
     * full band.  iters_hit < 0 leaves the hits at their grid points (test hook). */
    int Tb = cfg->iters_hit > 0 ? cfg->iters_hit : (cfg->iters_hit < 0 ? 0 : 2), Tc = cfg->iters_final > 0 ? cfg->iters_final : 7;
    double fall = cfg->mask_falloff > 0 ? cfg->mask_falloff : 20.0;
    double dstep = cfg->angular_step > 0 ? cfg->angular_step : 15.0;
    int en[5] = { cfg->refine_psi, cfg->refine_theta, cfg->refine_phi, cfg->refine_x, cfg->refine_y };
    const double bf = cfg->band_factor == 0 ? 3.0 : cfg->band_factor, rm_px = cfg->mask_radius / g.a;
    prior_t pr; prior_init(&pr, cfg, &g, en);
    size_t nb = (size_t)g.H * g.W;
    int half = (g.n_psi % 2 == 0);                 /* psi and psi+180 share a slice (conjugate) */
    int npsi_store = half ? g.n_psi / 2 : g.n_psi;
    cpx *bank = NULL;
    if (cfg->global_search) {
        bank = (cpx *)malloc((size_t)g.n_dir * npsi_store * nb * sizeof(cpx));
        if (!bank) return -12;
        for (int d = 0; d < g.n_dir; d++) {
            double th, ph; grid_direction(&g, dstep, d, &th, &ph);
            for (int k = 0; k < npsi_store; k++) {
                double m[6]; euler_cols(k * g.dpsi, th, ph, m);
                extract_slice(r, &g, m, g.r_s, bank + ((size_t)d * npsi_store + k) * nb);
            }
        }
    }
    long tot_g = 0, tot_l = 0;
    double tot_s = 0;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tot_g, tot_l, tot_s)
    for (int ip = 0; ip < n_img; ip++) {
        const double *row = rows_in + (size_t)ip * PPM_NCOL;
        double *out = rows_out + (size_t)ip * PPM_NCOL;
        memcpy(out, row, PPM_NCOL * sizeof(double));
        const float *img = images + (size_t)ip * g.N * g.N;
        ctf_t c; ctf_init(&c, row, g.N, g.a);
        cpx *I = (cpx *)malloc(nb * sizeof(cpx));
        double *wr = (double *)malloc((g.B + 2) * sizeof(double)), *wrs = wr, *wrsown = NULL;
        double disc[3]; const int focus_on = focus_disc(cfg, &g, row, disc);
        preprocess_row(img, &g, cfg->mask_radius, fall, cfg->normalize, cfg->invert, 1, 1, g.r_hi, I, wr, focus_on ? disc : NULL, row);
        long nev = 0; double sev = 0;
        cstate_t best; memset(&best, 0, sizeof(best));
        if (cfg->global_search) {
            cpx *Is = I, *Isown = NULL;
            if (!focus_on && cfg->search_mask_radius > 0 && cfg->search_mask_radius != cfg->mask_radius) {
                Isown = (cpx *)malloc(nb * sizeof(cpx)); wrsown = (double *)malloc((g.B + 2) * sizeof(double));
                preprocess_row(img, &g, cfg->search_mask_radius, fall, cfg->normalize, cfg->invert, 1, 1, g.r_hi, Isown, wrsown, NULL, row);
                Is = Isown; wrs = wrsown;
            }
            cpx *Wp = (cpx *)malloc(nb * sizeof(cpx)); float *C2 = (float *)malloc(nb * sizeof(float));
            cpx *work = (cpx *)malloc((size_t)g.Ns * g.Ns * sizeof(cpx));
            double nI = 0, rl2 = g.r_lo * g.r_lo, rs2 = g.r_s * g.r_s;
            for (int ky = -g.B; ky <= g.B; ky++) for (int kx = 0; kx <= g.B; kx++) {
                size_t i = (size_t)(ky + g.B) * g.W + kx;
                double k2 = (double)kx * kx + (double)ky * ky;
                if (k2 >= rs2 || k2 < rl2 || k2 == 0) { Wp[i].re = Wp[i].im = 0; C2[i] = 0; continue; }
                double cv = ctf_eval(&c, kx, ky) * wrs[(int)floor(sqrt(k2))], al = kx == 0 ? 1.0 : 2.0;
                Wp[i].re = (float)(al * cv * Is[i].re); Wp[i].im = (float)(al * cv * Is[i].im);
                C2[i] = (float)(al * cv * cv);
                nI += al * ((double)Is[i].re * Is[i].re + (double)Is[i].im * Is[i].im);
            }
            hit_t *hits = (hit_t *)malloc((size_t)g.n_orient * sizeof(hit_t));
            for (int d = 0; d < g.n_dir; d++) for (int k = 0; k < g.n_psi; k++) {
                int ks = half ? k % npsi_store : k, cj = half ? (k >= npsi_store) : 0;
                hit_t *h = &hits[d * g.n_psi + k];
                h->orient = d * g.n_psi + k;
                h->cc = ccf_peak(&g, Wp, C2, bank + ((size_t)d * npsi_store + ks) * nb, cj, nI, ccf_mode, work, &h->sx, &h->sy);
            }
            tot_g += g.n_orient;
            /* top-K by cc, ties -> lower orientation index */
            int Kk = K < g.n_orient ? K : g.n_orient;
            for (int a = 0; a < Kk; a++) {
                int bi = a;
                for (int b2 = a + 1; b2 < g.n_orient; b2++)
                    if (hits[b2].cc > hits[bi].cc || (hits[b2].cc == hits[bi].cc && hits[b2].orient < hits[bi].orient)) bi = b2;
                hit_t t = hits[a]; hits[a] = hits[bi]; hits[bi] = t;
            }
            int have = 0;
            for (int a = 0; a < Kk; a++) {
                cstate_t s;
                double th, ph; grid_direction(&g, dstep, hits[a].orient / g.n_psi, &th, &ph);
                euler_full((hits[a].orient % g.n_psi) * g.dpsi, th, ph, s.M);
                s.sh[0] = hits[a].sx * g.step; s.sh[1] = hits[a].sy * g.step;
                s.ha = 0.5 * dstep; s.hs = g.step;
                if (Tb > 0) {
                    for (int t = 0; t < Tb; t++) compass_iter(r, &g, &c, I, wr, g.r_s, rm_px, bf, en, &s, &nev, &sev, &pr);
                } else s.f = hits[a].cc;
                if (!have || s.f > best.f) { best = s; have = 1; }
            }
            if (cfg->local_refine)         /* the best hit continues at the full band */
                for (int t = 0; t < Tc; t++) compass_iter(r, &g, &c, I, wr, g.r_hi, rm_px, bf, en, &best, &nev, &sev, &pr);
            best.f = score_local(r, &g, &c, I, wr, g.r_hi, best.M, best.sh); nev++; sev += floor(ORC_PI * g.r_hi * g.r_hi / 2);
            free(hits); free(work); free(Wp); free(C2); free(Isown); free(wrsown);
        } else {
            euler_full(row[PPM_PSI], row[PPM_THETA], row[PPM_PHI], best.M);
            best.sh[0] = row[PPM_XSHIFT] / g.a; best.sh[1] = row[PPM_YSHIFT] / g.a;
            best.ha = cfg->local_angle_step > 0 ? cfg->local_angle_step : 2.5;
            best.hs = cfg->local_shift_step > 0 ? cfg->local_shift_step : 2.0;
            if (cfg->local_refine) for (int t = 0; t < Tb + Tc; t++) compass_iter(r, &g, &c, I, wr, g.r_hi, rm_px, bf, en, &best, &nev, &sev, &pr);
            best.f = score_local(r, &g, &c, I, wr, g.r_hi, best.M, best.sh); nev++; sev += floor(ORC_PI * g.r_hi * g.r_hi / 2);
        }
        ctf_t cfin = c;             /* the CTF the output row carries (moved by the defocus refinement) */
        /* defocus refinement (answers 33, 34, 45; frealign.py:3960-3961, :3978): offsets scored at the final pose */
        if (cfg->refine_defocus && cfg->defocus_step > 0 && cfg->defocus_range >= cfg->defocus_step) {
            int nt = (int)floor(cfg->defocus_range / cfg->defocus_step + 1e-6); if (nt > PPM_MAX_DEFOCUS_STEPS) nt = PPM_MAX_DEFOCUS_STEPS;
            double bestf = best.f; int bt = 0;
            for (int t = -nt; t <= nt; t++) {
                if (t == 0) continue;
                ctf_t c2 = c; c2.df1 += t * (double)cfg->defocus_step; c2.df2 += t * (double)cfg->defocus_step;
                double f = score_local(r, &g, &c2, I, wr, g.r_hi, best.M, best.sh);
                nev++; sev += floor(ORC_PI * g.r_hi * g.r_hi / 2);
                if (f > bestf) { bestf = f; bt = t; }
            }
            best.f = bestf;
            cfin.df1 += bt * (double)cfg->defocus_step; cfin.df2 += bt * (double)cfg->defocus_step;
            out[PPM_DF1] = row[PPM_DF1] + bt * (double)cfg->defocus_step; out[PPM_DF2] = row[PPM_DF2] + bt * (double)cfg->defocus_step;
        }
        tot_l += nev; tot_s += sev;
        angles_from_matrix(best.M, &out[PPM_PSI], &out[PPM_THETA], &out[PPM_PHI]);
        out[PPM_XSHIFT] = best.sh[0] * g.a; out[PPM_YSHIFT] = best.sh[1] * g.a;
        double cc = best.f;
        out[PPM_SCORE] = 100.0 * cc;
        /* answer 22 "classification resolution limit" (frealign.py:3945): LOGP / SIGMA come from the final pose scored over
         * r_lo .. r_cls (what the occupancy update of 3-D classification compares between class references) */
        if (g.r_cls < g.r_hi) { cc = score_local(r, &g, &cfin, I, wr, g.r_cls, best.M, best.sh); tot_l += 1; tot_s += floor(ORC_PI * g.r_cls * g.r_cls / 2); }
        double res = 1.0 - cc * cc; if (res < 1e-6) res = 1e-6;
        /* whitened image against the best model: residual variance 1 - cc^2 per sample, n = in-band samples of the
         * full plane; LogP = Gaussian log-likelihood at the maximum-likelihood sigma */
        double nsamp = ORC_PI * (g.r_cls * g.r_cls - g.r_lo * g.r_lo);
        out[PPM_SIGMA] = sqrt(res);
        out[PPM_LOGP] = -0.5 * nsamp * (log(2.0 * ORC_PI * res) + 1.0);
        free(I); free(wr);
    }
    free(bank);
    if (eval_counts) { eval_counts[0] = n_img ? tot_g / n_img : 0; eval_counts[1] = n_img ? tot_l / n_img : 0; eval_counts[2] = n_img ? (long)(tot_s / n_img) : 0; }
    return err;
}

/* score of given poses (no search): used by parity tests of the local-score kernel */
int orc_score_batch(void *refp, const ppm_refine_cfg *cfg, const float *images, int n_img,
                    const double *rows, double *scores) {
    fft_tables();
    oref_t *r = (oref_t *)refp; geom_t g;
    if (!r || geom_init(&g, cfg) || g.B > (r->B + 1) / r->pad - 1) return -22;
    double fall = cfg->mask_falloff > 0 ? cfg->mask_falloff : 20.0;
    size_t nb = (size_t)g.H * g.W;
#pragma omp parallel for schedule(dynamic, 1)
    for (int ip = 0; ip < n_img; ip++) {
        const double *row = rows + (size_t)ip * PPM_NCOL;
        ctf_t c; ctf_init(&c, row, g.N, g.a);
        cpx *I = (cpx *)malloc(nb * sizeof(cpx));
        double *wr = (double *)malloc((g.B + 2) * sizeof(double));
        double disc[3]; const int focus_on = focus_disc(cfg, &g, row, disc);
        preprocess_row(images + (size_t)ip * g.N * g.N, &g, cfg->mask_radius, fall, cfg->normalize, cfg->invert, 1, 1, g.r_hi, I, wr, focus_on ? disc : NULL, row);
        double M[9], sh[2] = { row[PPM_XSHIFT] / g.a, row[PPM_YSHIFT] / g.a };
        euler_full(row[PPM_PSI], row[PPM_THETA], row[PPM_PHI], M);
        scores[ip] = score_local(r, &g, &c, I, wr, g.r_hi, M, sh);
        free(I); free(wr);
    }
    return 0;
}

/* refine3d answers 8 / 43 "matching projections" (frealign.py:3929-3931): the reference projected at a row's pose, times the
 * row's CTF, at the row's shift, band-limited at res_high; out: n * N * N floats.  cube = FFT / N, so the unnormalised inverse
 * transform is divided by N once more. */
int orc_match_projections(void *refp, const ppm_refine_cfg *cfg, const double *rows, int n, float *out) {
    fft_tables();
    oref_t *r = (oref_t *)refp; geom_t g;
    ppm_refine_cfg c2 = *cfg; c2.global_search = 0;
    if (!r || geom_init(&g, &c2) || g.B > (r->B + 1) / r->pad - 1) return -22;
    const int N = g.N;
    const double rh2 = g.r_hi * g.r_hi;
#pragma omp parallel for schedule(dynamic, 1)
    for (int ip = 0; ip < n; ip++) {
        const double *row = rows + (size_t)ip * PPM_NCOL;
        ctf_t c; ctf_init(&c, row, N, g.a);
        double M[9]; euler_full(row[PPM_PSI], row[PPM_THETA], row[PPM_PHI], M);
        const double m[6] = { M[0], M[1], M[3], M[4], M[6], M[7] }, sx = row[PPM_XSHIFT] / g.a, sy = row[PPM_YSHIFT] / g.a;
        cpx *f = (cpx *)calloc((size_t)N * N, sizeof(cpx));
        for (int ky = -g.B; ky <= g.B; ky++) for (int kx = 0; kx <= g.B; kx++) {
            const double k2 = (double)kx * kx + (double)ky * ky;
            if (k2 >= rh2) continue;
            double pr, pi;
            sample_cube(r, m[0] * kx + m[1] * ky, m[2] * kx + m[3] * ky, m[4] * kx + m[5] * ky, &pr, &pi);
            double cv = ctf_eval(&c, kx, ky);
            if ((kx + ky) & 1) cv = -cv;                              /* projection centred on pixel (N/2, N/2) */
            const double ph = -2.0 * ORC_PI * (kx * sx + ky * sy) / N, cr = cos(ph), ci = sin(ph);
            const double vr = cv * (pr * cr - pi * ci), vi = cv * (pr * ci + pi * cr);
            cpx *o = &f[(size_t)((ky + N) % N) * N + kx];
            o->re = (float)vr; o->im = (float)vi;
            if (kx > 0) { cpx *q = &f[(size_t)((N - ky) % N) * N + (N - kx)]; q->re = (float)vr; q->im = (float)-vi; }
        }
        for (int y = 0; y < N; y++) fft1d(f + (size_t)y * N, N, 1, 1);
        for (int x = 0; x < N; x++) fft1d(f + x, N, N, 1);
        const double sc = (cfg->invert ? -1.0 : 1.0) / N;
        float *o = out + (size_t)ip * N * N;
        for (size_t i = 0; i < (size_t)N * N; i++) o[i] = (float)(f[i].re * sc);
        free(f);
    }
    return 0;
}

/* preprocessed (whitened, masked) band spectrum of one image: [2B+1][B+1] complex, for kernel tests */
int orc_preprocess(const ppm_refine_cfg *cfg, const float *img, float mask_radius, float *out_band, double *out_wring) {
    fft_tables();
    geom_t g; if (geom_init(&g, cfg)) return -22;
    double fall = cfg->mask_falloff > 0 ? cfg->mask_falloff : 20.0;
    preprocess(img, &g, mask_radius, fall, cfg->normalize, cfg->invert, 1, 1, g.r_hi, (cpx *)out_band, out_wring);
    return 0;
}

int orc_band_dims(const ppm_refine_cfg *cfg, int *B, int *n_orient, int *Ns, double *step, int *RSx, int *RSy) {
    geom_t g; if (geom_init(&g, cfg)) return -22;
    *B = g.B; *n_orient = g.n_orient; *Ns = g.Ns; *step = g.step; *RSx = g.RSx; *RSy = g.RSy;
    return 0;
}

/* one reference slice at (psi,theta,phi), band layout, for kernel tests */
int orc_extract_slice(void *refp, const ppm_refine_cfg *cfg, double psi, double theta, double phi, float *out_band) {
    oref_t *r = (oref_t *)refp; geom_t g;
    if (!r || geom_init(&g, cfg) || g.B > (r->B + 1) / r->pad - 1) return -22;
    double m[6]; euler_cols(psi, theta, phi, m);
    extract_slice(r, &g, m, g.r_hi, (cpx *)out_band);
    return 0;
}

/* ------------------------------------------------------------------ symmetry operators */
static void mat_mul(const double *a, const double *b, double *c) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
        c[i * 3 + j] = s;
    }
}
static void rot_axis(const double ax[3], double deg, double *m) {
    double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    double x = ax[0] / n, y = ax[1] / n, z = ax[2] / n, t = deg * ORC_PI / 180, c = cos(t), s = sin(t), C = 1 - c;
    double r[9] = { c + x * x * C, x * y * C - z * s, x * z * C + y * s, y * x * C + z * s, c + y * y * C, y * z * C - x * s,
                    z * x * C - y * s, z * y * C + x * s, c + z * z * C };
    memcpy(m, r, sizeof(r));
}
/* closure of a generator set; returns count (<= 60) */
static int sym_group(const double *gens, int ngen, double *ops) {
    int n = 1; double I3[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }; memcpy(ops, I3, sizeof(I3));
    for (int grew = 1; grew;) {
        grew = 0;
        for (int i = 0; i < n && n < 60; i++) for (int j = 0; j < ngen && n < 60; j++) {
            double c[9]; mat_mul(ops + i * 9, gens + j * 9, c);
            int found = 0;
            for (int k = 0; k < n && !found; k++) {
                double d = 0; for (int q = 0; q < 9; q++) d += fabs(ops[k * 9 + q] - c[q]);
                if (d < 1e-6) found = 1;
            }
            if (!found) { memcpy(ops + n * 9, c, sizeof(c)); n++; grew = 1; }
        }
    }
    return n;
}
int orc_symmetry_ops(const char *sym, double *ops /* 60*9 */) {
    double gens[3 * 9]; int ng = 0;
    double z[3] = { 0, 0, 1 }, x[3] = { 1, 0, 0 }, d111[3] = { 1, 1, 1 };
    char t = sym[0] >= 'a' ? sym[0] - 32 : sym[0];
    int n = atoi(sym + 1);
    if (t == 'C' && n >= 1) { rot_axis(z, 360.0 / n, gens); ng = 1; }
    else if (t == 'D' && n >= 1) { rot_axis(z, 360.0 / n, gens); rot_axis(x, 180, gens + 9); ng = 2; }
    else if (t == 'T') { rot_axis(z, 180, gens); rot_axis(d111, 120, gens + 9); ng = 2; }
    else if (t == 'O') { rot_axis(z, 90, gens); rot_axis(d111, 120, gens + 9); ng = 2; }
    else if (t == 'I') {
        double phi = (1 + sqrt(5.0)) / 2, a5[3] = { 0, 1, phi };   /* 2-fold axes on x,y,z; 5-fold on (0,1,phi) */
        rot_axis(z, 180, gens); rot_axis(d111, 120, gens + 9); rot_axis(a5, 72, gens + 18); ng = 3;
    } else return -22;
    return sym_group(gens, ng, ops);
}

/* ------------------------------------------------------------------ Fourier insertion */
/* K7: the reconstruct3d answers (frealign.py:1780-1824): resolution limit, weighting factor refine_bsc, score
 * threshold, normalise, invert, split even/odd, per-particle splitting by PIND (:1766, :1814-1815), symmetry (:1775-1778). */
/* acc: [2][N][N][N/2+1][3] floats {re, im, weight}; kz,ky stored at index k + N/2 */
int orc_insert_batch(float *acc, long *counts, const ppm_recon_cfg *cfg, const char *symmetry,
                     const float *images, int n_img, const double *rows) {
    fft_tables();
    int N = cfg->box; double a = cfg->pixel_size;
    if (!box_ok(N) || a <= 0) return -22;
    double ops[60 * 9]; int nsym = orc_symmetry_ops(symmetry && symmetry[0] ? symmetry : "C1", ops);
    if (nsym < 1) return -22;
    ppm_refine_cfg rc; memset(&rc, 0, sizeof(rc));
    rc.box = N; rc.pixel_size = (float)a; rc.res_high = cfg->res_limit > 0 ? cfg->res_limit : (float)(2 * a);
    rc.angular_step = 15;
    geom_t g; if (geom_init(&g, &rc)) return -22;
    size_t nb = (size_t)g.H * g.W, NX = N / 2 + 1, half_sz = (size_t)N * N * NX * 3;
    cpx *I = (cpx *)malloc(nb * sizeof(cpx));
    for (int ip = 0; ip < n_img; ip++) {
        const double *row = rows + (size_t)ip * PPM_NCOL;
        if (!(row[PPM_OCC] > 0) || row[PPM_SCORE] < cfg->score_threshold) continue;
        long key = cfg->split_by_pind ? (long)row[PPM_PIND] : (long)row[PPM_POS];
        int h = (int)(((key % 2) + 2) % 2);   /* odd keys -> half index 1, even -> 0 */
        float *A = acc + (size_t)h * half_sz;
        counts[h]++;
        preprocess_row(images + (size_t)ip * N * N, &g, cfg->mask_radius, 20.0, cfg->normalize, cfg->invert, 0, 0, g.r_hi, I, NULL, NULL, row);
        ctf_t c; ctf_init(&c, row, N, a);
        double m[6]; euler_cols(row[PPM_PSI], row[PPM_THETA], row[PPM_PHI], m);
        double sx = row[PPM_XSHIFT] / a, sy = row[PPM_YSHIFT] / a, na2 = (double)N * a * N * a;
        for (int ky = -g.B; ky <= g.B; ky++) for (int kx = 0; kx <= g.B; kx++) {
            double k2 = (double)kx * kx + (double)ky * ky;
            if (k2 >= g.r_hi * g.r_hi || k2 == 0) continue;
            double cv = ctf_eval(&c, kx, ky);
            double w = row[PPM_OCC] / 100.0;
            if (cfg->score_weight_bfactor != 0)
                w *= exp(-0.25 * cfg->score_weight_bfactor * (cfg->score_average - row[PPM_SCORE]) * k2 / na2);
            if (cfg->dose_weights && cfg->n_dose_weights > 0 && cfg->dose_exponent > 0) {
                /* data-driven dose weighting (frealign.py:1731-1753): exposure t = TIND attenuated by q_t^(F min(1, (s / (tr s_Nyq))^2)) */
                long t = (long)row[PPM_TIND];
                double dq = (t >= 0 && t < cfg->n_dose_weights) ? cfg->dose_weights[t] : 0.0;
                double tr = cfg->dose_transition > 0 && cfg->dose_transition <= 1 ? cfg->dose_transition : 1.0, cap2 = (tr * N / 2) * (tr * N / 2);
                if (dq > 0 && dq < 1) w *= exp(cfg->dose_exponent * log(dq) * (k2 < cap2 ? k2 : cap2) / cap2);
            }
            double ph = 2.0 * ORC_PI * (kx * sx + ky * sy) / N, cr = cos(ph), ci = sin(ph);
            const cpx *iv = &I[(size_t)(ky + g.B) * g.W + kx];
            double vr = w * cv * (iv->re * cr - iv->im * ci), vi = w * cv * (iv->re * ci + iv->im * cr), vw = w * cv * cv;
            double X0 = m[0] * kx + m[1] * ky, Y0 = m[2] * kx + m[3] * ky, Z0 = m[4] * kx + m[5] * ky;
            for (int s = 0; s < nsym; s++) {
                const double *S = ops + s * 9;
                double X = S[0] * X0 + S[1] * Y0 + S[2] * Z0, Y = S[3] * X0 + S[4] * Y0 + S[5] * Z0, Z = S[6] * X0 + S[7] * Y0 + S[8] * Z0;
                double ur = vr, ui = vi;
                if (X < 0) { X = -X; Y = -Y; Z = -Z; ui = -ui; }
                int x0 = (int)floor(X), y0 = (int)floor(Y), z0 = (int)floor(Z);
                double fx = X - x0, fy = Y - y0, fz = Z - z0;
                for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++) {
                    int xi = x0 + dx, yi = y0 + dy + N / 2, zi = z0 + dz + N / 2;
                    if (xi > N / 2 || yi < 0 || yi >= N || zi < 0 || zi >= N) continue;
                    double wt = (dx ? fx : 1 - fx) * (dy ? fy : 1 - fy) * (dz ? fz : 1 - fz);
                    float *v = A + (((size_t)zi * N + yi) * NX + xi) * 3;
                    v[0] += (float)(wt * ur); v[1] += (float)(wt * ui); v[2] += (float)(wt * vw);
                }
            }
        }
    }
    free(I);
    return 0;
}

/* ------------------------------------------------------------------ merge + finalise */
/* K8: merge3d (frealign.py:2075-2093): half maps, filtered map, the 7 statistics columns consumed at :2557-2567
 * and postprocess/core.py:203-221; particle volume from the 810 Da/nm^3 rule (docs/tutorials/tomo_empiar_10164.rst:289). */
static void ifft3_centered_real(cpx *f, int N, float *out) {
    /* f: full N^3 complex spectrum in FFT order (already carrying the centre phase); out = Re(IFFT)/N^2 */
    for (int z = 0; z < N; z++) for (int y = 0; y < N; y++) fft1d(f + ((size_t)z * N + y) * N, N, 1, 1);
    for (int z = 0; z < N; z++) for (int x = 0; x < N; x++) fft1d(f + (size_t)z * N * N + x, N, N, 1);
    for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) fft1d(f + (size_t)y * N + x, N, N * N, 1);
    double sc = 1.0 / ((double)N * N);
    for (size_t i = 0; i < (size_t)N * N * N; i++) out[i] = (float)(f[i].re * sc);
}

int orc_finalize(const float *acc_in, int N, double a, const ppm_final_cfg *cfg,
                 float *half1, float *half2, float *filt, double *stats) {
    fft_tables();
    if (!box_ok(N)) return -22;
    size_t NX = N / 2 + 1, half_sz = (size_t)N * N * NX * 3, n3 = (size_t)N * N * N;
    float *acc = (float *)malloc(2 * half_sz * sizeof(float));
    memcpy(acc, acc_in, 2 * half_sz * sizeof(float));
    /* kx = 0 plane holds both Friedel mates: fold them together */
    for (int h = 0; h < 2; h++) {
        float *A = acc + h * half_sz; const float *S = acc_in + h * half_sz;
        for (int z = -N / 2 + 1; z < N / 2; z++) for (int y = -N / 2 + 1; y < N / 2; y++) {
            float *v = A + (((size_t)(z + N / 2) * N + (y + N / 2)) * NX) * 3;
            const float *m = S + (((size_t)(-z + N / 2) * N + (-y + N / 2)) * NX) * 3;
            v[0] += m[0]; v[1] -= m[1]; v[2] += m[2];
        }
    }
    int ns = N / 2;
    double *sden = (double *)calloc(2 * ns + 2, sizeof(double)), *scnt = (double *)calloc(ns + 1, sizeof(double));
    double *sdt = (double *)calloc(ns + 1, sizeof(double));
    for (int z = -N / 2; z < N / 2; z++) for (int y = -N / 2; y < N / 2; y++) for (int x = 0; x <= N / 2; x++) {
        int b = (int)floor(sqrt((double)x * x + y * y + z * z) + 0.5);
        if (b >= ns) continue;
        size_t i = (((size_t)(z + N / 2) * N + (y + N / 2)) * NX + x) * 3;
        double al = x == 0 ? 1.0 : 2.0;
        sden[b] += al * acc[i + 2]; sden[ns + 1 + b] += al * acc[half_sz + i + 2]; scnt[b] += al;
        sdt[b] += al * (acc[i + 2] + acc[half_sz + i + 2]);
    }
    double *c12 = (double *)calloc(ns + 1, sizeof(double)), *c11 = (double *)calloc(ns + 1, sizeof(double)), *c22 = (double *)calloc(ns + 1, sizeof(double));
    for (int z = -N / 2; z < N / 2; z++) for (int y = -N / 2; y < N / 2; y++) for (int x = 0; x <= N / 2; x++) {
        int b = (int)floor(sqrt((double)x * x + y * y + z * z) + 0.5);
        if (b >= ns) continue;
        size_t i = (((size_t)(z + N / 2) * N + (y + N / 2)) * NX + x) * 3;
        double e1 = 1e-3 * sden[b] / scnt[b] + 1e-20, e2 = 1e-3 * sden[ns + 1 + b] / scnt[b] + 1e-20;
        double d1 = acc[i + 2] + e1, d2 = acc[half_sz + i + 2] + e2;
        double ar = acc[i] / d1, ai = acc[i + 1] / d1, br = acc[half_sz + i] / d2, bi = acc[half_sz + i + 1] / d2;
        double al = x == 0 ? 1.0 : 2.0;
        c12[b] += al * (ar * br + ai * bi); c11[b] += al * (ar * ar + ai * ai); c22[b] += al * (br * br + bi * bi);
    }
    double vfrac = cfg->molecular_mass_kda > 0 ? (cfg->molecular_mass_kda * 1000.0 / 0.81) / pow(N * a, 3.0) : 1.0;
    if (vfrac > 1) vfrac = 1; if (vfrac < 1e-6) vfrac = 1e-6;
    double *kap = (double *)calloc(ns + 1, sizeof(double));
    for (int b = 0; b < ns; b++) {
        double fsc = (c11[b] > 0 && c22[b] > 0) ? c12[b] / sqrt(c11[b] * c22[b]) : 0.0;
        double fc = fsc < 0 ? 0 : (fsc > 0.999 ? 0.999 : fsc);
        double rec = 2.0 * fc / (1.0 - fc), md = scnt[b] > 0 ? sdt[b] / scnt[b] : 0;
        kap[b] = b == 0 ? 1e-20 : md / (rec > 1e-6 ? rec : 1e-6);
        if (b >= 1 && stats) {
            double *s = stats + (size_t)(b - 1) * PPM_STATS_COLS;
            s[0] = b; s[1] = N * a / b; s[2] = b / (N * a); s[3] = fsc;
            s[4] = fc / (fc + vfrac * (1 - fc)); s[5] = md > 0 ? rec / md / vfrac : 0; s[6] = rec;
        }
    }
    cpx *f = (cpx *)malloc(n3 * sizeof(cpx));
    float *outs[3] = { half1, half2, filt };
    double rout = cfg->outer_radius / a, rin = cfg->inner_radius / a, fo = (cfg->mask_falloff > 0 ? cfg->mask_falloff : 10.0) / a;
    for (int which = 0; which < 3; which++) {
        if (!outs[which]) continue;
        memset(f, 0, n3 * sizeof(cpx));
        for (int z = -N / 2; z < N / 2; z++) for (int y = -N / 2; y < N / 2; y++) for (int x = 0; x <= N / 2; x++) {
            int b = (int)floor(sqrt((double)x * x + y * y + z * z) + 0.5);
            if (b >= ns) continue;
            size_t i = (((size_t)(z + N / 2) * N + (y + N / 2)) * NX + x) * 3;
            double nr, ni, dn;
            if (which < 2) { nr = acc[which * half_sz + i]; ni = acc[which * half_sz + i + 1]; dn = acc[which * half_sz + i + 2]; }
            else { nr = acc[i] + acc[half_sz + i]; ni = acc[i + 1] + acc[half_sz + i + 1]; dn = acc[i + 2] + acc[half_sz + i + 2]; }
            double d = dn + kap[b], sg = ((x + y + z) & 1) ? -1.0 : 1.0;
            double vr = sg * nr / d, vi = sg * ni / d;
            int ix = x % N, iy = (y + N) % N, iz = (z + N) % N;
            cpx *o = &f[((size_t)iz * N + iy) * N + ix]; o->re = (float)vr; o->im = (float)vi;
            if (x > 0 && x < N / 2) {
                cpx *q = &f[((size_t)((N - iz) % N) * N + ((N - iy) % N)) * N + (N - ix)];
                q->re = (float)vr; q->im = (float)-vi;
            }
        }
        ifft3_centered_real(f, N, outs[which]);
        for (int z = 0; z < N; z++) for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) {
            double dx = x - N / 2, dy = y - N / 2, dz = z - N / 2, g3 = 1.0;
            double t[3] = { dx / N, dy / N, dz / N };
            for (int q = 0; q < 3; q++) { double u = ORC_PI * t[q], sv = fabs(u) < 1e-9 ? 1.0 : sin(u) / u; g3 *= sv * sv; }
            double rho = sqrt(dx * dx + dy * dy + dz * dz), m = 1.0;
            if (rout > 0) {
                if (rho >= rout + 0.5 * fo) m = 0; else if (rho > rout - 0.5 * fo) m = 0.5 * (1 + cos(ORC_PI * (rho - rout + 0.5 * fo) / fo));
            }
            if (rin > 0 && rho < rin) m = 0;
            size_t i = ((size_t)z * N + y) * N + x;
            outs[which][i] = (float)(outs[which][i] / g3 * m);
        }
    }
    free(f); free(kap); free(c12); free(c11); free(c22); free(sden); free(scnt); free(sdt); free(acc);
    return 0;
}

/* ------------------------------------------------------------------ constrained refinement (csp) */
/* f-1 / H13: the absent `csp` program (argv at src/pyp/system/local_run.py:364-376, :451-463; modes at
 * src/pyp/align/core.py:1015-1023).  The row <-> (particle, tilt) geometry IS stated in the reference's Python
 * (csp_euler_angles, src/pyp/analysis/geometry/core.py:1081-1213) and is pinned by golden vectors produced with it
 * (tests/golden/golden_r02.json "csp_geometry"); the optimiser (compass search with frequency marching over the
 * constrained parameters, unit score = mean row score) is build-defined: PARITY UNPINNED like the refine3d restatement. */
static void rot_xyz(int k, double deg, double R[9]) {      /* right-handed rotation about x (0), y (1), z (2) */
    double t = deg * ORC_PI / 180, c = cos(t), s = sin(t);
    double rx[9] = { 1, 0, 0, 0, c, -s, 0, s, c }, ry[9] = { c, 0, s, 0, 1, 0, -s, 0, c }, rz[9] = { c, -s, 0, s, c, 0, 0, 0, 1 };
    memcpy(R, k == 0 ? rx : (k == 1 ? ry : rz), 9 * sizeof(double));
}

/* M_row = N Ry(-tilt) Rz(axis); geometric shift (pixels) = [Rz(-axis) Ry(tilt) (-p)]_xy + tilt shift */
static void csp_row_pose(const double N[9], const double p[3], double tilt, double axis, double tsx, double tsy, double M[9], double g[2]) {
    double a[9], b[9], t[9];
    rot_xyz(1, -tilt, a); rot_xyz(2, axis, b);
    mat_mul3(N, a, t); mat_mul3(t, b, M);
    rot_xyz(2, -axis, a); rot_xyz(1, tilt, b);
    double q[3] = { -p[0], -p[1], -p[2] }, u[3], v[3];
    for (int i = 0; i < 3; i++) u[i] = b[i * 3] * q[0] + b[i * 3 + 1] * q[1] + b[i * 3 + 2] * q[2];
    for (int i = 0; i < 3; i++) v[i] = a[i * 3] * u[0] + a[i * 3 + 1] * u[1] + a[i * 3 + 2] * u[2];
    g[0] = v[0] + tsx; g[1] = v[1] + tsy;
}

/* exported for the golden test: particle = {psi, theta, phi (as stored in the particle block), shift x, y, z};
 * out = {PSI, THETA, PHI, SHX, SHY} of the projection row */
void orc_csp_pose(double tilt, double axis, const double *particle, double *out) {
    double N[9], M[9], g[2];
    euler_full(-particle[0], -particle[1], -particle[2], N);
    csp_row_pose(N, particle + 3, tilt, axis, 0.0, 0.0, M, g);
    angles_from_matrix(M, &out[0], &out[1], &out[2]);
    out[3] = g[0]; out[4] = g[1];
}

typedef struct { double N[9], p[3], tl[4] /* angle, axis, sx, sy (tilts) */, acc[6]; } cunit_t;

/* state + displacement d (particles: rotations about specimen x, y, z in degrees, then shifts; tilts: angle, axis, -, sx, sy, -) */
static void csp_apply(int unit_kind, const cunit_t *s, const double d[6], cunit_t *o) {
    *o = *s;
    if (unit_kind == PPM_CSP_PARTICLES) {
        double R[9], T[9];
        for (int k = 0; k < 3; k++) if (d[k] != 0) { rot_xyz(k, d[k], R); mat_mul3(o->N, R, T); memcpy(o->N, T, sizeof(T)); }
        for (int k = 0; k < 3; k++) o->p[k] += d[3 + k];
    } else {
        o->tl[0] += d[0]; o->tl[1] += d[1]; o->tl[2] += d[3]; o->tl[3] += d[4];
    }
    for (int k = 0; k < 6; k++) o->acc[k] += d[k];
}

typedef struct {
    const oref_t *r; const geom_t *g; int n_proj; const double *rows; cpx **I; double **wr; ctf_t *ctf;
    const int *row_part, *row_tilt; const double *s0, *g0; const cunit_t *parts, *tls; const unsigned char *usable;
} cspctx_t;

static double csp_row_score(const cspctx_t *c, int j, const cunit_t *pu, const cunit_t *tu, double rmax, double M[9], double sh[2]) {
    double g[2];
    csp_row_pose(pu->N, pu->p, tu->tl[0], tu->tl[1], tu->tl[2], tu->tl[3], M, g);
    sh[0] = c->s0[2 * j] + g[0] - c->g0[2 * j]; sh[1] = c->s0[2 * j + 1] + g[1] - c->g0[2 * j + 1];
    return score_local(c->r, c->g, &c->ctf[j], c->I[j], c->wr[j], rmax, M, sh);
}

/* mean score of the usable rows of a unit for candidate state `cand` */
static double csp_unit_score(const cspctx_t *c, int kind, const int *urows, int nrows, const cunit_t *cand, double rmax, long *nev) {
    double s = 0; int n = 0, M9 = 0; (void)M9;
    for (int q = 0; q < nrows; q++) {
        int j = urows[q];
        if (!c->usable[j]) continue;
        double M[9], sh[2];
        const cunit_t *pu = kind == PPM_CSP_PARTICLES ? cand : &c->parts[c->row_part[j]];
        const cunit_t *tu = kind == PPM_CSP_PARTICLES ? &c->tls[c->row_tilt[j]] : cand;
        s += csp_row_score(c, j, pu, tu, rmax, M, sh); n++; (*nev)++;
    }
    return n ? s / n : -1e300;
}

int orc_csp_refine(void *refp, const ppm_refine_cfg *cfg, const ppm_csp_cfg *cc, const float *images, int n_proj,
                   double *rows, double *particles, int n_part, double *tilts, int n_tilt, long *eval_count) {
    fft_tables();
    oref_t *r = (oref_t *)refp; geom_t g;
    ppm_refine_cfg c2 = *cfg; c2.global_search = 0;
    if (!r || geom_init(&g, &c2) || g.B > (r->B + 1) / r->pad - 1 || r->N != g.N) return -22;
    if (cc->unit != PPM_CSP_PARTICLES && cc->unit != PPM_CSP_MICROGRAPHS) return -22;
    const double fall = cfg->mask_falloff > 0 ? cfg->mask_falloff : 20.0, bf = cfg->band_factor == 0 ? 3.0 : cfg->band_factor;
    const double rm_px = cfg->mask_radius / g.a;
    size_t nb = (size_t)g.H * g.W;
    cspctx_t c; memset(&c, 0, sizeof(c));
    c.r = r; c.g = &g; c.n_proj = n_proj; c.rows = rows;
    c.I = (cpx **)calloc(n_proj, sizeof(cpx *)); c.wr = (double **)calloc(n_proj, sizeof(double *)); c.ctf = (ctf_t *)calloc(n_proj, sizeof(ctf_t));
    int *row_part = (int *)malloc(n_proj * sizeof(int)), *row_tilt = (int *)malloc(n_proj * sizeof(int));
    double *s0 = (double *)malloc(2 * n_proj * sizeof(double)), *g0 = (double *)malloc(2 * n_proj * sizeof(double));
    unsigned char *usable = (unsigned char *)calloc(n_proj, 1);
    cunit_t *parts = (cunit_t *)calloc(n_part, sizeof(cunit_t)), *tls = (cunit_t *)calloc(n_tilt, sizeof(cunit_t));
    for (int i = 0; i < n_part; i++) {
        const double *P = particles + (size_t)i * PPM_NPCOL;
        euler_full(-P[4], -P[5], -P[6], parts[i].N);
        parts[i].p[0] = P[1]; parts[i].p[1] = P[2]; parts[i].p[2] = P[3];
    }
    for (int i = 0; i < n_tilt; i++) {
        const double *T = tilts + (size_t)i * PPM_NTCOL;
        tls[i].tl[0] = T[4]; tls[i].tl[1] = T[5]; tls[i].tl[2] = T[2]; tls[i].tl[3] = T[3];
    }
    int err = 0;
    for (int j = 0; j < n_proj && !err; j++) {
        const double *row = rows + (size_t)j * PPM_NCOL;
        row_part[j] = row_tilt[j] = -1;
        for (int i = 0; i < n_part; i++) if ((long)particles[(size_t)i * PPM_NPCOL] == (long)row[PPM_PIND]) { row_part[j] = i; break; }
        for (int i = 0; i < n_tilt; i++) if ((long)tilts[(size_t)i * PPM_NTCOL] == (long)row[PPM_TIND] && (long)tilts[(size_t)i * PPM_NTCOL + 1] == (long)row[28]) { row_tilt[j] = i; break; }
        if (row_part[j] < 0 || row_tilt[j] < 0) { err = -22; break; }
        long tind = (long)row[PPM_TIND];
        usable[j] = row[PPM_OCC] > 0 && tind >= cc->tind_min && (cc->tind_max < 0 || tind <= cc->tind_max);
        s0[2 * j] = row[PPM_XSHIFT] / g.a; s0[2 * j + 1] = row[PPM_YSHIFT] / g.a;
        double M[9];
        csp_row_pose(parts[row_part[j]].N, parts[row_part[j]].p, tls[row_tilt[j]].tl[0], tls[row_tilt[j]].tl[1], tls[row_tilt[j]].tl[2], tls[row_tilt[j]].tl[3], M, g0 + 2 * j);
    }
    c.row_part = row_part; c.row_tilt = row_tilt; c.s0 = s0; c.g0 = g0; c.parts = parts; c.tls = tls; c.usable = usable;
    if (!err) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int j = 0; j < n_proj; j++) {
            c.I[j] = (cpx *)malloc(nb * sizeof(cpx)); c.wr[j] = (double *)malloc((g.B + 2) * sizeof(double));
            preprocess_row(images + (size_t)j * g.N * g.N, &g, cfg->mask_radius, fall, cfg->normalize, cfg->invert, 1, 1, g.r_hi, c.I[j], c.wr[j], NULL, rows + (size_t)j * PPM_NCOL);
            ctf_init(&c.ctf[j], rows + (size_t)j * PPM_NCOL, g.N, g.a);
        }
    }
    if (cc->refine_defocus && cc->unit == PPM_CSP_MICROGRAPHS && !err) {
        /* csp mode 4: one defocus offset per tilt, scored at the rows' current poses and the full band */
        int nt = 0;
        const double step = cc->defocus_step > 0 ? cc->defocus_step : 50.0;
        if (cc->defocus_range >= step) { nt = (int)floor(cc->defocus_range / step + 1e-6); if (nt > PPM_MAX_DEFOCUS_STEPS) nt = PPM_MAX_DEFOCUS_STEPS; }
        long nev4 = 0;
        for (int u = 0; u < n_tilt; u++) {
            long id = (long)tilts[(size_t)u * PPM_NTCOL];
            if (id < cc->first || (cc->last >= 0 && id > cc->last)) continue;
            double best = -1e300; int bt = 0;
            for (int pass = 0; pass < 2; pass++) {           /* the unshifted values first (they win ties), then -nt .. nt */
                for (int t = (pass ? -nt : 0); t <= (pass ? nt : 0); t++) {
                    if (pass && t == 0) continue;
                    double ssum = 0; int sn = 0;
                    for (int j = 0; j < n_proj; j++) {
                        if (row_tilt[j] != u || !usable[j]) continue;
                        ctf_t c2 = c.ctf[j]; c2.df1 += t * step; c2.df2 += t * step;
                        double M[9], g2[2], sh[2];
                        csp_row_pose(parts[row_part[j]].N, parts[row_part[j]].p, tls[u].tl[0], tls[u].tl[1], tls[u].tl[2], tls[u].tl[3], M, g2);
                        sh[0] = s0[2 * j] + g2[0] - g0[2 * j]; sh[1] = s0[2 * j + 1] + g2[1] - g0[2 * j + 1];
                        ssum += score_local(r, &g, &c2, c.I[j], c.wr[j], g.r_hi, M, sh); sn++; nev4++;
                    }
                    if (sn && ssum / sn > best) { best = ssum / sn; bt = t; }
                }
            }
            for (int j = 0; j < n_proj; j++) {
                if (row_tilt[j] != u) continue;
                double *row = rows + (size_t)j * PPM_NCOL;
                row[PPM_DF1] += bt * step; row[PPM_DF2] += bt * step;
                ctf_t c2; ctf_init(&c2, row, g.N, g.a);
                double M[9], g2[2], sh[2];
                csp_row_pose(parts[row_part[j]].N, parts[row_part[j]].p, tls[u].tl[0], tls[u].tl[1], tls[u].tl[2], tls[u].tl[3], M, g2);
                sh[0] = s0[2 * j] + g2[0] - g0[2 * j]; sh[1] = s0[2 * j + 1] + g2[1] - g0[2 * j + 1];
                double cc2 = score_local(r, &g, &c2, c.I[j], c.wr[j], g.r_hi, M, sh), res = 1.0 - cc2 * cc2; nev4++;
                if (res < 1e-6) res = 1e-6;
                row[PPM_SCORE] = 100.0 * cc2; row[PPM_SIGMA] = sqrt(res);
                row[PPM_LOGP] = -0.5 * (ORC_PI * (g.r_hi * g.r_hi - g.r_lo * g.r_lo)) * (log(2.0 * ORC_PI * res) + 1.0);
            }
        }
        if (eval_count) *eval_count = nev4;
        for (int j = 0; j < n_proj; j++) { free(c.I[j]); free(c.wr[j]); }
        free(c.I); free(c.wr); free(c.ctf); free(row_part); free(row_tilt); free(s0); free(g0); free(usable); free(parts); free(tls);
        return 0;
    }
    const int kind = cc->unit, nu_all = kind == PPM_CSP_PARTICLES ? n_part : n_tilt;
    int en[6] = { 0, 0, 0, 0, 0, 0 }; double tol[6] = { 0, 0, 0, 0, 0, 0 };
    if (kind == PPM_CSP_PARTICLES) {
        for (int k = 0; k < 3; k++) { en[k] = cc->refine_rotation != 0; tol[k] = cc->tol_angle[k]; en[3 + k] = cc->refine_translation != 0; tol[3 + k] = cc->tol_shift; }
    } else {
        en[0] = en[1] = cc->refine_rotation != 0; tol[0] = cc->tol_angle[0]; tol[1] = cc->tol_angle[1];
        en[3] = en[4] = cc->refine_translation != 0; tol[3] = tol[4] = cc->tol_shift;
    }
    for (int k = 0; k < 6; k++) if (!(tol[k] > 0)) en[k] = 0;
    double ha0 = 0, hs0 = 0;
    for (int k = 0; k < 3; k++) if (en[k] && 0.5 * tol[k] > ha0) ha0 = 0.5 * tol[k];
    for (int k = 3; k < 6; k++) if (en[k] && 0.5 * tol[k] > hs0) hs0 = 0.5 * tol[k];
    const double steptol = cc->step_tolerance > 0 ? cc->step_tolerance : 0.01;
    int T = cc->max_iterations;
    if (T <= 0) { double m = ha0 > hs0 ? ha0 : hs0; T = m > steptol ? (int)ceil(log(m / steptol) / log(2.0)) : 1; if (T > 12) T = 12; if (T < 1) T = 1; }
    int en5[5] = { en[0] || en[1] || en[2], 0, 0, en[3] || en[4] || en[5], 0 };       /* for iter_band: any angle / any shift */
    long nev = 0;
    /* rows of every unit */
    int *ucount = (int *)calloc(nu_all, sizeof(int)), **urows = (int **)calloc(nu_all, sizeof(int *));
    for (int j = 0; j < n_proj && !err; j++) ucount[kind == PPM_CSP_PARTICLES ? row_part[j] : row_tilt[j]]++;
    for (int u = 0; u < nu_all; u++) { urows[u] = (int *)malloc((ucount[u] + 1) * sizeof(int)); ucount[u] = 0; }
    for (int j = 0; j < n_proj && !err; j++) { int u = kind == PPM_CSP_PARTICLES ? row_part[j] : row_tilt[j]; urows[u][ucount[u]++] = j; }
    cunit_t *units = kind == PPM_CSP_PARTICLES ? parts : tls;
    unsigned char *refined = (unsigned char *)calloc(nu_all, 1);
    for (int u = 0; u < nu_all && !err; u++) {
        long id = (long)(kind == PPM_CSP_PARTICLES ? particles[(size_t)u * PPM_NPCOL] : tilts[(size_t)u * PPM_NTCOL]);
        if (id < cc->first || (cc->last >= 0 && id > cc->last)) continue;
        int nus = 0; for (int q = 0; q < ucount[u]; q++) nus += usable[urows[u][q]];
        refined[u] = 1;
        if (!nus) continue;
        cunit_t s = units[u];
        double ha = ha0, hs = hs0;
        for (int it = 0; it < T; it++) {
            const double rmax = iter_band(&g, rm_px, bf, en5, ha, hs, g.r_hi);
            const double f0 = csp_unit_score(&c, kind, urows[u], ucount[u], &s, rmax, &nev);
            double fp[6], fm[6], d[6];
            int okp[6], okm[6];
            for (int i = 0; i < 6; i++) {
                d[i] = 0; fp[i] = fm[i] = -1e300; okp[i] = okm[i] = 0;
                if (!en[i]) continue;
                const double h = i < 3 ? ha : hs;
                for (int sg = 0; sg < 2; sg++) {
                    double dd[6] = { 0, 0, 0, 0, 0, 0 }; dd[i] = sg ? -h : h;
                    cunit_t q; csp_apply(kind, &s, dd, &q);
                    const int ok = fabs(q.acc[i]) <= tol[i] + 1e-9;
                    const double v = csp_unit_score(&c, kind, urows[u], ucount[u], &q, rmax, &nev);   /* evaluated like the device does; ignored when out of bounds */
                    if (sg) { fm[i] = ok ? v : -1e300; okm[i] = ok; } else { fp[i] = ok ? v : -1e300; okp[i] = ok; }
                }
                if (okp[i] && okm[i]) {
                    const double den = 2.0 * f0 - fp[i] - fm[i];
                    if (den > 1e-12) { double t = 0.5 * h * (fp[i] - fm[i]) / den; d[i] = t > h ? h : (t < -h ? -h : t); }
                    else { const double best = fp[i] > fm[i] ? fp[i] : fm[i]; d[i] = best > f0 ? (fp[i] > fm[i] ? h : -h) : 0.0; }
                } else if (okp[i]) d[i] = fp[i] > f0 ? h : 0.0;
                else if (okm[i]) d[i] = fm[i] > f0 ? -h : 0.0;
                if (s.acc[i] + d[i] > tol[i]) d[i] = tol[i] - s.acc[i];
                if (s.acc[i] + d[i] < -tol[i]) d[i] = -tol[i] - s.acc[i];
            }
            cunit_t tr; csp_apply(kind, &s, d, &tr);
            const double ft = csp_unit_score(&c, kind, urows[u], ucount[u], &tr, rmax, &nev);
            int bi = -1, bs = 0; double fb = f0;
            for (int i = 0; i < 6; i++) {
                if (!en[i]) continue;
                if (fp[i] > fb) { fb = fp[i]; bi = i; bs = 1; }
                if (fm[i] > fb) { fb = fm[i]; bi = i; bs = -1; }
            }
            if (ft > f0 && ft >= fb) s = tr;
            else if (bi >= 0) { double dd[6] = { 0, 0, 0, 0, 0, 0 }; dd[bi] = bs * (bi < 3 ? ha : hs); cunit_t q; csp_apply(kind, &s, dd, &q); s = q; }
            ha *= 0.5; hs *= 0.5;
        }
        units[u] = s;
    }
    /* write back: unit parameters, rows of refined units */
    for (int u = 0; u < nu_all && !err; u++) {
        if (!refined[u]) continue;
        if (kind == PPM_CSP_PARTICLES) {
            double *P = particles + (size_t)u * PPM_NPCOL, a1, a2, a3;
            angles_from_matrix(units[u].N, &a1, &a2, &a3);
            P[4] = -a1; P[5] = -a2; P[6] = -a3; P[1] = units[u].p[0]; P[2] = units[u].p[1]; P[3] = units[u].p[2];
        } else {
            double *Tt = tilts + (size_t)u * PPM_NTCOL;
            Tt[4] = units[u].tl[0]; Tt[5] = units[u].tl[1]; Tt[2] = units[u].tl[2]; Tt[3] = units[u].tl[3];
        }
        double ssum = 0; int sn = 0;
        for (int q = 0; q < ucount[u]; q++) {
            int j = urows[u][q];
            double *row = rows + (size_t)j * PPM_NCOL, M[9], sh[2];
            double cc2 = csp_row_score(&c, j, &parts[row_part[j]], &tls[row_tilt[j]], g.r_hi, M, sh); nev++;
            angles_from_matrix(M, &row[PPM_PSI], &row[PPM_THETA], &row[PPM_PHI]);
            row[PPM_XSHIFT] = sh[0] * g.a; row[PPM_YSHIFT] = sh[1] * g.a;
            double res = 1.0 - cc2 * cc2; if (res < 1e-6) res = 1e-6;
            row[PPM_SCORE] = 100.0 * cc2; row[PPM_SIGMA] = sqrt(res);
            row[PPM_LOGP] = -0.5 * (ORC_PI * (g.r_hi * g.r_hi - g.r_lo * g.r_lo)) * (log(2.0 * ORC_PI * res) + 1.0);
            if (usable[j]) { ssum += row[PPM_SCORE]; sn++; }
        }
        if (kind == PPM_CSP_PARTICLES) particles[(size_t)u * PPM_NPCOL + 10] = sn ? ssum / sn : -1.0;
    }
    if (eval_count) *eval_count = nev;
    for (int j = 0; j < n_proj; j++) { free(c.I[j]); free(c.wr[j]); }
    for (int u = 0; u < nu_all; u++) free(urows[u]);
    free(urows); free(ucount); free(refined); free(c.I); free(c.wr); free(c.ctf); free(row_part); free(row_tilt); free(s0); free(g0); free(usable); free(parts); free(tls);
    return err;
}

/* ------------------------------------------------------------------ sub-tomogram alignment (3DAVG) */
/* f-4, second half: the absent MPI_Classification (src/pyp/refine/tomo_avg/sub_tomo_avg.py:468-555; protocol fields
 * src/pyp/refine/3DAVG/iteration_002_mode_3.xml: image window, band-pass, missing wedge, search ranges).  Its spherical-
 * harmonics search is not visible; restated here is the documented objective (band-passed, wedge-weighted normalised
 * cross-correlation of 3-D transforms) with the compass search of the rest of this path - build-defined, PARITY UNPINNED. */
typedef struct { int kx, ky, kz; float w; } svs_t;

/* pass-band weight of a frequency s (cycles per pixel): Gaussian roll-offs outside [highpass, lowpass] */
static double sva_band_weight(const ppm_sva_cfg *c, double s) {
    double w = 1.0;
    if (c->highpass_cutoff > 0 && s < c->highpass_cutoff) {
        double d = c->highpass_cutoff - s;
        w *= c->highpass_decay > 0 ? exp(-d * d / (2.0 * c->highpass_decay * c->highpass_decay)) : 0.0;
    }
    if (c->lowpass_cutoff > 0 && s > c->lowpass_cutoff) {
        double d = s - c->lowpass_cutoff;
        w *= c->lowpass_decay > 0 ? exp(-d * d / (2.0 * c->lowpass_decay * c->lowpass_decay)) : 0.0;
    }
    return w;
}

/* largest Fourier radius (pixels) that still carries weight >= 1e-3 */
static double sva_band_radius(const ppm_sva_cfg *c) {
    int N = c->box;
    double s = c->lowpass_cutoff > 0 ? c->lowpass_cutoff + (c->lowpass_decay > 0 ? 3.7169 * c->lowpass_decay : 0.0) : 0.5;
    if (s > 0.5) s = 0.5;
    double r = s * N; if (r > N / 2 - 1) r = N / 2 - 1;
    return r;
}

/* a sample belongs to the measured region when the plane through the tilt axis (y) that contains it lies inside the tilt
 * range: angle of (kx, kz) from the kx axis, folded to (-90, 90] */
static int sva_in_wedge(int kx, int kz, double lw, double uw) {
    if (kx == 0 && kz == 0) return 1;
    double a = atan2((double)kz, (double)kx) * 180.0 / ORC_PI;
    if (a > 90.0) a -= 180.0; if (a <= -90.0) a += 180.0;
    return a >= lw && a <= uw;
}

static void fft3_inplace(cpx *f, int N, int inverse) {
    for (int z = 0; z < N; z++) for (int y = 0; y < N; y++) fft1d(f + ((size_t)z * N + y) * N, N, 1, inverse);
    for (int z = 0; z < N; z++) for (int x = 0; x < N; x++) fft1d(f + (size_t)z * N * N + x, N, N, inverse);
    for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) fft1d(f + (size_t)y * N + x, N, N * N, inverse);
}

/* score of one pose: sum over the sample list up to radius rmax */
static double sva_score(const oref_t *r, const svs_t *sl, int ns, const cpx *F, int N, double rmax, const double Nm[9], const double p[3]) {
    double A = 0, B = 0, C = 0, r2 = rmax * rmax;
    for (int i = 0; i < ns; i++) {
        const int kx = sl[i].kx, ky = sl[i].ky, kz = sl[i].kz;
        if ((double)kx * kx + (double)ky * ky + (double)kz * kz >= r2) continue;
        double pr, pi;
        sample_cube(r, Nm[0] * kx + Nm[1] * ky + Nm[2] * kz, Nm[3] * kx + Nm[4] * ky + Nm[5] * kz, Nm[6] * kx + Nm[7] * ky + Nm[8] * kz, &pr, &pi);
        double ph = 2.0 * ORC_PI * (kx * p[0] + ky * p[1] + kz * p[2]) / N, cr = cos(ph), ci = sin(ph);
        double mr = pr * cr - pi * ci, mi = pr * ci + pi * cr, w = sl[i].w;
        A += w * (F[i].re * mr + F[i].im * mi); B += w * (mr * mr + mi * mi); C += w * ((double)F[i].re * F[i].re + (double)F[i].im * F[i].im);
    }
    return (B > 0 && C > 0) ? A / sqrt(B * C) : 0.0;
}

/* `T` compass iterations of one sub-volume's pose (rotations about the specimen axes + 3-D shift, bounded by +-tol about where
 * s->acc started), steps ha / hs halved after every iteration; frequency marching under the cap `rband`. */
static void sva_compass(const oref_t *r, const svs_t *sl, int ns, const cpx *F, int N, double rband, double rm_px, double bf,
                        const int en[6], const double tol[6], cunit_t *sp, double *hap, double *hsp, int T, long *nevp) {
    cunit_t s = *sp;
    double ha = *hap, hs = *hsp;
    long nev = 0;
    int en5[5] = { en[0], 0, 0, en[3], 0 };
    geom_t g; memset(&g, 0, sizeof(g)); g.N = N;
    for (int it = 0; it < T && ns > 0; it++) {
        const double rmax = iter_band(&g, rm_px, bf, en5, ha, hs, rband);
        const double f0 = sva_score(r, sl, ns, F, N, rmax, s.N, s.p); nev++;
        double fp[6], fm[6], d[6]; int okp[6], okm[6];
        for (int i = 0; i < 6; i++) {
            d[i] = 0; fp[i] = fm[i] = -1e300; okp[i] = okm[i] = 0;
            if (!en[i]) continue;
            const double h = i < 3 ? ha : hs;
            for (int sg = 0; sg < 2; sg++) {
                double dd[6] = { 0, 0, 0, 0, 0, 0 }; dd[i] = sg ? -h : h;
                cunit_t q; csp_apply(PPM_CSP_PARTICLES, &s, dd, &q);
                const int ok = fabs(q.acc[i]) <= tol[i] + 1e-9;
                const double val = sva_score(r, sl, ns, F, N, rmax, q.N, q.p); nev++;
                if (sg) { fm[i] = ok ? val : -1e300; okm[i] = ok; } else { fp[i] = ok ? val : -1e300; okp[i] = ok; }
            }
            if (okp[i] && okm[i]) {
                const double den = 2.0 * f0 - fp[i] - fm[i];
                if (den > 1e-12) { double t = 0.5 * h * (fp[i] - fm[i]) / den; d[i] = t > h ? h : (t < -h ? -h : t); }
                else { const double best = fp[i] > fm[i] ? fp[i] : fm[i]; d[i] = best > f0 ? (fp[i] > fm[i] ? h : -h) : 0.0; }
            } else if (okp[i]) d[i] = fp[i] > f0 ? h : 0.0;
            else if (okm[i]) d[i] = fm[i] > f0 ? -h : 0.0;
            if (s.acc[i] + d[i] > tol[i]) d[i] = tol[i] - s.acc[i];
            if (s.acc[i] + d[i] < -tol[i]) d[i] = -tol[i] - s.acc[i];
        }
        cunit_t tr; csp_apply(PPM_CSP_PARTICLES, &s, d, &tr);
        const double ft = sva_score(r, sl, ns, F, N, rmax, tr.N, tr.p); nev++;
        int bi = -1, bs = 0; double fb = f0;
        for (int i = 0; i < 6; i++) {
            if (!en[i]) continue;
            if (fp[i] > fb) { fb = fp[i]; bi = i; bs = 1; }
            if (fm[i] > fb) { fb = fm[i]; bi = i; bs = -1; }
        }
        if (ft > f0 && ft >= fb) s = tr;
        else if (bi >= 0) { double dd[6] = { 0, 0, 0, 0, 0, 0 }; dd[bi] = bs * (bi < 3 ? ha : hs); cunit_t q; csp_apply(PPM_CSP_PARTICLES, &s, dd, &q); s = q; }
        ha *= 0.5; hs *= 0.5;
    }
    *sp = s; *hap = ha; *hsp = hs; *nevp += nev;
}

/* rotations of the global grid (include/ppm.h, ppm_sva_cfg.search_mode): G = Rz(phi) Ry(theta) Rz(psi) for grid point `idx` */
static int sva_grid_size(double step, int *n_theta, int *n_psi) {
    *n_theta = (int)floor(180.0 / step + 0.5) + 1; if (*n_theta < 2) *n_theta = 2;
    *n_psi = (int)floor(360.0 / step + 0.5); if (*n_psi < 1) *n_psi = 1;
    int nd = 0;
    for (int i = 0; i < *n_theta; i++) { int np = (int)floor(360.0 * sin(ORC_PI * i / (*n_theta - 1)) / step + 0.5); if (np < 1) np = 1; nd += np; }
    return nd * *n_psi;
}
static void sva_grid_rotation(double step, int n_theta, int n_psi, int idx, double G[9]) {
    int dir = idx / n_psi, k = idx - dir * n_psi, acc = 0;
    double th = 0, ph = 0;
    for (int i = 0; i < n_theta; i++) {
        double t = 180.0 * i / (n_theta - 1);
        int np = (int)floor(360.0 * sin(t * ORC_PI / 180.0) / step + 0.5); if (np < 1) np = 1;
        if (dir < acc + np) { th = t; ph = 360.0 * (dir - acc) / np; break; }
        acc += np;
    }
    euler_full(k * 360.0 / n_psi, th, ph, G);
}

/* translation-invariant score of a rotation: correlation of the AMPLITUDES |F(k)| and |Ref(N k)| (a shift only changes phases) */
static double sva_score_amp(const oref_t *r, const svs_t *sl, int ns, const cpx *F, double rmax, const double Nm[9]) {
    double A = 0, B = 0, C = 0, r2 = rmax * rmax;
    for (int i = 0; i < ns; i++) {
        const int kx = sl[i].kx, ky = sl[i].ky, kz = sl[i].kz;
        if ((double)kx * kx + (double)ky * ky + (double)kz * kz >= r2) continue;
        double pr, pi;
        sample_cube(r, Nm[0] * kx + Nm[1] * ky + Nm[2] * kz, Nm[3] * kx + Nm[4] * ky + Nm[5] * kz, Nm[6] * kx + Nm[7] * ky + Nm[8] * kz, &pr, &pi);
        const double m2 = pr * pr + pi * pi, f2 = (double)F[i].re * F[i].re + (double)F[i].im * F[i].im, w = sl[i].w;
        A += w * sqrt(m2 * f2); B += w * m2; C += w * f2;
    }
    return (B > 0 && C > 0) ? A / sqrt(B * C) : 0.0;
}

int orc_sva_align(void *refp, const ppm_sva_cfg *cfg, const float *volumes, int n_vol, const float *wedges, double *poses, double *scores,
                  long *eval_count) {
    fft_tables();
    oref_t *r = (oref_t *)refp;
    const int N = cfg->box;
    if (!r || r->N != N || !box_ok(N) || r->pad != 1) return -22;
    const double rband = sva_band_radius(cfg);
    if (rband > r->B) return -22;
    const size_t n3 = (size_t)N * N * N;
    const double bf = cfg->band_factor == 0 ? 3.0 : cfg->band_factor;
    double rm_px = 0; for (int k = 0; k < 3; k++) if (cfg->window[k] > rm_px) rm_px = cfg->window[k];
    if (!(rm_px > 0)) rm_px = 0.4 * N;
    const double steptol = cfg->step_tolerance > 0 ? cfg->step_tolerance : 0.05;
    const double ha0 = 0.5 * cfg->tol_angle, hs0 = 0.5 * cfg->tol_shift;
    int T = cfg->max_iterations;
    if (T <= 0) { double m = ha0 > hs0 ? ha0 : hs0; T = m > steptol ? (int)ceil(log(m / steptol) / log(2.0)) : 1; if (T > 12) T = 12; if (T < 1) T = 1; }
    int en[6]; double tol[6];
    for (int k = 0; k < 3; k++) { en[k] = cfg->tol_angle > 0 && cfg->search_mode != 2; tol[k] = cfg->tol_angle; en[3 + k] = cfg->tol_shift > 0; tol[3 + k] = cfg->tol_shift; }
    const int global = cfg->search_mode == 1;
    const double gstep = cfg->global_step > 0 ? cfg->global_step : 15.0;
    int g_nth = 0, g_nps = 0;
    const int n_grid = global ? sva_grid_size(gstep, &g_nth, &g_nps) : 0;
    int K = cfg->n_candidates > 0 ? cfg->n_candidates : 25; if (K > 64) K = 64; if (K > n_grid) K = n_grid;
    long nev = 0;
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : nev)
    for (int v = 0; v < n_vol; v++) {
        const float *vol = volumes + (size_t)v * n3;
        cpx *f = (cpx *)malloc(n3 * sizeof(cpx));
        /* normalise to mean 0 / sigma 1, window, centre-origin transform scaled 1/N^(3/2)... (scale drops out of the score) */
        double s1 = 0, s2 = 0;
        for (size_t i = 0; i < n3; i++) { s1 += vol[i]; s2 += (double)vol[i] * vol[i]; }
        double mu = s1 / n3, var = s2 / n3 - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
        for (int z = 0; z < N; z++) for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) {
            size_t i = ((size_t)z * N + y) * N + x;
            double wv = 1.0;
            const int c[3] = { x - N / 2, y - N / 2, z - N / 2 };
            for (int k = 0; k < 3; k++) {
                if (!(cfg->window[k] > 0)) continue;
                double d = fabs((double)c[k]) - cfg->window[k];
                if (d > 0) wv *= cfg->window_sigma > 0 ? exp(-d * d / (2.0 * cfg->window_sigma * cfg->window_sigma)) : 0.0;
            }
            f[i].re = (float)((vol[i] - mu) / sd * wv); f[i].im = 0;    /* the origin moves to the box centre on the spectrum side below */
        }
        fft3_inplace(f, N, 0);
        /* in-band half-space sample list, shell by shell */
        const double lw = wedges ? wedges[2 * v] : -90.0, uw = wedges ? wedges[2 * v + 1] : 90.0;
        int R = (int)ceil(rband), cap = 0, ns = 0;
        for (int kz = -R; kz <= R; kz++) for (int ky = -R; ky <= R; ky++) for (int kx = 0; kx <= R; kx++) cap++;
        svs_t *sl = (svs_t *)malloc((size_t)cap * sizeof(svs_t)); cpx *F = (cpx *)malloc((size_t)cap * sizeof(cpx));
        for (int sh = 0; sh <= R; sh++)
            for (int kz = -R; kz <= R; kz++) for (int ky = -R; ky <= R; ky++) for (int kx = 0; kx <= R; kx++) {
                double k2 = (double)kx * kx + (double)ky * ky + (double)kz * kz;
                if (k2 == 0 || k2 >= rband * rband || (int)floor(sqrt(k2)) != sh) continue;
                if (kx == 0 && (ky < 0 || (ky == 0 && kz < 0))) continue;                 /* one of each Friedel pair on the kx = 0 plane */
                if (cfg->use_missing_wedge && !sva_in_wedge(kx, kz, lw, uw)) continue;
                double w = sva_band_weight(cfg, sqrt(k2) / N);
                if (w < 1e-3) continue;
                size_t i = ((size_t)((kz + N) % N) * N + ((ky + N) % N)) * N + kx;
                double sg = ((kx + ky + kz) & 1) ? -1.0 : 1.0;
                sl[ns].kx = kx; sl[ns].ky = ky; sl[ns].kz = kz; sl[ns].w = (float)w;
                F[ns].re = (float)(f[i].re * sg); F[ns].im = (float)(f[i].im * sg); ns++;
            }
        free(f);
        cunit_t s; memset(&s, 0, sizeof(s));
        memcpy(s.N, poses + (size_t)v * 12, 9 * sizeof(double)); memcpy(s.p, poses + (size_t)v * 12 + 9, 3 * sizeof(double));
        if (!global) {
            double ha = ha0, hs = hs0;
            sva_compass(r, sl, ns, F, N, rband, rm_px, bf, en, tol, &s, &ha, &hs, T, &nev);
        } else if (ns > 0) {
            /* rotations ranked by the translation-invariant AMPLITUDE correlation on the coarse band the grid step allows */
            geom_t g; memset(&g, 0, sizeof(g)); g.N = N;
            const int enr[5] = { 1, 0, 0, 0, 0 };
            const double rg = iter_band(&g, rm_px, bf, enr, 0.5 * gstep, 0.0, rband);
            double *gs = (double *)malloc((size_t)n_grid * sizeof(double)); char *used = (char *)calloc((size_t)n_grid, 1);
            for (int q = 0; q < n_grid; q++) {
                double G[9], Nq[9]; sva_grid_rotation(gstep, g_nth, g_nps, q, G); mat_mul3(s.N, G, Nq);
                gs[q] = sva_score_amp(r, sl, ns, F, rg, Nq); nev++;
            }
            /* top-K (ties -> lower grid index): two compass iterations each from the start shift (first steps Delta / 2 and half the
             * shift tolerance, bounds +-Delta about the grid rotation and +-tolerance about the start shift); the best of them at
             * the full band is refined again from Delta / 4 and a quarter of the shift tolerance down to the step tolerance */
            const double tolg[6] = { gstep, gstep, gstep, tol[3], tol[4], tol[5] };
            const int eng[6] = { 1, 1, 1, en[3], en[4], en[5] };
            cunit_t bestc; double bestf = -1e300; int have = 0;
            for (int a2 = 0; a2 < K; a2++) {
                int bq = -1;
                for (int q = 0; q < n_grid; q++) if (!used[q] && (bq < 0 || gs[q] > gs[bq])) bq = q;
                if (bq < 0) break;
                used[bq] = 1;
                cunit_t c = s; double G[9], Nq[9]; sva_grid_rotation(gstep, g_nth, g_nps, bq, G); mat_mul3(s.N, G, Nq); memcpy(c.N, Nq, sizeof(Nq));
                double ha = 0.5 * gstep, hs = 0.5 * cfg->tol_shift;
                sva_compass(r, sl, ns, F, N, rband, rm_px, bf, eng, tolg, &c, &ha, &hs, 2, &nev);
                const double fc = sva_score(r, sl, ns, F, N, rband, c.N, c.p); nev++;
                if (!have || fc > bestf) { bestc = c; bestf = fc; have = 1; }
            }
            free(gs); free(used);
            if (have) {
                s = bestc;
                double ha = 0.25 * gstep, hs = 0.25 * cfg->tol_shift;
                const double m = ha > hs ? ha : hs;
                int Tf = m > steptol ? (int)ceil(log(m / steptol) / log(2.0)) : 0; if (Tf > 12) Tf = 12;
                sva_compass(r, sl, ns, F, N, rband, rm_px, bf, eng, tolg, &s, &ha, &hs, Tf, &nev);
            }
        }
        memcpy(poses + (size_t)v * 12, s.N, 9 * sizeof(double)); memcpy(poses + (size_t)v * 12 + 9, s.p, 3 * sizeof(double));
        if (scores) { scores[v] = ns > 0 ? sva_score(r, sl, ns, F, N, rband, s.N, s.p) : 0.0; nev++; }
        free(sl); free(F);
    }
    if (eval_count) *eval_count = nev;
    return err;
}

/* ------------------------------------------------------------------ sub-tomogram average */
/* The averaging step of a 3DAVG iteration (the absent MPI_Classification writes `<dataset>_iteration_%03d_refined_selected_average_0.mrc`,
 * src/pyp/refine/tomo_avg/sub_tomo_avg.py:79-94, src/pyp_main.py:3076-3100): restated as the 3-D analogue of K7 - every sub-volume's
 * normalised transform, taken into the reference frame by its aligned pose (F_v(k) = Ref(N k) e^{+2 pi i k.p / N}, the convention of
 * orc_sva_align), is added with its missing-wedge mask as the weight (include/ppm.h, ppm_sva_insert).  Gathered per voxel q of the
 * accumulator's half space: k = N^T q, trilinear interpolation of F_v.  Build-defined, PARITY UNPINNED.
 * acc / counts: the layout of orc_insert_batch; index (may be NULL): parity -> half map. */
int orc_sva_insert(float *acc, long *counts, const ppm_sva_cfg *cfg, const float *volumes, int n_vol, const float *wedges,
                   const double *poses, const long *index) {
    fft_tables();
    const int N = cfg->box;
    if (!box_ok(N)) return -22;
    const size_t n3 = (size_t)N * N * N, NX = N / 2 + 1, half_sz = (size_t)N * N * NX * 3;
    const double rmax = N / 2 - 1;
    cpx *f = (cpx *)malloc(n3 * sizeof(cpx));
    if (!f) return -12;
    for (int v = 0; v < n_vol; v++) {
        const float *vol = volumes + (size_t)v * n3;
        double s1 = 0, s2 = 0;
        for (size_t i = 0; i < n3; i++) { s1 += vol[i]; s2 += (double)vol[i] * vol[i]; }
        const double mu = s1 / n3, var = s2 / n3 - mu * mu, sd = var > 0 ? sqrt(var) : 1.0;
        for (size_t i = 0; i < n3; i++) { f[i].re = (float)((vol[i] - mu) / sd); f[i].im = 0; }
        fft3_inplace(f, N, 0);
        const long key = index ? index[v] : (long)v;
        const int h = (int)(((key % 2) + 2) % 2);
        counts[h]++;
        float *A = acc + (size_t)h * half_sz;
        const double *Nm = poses + (size_t)v * 12, *p = Nm + 9;
        const double lw = wedges ? wedges[2 * v] : -90.0, uw = wedges ? wedges[2 * v + 1] : 90.0;
#pragma omp parallel for schedule(static)
        for (int qz = -N / 2; qz < N / 2; qz++) for (int qy = -N / 2; qy < N / 2; qy++) for (int qx = 0; qx <= N / 2; qx++) {
            const double q2 = (double)qx * qx + (double)qy * qy + (double)qz * qz;
            if (q2 == 0 || q2 >= rmax * rmax) continue;
            if (qx == 0 && (qy < 0 || (qy == 0 && qz < 0))) continue;          /* the fold of orc_finalize supplies the mates */
            /* k = N^T q */
            const double kx = Nm[0] * qx + Nm[3] * qy + Nm[6] * qz, ky = Nm[1] * qx + Nm[4] * qy + Nm[7] * qz, kz = Nm[2] * qx + Nm[5] * qy + Nm[8] * qz;
            if (cfg->use_missing_wedge) {
                double a = atan2(kz, kx) * 180.0 / ORC_PI;
                if (a > 90.0) a -= 180.0;
                if (a <= -90.0) a += 180.0;
                if (!(a >= lw && a <= uw)) continue;
            }
            const int x0 = (int)floor(kx), y0 = (int)floor(ky), z0 = (int)floor(kz);
            const double fx = kx - x0, fy = ky - y0, fz = kz - z0;
            double sr = 0, si = 0;
            for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++) {
                int x = x0 + dx, y = y0 + dy, z = z0 + dz, mate = 0;
                if (x < 0) { x = -x; y = -y; z = -z; mate = 1; }
                const cpx *t = &f[((size_t)((z + N) % N) * N + ((y + N) % N)) * N + x];
                const double sg = ((x + y + z) & 1) ? -1.0 : 1.0, wt = (dx ? fx : 1 - fx) * (dy ? fy : 1 - fy) * (dz ? fz : 1 - fz);
                sr += wt * sg * t->re; si += wt * sg * (mate ? -t->im : t->im);
            }
            const double ph = -2.0 * ORC_PI * (kx * p[0] + ky * p[1] + kz * p[2]) / N, cr = cos(ph), ci = sin(ph);
            float *o = A + (((size_t)(qz + N / 2) * N + (qy + N / 2)) * NX + qx) * 3;
            o[0] += (float)((sr * cr - si * ci) / N); o[1] += (float)((sr * ci + si * cr) / N); o[2] += 1.0f;
        }
    }
    free(f);
    return 0;
}
