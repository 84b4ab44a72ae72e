// ppm_lib.hip — C-ABI entry points of libpypmatch.so (include/ppm.h) for MI355X (gfx950).
// Host glue only: workspace management, launch sequencing on one HIP stream, event timing.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ppm.h"
#include "ppm_geom.h"
#include "ppm_kernels2.h"
#include "ppm_csp_kernels.h"
#include "ppm_sva_kernels.h"
#include "ppm_gfft.h"

using namespace ppm;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = "ERROR: " + msg; return code; }

#define HIPCHK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_err = std::string("ERROR: HIP: ") + hipGetErrorString(e_) + " at " #call;        \
            return -5;                                                                        \
        }                                                                                     \
    } while (0)
#define HIPCHKP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_err = std::string("ERROR: HIP: ") + hipGetErrorString(e_) + " at " #call;        \
            return nullptr;                                                                   \
        }                                                                                     \
    } while (0)

struct Ctx {
    bool inited = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy = nullptr;     // uploads of the next chunk's images overlap the current chunk's kernels
    hipStream_t upload = nullptr;   // ppm_device_upload (may be called from a helper thread of the caller)
    struct PlanDev { FftPlan plan; bool ready = false; };
    PlanDev plans[513];             // FFT plans by length (tables live in device memory)
    bool prof_on = false;
    double prof_ms[PPM_K_COUNT] = { 0 };
    long prof_n[PPM_K_COUNT] = { 0 };
    struct Pending { int id; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
} g;

// Streams are per HANDLE (ppm_reference / ppm_accum own a compute and a copy stream each): an entry point that takes a handle makes
// them the calling thread's current streams for its duration (StreamScope), everything below launches on cur_stream().  Calls
// on DIFFERENT handles may therefore run concurrently from different threads; process-wide state (FFT plan tables, the profiling
// event lists) is guarded by g_mu.  Entry points without a handle use the library's own pair of streams.
thread_local hipStream_t tl_stream = nullptr, tl_copy = nullptr;
std::mutex g_mu;
inline hipStream_t cur_stream() { return tl_stream ? tl_stream : g.stream; }
inline hipStream_t cur_copy() { return tl_copy ? tl_copy : g.copy; }
struct StreamScope {
    hipStream_t ps, pc;
    StreamScope(hipStream_t s_, hipStream_t c_) : ps(tl_stream), pc(tl_copy) { tl_stream = s_; tl_copy = c_; if (g.inited) (void)hipSetDevice(g.device); }
    ~StreamScope() { tl_stream = ps; tl_copy = pc; }
};

hipEvent_t ev_get() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.pool.empty()) { hipEvent_t e = g.pool.back(); g.pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void prof_flush() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &p : g.pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b);
        g.prof_ms[p.id] += ms; g.prof_n[p.id] += 1;
        g.pool.push_back(p.a); g.pool.push_back(p.b);
    }
    g.pending.clear();
}
struct ProfScope {
    int id; hipEvent_t a = nullptr;
    explicit ProfScope(int id_) : id(id_) { if (g.prof_on) { a = ev_get(); (void)hipEventRecord(a, cur_stream()); } }
    ~ProfScope() { if (a) { hipEvent_t b = ev_get(); (void)hipEventRecord(b, cur_stream()); std::lock_guard<std::mutex> lk(g_mu); g.pending.push_back({ id, a, b }); } }
};

// PPM_TRACE=1: wall-clock marks of a call on stderr (the device is synchronised at every mark, so the phases do not overlap when tracing)
struct Trace {
    const char *who; bool on; std::chrono::steady_clock::time_point t0;
    explicit Trace(const char *w) : who(w), on(getenv("PPM_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what) const {
        if (!on) return;
        const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        (void)hipStreamSynchronize(cur_stream());
        fprintf(stderr, "%s: %8.2f ms  (host reached this mark)\n", who, host_ms);
        fprintf(stderr, "%s: %8.2f ms  %s\n", who, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
    }
};

int ensure_plan(int n) {
    if (n < 2 || n > 512) return fail(-22, "FFT length out of range");
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.plans[n].ready) return 0;
    std::vector<int> fac; std::vector<unsigned short> perm;
    fft_factors(n, fac, perm);
    int prod = 1; for (int f : fac) prod *= f;
    if (prod != n || fac.size() > 12) return fail(-22, "FFT length must have prime factors 2, 3, 5 and 7 only");
    std::vector<float2> t(n);
    for (int k = 0; k < n; k++) t[k] = make_float2((float)std::cos(2.0 * kPi * k / n), (float)std::sin(2.0 * kPi * k / n));
    float2 *dtw = nullptr; unsigned short *dperm = nullptr;
    HIPCHK(hipMalloc(&dtw, sizeof(float2) * n));
    HIPCHK(hipMalloc(&dperm, sizeof(unsigned short) * n));
    HIPCHK(hipMemcpy(dtw, t.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dperm, perm.data(), sizeof(unsigned short) * n, hipMemcpyHostToDevice));
    FftPlan &p = g.plans[n].plan;
    p.n = n; p.nfac = (int)fac.size(); for (size_t i = 0; i < fac.size(); i++) p.fac[i] = fac[i];
    p.tw = dtw; p.perm = dperm;
    g.plans[n].ready = true;
    return 0;
}

// 3-D FFT of an n^3 complex array in place (three strided passes through LDS)
int fft3d(float2 *d, int n, bool inverse) {
    if (int rc = ensure_plan(n)) return rc;
    int L = std::max(1, std::min(16, 7600 / (n + 1)));
    while (((long)n * n) % L) L--;
    long nlines = (long)n * n;
    for (int pass = 0; pass < 3; pass++) {
        FftLinesP P;
        P.data = d; P.plan = g.plans[n].plan; P.n = n; P.inverse = inverse ? 1 : 0; P.L = L; P.nlines = nlines;
        if (pass == 0) { P.inner = nlines; P.inner_stride = n; P.outer_stride = 0; P.elem_stride = 1; P.line_major = 0; }
        else if (pass == 1) { P.inner = n; P.inner_stride = 1; P.outer_stride = (long)n * n; P.elem_stride = n; P.line_major = 1; }
        else { P.inner = nlines; P.inner_stride = 1; P.outer_stride = 0; P.elem_stride = (long)n * n; P.line_major = 1; }
        unsigned blocks = (unsigned)((nlines + L - 1) / L);
        hipLaunchKernelGGL(k_fft_lines, dim3(blocks), dim3(256), (size_t)L * (n + 1) * sizeof(float2), cur_stream(), P);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// one pass of length-n transforms over strided lines of `d` (see FftLinesP)
static int fft_lines_pass(float2 *d, int n, long nlines, long inner, long inner_stride, long outer_stride, long elem_stride, int line_major, bool inverse) {
    if (nlines <= 0) return 0;
    if (int rc = ensure_plan(n)) return rc;
    int L = std::max(1, std::min(16, 7600 / (n + 1)));
    while (nlines % L) L--;
    FftLinesP P;
    P.data = d; P.plan = g.plans[n].plan; P.n = n; P.inverse = inverse ? 1 : 0; P.L = L; P.nlines = nlines;
    P.inner = inner; P.inner_stride = inner_stride; P.outer_stride = outer_stride; P.elem_stride = elem_stride; P.line_major = line_major;
    hipLaunchKernelGGL(k_fft_lines, dim3((unsigned)((nlines + L - 1) / L)), dim3(256), (size_t)L * (n + 1) * sizeof(float2), cur_stream(), P);
    HIPCHK(hipGetLastError());
    return 0;
}

template <typename T>
struct DevBuf {
    T *p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        HIPCHK(hipMalloc(&p, n * sizeof(T)));
        cap = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// device temporary freed on every exit path (error returns included)
template <typename T>
struct DevTmp {
    T *p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp &) = delete;
    DevTmp &operator=(const DevTmp &) = delete;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n * sizeof(T)); }
};


}  // namespace

struct ppm_ref {
    hipStream_t stream = nullptr, copy = nullptr;       // this handle's compute and copy streams (StreamScope)
    int N = 0, B = 0, CX = 0, CY = 0, NBX = 0, NBY = 0, pad = 1; unsigned LB = 0;   // B, CX, CY count samples of the padded transform
    float2 *cube = nullptr;
    // workspaces (grown on demand, reused across calls)
    DevBuf<double> rows_in, rows_out, dir_theta, dir_phi;
    DevBuf<float> images, wring, cw, C2, nP, nI, cc, mats, ddef;
    // constrained search (ppm_csp_refine)
    DevBuf<float2> c_Il, c_band; DevBuf<float> c_cw, c_img, c_wring; DevBuf<double> c_rows, c_N, c_p, c_tl, c_delta, c_s0, c_g0, c_out;
    DevBuf<int> c_eval, c_rp, c_rt, c_slot, c_uoff, c_active; DevBuf<LState> c_states; DevBuf<double> c_mean, c_tmean, c_acc, c_dtrial, c_fpm, c_delta_t;
    // sub-tomogram alignment (ppm_sva_align): the transforms' work array, the band-limited transforms of a chunk, staged host volumes
    // (GBs: allocating and freeing them on every call cost ~20 ms of a 120 ms call)
    DevBuf<float2> s_f, s_g, s_F; DevBuf<float> s_vols;
    // the band's sample list (built and sorted on the host: ~25 ms at 192^3 / 452 k samples) is kept while the band-pass settings stay
    struct { bool valid = false; float key[5] = { 0, 0, 0, 0, 0 }; int S = 0; std::vector<int> shell_off; DevBuf<uint32_t> samples, pos; DevBuf<float> bandw; DevBuf<float2> Fw; bool fw_valid = false; float wkey[4] = { 0, 0, 0, 0 }; } s_plan;    // Fw: the window's transform at the samples
    DevBuf<float2> band, Il, Wp, bank, twN;
    DevBuf<float2> spill;            // k_prep outside the scratch-free path: the half spectrum between the row and the column phase, [n][N][W]
    DevBuf<float4> rowtw;            // k_global's row-pair twiddles for this reference's current search grid
    DevBuf<int> sh;
    DevBuf<uint32_t> samples;
    DevBuf<Hit> hits, hits_t;        // hits_t: per-tile top-K lists of a shift window wider than the kernel's
    DevBuf<int> tile_c;
    DevBuf<LState> states, states2;
    // full-window correlation (k_gfft): the bank in the column pass's layout, window maxima per (particle, orientation), column penalties
    DevBuf<float4> bank4; DevBuf<float> part, gtw;       // gtw: twiddle tables of the search grid (butterfly table, line table), then the window's column penalties
    std::string bank_key, bank4_key; int gtw_ns = 0, gtw_rsx = -1;
    long last_counts[4] = { 0, 0, 0, 0 };
    std::string note;
};

struct ppm_accum {
    hipStream_t stream = nullptr, copy = nullptr;       // this handle's compute and copy streams (StreamScope)
    int N = 0; float pixel = 1.f;
    float *acc = nullptr; bool external = false;
    std::vector<double> symops; int nsym = 1;
    float *d_sym = nullptr;
    unsigned long long *d_counts = nullptr;
    unsigned *d_max = nullptr;       // chunk maxima for the fixed-point scales of k_insert_bricks
    long counts[2] = { 0, 0 };
    DevBuf<double> rows; DevBuf<float> images, dose; DevBuf<float2> band, spill; DevBuf<PartIns> pp; DevBuf<CullEnt> cull; DevBuf<BrickItem> items;
    DevBuf<float2> s_f, s_g; DevBuf<float> s_vols;      // ppm_sva_insert: the transforms' work arrays and staged host volumes
    std::vector<float> brick_load; float load_r = -1.f; int n_items = 0, items_cap = -1;
};

// Work items of the brick insertion (k_insert_bricks): expected load of a brick = share of random slice planes that cut
// its box, estimated with a fixed set of normals; heavy bricks are cut into up to `cap` particle slices and the
// items are sorted heavy-first.  Bricks wholly outside the band carry no item.
static int build_brick_items(ppm_accum *a, const Geom &gm, int BE, int nb) {
    const int N = gm.N, nbx = (N / 2 + 1 + BE - 1) / BE, nby = (N + BE - 1) / BE;
    const float r = (float)gm.r_hi, hh = 0.5f * BE;
    if (a->load_r != r || a->brick_load.empty()) {
        const int NS = 192;
        std::vector<float> nrm(NS * 3);
        for (int i = 0; i < NS; i++) {          // Fibonacci sphere
            double z = 1.0 - 2.0 * (i + 0.5) / NS, ph = i * 2.399963229728653, rr = std::sqrt(std::max(0.0, 1.0 - z * z));
            nrm[i * 3] = (float)(rr * std::cos(ph)); nrm[i * 3 + 1] = (float)(rr * std::sin(ph)); nrm[i * 3 + 2] = (float)z;
        }
        a->brick_load.assign((size_t)nbx * nby * nby, -1.f);
        for (int bz = 0; bz < nby; bz++) for (int by = 0; by < nby; by++) for (int bx = 0; bx < nbx; bx++) {
            const int x_lo = bx * BE, y_lo = by * BE - N / 2, z_lo = bz * BE - N / 2;
            const float dx = std::max(std::max((float)x_lo, -(float)(x_lo + BE)), 0.f), dy = std::max(std::max((float)y_lo, -(float)(y_lo + BE)), 0.f),
                        dz = std::max(std::max((float)z_lo, -(float)(z_lo + BE)), 0.f);
            if (dx * dx + dy * dy + dz * dz >= r * r) continue;
            const float cx = x_lo + hh, cy = y_lo + hh, cz = z_lo + hh;
            int cut = 0;
            for (int i = 0; i < NS; i++) {
                const float *n = &nrm[i * 3];
                if (std::fabs(n[0] * cx + n[1] * cy + n[2] * cz) <= (std::fabs(n[0]) + std::fabs(n[1]) + std::fabs(n[2])) * hh) cut++;
            }
            a->brick_load[((size_t)bz * nby + by) * nbx + bx] = 0.02f + (float)cut / NS;
        }
        a->load_r = r; a->items_cap = -1;
    }
    const int smax_env = getenv("PPM_BRICK_SLICES") ? atoi(getenv("PPM_BRICK_SLICES")) : 16;
    const int minp_env = getenv("PPM_BRICK_MINP") ? atoi(getenv("PPM_BRICK_MINP")) : 1024;
    const int cap = std::max(1, std::min(smax_env, nb / std::max(1, minp_env)));
    if (cap * 1000 + smax_env == a->items_cap) return 0;
    struct Tmp { BrickItem it; float load; };
    std::vector<Tmp> v;
    for (int bz = 0; bz < nby; bz++) for (int by = 0; by < nby; by++) for (int bx = 0; bx < nbx; bx++) {
        const float L = a->brick_load[((size_t)bz * nby + by) * nbx + bx];
        if (L < 0.f) continue;
        const int S = std::max(1, std::min(cap, (int)std::lround(L * smax_env)));
        for (int sl = 0; sl < S; sl++) {
            Tmp t; t.it.bx = (unsigned short)bx; t.it.by = (unsigned short)by; t.it.bz = (unsigned short)bz; t.it.s = (unsigned char)sl; t.it.S = (unsigned char)S;
            t.load = L / S; v.push_back(t);
        }
    }
    std::stable_sort(v.begin(), v.end(), [](const Tmp &x, const Tmp &y) { return x.load > y.load; });
    std::vector<BrickItem> items(v.size());
    for (size_t i = 0; i < v.size(); i++) items[i] = v[i].it;
    if (int rc = a->items.ensure(items.size())) return rc;
    HIPCHK(hipMemcpyAsync(a->items.p, items.data(), items.size() * sizeof(BrickItem), hipMemcpyHostToDevice, cur_stream()));
    HIPCHK(hipStreamSynchronize(cur_stream()));
    a->n_items = (int)items.size(); a->items_cap = cap * 1000 + smax_env;
    return 0;
}

// ------------------------------------------------------------------------------ pre-processing launch
static int launch_prep(DevBuf<float2> &spill /* the calling handle's scratch */, const float *d_images, const double *d_rows, int n_img, const Geom &gm, float Rm_px, float fall_px,
                       int normalize, int invert, int do_mask, int whiten, float2 *band, float *wring,
                       const uint32_t *samples, int S_pad, float2 *Il, float *cw, float2 *Wp, float *C2, float *nI,
                       unsigned *band_max = nullptr /* insertion: receives the chunk's largest |band| component */,
                       const float *focus_px = nullptr /* focus mask: sphere centre and radius in pixels, or null */) {
    if (int rc = ensure_plan(gm.N)) return rc;
    PrepP P;
    P.images = d_images; P.rows = d_rows; P.plan = g.plans[gm.N].plan;
    P.N = gm.N; P.B = gm.B; P.W = gm.W; P.H = gm.H;
    P.r_hi2 = (float)(gm.r_hi * gm.r_hi); P.Rm = Rm_px; P.wfall = fall_px; P.a = (float)gm.a;
    P.normalize = normalize; P.invert = invert; P.do_mask = do_mask; P.whiten = whiten;
    for (int k = 0; k < 4; k++) P.focus[k] = focus_px ? focus_px[k] : 0.f;
    // LDS plan: L row pairs per row pass (L N <= 8 x threads: the next pass is prefetched into <= 8 register pairs per
    // thread; L divides N/2) share their storage with the nc columns of one column chunk; the whole half spectrum goes
    // through a global scratch between the two phases.  512 threads / 80 KB -> two blocks per CU.
    // Block shape (A/B on one box, 100 k x 256^2 insertion workload, us per particle): 512 threads / 80 KB (233 VGPRs: ONE block per
    // CU resident) 0.456; 512 threads held to 128 VGPRs for two blocks 0.646 (spills); 1024 threads / 160 KB 0.69; 256 threads /
    // 52 KB at 233 VGPRs (two blocks) 0.399; 256 threads / 40 KB held to 168 VGPRs (three blocks, 252 B of scratch) 0.376 <- default.
    // The FFT stages are barrier-bound: several small independent blocks overlap each other's barrier waits.
    int PT = 256;
    if (const char *e = getenv("PPM_PREP_PT")) { const int v = atoi(e); if (v == 512 || v == 1024 || v == 256) PT = v; }
    // scratch-free path (N = 256): one 512-thread block per CU keeps the half spectrum in registers between the row and the column phase
    // (A/B on one box, 100 k x 256^2: reconstruction 0.38 -> 0.29 us per particle, refinement 0.44 -> 0.37; PPM_PREP_INREG=0 selects the scratch path)
    const bool inreg = gm.N == 256 && !getenv("PPM_PREP_GENERIC") && !getenv("PPM_PREP_PT") && !(getenv("PPM_PREP_INREG") && atoi(getenv("PPM_PREP_INREG")) == 0);
    if (inreg) PT = 512;
    // the same path with 1024 threads (16 waves per CU, 32 held values per thread) instead of 512 (8 waves, 64 values): PPM_PREP_INREG_PT=1024
    if (inreg && getenv("PPM_PREP_INREG_PT") && atoi(getenv("PPM_PREP_INREG_PT")) == 1024) PT = 1024;
    const int occ3 = !(getenv("PPM_PREP_OCC") && atoi(getenv("PPM_PREP_OCC")) == 2);
    const size_t budget = (PT == 1024 ? 160 : (PT == 256 ? (getenv("PPM_PREP_LDS") ? atoi(getenv("PPM_PREP_LDS")) : (occ3 ? 40 : 52)) : 80)) * 1024;
    const size_t lds_fixed = (size_t)(gm.B + 2) * 16 + 16 + 5 * (PT / 64) * sizeof(double) + (12 + PT / 64) * sizeof(float) + (size_t)gm.N * 12 + 16;
    P.fast256 = (gm.N == 256 && !getenv("PPM_PREP_GENERIC")) ? 1 : 0;
    P.inreg = inreg ? 1 : 0;
    P.TS = P.fast256 ? 273 : gm.N + 1; P.WS = P.fast256 ? 272 : gm.N;
    P.L = std::max(1, std::min(8 * PT / gm.N, gm.N / 2));
    if (getenv("PPM_PREP_L")) P.L = std::max(1, std::min(atoi(getenv("PPM_PREP_L")), 8 * PT / gm.N));
    while ((gm.N / 2) % P.L || (size_t)P.L * P.WS * sizeof(float2) + lds_fixed + P.TS * sizeof(float2) > budget / 2 + 8192) P.L--;     // the row pass walks the image 2 L rows at a time; leave about half of the LDS to the column chunk
    if (P.L < 1) return fail(-12, "pre-processing kernel: row buffer does not fit the LDS");
    {
        const size_t wk = (size_t)P.L * P.WS * sizeof(float2);
        const size_t left = budget - lds_fixed > wk ? budget - lds_fixed - wk : 0;
        P.nc = std::max(1, std::min(gm.W, (int)(left / (P.TS * sizeof(float2)))));
        P.nc = std::max(1, std::min(P.nc, 12 * PT / gm.N));        // k_prep prefetches one chunk into 12 registers pairs per thread
    }
    P.nchunks = (gm.W + P.nc - 1) / P.nc;
    P.nc = (gm.W + P.nchunks - 1) / P.nchunks;       // even chunks
    if (getenv("PPM_PREP_NCH")) { P.nchunks = std::max(P.nchunks, atoi(getenv("PPM_PREP_NCH"))); P.nc = (gm.W + P.nchunks - 1) / P.nchunks; P.nchunks = (gm.W + P.nc - 1) / P.nc; }
    if (!inreg) if (int rc = spill.ensure((size_t)n_img * gm.N * gm.W)) return rc;
    P.spill = spill.p;
    P.band_max = band_max;
    P.band = band; P.wring = wring; P.samples = samples; P.S_pad = S_pad; P.Il = Il; P.cw = cw;
    P.Wp = Wp; P.C2 = C2; P.nI = nI; P.Bs = gm.Bs; P.Hs = gm.Hs;
    P.r_s2 = (float)(gm.r_s * gm.r_s); P.r_lo2 = (float)(gm.r_lo * gm.r_lo);
    size_t lds = ((size_t)P.nc * P.TS + (size_t)P.L * P.WS) * sizeof(float2) + lds_fixed;
    if (inreg) {            // T (64 columns) and the row buffer (64 row pairs) share one 140 KB region
        P.L = 64; P.nc = 64; P.nchunks = (std::min(gm.W, 128) + 63) / 64;
        lds = (size_t)64 * P.TS * sizeof(float2) + lds_fixed;
        if (lds > (size_t)160 * 1024) return fail(-12, "pre-processing kernel: LDS plan exceeds 160 KB");
    } else
    if (lds > budget) return fail(-12, "pre-processing kernel: LDS plan exceeds its budget");
    static bool attr_set = false;
    std::unique_lock<std::mutex> lk_attr(g_mu);
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void *)k_prep<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_prep<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_prep<1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_prep<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_prep<256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_set = true;
    }
    lk_attr.unlock();
    const bool two_blocks = getenv("PPM_PREP_OCC") && atoi(getenv("PPM_PREP_OCC")) == 4;
    ProfScope ps(PPM_K_PREP);
    if (PT == 512 && two_blocks && !inreg) hipLaunchKernelGGL((k_prep<512, 4>), dim3(n_img), dim3(512), lds, cur_stream(), P);
    else if (PT == 512) hipLaunchKernelGGL((k_prep<512, 2>), dim3(n_img), dim3(512), lds, cur_stream(), P);
    else if (PT == 256 && occ3) hipLaunchKernelGGL((k_prep<256, 3>), dim3(n_img), dim3(256), lds, cur_stream(), P);
    else if (PT == 256) hipLaunchKernelGGL((k_prep<256, 2>), dim3(n_img), dim3(256), lds, cur_stream(), P);
    else hipLaunchKernelGGL((k_prep<1024, 1>), dim3(n_img), dim3(1024), lds, cur_stream(), P);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int R, bool HALF, bool TWO>
static int launch_global_k(const GlobP &P, int n_img, size_t lds) {
    static bool set = false;
    { std::lock_guard<std::mutex> lk_attr(g_mu); if (!set) { HIPCHK(hipFuncSetAttribute((const void *)k_global<R, HALF, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } }
    hipLaunchKernelGGL((k_global<R, HALF, TWO>), dim3((n_img + global_particles(R) - 1) / global_particles(R)), dim3(global_threads(R)), lds, cur_stream(), P);
    HIPCHK(hipGetLastError());
    return 0;
}
template <int R>
static int launch_global_r(const GlobP &P, int n_img, bool half, size_t lds) {
    // search bands of at most 32 pixels: two slices per wave (k_global<.., TWO>); PPM_GLOBAL_TWO=0 keeps one
    const bool two = P.Bs <= 31 && !(getenv("PPM_GLOBAL_TWO") && atoi(getenv("PPM_GLOBAL_TWO")) == 0);
    if (two) return half ? launch_global_k<R, true, true>(P, n_img, lds) : launch_global_k<R, false, true>(P, n_img, lds);
    return half ? launch_global_k<R, true, false>(P, n_img, lds) : launch_global_k<R, false, false>(P, n_img, lds);
}

static int launch_global(GlobP &P, int n_img, bool half, int R) {
    size_t lds = (size_t)P.HsP * 64 * sizeof(float2) * global_particles(R);
    P.n = n_img;
    if (lds < 1024) lds = 1024;
    // the top-K pass re-uses the block's LDS for a copy of the particle's n_orient scores when they fit (160 KB = 40 928
    // orientations, e.g. 8 deg at C1); finer grids select on the global scratch instead
    const size_t lds_topk = (size_t)(32 + P.n_orient) * sizeof(float);
    P.topk_lds = lds_topk <= (size_t)160 * 1024 ? 1 : 0;
    if (P.topk_lds && lds_topk > lds) lds = lds_topk;
    ProfScope ps(PPM_K_GLOBAL);
    switch (R) {
        case 1: return launch_global_r<1>(P, n_img, half, lds);
        case 2: return launch_global_r<2>(P, n_img, half, lds);
        case 3: return launch_global_r<3>(P, n_img, half, lds);
        case 4: return launch_global_r<4>(P, n_img, half, lds);
        case 5: return launch_global_r<5>(P, n_img, half, lds);
        default: return launch_global_r<6>(P, n_img, half, lds);      // wider windows: k_gfft, or tiles of this one (ppm_refine_batch)
    }
}

// Full-window correlation (ppm_gfft.h): LDS plan and launch.  Returns -1 when the search grid is outside what the kernel is built
// for (Ns = 16 .. 128), in which case the caller keeps the tiled k_global.
struct GfftPlan { int LN = 0, L = 0, RC = 0, nchunk = 1; size_t t_bytes = 0, lds = 0; int topk_lds = 0; };
static bool gfft_plan(const Geom &gm, GfftPlan &pl) {
    int LN = 0; while ((1 << LN) < gm.Ns) LN++;
    if ((1 << LN) != gm.Ns || LN < 4 || LN > 7) return false;
    pl.LN = LN; pl.L = gm.Ns / 2;
    const int L = pl.L, G = gfft_slices_per_pass(L), NR = 2 * gm.RSy + 1;
    const size_t row = (size_t)G * 2 * gfft_row_stride(L) * sizeof(float2), fixed = (size_t)L * L * sizeof(float4) + gfft_small_bytes(L);
    const size_t room = (size_t)160 * 1024 - fixed;
    int RC = NR;
    if (const char *e = getenv("PPM_GFFT_ROWS")) { const int v = atoi(e); if (v > 0 && v < RC) RC = v; }       // tests: force several row chunks
    if ((size_t)RC * row > room) RC = (int)(room / row);
    if (RC > 2 * L) RC = 2 * L;
    pl.RC = RC; pl.nchunk = (NR + RC - 1) / RC;
    pl.RC = (NR + pl.nchunk - 1) / pl.nchunk;        // even chunks
    pl.t_bytes = std::max((size_t)pl.RC * row, (size_t)G * L * L * sizeof(float4));        // T doubles as the staging area of the bank slice(s) of a pass
    const size_t topk = (size_t)gm.n_orient * sizeof(float);
    pl.topk_lds = 0;
    if (topk <= pl.t_bytes) pl.topk_lds = 1;
    else if (fixed + topk <= (size_t)160 * 1024) { pl.topk_lds = 1; pl.t_bytes = (topk + 15) & ~(size_t)15; }
    pl.lds = fixed + pl.t_bytes;
    return true;
}
template <int LN, bool CHUNKED>
static int launch_gfft_k(const GfftP &P, int n_img, size_t lds) {
    static bool set = false;
    { std::lock_guard<std::mutex> lk_attr(g_mu); if (!set) { HIPCHK(hipFuncSetAttribute((const void *)k_gfft<LN, CHUNKED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } }
    hipLaunchKernelGGL((k_gfft<LN, CHUNKED>), dim3(n_img), dim3(256), lds, cur_stream(), P);
    HIPCHK(hipGetLastError());
#ifdef PPM_GFFT_STAMPS
    {   // diagnostic build: cycles per phase and wave of blocks 0 .. 3 (ppm_gfft.h)
        float st[4 * 4 * 12];
        HIPCHK(hipStreamSynchronize(cur_stream()));
        HIPCHK(hipMemcpy(st, P.cc, sizeof(st), hipMemcpyDeviceToHost));
        const char *names[12] = { "products", "fft", "stores", "barrier A", "row tail", "barrier B", "twiddles", "loop", "row reads + pairs", "row fft", "row max", "-" };
        const int nsl = P.n_dir * P.npsi_store;
        for (int b = 0; b < 4 && b < n_img; b++) for (int w = 0; w < 4; w++) {
            fprintf(stderr, "k_gfft stamps block %d wave %d (cycles per slice):", b, w);
            for (int i = 0; i < 11; i++) fprintf(stderr, " | %s %.0f", names[i], st[(b * 4 + w) * 12 + i] / nsl);
            fprintf(stderr, "\n");
        }
    }
#endif
    return 0;
}
static int launch_gfft(GfftP &P, int n_img, const GfftPlan &pl) {
    P.n = n_img; P.RC = pl.RC; P.nchunk = pl.nchunk; P.t_bytes = (int)pl.t_bytes; P.topk_lds = pl.topk_lds;
    ProfScope ps(PPM_K_GLOBAL);
    if (pl.nchunk > 1) {
        switch (pl.LN) {
            case 4: return launch_gfft_k<4, true>(P, n_img, pl.lds);
            case 5: return launch_gfft_k<5, true>(P, n_img, pl.lds);
            case 6: return launch_gfft_k<6, true>(P, n_img, pl.lds);
            default: return launch_gfft_k<7, true>(P, n_img, pl.lds);
        }
    }
    switch (pl.LN) {
        case 4: return launch_gfft_k<4, false>(P, n_img, pl.lds);
        case 5: return launch_gfft_k<5, false>(P, n_img, pl.lds);
        case 6: return launch_gfft_k<6, false>(P, n_img, pl.lds);
        default: return launch_gfft_k<7, false>(P, n_img, pl.lds);
    }
}

static __global__ void k_noop() {}

extern "C" {

const char *ppm_last_error(void) { return g_err.c_str(); }
const char *ppm_version(void) { return "pypmatch 0.1 (gfx950)"; }
const char *ppm_build_id(void) { return "pypmatch " __DATE__ " " __TIME__; }
int ppm_device_mem_info(size_t *free_bytes, size_t *total_bytes) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    (void)hipSetDevice(g.device);
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

int ppm_init(int device) {
    static std::mutex init_mu;                       // a caller may start the device from a helper thread and call again from its main thread
    std::lock_guard<std::mutex> lk(init_mu);
    const auto t_init0 = std::chrono::steady_clock::now();
    if (g.inited && g.device == device) { (void)hipSetDevice(device); return 0; }      // the current device is a per-thread setting
    if (g.inited) return fail(-16, "libpypmatch is bound to device " + std::to_string(g.device) + " in this process (one process per GPU); start another process for device " + std::to_string(device));
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(-19, "no HIP device visible; libpypmatch has no CPU path");
    if (device < 0 || device >= count) return fail(-22, "device index out of range");
    // how a host thread waits for the device: PPM_SYNC=block sleeps on an interrupt instead of spinning (the drop-in executables
    // set it: their reader threads need the cores a spinning wait would burn); default = the runtime's own choice
    if (const char *e = getenv("PPM_SYNC")) {
        const std::string v(e);
        (void)hipSetDeviceFlags(v == "block" ? hipDeviceScheduleBlockingSync : (v == "yield" ? hipDeviceScheduleYield : (v == "spin" ? hipDeviceScheduleSpin : hipDeviceScheduleAuto)));
    }
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(-19, std::string("device is ") + prop.gcnArchName + ", libpypmatch is built for gfx950 only");
    if (!g.stream) HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    if (!g.copy) HIPCHK(hipStreamCreateWithFlags(&g.copy, hipStreamNonBlocking));
    if (!g.upload) HIPCHK(hipStreamCreateWithFlags(&g.upload, hipStreamNonBlocking));
    const auto t_ctx = std::chrono::steady_clock::now();
    // the code object is loaded at the first launch (tens of ms): here, where a caller can overlap it with its own start-up
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, g.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    g.device = device; g.inited = true;
    if (getenv("PPM_TRACE"))
        fprintf(stderr, "ppm_init: context + streams %.1f ms, code object + first launch %.1f ms\n",
                std::chrono::duration<double, std::milli>(t_ctx - t_init0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ctx).count());
    return 0;
}

void ppm_profile_enable(int on) { g.prof_on = on != 0; }
void ppm_profile_reset(void) { prof_flush(); for (int i = 0; i < PPM_K_COUNT; i++) { g.prof_ms[i] = 0; g.prof_n[i] = 0; } }
int ppm_profile_get(int id, double *ms, long *n) {
    if (id < 0 || id >= PPM_K_COUNT) return fail(-22, "bad kernel id");
    prof_flush();
    if (ms) *ms = g.prof_ms[id];
    if (n) *n = g.prof_n[id];
    return 0;
}

// (the current device is a per-thread setting of the runtime: helper threads of the caller get the library's device here)
void *ppm_device_alloc(size_t bytes) { if (g.inited) (void)hipSetDevice(g.device); void *p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) { g_err = "ERROR: device allocation failed"; return nullptr; } return p; }
void ppm_device_free(void *p) { if (p) (void)hipFree(p); }
// own stream: a helper thread of the caller may upload the next chunk while another thread's library call computes (and uses
// cur_copy() for its internal double buffering); returns when the copy has completed
int ppm_device_upload(void *dst, const void *src, size_t bytes) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    HIPCHK(hipSetDevice(g.device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g.upload));
    HIPCHK(hipStreamSynchronize(g.upload));
    return 0;
}
void *ppm_host_alloc(size_t bytes) { if (g.inited) (void)hipSetDevice(g.device); void *p = nullptr; if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { g_err = "ERROR: pinned host allocation failed"; return nullptr; } return p; }
void ppm_host_free(void *p) { if (p) (void)hipHostFree(p); }
// ---- file reads for the executables' reader stage: a persistent pool, one pread loop per part
extern "C++" {
namespace {
struct ReadPool {
    std::mutex mu;                      // one ppm_host_read at a time
    std::mutex qmu;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    struct Part { int fd; long long off; char *dst; size_t bytes; };
    std::vector<Part> parts;
    size_t next = 0, pending = 0;
    int err = 0;
    bool quit = false;
    void worker() {
        std::unique_lock<std::mutex> lk(qmu);
        for (;;) {
            wake.wait(lk, [&] { return quit || next < parts.size(); });
            if (quit) return;
            Part p = parts[next++];
            lk.unlock();
            int e = 0;
            size_t got = 0;
            while (got < p.bytes) {
                ssize_t r = pread(p.fd, p.dst + got, std::min(p.bytes - got, (size_t)64 << 20), p.off + (long long)got);
                if (r < 0) { if (errno == EINTR) continue; e = -errno; break; }
                if (r == 0) { e = -5; break; }
                got += (size_t)r;
            }
            lk.lock();
            if (e && !err) err = e;
            if (--pending == 0) done.notify_all();
        }
    }
    int run(int fd, long long off, char *dst, size_t bytes, int nt) {
        std::lock_guard<std::mutex> one(mu);
        nt = std::max(1, std::min(nt, 16));
        std::unique_lock<std::mutex> lk(qmu);
        while ((int)threads.size() < nt) threads.emplace_back([this] { worker(); });
        // parts of whole MB so that every pread starts on a page boundary of the destination
        const size_t per = std::max((size_t)1 << 20, ((bytes + nt - 1) / nt + ((size_t)1 << 20) - 1) >> 20 << 20);
        parts.clear(); next = 0; err = 0;
        for (size_t a = 0; a < bytes; a += per) parts.push_back({fd, off + (long long)a, dst + a, std::min(per, bytes - a)});
        pending = parts.size();
        if (!pending) return 0;
        wake.notify_all();
        done.wait(lk, [&] { return pending == 0; });
        parts.clear(); next = 0;
        return err;
    }
    ~ReadPool() {
        { std::lock_guard<std::mutex> lk(qmu); quit = true; }
        wake.notify_all();
        for (auto &t : threads) t.join();
    }
};
ReadPool &read_pool() { static ReadPool *p = new ReadPool(); return *p; }      // leaked on purpose: no joins at process exit
}
}
int ppm_host_read(int fd, long long offset, void *dst, size_t bytes, int n_threads) {
    if (fd < 0 || offset < 0 || (!dst && bytes)) return fail(-22, "ppm_host_read: bad argument");
    int e = read_pool().run(fd, offset, (char *)dst, bytes, n_threads);
    if (e == -5) return fail(-5, "short read from the particle stack");
    if (e) return fail(e, std::string("reading the particle stack failed: ") + strerror(-e));
    return 0;
}

int ppm_device_sync(void) { if (cur_stream()) HIPCHK(hipStreamSynchronize(cur_stream())); HIPCHK(hipDeviceSynchronize()); return 0; }

// ------------------------------------------------------------------------------ reference
ppm_ref_t *ppm_reference_create_weighted(const float *vol, int n, float max_band_px, int pad, const float *ring_weight, int n_weight) {
    if (!g.inited) { fail(-1, "ppm_init has not been called"); return nullptr; }
    if (!vol || !box_ok(n) || !(max_band_px > 0)) { fail(-22, "reference box must be even, 32..512, with prime factors 2, 3, 5, 7, and the band positive"); return nullptr; }
    if ((pad != 1 && pad != 2 && pad != 4) || n * pad > 512) { fail(-22, "padding factor must be 1, 2 or 4 with padded box <= 512"); return nullptr; }
    if (max_band_px > n / 2) max_band_px = (float)(n / 2);
    const int np = n * pad;
    int B = (int)std::ceil((double)max_band_px * pad) - 1;
    if (B > np / 2 - 1) B = np / 2 - 1;
    size_t n3 = (size_t)n * n * n, np3 = (size_t)np * np * np;
    std::unique_ptr<ppm_ref, void (*)(ppm_ref_t *)> guard(new ppm_ref(), ppm_reference_destroy);     // freed on every error return
    ppm_ref *r = guard.get();
    HIPCHKP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    HIPCHKP(hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking));
    StreamScope ss_(r->stream, r->copy);          // the preparation runs on the new handle's own stream: references may be made concurrently
    DevTmp<float> t_vol, t_w; DevTmp<float2> t_f;
    HIPCHKP(t_vol.alloc(n3));
    HIPCHKP(t_f.alloc(np3));
    float *d_vol = t_vol.p; float2 *d_f = t_f.p;
    HIPCHKP(hipMemcpy(d_vol, vol, n3 * sizeof(float), hipMemcpyHostToDevice));
    if (pad > 1) HIPCHKP(hipMemsetAsync(d_f, 0, np3 * sizeof(float2), cur_stream()));
    float *d_w = nullptr;
    if (ring_weight && n_weight > 0) {
        HIPCHKP(t_w.alloc((size_t)n_weight));
        d_w = t_w.p;
        HIPCHKP(hipMemcpyAsync(d_w, ring_weight, (size_t)n_weight * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
    }
    r->N = n; r->pad = pad; r->B = B; r->CX = B + 2; r->CY = 2 * B + 3;
    size_t cube_n = (size_t)r->CX * r->CY * r->CY;
    r->NBX = (r->CX + 3) / 4; r->NBY = (r->CY + 1) / 2;
    const size_t copy_n = (size_t)r->NBX * r->NBY * r->NBY * 16;         // blocked layout, two copies (ppm_dev.h)
    if (2 * copy_n * sizeof(float2) >= ((size_t)1 << 32)) {        // byte offsets of the buffer loads are 32-bit
    fail(-22, "reference cube too large"); return nullptr; }
    r->LB = (unsigned)copy_n;
    if (hipMalloc(&r->cube, 2 * copy_n * sizeof(float2)) != hipSuccess) { r->cube = nullptr; fail(-12, "out of device memory for the reference cube"); return nullptr; }
    HIPCHKP(hipMemsetAsync(r->cube, 0, 2 * copy_n * sizeof(float2), cur_stream()));
    {
        ProfScope ps(PPM_K_BANK);
        hipLaunchKernelGGL(k_ref_load, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, cur_stream(), d_vol, d_f, n, np);
        if (fft3d(d_f, np, false)) return nullptr;
        hipLaunchKernelGGL(k_ref_crop, dim3((unsigned)((cube_n + 255) / 256)), dim3(256), 0, cur_stream(), d_f, r->cube, np, n, B, r->CX, r->CY, r->NBX, r->NBY, r->LB, d_w, n_weight);
    }
    if (hipStreamSynchronize(cur_stream()) != hipSuccess || hipGetLastError() != hipSuccess) { fail(-5, "reference preparation failed on the device"); return nullptr; }
    return guard.release();
}

ppm_ref_t *ppm_reference_create_padded(const float *vol, int n, float max_band_px, int pad) { return ppm_reference_create_weighted(vol, n, max_band_px, pad, nullptr, 0); }
ppm_ref_t *ppm_reference_create(const float *vol, int n, float max_band_px) { return ppm_reference_create_weighted(vol, n, max_band_px, 1, nullptr, 0); }

void ppm_reference_destroy(ppm_ref_t *r) {
    if (!r) return;
    if (r->cube) (void)hipFree(r->cube);
    r->rows_in.release(); r->rows_out.release(); r->dir_theta.release(); r->dir_phi.release();
    r->images.release(); r->wring.release(); r->cw.release(); r->C2.release(); r->nP.release(); r->nI.release();
    r->s_f.release(); r->s_g.release(); r->s_F.release(); r->s_vols.release(); r->s_plan.samples.release(); r->s_plan.pos.release(); r->s_plan.bandw.release(); r->s_plan.Fw.release();
    r->c_Il.release(); r->c_band.release(); r->c_cw.release(); r->c_img.release(); r->c_wring.release(); r->c_rows.release(); r->c_N.release(); r->c_p.release(); r->c_tl.release();
    r->c_delta.release(); r->c_s0.release(); r->c_g0.release(); r->c_out.release(); r->c_eval.release(); r->c_rp.release(); r->c_rt.release(); r->c_slot.release(); r->c_states.release(); r->c_uoff.release(); r->c_mean.release(); r->c_active.release(); r->c_tmean.release(); r->c_acc.release(); r->c_dtrial.release(); r->c_fpm.release(); r->c_delta_t.release(); r->cc.release(); r->mats.release(); r->ddef.release();
    r->band.release(); r->spill.release(); r->Il.release(); r->Wp.release(); r->bank.release(); r->twN.release(); r->rowtw.release(); r->sh.release(); r->samples.release();
    r->hits.release(); r->states.release(); r->states2.release();
    r->hits_t.release(); r->tile_c.release(); r->bank4.release(); r->part.release(); r->gtw.release();
    if (r->stream) (void)hipStreamDestroy(r->stream);
    if (r->copy) (void)hipStreamDestroy(r->copy);
    delete r;
}

// 2-D FFTs of `nimg` complex n x n images in place (rows, then columns)
static int fft2d_batch(float2 *d, int n, long nimg, bool inverse) {
    if (int rc = ensure_plan(n)) return rc;
    int L = std::max(1, std::min(16, 7600 / (n + 1)));
    const long nlines = nimg * n;
    while (nlines % L) L--;
    for (int pass = 0; pass < 2; pass++) {
        FftLinesP P;
        P.data = d; P.plan = g.plans[n].plan; P.n = n; P.inverse = inverse ? 1 : 0; P.L = L; P.nlines = nlines;
        if (pass == 0) { P.inner = nlines; P.inner_stride = n; P.outer_stride = 0; P.elem_stride = 1; P.line_major = 0; }
        else { P.inner = n; P.inner_stride = 1; P.outer_stride = (long)n * n; P.elem_stride = n; P.line_major = 1; }
        hipLaunchKernelGGL(k_fft_lines, dim3((unsigned)((nlines + L - 1) / L)), dim3(256), (size_t)L * (n + 1) * sizeof(float2), cur_stream(), P);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------ matching projections
int ppm_match_projections(ppm_ref_t *ref, const ppm_refine_cfg *cfg, const double *rows, int n_rows, float *out) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!ref || !cfg || !rows || !out) return fail(-22, "null argument");
    StreamScope ss_(ref->stream, ref->copy);
    if (n_rows <= 0) return 0;
    ppm_refine_cfg c2 = *cfg; c2.global_search = 0;          // only box, pixel size and the high-resolution limit matter here
    Geom gm; std::string err;
    if (!geom_init(gm, c2, err)) return fail(-22, err);
    if (gm.N != ref->N) return fail(-22, "particle box differs from the reference box");
    if (gm.B > (ref->B + 1) / ref->pad - 1) return fail(-22, "high-resolution limit exceeds the band the reference was prepared for");
    const size_t NN = (size_t)gm.N * gm.N;
    const int CH = (int)std::min<size_t>((size_t)n_rows, std::max<size_t>(1, ((size_t)1 << 30) / (NN * 12)));
    DevTmp<float2> d_f; DevTmp<float> d_o; DevTmp<MatchRow> d_rows;
    HIPCHK(d_f.alloc(NN * CH)); HIPCHK(d_o.alloc(NN * CH)); HIPCHK(d_rows.alloc(CH));
    MatchP MP;
    MP.cv.cube = ref->cube; MP.cv.NBX = ref->NBX; MP.cv.NBY = ref->NBY; MP.cv.LB = ref->LB; MP.cv.off = ref->B + 1; MP.cv.scale = (float)ref->pad;
    MP.rows = d_rows.p; MP.f = d_f.p; MP.N = gm.N; MP.B = gm.B; MP.r_hi2 = (float)(gm.r_hi * gm.r_hi);
    std::vector<MatchRow> hr(CH);
    const float scale = (cfg->invert ? -1.f : 1.f) / (float)gm.N;     // cube = FFT / N: the unnormalised inverse transform needs 1 / N more
    for (int c0 = 0; c0 < n_rows; c0 += CH) {
        const int nb = std::min(CH, n_rows - c0);
        for (int i = 0; i < nb; i++) {
            const double *row = rows + (size_t)(c0 + i) * PPM_NCOL;
            double M[9]; euler_matrix(row[PPM_PSI], row[PPM_THETA], row[PPM_PHI], M);
            MatchRow &q = hr[i];
            q.m[0] = (float)M[0]; q.m[1] = (float)M[1]; q.m[2] = (float)M[3]; q.m[3] = (float)M[4]; q.m[4] = (float)M[6]; q.m[5] = (float)M[7];
            q.sx = (float)(row[PPM_XSHIFT] / gm.a); q.sy = (float)(row[PPM_YSHIFT] / gm.a);
            q.ctf = ctf_from_row(row, gm.N, gm.a);
        }
        HIPCHK(hipMemcpyAsync(d_rows.p, hr.data(), (size_t)nb * sizeof(MatchRow), hipMemcpyHostToDevice, cur_stream()));
        MP.n = nb;
        hipLaunchKernelGGL(k_match_fill, dim3((unsigned)((NN * nb + 255) / 256)), dim3(256), 0, cur_stream(), MP);
        if (int rc = fft2d_batch(d_f.p, gm.N, nb, true)) return rc;
        hipLaunchKernelGGL(k_match_real, dim3((unsigned)((NN * nb + 255) / 256)), dim3(256), 0, cur_stream(), d_f.p, d_o.p, NN * nb, scale);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out + (size_t)c0 * NN, d_o.p, NN * nb * sizeof(float), hipMemcpyDeviceToHost, cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_stream()));            // `hr` is reused by the next chunk
    }
    return 0;
}

// ------------------------------------------------------------------------------ refine
int ppm_refine_batch(ppm_ref_t *ref, const ppm_refine_cfg *cfg, const void *images, int images_on_device,
                     int n_img, const double *rows_in, double *rows_out) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!ref || !cfg || !images || !rows_in || !rows_out) return fail(-22, "null argument");
    StreamScope ss_(ref->stream, ref->copy);
    if (n_img <= 0) return 0;
    Geom gm; std::string err;
    if (!geom_init(gm, *cfg, err)) return fail(-22, err);
    if (gm.N != ref->N) return fail(-22, "particle box differs from the reference box");
    ref->note.clear();
    if (gm.r_s_asked > gm.r_s) {
        char b[256];
        std::snprintf(b, sizeof(b), "NOTE: global search band lowered from %.1f to %.1f Fourier pixels (%.2f A instead of %.2f A): the grid-search "
                      "kernel covers 64 pixels; the top hits are refined up to the high-resolution limit as asked", gm.r_s_asked, gm.r_s,
                      gm.N * gm.a / gm.r_s, gm.N * gm.a / gm.r_s_asked);
        ref->note = b;
    }
    if (cfg->global_search && gm.range_capped) {
        char b[320];
        std::snprintf(b, sizeof(b), "%sNOTE: shift window of the grid search: +-%.0f x +-%.0f pixels (%d x %d search-grid steps of %.1f pixels, the most the "
                      "search grid of %d points holds; asked: %.0f pixels); the refinement of the hits is not limited to it", ref->note.empty() ? "" : "\n",
                      gm.RSx * gm.step, gm.RSy * gm.step, gm.RSx, gm.RSy, gm.step, gm.Ns, gm.range_asked_px);
        ref->note += b;
    }
    if (gm.B > (ref->B + 1) / ref->pad - 1) return fail(-22, "high-resolution limit exceeds the band the reference was prepared for");
    if (cfg->global_search && gm.Bs + 1 > 64)
        return fail(-22, "global search band wider than 64 Fourier pixels is not supported; lower the 'resolution limit for search'");
    if (!cfg->global_search && !cfg->local_refine) { /* score only */ }
    int K = cfg->top_hits > 0 ? cfg->top_hits : 20;
    if (K > PPM_MAX_TOP_HITS) K = PPM_MAX_TOP_HITS;
    if (K > gm.n_orient) K = gm.n_orient;
    // answers 36 / 37 (ppm.h): a global search always refines its top hits (Tb iterations each); answer 37 decides whether the
    // best hit continues at the full band (Tc iterations).  iters_hit < 0: hits stay at their grid points (test hook).
    const int Tb = cfg->iters_hit > 0 ? cfg->iters_hit : (cfg->iters_hit < 0 ? 0 : 2), Tc = cfg->iters_final > 0 ? cfg->iters_final : 7;
    const double fall = cfg->mask_falloff > 0 ? cfg->mask_falloff : 20.0;
    const float fall_px = (float)(fall / gm.a), Rm_px = (float)(cfg->mask_radius / gm.a);
    const bool focus_on = cfg->focus[3] > 0.f;     // a focus mask replaces the centred masks of both stages
    const float focus_px[4] = { (float)(cfg->focus[0] / gm.a), (float)(cfg->focus[1] / gm.a), (float)(cfg->focus[2] / gm.a), (float)(cfg->focus[3] / gm.a) };
    const bool sep_search = !focus_on && cfg->global_search && cfg->search_mask_radius > 0 && cfg->search_mask_radius != cfg->mask_radius;

    SampleList sl; build_samples(gm, sl);
    const int S_pad = (int)sl.packed.size();
    const int nrings = gm.B + 2;
    int ring_s = (int)std::ceil(gm.r_s); if (ring_s > gm.B + 1) ring_s = gm.B + 1;
    (void)ring_s;
    if (int rc = ref->samples.ensure(S_pad)) return rc;
    HIPCHK(hipMemcpyAsync(ref->samples.p, sl.packed.data(), S_pad * sizeof(uint32_t), hipMemcpyHostToDevice, cur_stream()));

    const size_t NN = (size_t)gm.N * gm.N, HW = (size_t)gm.H * gm.W, HS = (size_t)gm.Hs * 64;
    // shift window: the kernel searches +-PPM_MAX_SHIFT_STEPS steps; a wider window is covered by overlapping tiles of that
    // half-width whose union is exactly [-RS, RS] (centres cxs / cys, in steps)
    // k_global keeps its shift window in registers: up to kTileR steps either side without scratch.  Anything wider — PYP's default
    // "search range 0 = mask radius" is +-41 steps at a 256 box and 4 A — goes to the full-window transform (k_gfft, ppm_gfft.h);
    // PPM_GLOBAL_PATH=tiles keeps the tiled k_global (A/B runs and the tests that hold one path against the other), =fft forces
    // the transform for narrow windows too.
    constexpr int kTileR = 6;
    GfftPlan gpl;
    bool use_fft = false;
    if (cfg->global_search && gfft_plan(gm, gpl)) {
        const char *gp = getenv("PPM_GLOBAL_PATH");
        const bool force_fft = gp && !strcmp(gp, "fft"), force_tiles = gp && !strcmp(gp, "tiles");
        use_fft = force_fft || (!force_tiles && std::max(gm.RSx, gm.RSy) > kTileR);
    }
    const int Rtx = std::min(gm.RSx, kTileR), Rty = std::min(gm.RSy, kTileR);
    auto tile_centres = [](int RS, int Rt) {
        const int T = (2 * RS + 1 + 2 * Rt) / (2 * Rt + 1);
        std::vector<int> c(T, 0);
        for (int i = 0; i < T && T > 1; i++) c[i] = -RS + Rt + (int)(((long)i * 2 * (RS - Rt)) / (T - 1));
        return c;
    };
    const std::vector<int> cxs = tile_centres(gm.RSx, Rtx), cys = tile_centres(gm.RSy, Rty);
    const int ntiles = (int)(cxs.size() * cys.size());
    const int Rwin = std::max(Rtx, Rty);
    // bank rows per slice in the paired order of k_global: row 0 = ky 0, row 1 = empty, rows 2t / 2t+1 = ky +t / -t
    const int HsP = ((2 * (gm.Bs + 1) + 2 * global_unroll(Rwin) - 1) / (2 * global_unroll(Rwin))) * (2 * global_unroll(Rwin));   // k_global walks 2 U rows per trip
    const size_t HSP = (size_t)HsP * 64;
    const int nslices = gm.n_dir * gm.npsi_store;
    // chunk so that the scratch stays well inside HBM
    size_t per = NN * 4 + HW * 8 + (size_t)S_pad * 12 + 2 * PPM_NCOL * 8 + (gm.B + 2) * 4;
    if (cfg->global_search) per += HS * 12 + (size_t)nslices * 4 + (size_t)gm.n_orient * 8 + (size_t)K * (sizeof(Hit) + sizeof(LState)) + sizeof(LState);
    int CH = (int)std::min<size_t>((size_t)n_img, std::max<size_t>(64, ((size_t)4 << 30) / per));
    CH = std::min(CH, 8192);
    if (CH >= 2048) CH &= ~1023;        // whole rounds of blocks: 256 CUs x 1 (k_global) and x 4 (k_local, one block per particle)
    if (const char *e = std::getenv("PPM_CHUNK")) { int v = std::atoi(e); if (v > 0) CH = std::min(CH, v); }   // tests: force several chunks

    if (int rc = ref->rows_in.ensure((size_t)CH * PPM_NCOL)) return rc;
    if (int rc = ref->rows_out.ensure((size_t)CH * PPM_NCOL)) return rc;
    if (!images_on_device) if (int rc = ref->images.ensure((size_t)2 * CH * NN)) return rc;     // double-buffered staging
    if (int rc = ref->band.ensure((size_t)CH * HW)) return rc;
    if (int rc = ref->wring.ensure((size_t)CH * (gm.B + 2))) return rc;
    if (int rc = ref->Il.ensure((size_t)CH * S_pad)) return rc;
    if (int rc = ref->cw.ensure((size_t)CH * S_pad)) return rc;
    if (int rc = ref->states2.ensure(CH)) return rc;
    CubeView cv; cv.cube = ref->cube; cv.NBX = ref->NBX; cv.NBY = ref->NBY; cv.LB = ref->LB; cv.off = ref->B + 1; cv.scale = (float)ref->pad;

    if (cfg->global_search) {
        if (int rc = ref->Wp.ensure((size_t)CH * HS)) return rc;
        if (int rc = ref->C2.ensure((size_t)CH * HS)) return rc;
        if (int rc = ref->nP.ensure((size_t)CH * nslices)) return rc;
        if (int rc = ref->nI.ensure(CH)) return rc;
        if (int rc = ref->cc.ensure((size_t)CH * gm.n_orient)) return rc;
        if (int rc = ref->sh.ensure((size_t)CH * gm.n_orient)) return rc;
        if (int rc = ref->hits.ensure((size_t)CH * K)) return rc;
        if (int rc = ref->states.ensure((size_t)CH * K)) return rc;
        // slice bank, twiddles and direction tables: rebuilt only when the grid / band changes
        char key[160];
        std::snprintf(key, sizeof(key), "%d/%.6f/%.6f/%d/%d/%d/%.3f/%.3f", gm.N, gm.r_s, gm.dstep, gm.Ns, gm.npsi_store, HsP, gm.phi_max, gm.theta_max);
        if (ref->bank_key != key) {
            std::vector<float> mats((size_t)nslices * 6);
            std::vector<double> dth(gm.n_dir), dph(gm.n_dir);
            for (int d = 0; d < gm.n_dir; d++) {
                grid_direction(gm, d, dth[d], dph[d]);
                for (int k = 0; k < gm.npsi_store; k++) {
                    double M[9]; euler_matrix(k * gm.dpsi, dth[d], dph[d], M);
                    float *m = &mats[((size_t)d * gm.npsi_store + k) * 6];
                    m[0] = (float)M[0]; m[1] = (float)M[1]; m[2] = (float)M[3]; m[3] = (float)M[4]; m[4] = (float)M[6]; m[5] = (float)M[7];
                }
            }
            std::vector<float2> tw(gm.Ns);
            for (int t = 0; t < gm.Ns; t++) tw[t] = make_float2((float)std::cos(2.0 * kPi * t / gm.Ns), (float)std::sin(2.0 * kPi * t / gm.Ns));
            if (int rc = ref->mats.ensure(mats.size())) return rc;
            if (int rc = ref->dir_theta.ensure(gm.n_dir)) return rc;
            if (int rc = ref->dir_phi.ensure(gm.n_dir)) return rc;
            if (int rc = ref->twN.ensure(gm.Ns)) return rc;
            if (int rc = ref->bank.ensure((size_t)nslices * HSP)) return rc;
            HIPCHK(hipMemcpyAsync(ref->mats.p, mats.data(), mats.size() * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
            HIPCHK(hipMemcpyAsync(ref->dir_theta.p, dth.data(), dth.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
            HIPCHK(hipMemcpyAsync(ref->dir_phi.p, dph.data(), dph.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
            HIPCHK(hipMemcpyAsync(ref->twN.p, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice, cur_stream()));
            // row twiddles of the shift window (scalar loads in k_global)
            {
                std::vector<float4> rt((size_t)kRowTwRows * PPM_MAX_SHIFT_STEPS, make_float4(1.f, 1.f, 0.f, 0.f));
                for (int tp = 0; tp <= gm.Bs && tp < kRowTwRows; tp++) for (int j = 1; j <= PPM_MAX_SHIFT_STEPS; j++) {
                    int t = ((tp * j) % gm.Ns + gm.Ns) % gm.Ns;
                    const float c = (float)std::cos(2.0 * kPi * t / gm.Ns), sn = (float)std::sin(2.0 * kPi * t / gm.Ns);
                    rt[(size_t)tp * PPM_MAX_SHIFT_STEPS + j - 1] = make_float4(c, c, sn, sn);
                }
                if (int rc = ref->rowtw.ensure(rt.size())) return rc;
                HIPCHK(hipMemcpyAsync(ref->rowtw.p, rt.data(), rt.size() * sizeof(float4), hipMemcpyHostToDevice, cur_stream()));
                HIPCHK(hipStreamSynchronize(cur_stream()));
            }
            BankP BP; BP.cv = cv; BP.mats = ref->mats.p; BP.bank = ref->bank.p; BP.nslices = nslices; BP.Bs = gm.Bs; BP.Hs = HsP;
            BP.r_s2 = (float)(gm.r_s * gm.r_s);
            {
                ProfScope ps(PPM_K_BANK);
                size_t tot = (size_t)nslices * HSP;
                hipLaunchKernelGGL(k_bank, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, cur_stream(), BP);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(cur_stream()));   // host vectors go out of scope
            ref->bank_key = key;
        }
    }
    if (use_fft) {
        const int L = gpl.L;
        if (int rc = ref->part.ensure((size_t)CH * gm.n_orient * 2)) return rc;
        char key[200];
        std::snprintf(key, sizeof(key), "%s/L%d", ref->bank_key.c_str(), L);
        if (ref->bank4_key != key) {
            if ((size_t)nslices * L * L * sizeof(float4) >= ((size_t)1 << 32)) return fail(-22, "slice bank of the grid search exceeds 4 GB: use a coarser angular step or a narrower search band");
            if (int rc = ref->bank4.ensure((size_t)nslices * L * L)) return rc;
            Bank4P BP; BP.cv = cv; BP.mats = ref->mats.p; BP.bank4 = ref->bank4.p; BP.nslices = nslices; BP.Bs = gm.Bs; BP.L = L; BP.r_s2 = (float)(gm.r_s * gm.r_s);
            ProfScope ps(PPM_K_BANK);
            const size_t tot = (size_t)nslices * L * L;
            hipLaunchKernelGGL(k_bank4, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, cur_stream(), BP);
            HIPCHK(hipGetLastError());
            ref->bank4_key = key;
        }
        if (ref->gtw_ns != gm.Ns || ref->gtw_rsx != gm.RSx) {
            // twiddle tables of the in-register transforms (ppm_fft_reg.h), (cos, sin) pairs: the butterfly table of the L-point transform
            // (8 floats per entry: w^k, w^2k, w^3k, padding), then the line table w^0 .. w^(L-1) of the Ns-point grid
            const int nb = fr::bfly_entries(L);
            std::vector<float> tw((size_t)fr::tw_table_floats(L) + gm.Ns, 0.f);
            auto put = [&](float *d, double ang) { d[0] = (float)std::cos(ang); d[1] = (float)std::sin(ang); };
            for (int M = L; M >= 8; M /= 4)
                for (int k = 1; k < M / 4; k++)
                    for (int j = 1; j <= 3; j++) put(&tw[(size_t)fr::bfly_entry(L, M, k) * 8 + (j - 1) * 2], 2.0 * kPi * j * k / M);
            for (int t = 0; t < L; t++) put(&tw[(size_t)nb * 8 + (size_t)t * 2], 2.0 * kPi * t / gm.Ns);
            // column penalties of the row pass, in the order the L-point transform leaves its outputs: position p holds the columns
            // j = 2 f, 2 f + 1 (f = freq_at(L, p)), column j is the shift sx = j (j < L) or j - Ns
            for (int pp = 0; pp < L; pp++)
                for (int h = 0; h < 2; h++) {
                    const int j = 2 * fr::freq_at(L, pp) + h, sx = j < L ? j : j - gm.Ns;
                    tw[(size_t)fr::tw_table_floats(L) + 2 * pp + h] = std::abs(sx) <= gm.RSx ? 0.f : -3.0e38f;
                }
            if (int rc = ref->gtw.ensure(tw.size())) return rc;
            HIPCHK(hipMemcpyAsync(ref->gtw.p, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
            HIPCHK(hipStreamSynchronize(cur_stream()));       // the host vector goes out of scope
            ref->gtw_ns = gm.Ns; ref->gtw_rsx = gm.RSx;
        }
    }
    HIPCHK(hipStreamSynchronize(cur_stream()));

    // frequency marching: band of a compass iteration from its probe displacement (same rule as the oracle's iter_band)
    const double bf = cfg->band_factor == 0 ? 3.0 : cfg->band_factor, rm_px = cfg->mask_radius / gm.a;
    const bool any_ang = cfg->refine_psi || cfg->refine_theta || cfg->refine_phi, any_sh = cfg->refine_x || cfg->refine_y;
    auto iter_band = [&](double ha, double hs, double rcap) {
        if (bf < 0) return rcap;
        double d = 0;
        if (any_ang) d = rm_px * ha * kPi / 180.0;
        if (any_sh && hs > d) d = hs;
        if (!(d > 0)) return rcap;
        double rit = bf * gm.N / (2.0 * kPi * d);
        if (rit < 4.0) rit = 4.0;
        return rit < rcap ? rit : rcap;
    };
    auto prefix_of = [&](double rband) { int rg = (int)std::ceil(rband); if (rg > gm.B + 1) rg = gm.B + 1; return sl.ring_off[rg]; };
    double sample_evals = 0;   // in-band samples summed over all local score evaluations of one particle
    int ndef = 0;              // defocus offsets tried on either side of the row's values
    if (cfg->refine_defocus && cfg->defocus_step > 0 && cfg->defocus_range >= cfg->defocus_step)
        ndef = std::min((int)std::floor(cfg->defocus_range / cfg->defocus_step + 1e-6), PPM_MAX_DEFOCUS_STEPS);
    LocalP LP;
    LP.cv = cv; LP.samples = ref->samples.p; LP.Il = ref->Il.p; LP.cw = ref->cw.p; LP.S_pad = S_pad; LP.nrings = nrings; LP.N = gm.N;
    LP.tabR = cube_tab_radius(gm.B, cv.scale);
    // tap addresses from LDS tables (ppm_dev.h) unless the tables would crowd the ring sums out of a CU (PPM_LOCAL_TABLES=0: arithmetic)
    const bool local_tab = !(getenv("PPM_LOCAL_TABLES") && atoi(getenv("PPM_LOCAL_TABLES")) == 0) && cube_tab_bytes(LP.tabR) <= 16 * 1024;
    const int final_threads = (getenv("PPM_LOCAL_FINAL_THREADS") && atoi(getenv("PPM_LOCAL_FINAL_THREADS")) == 128) ? 128 : 256;      // A/B knob
    auto launch_local = [&](unsigned grid, int threads) {
        const size_t ring = ring_lds_bytes8(threads / 64, kMaxCand, LP.nr);
        const bool tab = local_tab && ring + cube_tab_bytes(LP.tabR) + 2048 <= (size_t)64 * 1024;      // with the kernel's static LDS inside the 64 KB a launch may ask for (box 512 at the full band: arithmetic)
        size_t lds = ring + (tab ? cube_tab_bytes(LP.tabR) : 0);
        if (const char *e = getenv("PPM_LOCAL_BLOCKS_PER_CU")) {      // A/B knob: fewer blocks per CU through a larger LDS request (scripts/ab_local.sh)
            const int bpc = atoi(e);
            if (bpc > 0 && bpc < 8) lds = std::min((size_t)63 * 1024, std::max(lds, (size_t)(160 * 1024 / (bpc + 1) + 1024) & ~(size_t)1023));
        }
        if (tab) hipLaunchKernelGGL(k_local<true>, dim3(grid), dim3(threads), lds, cur_stream(), LP);
        else hipLaunchKernelGGL(k_local<false>, dim3(grid), dim3(threads), lds, cur_stream(), LP);
    };
    LP.rlo2 = (float)(gm.r_lo * gm.r_lo); LP.ring_signed = (float)std::min(gm.ring_signed, 1e30);
    LP.en[0] = cfg->refine_psi; LP.en[1] = cfg->refine_theta; LP.en[2] = cfg->refine_phi; LP.en[3] = cfg->refine_x; LP.en[4] = cfg->refine_y;
    LP.use_priors = 0;
    for (int i = 0; i < 5; i++) { LP.pmean[i] = 0; LP.pw[i] = 0; }
    if (cfg->use_priors) {          // Gaussian restraint on the refined parameters (include/ppm.h; same numbers as the oracle's prior_init)
        const double ns = kPi * (gm.r_hi * gm.r_hi - gm.r_lo * gm.r_lo);
        for (int i = 0; i < 5; i++) {
            double var = cfg->prior_var[i], mean = cfg->prior_mean[i];
            if (i >= 3) { mean /= gm.a; var /= gm.a * gm.a; }
            LP.pmean[i] = mean;
            if (LP.en[i] && var > 0 && ns > 0) { LP.pw[i] = 1.0 / (2.0 * var * ns); LP.use_priors = 1; }
        }
    }

    const int nfree = (cfg->refine_psi != 0) + (cfg->refine_theta != 0) + (cfg->refine_phi != 0) + (cfg->refine_x != 0) + (cfg->refine_y != 0);
    const int per_iter = nfree ? 2 * nfree + 2 : 0;     // centre + 2 per free parameter + trial
    if (Tb + Tc > kMaxIters) return fail(-22, "too many compass iterations requested");
    auto fill_schedule = [&](double ha, double hs, int t0, int T, double rcap, double mult) {
        for (int t = 0; t < T; t++) {
            double rb = iter_band(ha, hs, rcap);
            LP.rmax2_it[t] = (float)(rb * rb); LP.S_it[t] = prefix_of(rb);
            sample_evals += mult * per_iter * std::floor(kPi * rb * rb / 2);
            ha *= 0.5; hs *= 0.5;
        }
        (void)t0;
    };
    LP.rmax2_final = (float)(gm.r_hi * gm.r_hi); LP.S_final = S_pad;
    // answer 22: LOGP / SIGMA over r_lo .. r_cls; without a defocus refinement the final k_local launch scores it, with one k_defocus does
    const bool cls_on = gm.r_cls < gm.r_hi;
    LP.rmax2_class = (float)(gm.r_cls * gm.r_cls); LP.S_class = (cls_on && ndef == 0) ? prefix_of(gm.r_cls) : 0;
    if (!images_on_device) {        // first chunk's images
        HIPCHK(hipMemcpyAsync(ref->images.p, images, (size_t)std::min(CH, n_img) * NN * sizeof(float), hipMemcpyHostToDevice, cur_copy()));
        HIPCHK(hipStreamSynchronize(cur_copy()));
    }
    for (int c0 = 0, ci = 0; c0 < n_img; c0 += CH, ci++) {
        const int nb = std::min(CH, n_img - c0);
        HIPCHK(hipMemcpyAsync(ref->rows_in.p, rows_in + (size_t)c0 * PPM_NCOL, (size_t)nb * PPM_NCOL * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
        const float *d_img = images_on_device ? (const float *)images + (size_t)c0 * NN : ref->images.p + (size_t)(ci & 1) * CH * NN;
        // refinement spectra (+ search tables when the same mask serves both)
        if (int rc = launch_prep(ref->spill, d_img, ref->rows_in.p, nb, gm, Rm_px, fall_px, cfg->normalize, cfg->invert, 1, 1, ref->band.p, ref->wring.p,
                                 ref->samples.p, S_pad, ref->Il.p, ref->cw.p,
                                 (cfg->global_search && !sep_search) ? ref->Wp.p : nullptr, ref->C2.p, ref->nI.p, nullptr, focus_on ? focus_px : nullptr)) return rc;
        if (sep_search)
            if (int rc = launch_prep(ref->spill, d_img, ref->rows_in.p, nb, gm, (float)(cfg->search_mask_radius / gm.a), fall_px, cfg->normalize, cfg->invert, 1, 1,
                                     ref->band.p, nullptr, nullptr, 0, nullptr, nullptr, ref->Wp.p, ref->C2.p, ref->nI.p)) return rc;
        LState *final_states = ref->states2.p;
        if (cfg->global_search) {
            GlobP GP;
            GP.bank = ref->bank.p; GP.Wp = ref->Wp.p; GP.nP = ref->nP.p; GP.nI = ref->nI.p; GP.twN = ref->twN.p; GP.rowtw = ref->rowtw.p;
            GP.cc = ref->cc.p; GP.sh = ref->sh.p; GP.hits = ref->hits.p;
            GP.Bs = gm.Bs; GP.Hs = gm.Hs; GP.HsP = HsP; GP.Ns = gm.Ns; GP.RSx = Rtx; GP.RSy = Rty;
            GP.n_dir = gm.n_dir; GP.n_psi = gm.n_psi; GP.npsi_store = gm.npsi_store; GP.n_orient = gm.n_orient; GP.K = K;
            {
                ProfScope ps(PPM_K_NORMS);
                NormP NP; NP.C2 = ref->C2.p; NP.bank = ref->bank.p; NP.nP = ref->nP.p; NP.n = nb; NP.nslices = nslices; NP.Bs = gm.Bs; NP.Hs = gm.Hs; NP.HsP = HsP;
                hipLaunchKernelGGL(k_slice_norms, dim3((nb + 127) / 128, (nslices + 127) / 128), dim3(256), 0, cur_stream(), NP);
            }
            if (use_fft) {
                GfftP FP;
                FP.bank4 = ref->bank4.p; FP.bank4_bytes = (unsigned)((size_t)nslices * gpl.L * gpl.L * sizeof(float4)); FP.Wp = ref->Wp.p; FP.nP = ref->nP.p; FP.nI = ref->nI.p; FP.tw = ref->gtw.p;
                FP.part = ref->part.p; FP.cc = ref->cc.p; FP.hits = ref->hits.p;
                FP.Bs = gm.Bs; FP.Hs = gm.Hs; FP.RSx = gm.RSx; FP.RSy = gm.RSy;
                FP.n_dir = gm.n_dir; FP.n_psi = gm.n_psi; FP.npsi_store = gm.npsi_store; FP.n_orient = gm.n_orient; FP.K = K;
                if (int rc = launch_gfft(FP, nb, gpl)) return rc;
            } else if (ntiles == 1) {
                if (int rc = launch_global(GP, nb, gm.half != 0, Rwin)) return rc;
            } else {
                // tiles of the shift window: ramp the search tables to the tile's centre, search, keep the tile's top-K; then merge
                if (int rc = ref->hits_t.ensure((size_t)ntiles * nb * K)) return rc;
                if (int rc = ref->tile_c.ensure((size_t)2 * ntiles)) return rc;
                std::vector<int> tc(2 * ntiles);
                for (int ty = 0, t = 0; ty < (int)cys.size(); ty++) for (int tx = 0; tx < (int)cxs.size(); tx++, t++) { tc[t] = cxs[tx]; tc[ntiles + t] = cys[ty]; }
                HIPCHK(hipMemcpyAsync(ref->tile_c.p, tc.data(), tc.size() * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
                int px = 0, py = 0;
                const size_t tot = (size_t)nb * HS;
                for (int t = 0; t < ntiles; t++) {
                    const int dcx = tc[t] - px, dcy = tc[ntiles + t] - py;
                    if (dcx || dcy) hipLaunchKernelGGL(k_wp_ramp, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, cur_stream(), ref->Wp.p, tot, gm.Bs, gm.Ns, dcx, dcy, ref->twN.p);
                    px = tc[t]; py = tc[ntiles + t];
                    GP.hits = ref->hits_t.p + (size_t)t * nb * K;
                    if (int rc = launch_global(GP, nb, gm.half != 0, Rwin)) return rc;
                }
                GP.hits = ref->hits.p;
                hipLaunchKernelGGL(k_merge_hits, dim3((nb + 127) / 128), dim3(128), 0, cur_stream(), ref->hits_t.p, ref->hits.p, nb, K, ntiles, ref->tile_c.p, ref->tile_c.p + ntiles);
                HIPCHK(hipStreamSynchronize(cur_stream()));      // the host vector of the centres goes out of scope
            }
            {
                ProfScope ps(PPM_K_TOPK);
                hipLaunchKernelGGL(k_states_from_hits, dim3((nb * K + 255) / 256), dim3(256), 0, cur_stream(), ref->hits.p, ref->states.p, nb, K,
                                   ref->dir_theta.p, ref->dir_phi.p, gm.n_psi, gm.dpsi, gm.step, 0.5 * gm.dstep, gm.step);
            }
            sample_evals = 0;
            if (Tb > 0) {
                LP.states = ref->states.p; LP.T = Tb; LP.final_rescore = 0;
                LP.nr = std::min(nrings, (int)std::ceil(gm.r_s) + 1);
                fill_schedule(0.5 * gm.dstep, gm.step, 0, Tb, gm.r_s, (double)K);
                ProfScope ps(PPM_K_LOCAL);
                // small blocks for the hit stage: one wave up to 1 024 samples per sweep (no cross-wave steps, 64-sample granularity: k_local
                // 83.9 -> 80.9 ms per 28 672 particles against two waves, 93.2 with four; scripts/ab_hit_threads.sh), two waves above; 256 threads below
                const int hit_threads = getenv("PPM_LOCAL_HIT_THREADS") ? atoi(getenv("PPM_LOCAL_HIT_THREADS")) : (LP.S_it[0] <= 1024 ? 64 : 128);
                launch_local((unsigned)(nb * K), hit_threads == 64 || hit_threads == 256 ? hit_threads : 128);
            }
            {
                ProfScope ps(PPM_K_TOPK);
                hipLaunchKernelGGL(k_select_best, dim3((nb + 255) / 256), dim3(256), 0, cur_stream(), ref->states.p, ref->states2.p, nb, K);
            }
            {
                LP.states = ref->states2.p; LP.T = cfg->local_refine ? Tc : 0; LP.final_rescore = 1; LP.nr = nrings;
                fill_schedule(0.5 * gm.dstep / (double)(1 << Tb), gm.step / (double)(1 << Tb), Tb, LP.T, gm.r_hi, 1.0);
                sample_evals += std::floor(kPi * gm.r_hi * gm.r_hi / 2);
                ProfScope ps(PPM_K_LOCAL);
                launch_local((unsigned)nb, final_threads);
            }
        } else {
            double ha0 = cfg->local_angle_step > 0 ? cfg->local_angle_step : 2.5, hs0 = cfg->local_shift_step > 0 ? cfg->local_shift_step : 2.0;
            hipLaunchKernelGGL(k_states_from_rows, dim3((nb + 255) / 256), dim3(256), 0, cur_stream(), ref->rows_in.p, ref->states2.p, nb, gm.a, ha0, hs0);
            sample_evals = 0;
            LP.states = ref->states2.p; LP.T = cfg->local_refine ? Tb + Tc : 0; LP.final_rescore = 1; LP.nr = nrings;
            fill_schedule(ha0, hs0, 0, LP.T, gm.r_hi, 1.0);
            sample_evals += std::floor(kPi * gm.r_hi * gm.r_hi / 2);
            ProfScope ps(PPM_K_LOCAL);
            launch_local((unsigned)nb, final_threads);
        }
        const float *d_ddef = nullptr;
        if (ndef > 0) {                                 // defocus offsets at the final pose
            if (int rc = ref->ddef.ensure(CH)) return rc;
            DefocusP DP;
            DP.cv = cv; DP.samples = ref->samples.p; DP.Il = ref->Il.p; DP.wring = ref->wring.p; DP.S_pad = S_pad; DP.nrings = nrings; DP.N = gm.N; DP.B = gm.B;
            DP.rlo2 = (float)(gm.r_lo * gm.r_lo); DP.rmax2 = (float)(gm.r_hi * gm.r_hi); DP.ring_signed = LP.ring_signed; DP.a = (float)gm.a;
            DP.rows = ref->rows_in.p; DP.states = final_states; DP.ddef = ref->ddef.p; DP.nt = ndef; DP.step = cfg->defocus_step; DP.all_scores = nullptr;
            DP.rcls2 = cls_on ? (float)(gm.r_cls * gm.r_cls) : 0.f;
            const int T = 2 * ndef + 1;
            DP.tchunk = std::max(1, std::min(T, (int)(60000 / (16 * (size_t)nrings))));      // per-wave ring tables of one pass stay below 64 KB
            ProfScope ps(PPM_K_LOCAL);
            hipLaunchKernelGGL(k_defocus, dim3(nb), dim3(256), ring_lds_bytes(4, DP.tchunk, nrings), cur_stream(), DP);
            d_ddef = ref->ddef.p;
        }
        hipLaunchKernelGGL(k_rows_out, dim3((nb + 255) / 256), dim3(256), 0, cur_stream(), final_states, ref->rows_in.p, ref->rows_out.p, nb, gm.a, gm.r_cls, gm.r_lo, d_ddef);
        HIPCHK(hipGetLastError());
        if (!images_on_device && c0 + CH < n_img) {     // next chunk's images travel while this chunk computes
            const int nn = std::min(CH, n_img - (c0 + CH));
            HIPCHK(hipMemcpyAsync(ref->images.p + (size_t)((ci + 1) & 1) * CH * NN, (const float *)images + (size_t)(c0 + CH) * NN,
                                  (size_t)nn * NN * sizeof(float), hipMemcpyHostToDevice, cur_copy()));
        }
        HIPCHK(hipMemcpyAsync(rows_out + (size_t)c0 * PPM_NCOL, ref->rows_out.p, (size_t)nb * PPM_NCOL * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_copy()));
    }
    // evaluation counts per particle, for the roofline's algorithmic bytes
    long nl;
    if (cfg->global_search) nl = (long)K * Tb * per_iter + (cfg->local_refine ? (long)Tc * per_iter : 0) + 1;
    else nl = 1 + (cfg->local_refine ? (long)(Tb + Tc) * per_iter : 0);
    ref->last_counts[0] = cfg->global_search ? gm.n_orient : 0;
    nl += 2L * ndef + (cls_on ? 1 : 0);
    if (cls_on) sample_evals += std::floor(kPi * gm.r_cls * gm.r_cls / 2);
    sample_evals += 2.0 * ndef * std::floor(kPi * gm.r_hi * gm.r_hi / 2);
    ref->last_counts[1] = nl;
    ref->last_counts[2] = (long)std::floor(kPi * gm.r_s * gm.r_s / 2);
    ref->last_counts[3] = (long)sample_evals;      // sum over the local evaluations of their in-band sample counts
    return 0;
}

const char *ppm_refine_note(ppm_ref_t *ref) { return ref ? ref->note.c_str() : ""; }

int ppm_refine_last_counts(ppm_ref_t *ref, long *n_global, long *n_local, long *samples_global, long *samples_local) {
    if (!ref) return fail(-22, "null reference");
    if (n_global) *n_global = ref->last_counts[0];
    if (n_local) *n_local = ref->last_counts[1];
    if (samples_global) *samples_global = ref->last_counts[2];
    if (samples_local) *samples_local = ref->last_counts[3];
    return 0;
}

// ------------------------------------------------------------------------------ reconstruction
size_t ppm_accum_floats(int box) { return (size_t)2 * box * box * (box / 2 + 1) * 3; }

ppm_accum_t *ppm_accum_create(int box, float pixel_size, const char *symmetry, void *ext) {
    if (!g.inited) { fail(-1, "ppm_init has not been called"); return nullptr; }
    if (!box_ok(box) || !(pixel_size > 0)) { fail(-22, "box must be even, 32..512, with prime factors 2, 3, 5, 7, and the pixel size positive"); return nullptr; }
    std::unique_ptr<ppm_accum, void (*)(ppm_accum_t *)> guard(new ppm_accum(), ppm_accum_destroy);      // freed on every error return
    ppm_accum *a = guard.get();
    HIPCHKP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
    HIPCHKP(hipStreamCreateWithFlags(&a->copy, hipStreamNonBlocking));
    a->N = box; a->pixel = pixel_size;
    a->nsym = symmetry_ops(symmetry, a->symops);
    if (a->nsym < 1) { fail(-22, std::string("unknown symmetry symbol '") + (symmetry ? symmetry : "") + "'"); return nullptr; }
    size_t nf = ppm_accum_floats(box);
    if (ext) { a->acc = (float *)ext; a->external = true; }
    else {
        if (hipMalloc(&a->acc, nf * sizeof(float)) != hipSuccess) { a->acc = nullptr; fail(-12, "out of device memory for the accumulators"); return nullptr; }
        (void)hipMemset(a->acc, 0, nf * sizeof(float));
    }
    std::vector<float> s(a->symops.begin(), a->symops.end());
    HIPCHKP(hipMalloc(&a->d_sym, s.size() * sizeof(float)));
    HIPCHKP(hipMemcpy(a->d_sym, s.data(), s.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHKP(hipMalloc(&a->d_counts, 2 * sizeof(unsigned long long)));
    HIPCHKP(hipMemset(a->d_counts, 0, 2 * sizeof(unsigned long long)));
    HIPCHKP(hipMalloc(&a->d_max, 2 * sizeof(unsigned)));
    static bool attr_set = false;
    std::lock_guard<std::mutex> lk_attr(g_mu);
    if (!attr_set) { HIPCHKP(hipFuncSetAttribute((const void *)k_insert_bricks<16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 17 * (17 * 52 + 3) * 8)); attr_set = true; }
    return guard.release();
}

void ppm_accum_destroy(ppm_accum_t *a) {
    if (!a) return;
    if (a->acc && !a->external) (void)hipFree(a->acc);
    if (a->d_sym) (void)hipFree(a->d_sym);
    if (a->d_counts) (void)hipFree(a->d_counts);
    if (a->d_max) (void)hipFree(a->d_max);
    a->rows.release(); a->images.release(); a->dose.release(); a->band.release(); a->spill.release(); a->s_f.release(); a->s_g.release(); a->s_vols.release(); a->pp.release(); a->cull.release(); a->items.release();
    if (a->stream) (void)hipStreamDestroy(a->stream);
    if (a->copy) (void)hipStreamDestroy(a->copy);
    delete a;
}

int ppm_insert_batch(ppm_accum_t *a, const ppm_recon_cfg *cfg, const void *images, int images_on_device, int n_img, const double *rows) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!a || !cfg || !images || !rows) return fail(-22, "null argument");
    StreamScope ss_(a->stream, a->copy);
    if (cfg->box != a->N) return fail(-22, "box differs from the accumulator's");
    if (n_img <= 0) return 0;
    ppm_refine_cfg rc; std::memset(&rc, 0, sizeof(rc));
    rc.box = a->N; rc.pixel_size = cfg->pixel_size; rc.res_high = cfg->res_limit > 0 ? cfg->res_limit : 2.f * cfg->pixel_size; rc.angular_step = 15.f;
    Geom gm; std::string err;
    if (!geom_init(gm, rc, err)) return fail(-22, err);
    const size_t NN = (size_t)gm.N * gm.N, HW = (size_t)gm.H * gm.W;
    // particles per k_prep / k_insert_bricks launch: 16 GB of images + band spectra (32 k particles at 256^2) where the device has them to
    // spare, 8 GB otherwise; swept on the 500 k x 256^2 reconstruction (scripts/sweep_insert2.sh): 4 / 8 / 16 / 24 / 32 / 48 GB -> 1.44 / 1.51 /
    // 1.55 / 1.54 / 1.54 / 1.54 M particles/s
    size_t chunk_gb = 8;
    {
        size_t free_b = 0, total_b = 0;
        // ... and only for calls of at least four such chunks: the buffers are allocated per accumulator, and a 100 k-particle call
        // through the resident server (0.18 s in all) lost more to the larger allocation than the launches gained
        const bool big_call = (size_t)n_img * (NN * 4 + HW * 8) >= ((size_t)64 << 30);
        if (big_call && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b >= ((size_t)128 << 30) && free_b >= ((size_t)64 << 30)) chunk_gb = 16;
    }
    if (getenv("PPM_INSERT_GB")) chunk_gb = (size_t)std::max(1, atoi(getenv("PPM_INSERT_GB")));
    int CH = (int)std::min<size_t>((size_t)n_img, std::max<size_t>(32, (chunk_gb << 30) / (NN * 4 + HW * 8)));
    CH = std::min(CH, 32768);
    if (const char *e = std::getenv("PPM_CHUNK")) { int v = std::atoi(e); if (v > 0) CH = std::min(CH, v); }   // tests: force several chunks
    if (int r = a->rows.ensure((size_t)CH * PPM_NCOL)) return r;
    if (!images_on_device) if (int r = a->images.ensure((size_t)2 * CH * NN)) return r;       // double-buffered staging
    if (int r = a->band.ensure((size_t)CH * HW)) return r;
    const float *d_dose = nullptr;
    float dose_cap2 = 1.f;
    if (cfg->dose_weights && cfg->n_dose_weights > 0 && cfg->dose_exponent > 0) {
        if (int r = a->dose.ensure(cfg->n_dose_weights)) return r;
        HIPCHK(hipMemcpyAsync(a->dose.p, cfg->dose_weights, (size_t)cfg->n_dose_weights * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
        d_dose = a->dose.p;
        const float tr = cfg->dose_transition > 0 && cfg->dose_transition <= 1 ? cfg->dose_transition : 1.f;
        dose_cap2 = (tr * gm.N / 2) * (tr * gm.N / 2);
    }
    if (!images_on_device) {
        HIPCHK(hipMemcpyAsync(a->images.p, images, (size_t)std::min(CH, n_img) * NN * sizeof(float), hipMemcpyHostToDevice, cur_copy()));
        HIPCHK(hipStreamSynchronize(cur_copy()));
    }
    for (int c0 = 0, ci = 0; c0 < n_img; c0 += CH, ci++) {
        const int nb = std::min(CH, n_img - c0);
        HIPCHK(hipMemcpyAsync(a->rows.p, rows + (size_t)c0 * PPM_NCOL, (size_t)nb * PPM_NCOL * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
        const float *d_img = images_on_device ? (const float *)images + (size_t)c0 * NN : a->images.p + (size_t)(ci & 1) * CH * NN;
        // the chunk's value bounds ([0] max |band| from k_prep, [1] max weight from k_insert_params) scale the fixed point
        HIPCHK(hipMemsetAsync(a->d_max, 0, 2 * sizeof(unsigned), cur_stream()));
        if (int prc = launch_prep(a->spill, d_img, a->rows.p, nb, gm, cfg->mask_radius / cfg->pixel_size, 1.f, cfg->normalize, cfg->invert, 0, 0,
                                  a->band.p, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, a->d_max)) return prc;
        // per-particle constants, then one block per (brick, particle slice, half)
        if (int r = a->pp.ensure(nb)) return r;
        if (int r = a->cull.ensure((size_t)nb * a->nsym)) return r;
        hipLaunchKernelGGL(k_insert_params, dim3((nb + 255) / 256), dim3(256), 0, cur_stream(), a->rows.p, a->pp.p, a->cull.p, a->d_sym, a->nsym, nb, gm.N, (double)cfg->pixel_size,
                           (double)cfg->score_weight_bfactor, (double)cfg->score_average, (double)cfg->score_threshold, cfg->split_by_pind,
                           gm.r_hi * gm.r_hi, a->d_counts, a->d_max, d_dose, cfg->n_dose_weights, cfg->dose_exponent, dose_cap2);
        const int BE = gm.N >= 128 ? 16 : 8;
        if (int r = build_brick_items(a, gm, BE, nb)) return r;
        InsertBrickP IP;
        IP.band = a->band.p; IP.pp = a->pp.p; IP.cull = a->cull.p; IP.symops = a->d_sym; IP.nsym = a->nsym; IP.acc = a->acc;
        IP.N = gm.N; IP.B = gm.B; IP.W = gm.W; IP.H = gm.H; IP.n_img = nb; IP.items = a->items.p; IP.maxima = a->d_max;
        IP.r2 = (float)(gm.r_hi * gm.r_hi);
        {
            ProfScope ps(PPM_K_INSERT);
            dim3 grid((unsigned)a->n_items, 2);
            if (BE == 16) hipLaunchKernelGGL((k_insert_bricks<16, 16>), grid, dim3(1024), 17 * (17 * 52 + 3) * sizeof(long long), cur_stream(), IP);
            else hipLaunchKernelGGL((k_insert_bricks<8, 4>), grid, dim3(256), 9 * (9 * 28 + 3) * sizeof(long long), cur_stream(), IP);
        }
        HIPCHK(hipGetLastError());
#ifdef PPM_INS_STAMPS
        {   // diagnostic build: cycles per phase summed over the waves of this launch (ppm_kernels2.h)
            unsigned long long st[24], z[24] = { 0 };
            HIPCHK(hipStreamSynchronize(cur_stream()));
            HIPCHK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ins_stamps), sizeof(st)));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_ins_stamps), z, sizeof(z)));
            const char *names[10] = { "zero brick", "cull", "wait after cull", "cut set-up", "row intervals + prefix", "deal-out + test", "evaluate 64", "evaluate tail", "wait at round end", "write-back" };
            double tot = 0; for (int i = 0; i < 10; i++) tot += (double)st[i];
            fprintf(stderr, "k_insert_bricks stamps, %d particles, %d items:", nb, a->n_items);
            for (int i = 0; i < 10; i++) fprintf(stderr, " | %s %.1f%%", names[i], 100.0 * (double)st[i] / tot);
            fprintf(stderr, " || wave-cycles per particle %.0f, cuts per particle %.1f, candidates per cut %.1f, hits per cut %.1f, full evaluations per cut %.2f, tails per cut %.2f\n",
                    tot / nb, (double)st[12] / nb, (double)st[13] / (double)st[12], (double)st[14] / (double)st[12], (double)st[15] / (double)st[12], (double)st[16] / (double)st[12]);
        }
#endif
        if (!images_on_device && c0 + CH < n_img) {
            const int nn = std::min(CH, n_img - (c0 + CH));
            HIPCHK(hipMemcpyAsync(a->images.p + (size_t)((ci + 1) & 1) * CH * NN, (const float *)images + (size_t)(c0 + CH) * NN,
                                  (size_t)nn * NN * sizeof(float), hipMemcpyHostToDevice, cur_copy()));
        }
        HIPCHK(hipStreamSynchronize(cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_copy()));
    }
    unsigned long long c[2];
    HIPCHK(hipMemcpy(c, a->d_counts, sizeof(c), hipMemcpyDeviceToHost));
    a->counts[0] = (long)c[0]; a->counts[1] = (long)c[1];
    return 0;
}

long ppm_accum_count(ppm_accum_t *a, int half) { return (a && (half == 0 || half == 1)) ? a->counts[half] : -1; }
void ppm_accum_set_count(ppm_accum_t *a, int half, long count) {
    if (!a || (half != 0 && half != 1)) return;
    a->counts[half] = count;
    unsigned long long c = (unsigned long long)count;
    (void)hipMemcpy(a->d_counts + half, &c, sizeof(c), hipMemcpyHostToDevice);
}

int ppm_accum_download(ppm_accum_t *a, float *host) {
    if (!a || !host) return fail(-22, "null argument");
    StreamScope ss_(a->stream, a->copy);
    HIPCHK(hipStreamSynchronize(cur_stream()));
    HIPCHK(hipMemcpy(host, a->acc, ppm_accum_floats(a->N) * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int ppm_accum_download_range(ppm_accum_t *a, float *host, size_t first, size_t count) {
    if (!a || !host) return fail(-22, "null argument");
    if (first > ppm_accum_floats(a->N) || count > ppm_accum_floats(a->N) - first) return fail(-22, "range beyond the accumulators");
    StreamScope ss_(a->stream, a->copy);
    HIPCHK(hipStreamSynchronize(cur_stream()));
    HIPCHK(hipMemcpy(host, a->acc + first, count * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int ppm_accum_add(ppm_accum_t *a, const float *host) {
    if (!a || !host) return fail(-22, "null argument");
    StreamScope ss_(a->stream, a->copy);
    size_t nf = ppm_accum_floats(a->N);
    DevTmp<float> tmp;
    HIPCHK(tmp.alloc(nf));
    HIPCHK(hipMemcpy(tmp.p, host, nf * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_axpy, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, cur_stream(), a->acc, tmp.p, nf);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(cur_stream()));
    return 0;
}

// ------------------------------------------------------------------------------ the one collective of the path (RCCL)
// librccl is opened on first use (dlopen), not linked: the single-GPU executables never pay for loading it.  Only the plain C
// entry points of rccl.h are used; their prototypes are restated here so that the library builds without the RCCL headers.
namespace {
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, ppm_comm_id, int) = nullptr;      // ncclUniqueId is passed by value: 128 opaque bytes
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(const void *, int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
static void rccl_load(Rccl &r) {
    const char *names[] = { getenv("PPM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *n : names) { if (!n) continue; r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) { r.err = std::string("librccl could not be opened: ") + dlerror(); return; }
    auto sym = [&](const char *n) { void *p = dlsym(r.h, n); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n; return p; };
    r.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
    r.CommInitRank = (int (*)(void **, int, ppm_comm_id, int))sym("ncclCommInitRank");
    r.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
    r.CommCount = (int (*)(const void *, int *))sym("ncclCommCount");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclAllReduce");
    r.Reduce = (int (*)(const void *, void *, size_t, int, int, int, void *, hipStream_t))sym("ncclReduce");
    r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
}
static Rccl &rccl() {           // opened once, whichever thread asks first (the handle-less entry points are thread-safe, include/ppm.h)
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, rccl_load, std::ref(r));
    return r;
}
constexpr int kNcclInt64 = 4, kNcclFloat32 = 7, kNcclSum = 0;      // ncclDataType_t / ncclRedOp_t values of rccl.h
int rccl_fail(Rccl &r, int rc, const char *what) {
    return fail(-5, std::string(what) + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}
}  // namespace

int ppm_comm_unique_id(ppm_comm_id *id) {
    if (!id) return fail(-22, "null argument");
    Rccl &r = rccl();
    if (!r.err.empty()) return fail(-38, r.err);
    static_assert(sizeof(ppm_comm_id) == 128, "ncclUniqueId is 128 bytes");
    if (int rc = r.GetUniqueId(id)) return rccl_fail(r, rc, "ncclGetUniqueId");
    return 0;
}

void *ppm_comm_create(int n_ranks, int rank, const ppm_comm_id *id) {
    if (!g.inited) { fail(-1, "ppm_init has not been called"); return nullptr; }
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) { fail(-22, "bad communicator arguments"); return nullptr; }
    Rccl &r = rccl();
    if (!r.err.empty()) { fail(-38, r.err); return nullptr; }
    void *comm = nullptr;
    (void)hipSetDevice(g.device);                    // the communicator binds to the calling thread's current device
    if (int rc = r.CommInitRank(&comm, n_ranks, *id, rank)) { rccl_fail(r, rc, "ncclCommInitRank"); return nullptr; }
    return comm;
}

int ppm_comm_count(void *comm) {
    if (!comm) return fail(-22, "null communicator");
    Rccl &r = rccl();
    if (!r.err.empty()) return fail(-38, r.err);
    int n = 0;
    if (int rc = r.CommCount(comm, &n)) return rccl_fail(r, rc, "ncclCommCount");
    return n;
}

void ppm_comm_destroy(void *comm) {
    Rccl &r = rccl();
    if (comm && r.CommDestroy) (void)r.CommDestroy(comm);
}

int ppm_accum_reduce(ppm_accum_t *a, void *comm, int root) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!a || !comm) return fail(-22, "null argument");
    StreamScope ss_(a->stream, a->copy);
    Rccl &r = rccl();
    if (!r.err.empty()) return fail(-38, r.err);
    const size_t nf = ppm_accum_floats(a->N);
    // the particle counters travel with the sums: brought up to date on the device, reduced as two int64
    unsigned long long c[2] = { (unsigned long long)a->counts[0], (unsigned long long)a->counts[1] };
    HIPCHK(hipMemcpyAsync(a->d_counts, c, sizeof(c), hipMemcpyHostToDevice, cur_stream()));
    int rc;
    if (root < 0) {
        rc = r.AllReduce(a->acc, a->acc, nf, kNcclFloat32, kNcclSum, comm, cur_stream());
        if (!rc) rc = r.AllReduce(a->d_counts, a->d_counts, 2, kNcclInt64, kNcclSum, comm, cur_stream());
    } else {
        rc = r.Reduce(a->acc, a->acc, nf, kNcclFloat32, kNcclSum, root, comm, cur_stream());
        if (!rc) rc = r.Reduce(a->d_counts, a->d_counts, 2, kNcclInt64, kNcclSum, root, comm, cur_stream());
    }
    if (rc) return rccl_fail(r, rc, root < 0 ? "ncclAllReduce" : "ncclReduce");
    HIPCHK(hipMemcpyAsync(c, a->d_counts, sizeof(c), hipMemcpyDeviceToHost, cur_stream()));
    HIPCHK(hipStreamSynchronize(cur_stream()));
    a->counts[0] = (long)c[0]; a->counts[1] = (long)c[1];      // on ranks other than a root the values are undefined, as ncclReduce leaves them
    return 0;
}

int ppm_extract_boxes(const void *image, int image_on_device, int rows, int cols, const double *coords, int m,
                      int box, double coordinate_binning, double radius_px, int normalize, int fix_empty,
                      void *out, int out_on_device) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!image || !coords || !out) return fail(-22, "null argument");
    if (rows <= 0 || cols <= 0 || box < 2 || box > 4096 || !(coordinate_binning > 0)) return fail(-22, "bad extraction geometry");
    if (m <= 0) return 0;
    if (radius_px > box / 2.0) radius_px = box / 2.0;        // "Particle radius falls outside box" (image.py:323-331)
    float *d_img = nullptr, *d_out = nullptr;
    DevTmp<float> t_img, t_out; DevTmp<double> t_xy;
    const size_t npix = (size_t)rows * cols, nout = (size_t)m * box * box;
    if (image_on_device) d_img = (float *)image;
    else { HIPCHK(t_img.alloc(npix)); d_img = t_img.p; HIPCHK(hipMemcpy(d_img, image, npix * sizeof(float), hipMemcpyHostToDevice)); }
    if (out_on_device) d_out = (float *)out; else { HIPCHK(t_out.alloc(nout)); d_out = t_out.p; }
    HIPCHK(t_xy.alloc((size_t)m * 2));
    double *d_xy = t_xy.p;
    HIPCHK(hipMemcpyAsync(d_xy, coords, (size_t)m * 2 * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
    ExtractP P; P.image = d_img; P.rows = rows; P.cols = cols; P.coords = d_xy; P.box = box; P.cbin = coordinate_binning;
    P.radius2 = (float)(radius_px * radius_px); P.normalize = normalize; P.fix_empty = fix_empty; P.out = d_out;
    {
        ProfScope ps(PPM_K_EXTRACT);
        hipLaunchKernelGGL(k_extract, dim3(m), dim3(256), 0, cur_stream(), P);
    }
    HIPCHK(hipGetLastError());
    if (!out_on_device) HIPCHK(hipMemcpyAsync(out, d_out, nout * sizeof(float), hipMemcpyDeviceToHost, cur_stream()));
    HIPCHK(hipStreamSynchronize(cur_stream()));
    return 0;
}

int ppm_finalize(ppm_accum_t *a, const ppm_final_cfg *cfg, float *half1, float *half2, float *filtered, double *stats) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!a || !cfg) return fail(-22, "null argument");
    StreamScope ss_(a->stream, a->copy);
    const int N = a->N, ns = N / 2;
    const double px = a->pixel;
    const size_t nf = ppm_accum_floats(N), n3 = (size_t)N * N * N, tot = (size_t)N * N * (N / 2 + 1);
    DevTmp<float> t_tmp, t_out; DevTmp<double> t_s; DevTmp<float2> t_f;
    HIPCHK(t_tmp.alloc(nf));
    HIPCHK(t_s.alloc((size_t)8 * ns));
    float *tmp = t_tmp.p; double *d_s = t_s.p;
    HIPCHK(hipMemset(d_s, 0, 8 * ns * sizeof(double)));
    HIPCHK(hipMemcpyAsync(tmp, a->acc, nf * sizeof(float), hipMemcpyDeviceToDevice, cur_stream()));
    {
    ProfScope ps(PPM_K_FINAL);
    hipLaunchKernelGGL(k_fold_plane, dim3((unsigned)(((size_t)2 * N * N + 255) / 256)), dim3(256), 0, cur_stream(), a->acc, tmp, N);
    hipLaunchKernelGGL(k_shell_den, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, cur_stream(), tmp, d_s, N);
    hipLaunchKernelGGL(k_shell_fsc, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, cur_stream(), tmp, d_s, d_s + 4 * ns, N);
    }
    std::vector<double> hs(8 * ns);
    HIPCHK(hipMemcpyAsync(hs.data(), d_s, 8 * ns * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
    HIPCHK(hipStreamSynchronize(cur_stream()));
    double vfrac = cfg->molecular_mass_kda > 0 ? (cfg->molecular_mass_kda * 1000.0 / 0.81) / std::pow(N * px, 3.0) : 1.0;
    vfrac = std::min(1.0, std::max(1e-6, vfrac));
    std::vector<double> kap(ns);
    for (int b = 0; b < ns; b++) {
        double c12 = hs[4 * ns + b], c11 = hs[5 * ns + b], c22 = hs[6 * ns + b], cnt = hs[2 * ns + b], sdt = hs[3 * ns + b];
        double fsc = (c11 > 0 && c22 > 0) ? c12 / std::sqrt(c11 * c22) : 0.0;
        double fc = fsc < 0 ? 0 : (fsc > 0.999 ? 0.999 : fsc);
        double rec = 2.0 * fc / (1.0 - fc), md = cnt > 0 ? sdt / cnt : 0;
        kap[b] = b == 0 ? 1e-20 : md / (rec > 1e-6 ? rec : 1e-6);
        if (b >= 1 && stats) {
            double *s = stats + (size_t)(b - 1) * PPM_STATS_COLS;
            s[0] = b; s[1] = N * px / b; s[2] = b / (N * px); s[3] = fsc;
            s[4] = fc / (fc + vfrac * (1 - fc)); s[5] = md > 0 ? rec / md / vfrac : 0; s[6] = rec;
        }
    }
    HIPCHK(hipMemcpy(d_s, kap.data(), ns * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(t_f.alloc(n3));
    HIPCHK(t_out.alloc(n3));
    float2 *d_f = t_f.p; float *d_out = t_out.p;
    float *outs[3] = { half1, half2, filtered };
    const float rout = (float)(cfg->outer_radius / px), rin = (float)(cfg->inner_radius / px);
    const float fo = (float)((cfg->mask_falloff > 0 ? cfg->mask_falloff : 10.0) / px);
    for (int which = 0; which < 3; which++) {
        if (!outs[which]) continue;
        {
            ProfScope p2(PPM_K_FINAL);
            HIPCHK(hipMemsetAsync(d_f, 0, n3 * sizeof(float2), cur_stream()));
            hipLaunchKernelGGL(k_wiener, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, cur_stream(), tmp, d_s, d_f, N, which);
            if (int rc = fft3d(d_f, N, true)) return rc;
            hipLaunchKernelGGL(k_map_post, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, cur_stream(), d_f, d_out, N, rout, rin, fo);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(outs[which], d_out, n3 * sizeof(float), hipMemcpyDeviceToHost, cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_stream()));
    }
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------ constrained refinement (csp)
namespace {
struct CUnit { double N[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, p[3] = { 0, 0, 0 }, tl[4] = { 0, 0, 0, 0 }, acc[6] = { 0, 0, 0, 0, 0, 0 }; };

void csp_apply(int kind, const CUnit &s, const double d[6], CUnit &o) {
    o = s;
    if (kind == PPM_CSP_PARTICLES) {
        double R[9], T[9];
        for (int k = 0; k < 3; k++) if (d[k] != 0) { rot_xyz(k, d[k], R); mat_mul3h(o.N, R, T); std::memcpy(o.N, T, sizeof(T)); }
        for (int k = 0; k < 3; k++) o.p[k] += d[3 + k];
    } else { o.tl[0] += d[0]; o.tl[1] += d[1]; o.tl[2] += d[3]; o.tl[3] += d[4]; }
    for (int k = 0; k < 6; k++) o.acc[k] += d[k];
}
}  // namespace

extern "C" int ppm_csp_refine(ppm_ref_t *ref, const ppm_refine_cfg *cfg, const ppm_csp_cfg *cc, const void *images, int images_on_device,
                              int n_proj, double *rows, double *particles, int n_part, double *tilts, int n_tilt) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!ref || !cfg || !cc || !images || !rows || !particles || !tilts) return fail(-22, "null argument");
    StreamScope ss_(ref->stream, ref->copy);
    if (cc->unit != PPM_CSP_PARTICLES && cc->unit != PPM_CSP_MICROGRAPHS) return fail(-22, "csp: unit must be particles (1) or micrographs (2)");
    if (n_proj <= 0) return 0;
    if (n_part <= 0 || n_tilt <= 0) return fail(-22, "csp: the extended parameters hold no particles or no tilts");
    ppm_refine_cfg c2 = *cfg; c2.global_search = 0;
    Geom gm; std::string err;
    if (!geom_init(gm, c2, err)) return fail(-22, err);
    if (gm.N != ref->N) return fail(-22, "particle box differs from the reference box");
    if (gm.B > (ref->B + 1) / ref->pad - 1) return fail(-22, "high-resolution limit exceeds the band the reference was prepared for");
    const int kind = cc->unit;
    const Trace trace_("ppm_csp_refine");
    // ---- device first: sample list and the prepared spectra of all rows are enqueued before the host builds its unit tables, which then
    // happens while the device works (20 k rows: ~1.5 ms of hash maps and poses against ~3 ms of pre-processing)
    const double rm_px = cfg->mask_radius / gm.a;
    SampleList sl; build_samples(gm, sl);
    const int S_pad = (int)sl.packed.size(), nrings = gm.B + 2;
    auto prefix_of = [&](double rband) { int rg = (int)std::ceil(rband); if (rg > gm.B + 1) rg = gm.B + 1; return sl.ring_off[rg]; };
    const size_t NN = (size_t)gm.N * gm.N, HW = (size_t)gm.H * gm.W;
    if (int rc = ref->samples.ensure(S_pad)) return rc;
    HIPCHK(hipMemcpyAsync(ref->samples.p, sl.packed.data(), S_pad * sizeof(uint32_t), hipMemcpyHostToDevice, cur_stream()));
    // scratch of the constrained search lives in the reference handle (grown on demand, freed with it): allocating and freeing
    // several hundred MB per call cost a third of a call on a 20 k-projection series
    DevBuf<float2> &Il = ref->c_Il, &band = ref->c_band; DevBuf<float> &cw = ref->c_cw, &img = ref->c_img, &wring = ref->c_wring;
    DevBuf<double> &d_rows = ref->c_rows, &d_N = ref->c_N, &d_p = ref->c_p, &d_tl = ref->c_tl, &d_delta = ref->c_delta, &d_s0 = ref->c_s0, &d_g0 = ref->c_g0, &d_out = ref->c_out;
    DevBuf<int> &d_eval = ref->c_eval, &d_rp = ref->c_rp, &d_rt = ref->c_rt, &d_slot = ref->c_slot;
    DevBuf<LState> &d_states = ref->c_states;
    const bool mode4 = cc->refine_defocus != 0;
    if (mode4 && kind != PPM_CSP_MICROGRAPHS) return fail(-22, "csp: defocus refinement works on tilts (unit = micrographs)");
    const int CH = (int)std::min<size_t>((size_t)n_proj, std::max<size_t>(64, ((size_t)2 << 30) / (NN * 4 + HW * 8)));
    if (int rc = Il.ensure((size_t)n_proj * S_pad)) return rc;
    if (int rc = cw.ensure((size_t)n_proj * S_pad)) return rc;
    if (int rc = band.ensure((size_t)CH * HW)) return rc;
    if (int rc = d_rows.ensure((size_t)n_proj * PPM_NCOL)) return rc;
    if (mode4) if (int rc = wring.ensure((size_t)n_proj * (gm.B + 2))) return rc;
    if (!images_on_device) if (int rc = img.ensure((size_t)CH * NN)) return rc;
    HIPCHK(hipMemcpyAsync(d_rows.p, rows, (size_t)n_proj * PPM_NCOL * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
    const double fall = cfg->mask_falloff > 0 ? cfg->mask_falloff : 20.0;
    for (int c0 = 0; c0 < n_proj; c0 += CH) {
        const int nb = std::min(CH, n_proj - c0);
        const float *d_img = (const float *)images + (size_t)c0 * NN;
        if (!images_on_device) {
            HIPCHK(hipMemcpyAsync(img.p, (const float *)images + (size_t)c0 * NN, (size_t)nb * NN * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
            d_img = img.p;
        }
        if (int rc = launch_prep(ref->spill, d_img, d_rows.p + (size_t)c0 * PPM_NCOL, nb, gm, (float)rm_px, (float)(fall / gm.a), cfg->normalize, cfg->invert, 1, 1,
                                 band.p, mode4 ? wring.p + (size_t)c0 * (gm.B + 2) : nullptr, ref->samples.p, S_pad, Il.p + (size_t)c0 * S_pad, cw.p + (size_t)c0 * S_pad, nullptr, nullptr, nullptr)) return rc;
        if (!images_on_device) HIPCHK(hipStreamSynchronize(cur_stream()));     // the staging buffer is reused by the next chunk
    }
    // ---- rows -> units
    std::unordered_map<long, int> pmap, tmap;          // tilt key: (TIND, RIND) folded into one integer
    pmap.reserve((size_t)n_part * 2); tmap.reserve((size_t)n_tilt * 2);
    auto tkey = [](long tind, long rind) { return tind * 1000003L + rind; };
    for (int i = 0; i < n_part; i++) pmap[(long)particles[(size_t)i * PPM_NPCOL]] = i;
    for (int i = 0; i < n_tilt; i++) tmap[tkey((long)tilts[(size_t)i * PPM_NTCOL], (long)tilts[(size_t)i * PPM_NTCOL + 1])] = i;
    std::vector<int> row_part(n_proj), row_tilt(n_proj);
    std::vector<unsigned char> usable(n_proj);
    std::vector<CUnit> parts(n_part), tls(n_tilt);
    for (int i = 0; i < n_part; i++) {
        const double *P = particles + (size_t)i * PPM_NPCOL;
        euler_matrix(-P[4], -P[5], -P[6], parts[i].N);
        parts[i].p[0] = P[1]; parts[i].p[1] = P[2]; parts[i].p[2] = P[3];
    }
    for (int i = 0; i < n_tilt; i++) {
        const double *T = tilts + (size_t)i * PPM_NTCOL;
        tls[i].tl[0] = T[4]; tls[i].tl[1] = T[5]; tls[i].tl[2] = T[2]; tls[i].tl[3] = T[3];
    }
    std::vector<double> s0((size_t)2 * n_proj), g0((size_t)2 * n_proj);
    std::vector<TiltRot> trot(n_tilt);                  // one set of rotations per tilt instead of four sin / cos pairs per row
    for (int i = 0; i < n_tilt; i++) tilt_rotations(tls[i].tl[0], tls[i].tl[1], trot[i]);
    for (int j = 0; j < n_proj; j++) {
        const double *row = rows + (size_t)j * PPM_NCOL;
        auto ip = pmap.find((long)row[PPM_PIND]); auto it = tmap.find(tkey((long)row[PPM_TIND], (long)row[28]));
        if (ip == pmap.end() || it == tmap.end()) return fail(-22, "csp: row " + std::to_string(j + 1) + " refers to a particle or tilt missing from the extended parameters");
        row_part[j] = ip->second; row_tilt[j] = it->second;
        const long tind = (long)row[PPM_TIND];
        usable[j] = row[PPM_OCC] > 0 && tind >= cc->tind_min && (cc->tind_max < 0 || tind <= cc->tind_max);
        s0[2 * j] = row[PPM_XSHIFT] / gm.a; s0[2 * j + 1] = row[PPM_YSHIFT] / gm.a;
        double M[9];
        const CUnit &pu = parts[row_part[j]], &tu = tls[row_tilt[j]];
        csp_row_pose(pu.N, pu.p, trot[row_tilt[j]], tu.tl[2], tu.tl[3], M, &g0[2 * j]);
    }
    const int nu_all = kind == PPM_CSP_PARTICLES ? n_part : n_tilt;
    std::vector<CUnit> &units = kind == PPM_CSP_PARTICLES ? parts : tls;
    std::vector<std::vector<int>> urows(nu_all);
    for (int j = 0; j < n_proj; j++) urows[kind == PPM_CSP_PARTICLES ? row_part[j] : row_tilt[j]].push_back(j);
    std::vector<int> unit_slot(nu_all, -1), active;        // active: refined units with at least one usable row
    std::vector<unsigned char> refined(nu_all, 0);
    for (int u = 0; u < nu_all; u++) {
        const long id = (long)(kind == PPM_CSP_PARTICLES ? particles[(size_t)u * PPM_NPCOL] : tilts[(size_t)u * PPM_NTCOL]);
        if (id < cc->first || (cc->last >= 0 && id > cc->last)) continue;
        refined[u] = 1;
        int nus = 0; for (int j : urows[u]) nus += usable[j];
        if (nus) { unit_slot[u] = (int)active.size(); active.push_back(u); }
    }
    std::vector<int> eval_rows, final_rows;
    for (int u : active) for (int j : urows[u]) if (usable[j]) eval_rows.push_back(j);
    for (int u = 0; u < nu_all; u++) if (refined[u]) for (int j : urows[u]) final_rows.push_back(j);
    if (final_rows.empty()) return 0;
    // units without usable rows still need a slot for the final scoring of their rows (zero displacement)
    int n_slots = (int)active.size();
    for (int u = 0; u < nu_all; u++) if (refined[u] && unit_slot[u] < 0) unit_slot[u] = n_slots++;
    // in the other kind's lookups (a particle sweep reads the tilt of a row and vice versa) no slot is needed

    int en[6] = { 0, 0, 0, 0, 0, 0 }; double tol[6] = { 0, 0, 0, 0, 0, 0 };
    if (kind == PPM_CSP_PARTICLES) {
        for (int k = 0; k < 3; k++) { en[k] = cc->refine_rotation != 0; tol[k] = cc->tol_angle[k]; en[3 + k] = cc->refine_translation != 0; tol[3 + k] = cc->tol_shift; }
    } else {
        en[0] = en[1] = cc->refine_rotation != 0; tol[0] = cc->tol_angle[0]; tol[1] = cc->tol_angle[1];
        en[3] = en[4] = cc->refine_translation != 0; tol[3] = tol[4] = cc->tol_shift;
    }
    int nfree = 0;
    for (int k = 0; k < 6; k++) { if (!(tol[k] > 0)) en[k] = 0; nfree += en[k]; }
    double ha0 = 0, hs0 = 0;
    for (int k = 0; k < 3; k++) if (en[k] && 0.5 * tol[k] > ha0) ha0 = 0.5 * tol[k];
    for (int k = 3; k < 6; k++) if (en[k] && 0.5 * tol[k] > hs0) hs0 = 0.5 * tol[k];
    const double steptol = cc->step_tolerance > 0 ? cc->step_tolerance : 0.01;
    int T = cc->max_iterations;
    if (T <= 0) { const double m = std::max(ha0, hs0); T = m > steptol ? (int)std::ceil(std::log(m / steptol) / std::log(2.0)) : 1; T = std::min(12, std::max(1, T)); }
    if (!nfree || active.empty()) T = 0;
    const double bf = cfg->band_factor == 0 ? 3.0 : cfg->band_factor;
    const bool any_ang = en[0] || en[1] || en[2], any_sh = en[3] || en[4] || en[5];
    auto iter_band = [&](double ha, double hs) {
        if (bf < 0) return gm.r_hi;
        double d = 0;
        if (any_ang) d = rm_px * ha * kPi / 180.0;
        if (any_sh && hs > d) d = hs;
        if (!(d > 0)) return gm.r_hi;
        double rit = bf * gm.N / (2.0 * kPi * d);
        if (rit < 4.0) rit = 4.0;
        return rit < gm.r_hi ? rit : gm.r_hi;
    };

    trace_.mark("host tables");
    if (mode4) {
        // ---- csp mode 4: every row's score for every defocus offset in one sweep (k_defocus), averaged per tilt on the host
        const double step = cc->defocus_step > 0 ? cc->defocus_step : 50.0;
        int nt = 0;
        if (cc->defocus_range >= step) nt = std::min((int)std::floor(cc->defocus_range / step + 1e-6), PPM_MAX_DEFOCUS_STEPS);
        const int Tn = 2 * nt + 1;
        if (int rc = d_states.ensure(n_proj)) return rc;
        if (int rc = d_out.ensure((size_t)n_proj * Tn)) return rc;
        hipLaunchKernelGGL(k_states_from_rows, dim3((n_proj + 255) / 256), dim3(256), 0, cur_stream(), d_rows.p, d_states.p, n_proj, gm.a, 1.0, 1.0);
        DefocusP DP;
        DP.cv.cube = ref->cube; DP.cv.NBX = ref->NBX; DP.cv.NBY = ref->NBY; DP.cv.LB = ref->LB; DP.cv.off = ref->B + 1; DP.cv.scale = (float)ref->pad;
        DP.samples = ref->samples.p; DP.Il = Il.p; DP.wring = wring.p; DP.S_pad = S_pad; DP.nrings = nrings; DP.N = gm.N; DP.B = gm.B;
        DP.rlo2 = (float)(gm.r_lo * gm.r_lo); DP.rmax2 = (float)(gm.r_hi * gm.r_hi); DP.ring_signed = (float)std::min(gm.ring_signed, 1e30); DP.a = (float)gm.a;
        DP.rows = d_rows.p; DP.states = d_states.p; DP.ddef = nullptr; DP.nt = nt; DP.step = (float)step; DP.all_scores = d_out.p; DP.rcls2 = 0.f;
        DP.tchunk = std::max(1, std::min(Tn, (int)(60000 / (16 * (size_t)nrings))));
        {
            ProfScope ps(PPM_K_LOCAL);
            hipLaunchKernelGGL(k_defocus, dim3(n_proj), dim3(256), ring_lds_bytes(4, DP.tchunk, nrings), cur_stream(), DP);
        }
        HIPCHK(hipGetLastError());
        std::vector<double> sc((size_t)n_proj * Tn);
        HIPCHK(hipMemcpyAsync(sc.data(), d_out.p, sc.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_stream()));
        for (int u = 0; u < n_tilt; u++) {
            if (!refined[u]) continue;
            double best = -1e300; int bt = nt;
            for (int pass = 0; pass < 2; pass++)
                for (int t = (pass ? 0 : nt); t < (pass ? Tn : nt + 1); t++) {
                    if (pass && t == nt) continue;
                    double ssum = 0; int sn = 0;
                    for (int j : urows[u]) if (usable[j]) { ssum += sc[(size_t)j * Tn + t]; sn++; }
                    if (sn && ssum / sn > best) { best = ssum / sn; bt = t; }
                }
            for (int j : urows[u]) {
                double *row = rows + (size_t)j * PPM_NCOL;
                row[PPM_DF1] += (bt - nt) * step; row[PPM_DF2] += (bt - nt) * step;
                const double ccv = sc[(size_t)j * Tn + bt]; double res = 1.0 - ccv * ccv; if (res < 1e-6) res = 1e-6;
                row[PPM_SCORE] = 100.0 * ccv; row[PPM_SIGMA] = std::sqrt(res);
                row[PPM_LOGP] = -0.5 * (kPi * (gm.r_hi * gm.r_hi - gm.r_lo * gm.r_lo)) * (std::log(2.0 * kPi * res) + 1.0);
            }
        }
        return 0;
    }
    // ---- static tables
    if (int rc = d_rp.ensure(n_proj)) return rc;
    if (int rc = d_rt.ensure(n_proj)) return rc;
    if (int rc = d_slot.ensure(nu_all)) return rc;
    if (int rc = d_s0.ensure((size_t)2 * n_proj)) return rc;
    if (int rc = d_g0.ensure((size_t)2 * n_proj)) return rc;
    if (int rc = d_N.ensure((size_t)9 * n_part)) return rc;
    if (int rc = d_p.ensure((size_t)3 * n_part)) return rc;
    if (int rc = d_tl.ensure((size_t)4 * n_tilt)) return rc;
    const int ncand_max = 1 + 2 * nfree;
    if (ncand_max > kMaxCand) return fail(-22, "csp: too many free parameters");
    if (int rc = d_delta.ensure((size_t)std::max(n_slots, 1) * ncand_max * 6)) return rc;
    if (int rc = d_eval.ensure(std::max(eval_rows.size(), final_rows.size()))) return rc;
    if (int rc = d_out.ensure(std::max(eval_rows.size() * ncand_max, final_rows.size()))) return rc;
    HIPCHK(hipMemcpyAsync(d_rp.p, row_part.data(), n_proj * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
    HIPCHK(hipMemcpyAsync(d_rt.p, row_tilt.data(), n_proj * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
    HIPCHK(hipMemcpyAsync(d_slot.p, unit_slot.data(), nu_all * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
    HIPCHK(hipMemcpyAsync(d_s0.p, s0.data(), s0.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
    HIPCHK(hipMemcpyAsync(d_g0.p, g0.data(), g0.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
    std::vector<double> hN((size_t)9 * n_part), hp((size_t)3 * n_part), htl((size_t)4 * n_tilt);
    auto upload_units = [&]() -> int {
        for (int i = 0; i < n_part; i++) { std::memcpy(&hN[(size_t)9 * i], parts[i].N, 9 * sizeof(double)); std::memcpy(&hp[(size_t)3 * i], parts[i].p, 3 * sizeof(double)); }
        for (int i = 0; i < n_tilt; i++) std::memcpy(&htl[(size_t)4 * i], tls[i].tl, 4 * sizeof(double));
        HIPCHK(hipMemcpyAsync(d_N.p, hN.data(), hN.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipMemcpyAsync(d_p.p, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipMemcpyAsync(d_tl.p, htl.data(), htl.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_stream()));
        return 0;
    };
    CspEvalP EP;
    EP.cv.cube = ref->cube; EP.cv.NBX = ref->NBX; EP.cv.NBY = ref->NBY; EP.cv.LB = ref->LB; EP.cv.off = ref->B + 1; EP.cv.scale = (float)ref->pad;
    EP.samples = ref->samples.p; EP.Il = Il.p; EP.cw = cw.p; EP.S_pad = S_pad; EP.N = gm.N; EP.nr = nrings;
    EP.tabR = cube_tab_radius(gm.B, EP.cv.scale);
    // threads per block of k_csp_eval by the samples of a sweep (PPM_CSP_THREADS = 64 / 128 / 256 forces one)
    auto csp_threads = [](int S_used) {
        if (const char *e = getenv("PPM_CSP_THREADS")) { const int v = atoi(e); if (v == 64 || v == 128 || v == 256) return v; }
        return 256;
    };
    const int csp_bpc = getenv("PPM_CSP_BLOCKS_PER_CU") ? atoi(getenv("PPM_CSP_BLOCKS_PER_CU")) : 0;      // blocks of k_csp_eval per CU through the LDS request (0: what registers and LDS allow)
    const bool csp_tab = !(getenv("PPM_LOCAL_TABLES") && atoi(getenv("PPM_LOCAL_TABLES")) == 0) && cube_tab_bytes(EP.tabR) <= 16 * 1024 &&
                         ring_lds_bytes8(4, kMaxCand, nrings) + cube_tab_bytes(EP.tabR) + 2048 <= (size_t)64 * 1024;
    EP.rlo2 = (float)(gm.r_lo * gm.r_lo); EP.ring_signed = (float)std::min(gm.ring_signed, 1e30);
    EP.kind = kind; EP.eval_rows = d_eval.p; EP.row_part = d_rp.p; EP.row_tilt = d_rt.p; EP.unit_slot = d_slot.p;
    EP.Nmat = d_N.p; EP.pshift = d_p.p; EP.tl = d_tl.p; EP.delta = d_delta.p; EP.s0 = d_s0.p; EP.g0 = d_g0.p; EP.out = d_out.p;
    std::vector<double> hdelta, hout;
    // evaluation list of the search (the usable rows of the active units, grouped by unit) and the units' offsets in it: uploaded once
    std::vector<int> uoff(active.size() + 1, 0);
    for (size_t a = 0; a < active.size(); a++) { int n = 0; for (int j : urows[active[a]]) n += usable[j] ? 1 : 0; uoff[a + 1] = uoff[a] + n; }
    if (int rc = ref->c_uoff.ensure(uoff.size())) return rc;
    if (int rc = ref->c_mean.ensure(std::max<size_t>(active.size() * ncand_max, 1))) return rc;
    HIPCHK(hipMemcpyAsync(ref->c_uoff.p, uoff.data(), uoff.size() * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
    const std::vector<int> *uploaded_list = nullptr;
    double acct_gathers = 0; long acct_sweeps = 0;
    // one sweep: `ncand` candidates per unit (hdelta laid out [slot][ncand][6]) over `rows_list`.  unit_means: the per-unit means of
    // the scores -> `means` [active unit][ncand] (reduced on the device); otherwise the per-row scores -> hout [row][ncand]
    auto sweep = [&](const std::vector<int> &rows_list, int ncand, double rband, std::vector<double> *means) -> int {
        HIPCHK(hipMemcpyAsync(d_delta.p, hdelta.data(), (size_t)n_slots * ncand * 6 * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
        if (uploaded_list != &rows_list) {
            HIPCHK(hipMemcpyAsync(d_eval.p, rows_list.data(), rows_list.size() * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
            uploaded_list = &rows_list;
        }
        EP.ncand = ncand; EP.S_used = prefix_of(rband); EP.rmax2 = (float)(rband * rband);
        {   // accounting for the roofline (ppm_refine_last_counts): gathers = samples x rotations that differ (shift candidates share the centre's)
            int nrot_c = 0;
            if (ncand > 1) for (int i = 0; i < 3; i++) nrot_c += en[i] ? 2 : 0;
            acct_gathers += (double)rows_list.size() * EP.S_used * (1 + nrot_c); acct_sweeps++;
        }
        {
            ProfScope ps(PPM_K_LOCAL);
            {
                const int thr = csp_threads(EP.S_used);
                size_t lds = ring_lds_bytes8(thr / 64, kMaxCand, nrings) + (csp_tab ? cube_tab_bytes(EP.tabR) : 0);
                if (csp_bpc > 0 && csp_bpc < 8) lds = std::min((size_t)63 * 1024, std::max(lds, (size_t)(160 * 1024 / (csp_bpc + 1) + 1024) & ~(size_t)1023));
                if (csp_tab) hipLaunchKernelGGL(k_csp_eval<true>, dim3((unsigned)rows_list.size()), dim3(thr), lds, cur_stream(), EP);
                else hipLaunchKernelGGL(k_csp_eval<false>, dim3((unsigned)rows_list.size()), dim3(thr), lds, cur_stream(), EP);
            }
        }
        if (means) {
            const int nm = (int)active.size() * ncand;
            hipLaunchKernelGGL(k_csp_unit_means, dim3((nm + 255) / 256), dim3(256), 0, cur_stream(), d_out.p, ref->c_uoff.p, (int)active.size(), ncand, ref->c_mean.p);
            HIPCHK(hipGetLastError());
            means->resize((size_t)nm);
            HIPCHK(hipMemcpyAsync(means->data(), ref->c_mean.p, (size_t)nm * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
        } else {
            HIPCHK(hipGetLastError());
            hout.resize(rows_list.size() * (size_t)ncand);
            HIPCHK(hipMemcpyAsync(hout.data(), d_out.p, hout.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
        }
        HIPCHK(hipStreamSynchronize(cur_stream()));
        return 0;
    };
    if (int rc = upload_units()) return rc;
    trace_.mark("spectra prepared, units uploaded");
    // ---- the compass search: state and decisions on the device (ppm_csp_kernels.h), the iterations enqueued back to back — six launches
    // each (candidates, unit means, trial step, its score, its means, accept) and no host wait until the units come back at the end.
    // The bands follow from the step schedule alone, so the host knows them up front.
    if (T > 0) {
        const int na = (int)active.size();
        DevBuf<double> &d_acc = ref->c_acc, &d_dtrial = ref->c_dtrial, &d_fpm = ref->c_fpm, &d_delta_t = ref->c_delta_t, &d_tmean = ref->c_tmean;
        DevBuf<int> &d_active = ref->c_active;
        if (int rc = d_acc.ensure((size_t)na * 6)) return rc;
        if (int rc = d_dtrial.ensure((size_t)na * 6)) return rc;
        if (int rc = d_fpm.ensure((size_t)na * 12)) return rc;
        if (int rc = d_delta_t.ensure((size_t)std::max(n_slots, 1) * 6)) return rc;
        if (int rc = d_tmean.ensure((size_t)na)) return rc;
        if (int rc = d_active.ensure((size_t)na)) return rc;
        HIPCHK(hipMemcpyAsync(d_active.p, active.data(), (size_t)na * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipMemcpyAsync(d_eval.p, eval_rows.data(), eval_rows.size() * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
        uploaded_list = &eval_rows;
        HIPCHK(hipMemsetAsync(d_delta.p, 0, (size_t)std::max(n_slots, 1) * ncand_max * 6 * sizeof(double), cur_stream()));      // slots of units without usable rows stay zero
        HIPCHK(hipMemsetAsync(d_delta_t.p, 0, (size_t)std::max(n_slots, 1) * 6 * sizeof(double), cur_stream()));
        int ncand = 1;
        for (int i = 0; i < 6; i++) ncand += en[i] ? 2 : 0;
        CspStepP SP;
        SP.kind = kind; SP.n_active = na; SP.ncand = ncand; SP.active = d_active.p; SP.unit_slot = d_slot.p;
        for (int i = 0; i < 6; i++) { SP.en[i] = en[i]; SP.tol[i] = tol[i]; }
        SP.mean = ref->c_mean.p; SP.tmean = d_tmean.p; SP.acc = d_acc.p; SP.dtrial = d_dtrial.p; SP.fpm = d_fpm.p;
        SP.delta_c = d_delta.p; SP.delta_t = d_delta_t.p; SP.Nmat = d_N.p; SP.pshift = d_p.p; SP.tl = d_tl.p; SP.nstride = 9; SP.pstride = 3;
        HIPCHK(hipMemsetAsync(d_acc.p, 0, (size_t)na * 6 * sizeof(double), cur_stream()));
        const unsigned gstep = (unsigned)((na + 127) / 128);
        int nrot_c = 0;
        for (int i = 0; i < 3; i++) nrot_c += en[i] ? 2 : 0;
        auto eval_async = [&](const double *delta, int nc, double rband, double *means) {
            EP.delta = delta; EP.ncand = nc; EP.S_used = prefix_of(rband); EP.rmax2 = (float)(rband * rband);
            acct_gathers += (double)eval_rows.size() * EP.S_used * (nc > 1 ? 1 + nrot_c : 1); acct_sweeps++;
            {
                ProfScope ps(PPM_K_LOCAL);
                const int thr = csp_threads(EP.S_used);
                size_t lds = ring_lds_bytes8(thr / 64, kMaxCand, nrings) + (csp_tab ? cube_tab_bytes(EP.tabR) : 0);
                if (csp_bpc > 0 && csp_bpc < 8) lds = std::min((size_t)63 * 1024, std::max(lds, (size_t)(160 * 1024 / (csp_bpc + 1) + 1024) & ~(size_t)1023));
                if (csp_tab) hipLaunchKernelGGL(k_csp_eval<true>, dim3((unsigned)eval_rows.size()), dim3(thr), lds, cur_stream(), EP);
                else hipLaunchKernelGGL(k_csp_eval<false>, dim3((unsigned)eval_rows.size()), dim3(thr), lds, cur_stream(), EP);
            }
            const int nm = na * nc;
            hipLaunchKernelGGL(k_csp_unit_means, dim3((nm + 255) / 256), dim3(256), 0, cur_stream(), d_out.p, ref->c_uoff.p, na, nc, means);
        };
        double ha = ha0, hs = hs0;
        SP.ha = ha; SP.hs = hs; SP.ha_next = ha; SP.hs_next = hs;
        hipLaunchKernelGGL(k_csp_step_init, dim3(gstep), dim3(128), 0, cur_stream(), SP);
        for (int it = 0; it < T; it++) {
            const double rband = iter_band(ha, hs);
            SP.ha = ha; SP.hs = hs; SP.ha_next = 0.5 * ha; SP.hs_next = 0.5 * hs;
            eval_async(d_delta.p, ncand, rband, ref->c_mean.p);
            hipLaunchKernelGGL(k_csp_step_trial, dim3(gstep), dim3(128), 0, cur_stream(), SP);
            eval_async(d_delta_t.p, 1, rband, d_tmean.p);
            hipLaunchKernelGGL(k_csp_step_accept, dim3(gstep), dim3(128), 0, cur_stream(), SP);
            ha *= 0.5; hs *= 0.5;
        }
        HIPCHK(hipGetLastError());
        // the units as the search left them
        HIPCHK(hipMemcpyAsync(hN.data(), d_N.p, hN.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
        HIPCHK(hipMemcpyAsync(hp.data(), d_p.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
        HIPCHK(hipMemcpyAsync(htl.data(), d_tl.p, htl.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
        HIPCHK(hipStreamSynchronize(cur_stream()));
        for (int i = 0; i < n_part; i++) { std::memcpy(parts[i].N, &hN[(size_t)9 * i], 9 * sizeof(double)); std::memcpy(parts[i].p, &hp[(size_t)3 * i], 3 * sizeof(double)); }
        for (int i = 0; i < n_tilt; i++) std::memcpy(tls[i].tl, &htl[(size_t)4 * i], 4 * sizeof(double));
        EP.delta = d_delta.p;
    }
    trace_.mark("searched");
    // ---- final scores of every row of the refined units at the full band; write-back
    hdelta.assign((size_t)std::max(n_slots, 1) * 6, 0.0);
    if (int rc = sweep(final_rows, 1, gm.r_hi, nullptr)) return rc;
    trace_.mark("final scores");
    std::vector<double> row_score(n_proj, 0.0);
    for (size_t q = 0; q < final_rows.size(); q++) row_score[final_rows[q]] = hout[q];
    for (int i = 0; i < n_tilt; i++) tilt_rotations(tls[i].tl[0], tls[i].tl[1], trot[i]);       // the tilts may have moved
    for (int u = 0; u < nu_all; u++) {
        if (!refined[u]) continue;
        if (kind == PPM_CSP_PARTICLES) {
            double *P = particles + (size_t)u * PPM_NPCOL, a1, a2, a3;
            angles_from_matrix(units[u].N, a1, a2, a3);
            P[4] = -a1; P[5] = -a2; P[6] = -a3; P[1] = units[u].p[0]; P[2] = units[u].p[1]; P[3] = units[u].p[2];
        } else {
            double *Tt = tilts + (size_t)u * PPM_NTCOL;
            Tt[4] = units[u].tl[0]; Tt[5] = units[u].tl[1]; Tt[2] = units[u].tl[2]; Tt[3] = units[u].tl[3];
        }
        double ssum = 0; int sn = 0;
        for (int j : urows[u]) {
            double *row = rows + (size_t)j * PPM_NCOL, M[9], gq[2];
            const CUnit &pu = parts[row_part[j]], &tu = tls[row_tilt[j]];
            csp_row_pose(pu.N, pu.p, trot[row_tilt[j]], tu.tl[2], tu.tl[3], M, gq);
            angles_from_matrix(M, row[PPM_PSI], row[PPM_THETA], row[PPM_PHI]);
            row[PPM_XSHIFT] = (s0[2 * j] + gq[0] - g0[2 * j]) * gm.a; row[PPM_YSHIFT] = (s0[2 * j + 1] + gq[1] - g0[2 * j + 1]) * gm.a;
            const double ccv = row_score[j]; double res = 1.0 - ccv * ccv; if (res < 1e-6) res = 1e-6;
            row[PPM_SCORE] = 100.0 * ccv; row[PPM_SIGMA] = std::sqrt(res);
            row[PPM_LOGP] = -0.5 * (kPi * (gm.r_hi * gm.r_hi - gm.r_lo * gm.r_lo)) * (std::log(2.0 * kPi * res) + 1.0);
            if (usable[j]) { ssum += row[PPM_SCORE]; sn++; }
        }
        if (kind == PPM_CSP_PARTICLES) particles[(size_t)u * PPM_NPCOL + 10] = sn ? ssum / sn : -1.0;
    }
    // ppm_refine_last_counts after a constrained refinement: 0, sweeps (k_csp_eval launches), in-band samples of the full band, gathered
    // samples per projection summed over the sweeps
    ref->last_counts[0] = 0; ref->last_counts[1] = acct_sweeps; ref->last_counts[2] = (long)std::floor(kPi * gm.r_hi * gm.r_hi / 2);
    ref->last_counts[3] = (long)(acct_gathers / std::max(n_proj, 1));
    return 0;
}

// ------------------------------------------------------------------------------ sub-tomogram alignment (3DAVG)
namespace {
double sva_band_weight(const ppm_sva_cfg &c, double s) {
    double w = 1.0;
    if (c.highpass_cutoff > 0 && s < c.highpass_cutoff) { const double d = c.highpass_cutoff - s; w *= c.highpass_decay > 0 ? std::exp(-d * d / (2.0 * c.highpass_decay * c.highpass_decay)) : 0.0; }
    if (c.lowpass_cutoff > 0 && s > c.lowpass_cutoff) { const double d = s - c.lowpass_cutoff; w *= c.lowpass_decay > 0 ? std::exp(-d * d / (2.0 * c.lowpass_decay * c.lowpass_decay)) : 0.0; }
    return w;
}
double sva_band_radius(const ppm_sva_cfg &c) {
    const int N = c.box;
    double s = c.lowpass_cutoff > 0 ? c.lowpass_cutoff + (c.lowpass_decay > 0 ? 3.7169 * c.lowpass_decay : 0.0) : 0.5;
    if (s > 0.5) s = 0.5;
    double r = s * N; if (r > N / 2 - 1) r = N / 2 - 1;
    return r;
}
}  // namespace

static int sva_insert_device(ppm_accum_t *a, const ppm_sva_cfg *cfg, const float *d_vols, int n_vol, const float *wedges, const double *poses,
                             const long *index, long index_base);

// ppm_sva_align and ppm_sva_align_average: with an accumulator every chunk is added to the average at its refined poses while it is
// still in device memory (host volumes cross PCIe once per iteration)
static int sva_align_impl(ppm_ref_t *ref, ppm_accum_t *avg, const ppm_sva_cfg *cfg, const void *volumes, int volumes_on_device, int n_vol, const float *wedges,
                          double *poses, double *scores, const long *index) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!ref || !cfg || !volumes || !poses) return fail(-22, "null argument");
    StreamScope ss_(ref->stream, ref->copy);
    if (n_vol <= 0) return 0;
    const Trace trace_("ppm_sva_align");
    auto mark = [&](const char *what) { trace_.mark(what); };
    const int N = cfg->box;
    if (!box_ok(N) || N != ref->N) return fail(-22, "sub-volume box differs from the reference box (even, 32..512, prime factors 2, 3, 5, 7)");
    if (ref->pad != 1) return fail(-22, "sub-tomogram alignment needs a reference prepared with padding 1");
    const double rband = sva_band_radius(*cfg);
    if (rband > ref->B) return fail(-22, "low-pass limit exceeds the band the reference was prepared for");
    const size_t n3 = (size_t)N * N * N;
    // ---- sample list of the band (half space, shell by shell), common to all sub-volumes; the wedge is applied per volume
    const int R = (int)std::ceil(rband);
    std::vector<uint32_t> samples; std::vector<float> bandw; std::vector<int> shell_off(R + 2, 0);
    const float plan_key[5] = { (float)N, cfg->highpass_cutoff, cfg->highpass_decay, cfg->lowpass_cutoff, cfg->lowpass_decay };
    const bool plan_cached = ref->s_plan.valid && std::memcmp(plan_key, ref->s_plan.key, sizeof(plan_key)) == 0;
    if (plan_cached) shell_off = ref->s_plan.shell_off;
    else {
        // one pass over the half space, bucketed by shell (the order inside a shell is the scan order kz, ky, kx)
        std::vector<std::vector<uint32_t>> sh_s(R + 1); std::vector<std::vector<float>> sh_w(R + 1);
        for (int kz = -R; kz <= R; kz++) for (int ky = -R; ky <= R; ky++) for (int kx = 0; kx <= R; kx++) {
            const double k2 = (double)kx * kx + (double)ky * ky + (double)kz * kz;
            if (k2 == 0 || k2 >= rband * rband) continue;
            if (kx == 0 && (ky < 0 || (ky == 0 && kz < 0))) continue;
            const double kr = std::sqrt(k2);
            const int sh = (int)std::floor(kr);
            if (sh > R) continue;
            const double w = sva_band_weight(*cfg, kr / N);
            if (w < 1e-3) continue;
            sh_s[sh].push_back(sva_pack(kx, ky, kz)); sh_w[sh].push_back((float)w);
        }
        // inside a shell the samples are grouped by tilt angle and follow a Z-order curve inside a group: the 64 lanes of a wave gather
        // from a compact patch of the reference cube
        auto spread = [](uint32_t v) { uint64_t x = v & 0x3ffu; x = (x | x << 16) & 0x30000ffull; x = (x | x << 8) & 0x300f00full; x = (x | x << 4) & 0x30c30c3ull; x = (x | x << 2) & 0x9249249ull; return x; };
        for (int sh = 0; sh <= R; sh++) {
            std::vector<std::pair<uint64_t, int>> key(sh_s[sh].size());
            for (size_t i = 0; i < key.size(); i++) {
                int kx, ky, kz; sva_unpack(sh_s[sh][i], kx, ky, kz);
                // major key: the tilt angle of the sample's (kx, kz) direction in 4-degree bins, so that the samples a missing wedge
                // removes are whole waves (k_sva_eval skips zero weights)
                double ang = (kx == 0 && kz == 0) ? 0.0 : std::atan2((double)kz, (double)kx) * 180.0 / kPi;
                if (ang > 90.0) ang -= 180.0;
                if (ang <= -90.0) ang += 180.0;
                const uint64_t bin = (uint64_t)std::floor((ang + 90.0) / 4.0);
                key[i] = { bin << 40 | spread((uint32_t)kx) | spread((uint32_t)(ky + R)) << 1 | spread((uint32_t)(kz + R)) << 2, (int)i };
            }
            std::sort(key.begin(), key.end());
            for (const auto &k : key) { samples.push_back(sh_s[sh][k.second]); bandw.push_back(sh_w[sh][k.second]); }
            shell_off[sh + 1] = (int)samples.size();
        }
    }
    const int S = plan_cached ? ref->s_plan.S : (int)samples.size();
    if (S == 0) return fail(-22, "the band-pass filter leaves no Fourier samples");
    auto prefix_of = [&](double rb) { int rg = (int)std::ceil(rb); if (rg > R + 1) rg = R + 1; return shell_off[rg]; };
    // ---- search plan (the particle unit of the constrained search: rotations about the specimen axes + 3-D shift)
    int en[6]; double tol[6];
    for (int k = 0; k < 3; k++) { en[k] = cfg->tol_angle > 0 && cfg->search_mode != 2; tol[k] = cfg->tol_angle; en[3 + k] = cfg->tol_shift > 0; tol[3 + k] = cfg->tol_shift; }
    // global rotation + translation search (ppm_sva_cfg.search_mode 1, include/ppm.h)
    const bool global = cfg->search_mode == 1;
    const double gstep = cfg->global_step > 0 ? cfg->global_step : 15.0;
    std::vector<double> grid_d; int n_grid = 0;
    if (global) {
        int n_theta = (int)std::floor(180.0 / gstep + 0.5) + 1; if (n_theta < 2) n_theta = 2;
        int n_psi = (int)std::floor(360.0 / gstep + 0.5); if (n_psi < 1) n_psi = 1;
        for (int i = 0; i < n_theta; i++) {
            const double th = 180.0 * i / (n_theta - 1);
            int np = (int)std::floor(360.0 * std::sin(th * kPi / 180.0) / gstep + 0.5); if (np < 1) np = 1;
            for (int j = 0; j < np; j++) for (int k = 0; k < n_psi; k++) {
                double G[9]; euler_matrix(k * 360.0 / n_psi, th, 360.0 * j / np, G);
                grid_d.insert(grid_d.end(), G, G + 9);
            }
        }
        n_grid = (int)(grid_d.size() / 9);
    }
    int Kc = cfg->n_candidates > 0 ? cfg->n_candidates : 25; Kc = std::min(std::min(Kc, 64), std::max(n_grid, 1));
    const int eng[6] = { 1, 1, 1, en[3], en[4], en[5] };
    const double tolg[6] = { gstep, gstep, gstep, tol[3], tol[4], tol[5] };
    const int nrot = (en[0] || global) ? 6 : 0, nsh = en[3] ? 6 : 0, ncand = 1 + nrot + nsh;
    const double steptol = cfg->step_tolerance > 0 ? cfg->step_tolerance : 0.05;
    const double ha0 = 0.5 * cfg->tol_angle, hs0 = 0.5 * cfg->tol_shift;
    int T = cfg->max_iterations;
    if (T <= 0) { const double m = std::max(ha0, hs0); T = m > steptol ? (int)std::ceil(std::log(m / steptol) / std::log(2.0)) : 1; T = std::min(12, std::max(1, T)); }
    if (ncand == 1) T = 0;
    const double bf = cfg->band_factor == 0 ? 3.0 : cfg->band_factor;
    double rm_px = std::max(cfg->window[0], std::max(cfg->window[1], cfg->window[2]));
    if (!(rm_px > 0)) rm_px = 0.4 * N;
    // coarse band the grid step allows (probe Delta / 2, rotations only)
    double rg = rband;
    if (global && bf >= 0) { const double d = rm_px * 0.5 * gstep * kPi / 180.0; double rit = bf * N / (2.0 * kPi * d); if (rit < 4.0) rit = 4.0; rg = std::min(rit, rband); }
    // ---- device buffers (RAII), chunks of sub-volumes
    // chunks of sub-volumes: the search kernel runs one block per sub-volume, so a chunk should fill the chip (>= 256 blocks).  Resident
    // volumes: limited by the band transforms (S float2 each, 4 GB); host volumes: two staging buffers of a chunk each (2 x 7 GB at
    // 192^3 — small change on a 288 GB device), the next chunk uploaded while this one is searched.
    int CH = (int)std::min<size_t>((size_t)n_vol, std::max<size_t>(1, ((size_t)4 << 30) / ((size_t)S * 8)));
    if (!volumes_on_device) {
        const int hc = getenv("PPM_SVA_CHUNK") ? std::max(1, atoi(getenv("PPM_SVA_CHUNK"))) : (int)std::max<size_t>(1, ((size_t)7 << 30) / (n3 * 4));
        CH = std::min(CH, hc);
    }
    struct { uint32_t *p; } d_samples{nullptr}; struct { float *p; } d_bandw{nullptr};       // the handle's cached sample plan
    DevTmp<float> d_wedges, d_grid, d_gscore; DevTmp<double> d_stats, d_poses, d_delta, d_out, d_partial;
    struct { float2 *p; } d_f{nullptr}, d_F{nullptr};       // views of the handle's cached work arrays
    struct { float *p; } d_vols{nullptr};
    DevTmp<int> d_vmap;
    const int KX = std::min(N / 2 + 1, R + 1);          // x coefficients kept; |ky|, |kz| <= R are the lines the later passes touch
    const int NB = std::min(CH, 32);                     // sub-volumes transformed per launch (work array: NB x N x N x KX complex)
    if (!plan_cached) {
        ref->s_plan.valid = false; ref->s_plan.fw_valid = false;
        if (int rc = ref->s_plan.samples.ensure(S)) return rc;
        if (int rc = ref->s_plan.bandw.ensure(S)) return rc;
        HIPCHK(hipMemcpyAsync(ref->s_plan.samples.p, samples.data(), (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipMemcpyAsync(ref->s_plan.bandw.p, bandw.data(), (size_t)S * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
        {   // where a coefficient (kx, kyi, kzi) of the pruned transform sits in the sample list (k_sva_yz16's z pass emits the samples itself)
            const int KYp = std::min(N, 2 * R + 1);
            std::vector<unsigned> pos((size_t)KX * KYp * KYp, 0x7fffffffu);
            for (int i = 0; i < S; i++) {
                int kx, ky, kz; sva_unpack(samples[i], kx, ky, kz);
                if (kx >= KX) continue;
                const int kyi = ky >= 0 ? ky : ky + KYp, kzi = kz >= 0 ? kz : kz + KYp;
                pos[((size_t)kx * KYp + kyi) * KYp + kzi] = (unsigned)i | (((kx + ky + kz) & 1) ? 0x80000000u : 0u);
            }
            if (int rc = ref->s_plan.pos.ensure(pos.size())) return rc;
            HIPCHK(hipMemcpyAsync(ref->s_plan.pos.p, pos.data(), pos.size() * sizeof(unsigned), hipMemcpyHostToDevice, cur_stream()));
        }
        HIPCHK(hipStreamSynchronize(cur_stream()));        // the host vectors go out of use here
        std::memcpy(ref->s_plan.key, plan_key, sizeof(plan_key)); ref->s_plan.S = S; ref->s_plan.shell_off = shell_off; ref->s_plan.valid = true;
    }
    d_samples.p = ref->s_plan.samples.p; d_bandw.p = ref->s_plan.bandw.p;
    if (int rc = ref->s_f.ensure((size_t)NB * N * N * KX)) return rc;
    if (int rc = ref->s_F.ensure((size_t)CH * S)) return rc;
    d_f.p = ref->s_f.p; d_F.p = ref->s_F.p;
    // box sizes that are multiples of 16 take the two-step transforms (k_sva_x16 / k_sva_yz16): a second work array B[kx][kyi][z]
    const bool fast16 = N % 16 == 0 && getenv("PPM_SVA_GENERIC_FFT") == nullptr;
    const int KY = std::min(N, 2 * R + 1);
    const bool sva_fold = !(getenv("PPM_SVA_FOLD") && atoi(getenv("PPM_SVA_FOLD")) == 0);      // 0: the z pass writes B back and k_sva_gather16 picks the samples (A/B, tests)
    DevTmp<double> d_spart;                              // per-block partial sums of the two-step x pass
    if (fast16) {
        if (int rc = ref->s_g.ensure((size_t)NB * KX * KY * N)) return rc;
        HIPCHK(d_spart.alloc((size_t)2 * NB * ((size_t)N * N / (N <= 256 ? 16 : 8))));
    }
    const size_t CHS = (size_t)CH * (global ? Kc : 1);       // states per chunk: the global search refines Kc candidates per sub-volume
    HIPCHK(d_stats.alloc((size_t)2 * CH)); HIPCHK(d_poses.alloc((size_t)12 * CHS)); HIPCHK(d_delta.alloc(CHS * ncand * 6)); HIPCHK(d_out.alloc(CHS * ncand));
    HIPCHK(d_vmap.alloc(CHS)); HIPCHK(d_partial.alloc(CHS * kSvaParts * (2 * kMaxCand + 1)));
    // the compass search's state on the device (ppm_csp_kernels.h: k_csp_step_*), kept in the handle like the constrained search's
    // (allocating and freeing five more buffers per call cost 7 ms of a 57 ms call: hipFree waits for the device)
    DevBuf<double> &d_acc = ref->c_acc, &d_dtrial = ref->c_dtrial, &d_fpm = ref->c_fpm, &d_delta_t = ref->c_delta_t, &d_tout = ref->c_tmean;
    if (int rc = d_acc.ensure(CHS * 6)) return rc;
    if (int rc = d_dtrial.ensure(CHS * 6)) return rc;
    if (int rc = d_fpm.ensure(CHS * 12)) return rc;
    if (int rc = d_delta_t.ensure(CHS * 6)) return rc;
    if (int rc = d_tout.ensure(CHS)) return rc;
    if (global) {
        std::vector<float> gf(grid_d.begin(), grid_d.end());
        HIPCHK(d_grid.alloc(gf.size())); HIPCHK(d_gscore.alloc((size_t)CH * n_grid));
        HIPCHK(hipMemcpy(d_grid.p, gf.data(), gf.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHK(d_wedges.alloc((size_t)2 * CH));
    const bool two_bufs = !volumes_on_device && n_vol > CH;      // host volumes: the next chunk is uploaded by a helper thread while this one is searched
    if (!volumes_on_device) { if (int rc = ref->s_vols.ensure((size_t)(two_bufs ? 2 : 1) * CH * n3)) return rc; d_vols.p = ref->s_vols.p; }
    SvaWin W; for (int k = 0; k < 3; k++) W.w[k] = cfg->window[k]; W.sigma = cfg->window_sigma;
    SvaEvalP EP;
    EP.cv.cube = ref->cube; EP.cv.NBX = ref->NBX; EP.cv.NBY = ref->NBY; EP.cv.LB = ref->LB; EP.cv.off = ref->B + 1; EP.cv.scale = 1.f;
    EP.samples = d_samples.p; EP.bandw = d_bandw.p; EP.F = d_F.p; EP.S = S; EP.N = N; EP.use_wedge = cfg->use_missing_wedge != 0;
    EP.tabR = ref->B + 4;           // every sample of the band (|k| <= B + 1) and its upper taps
    EP.wedges = d_wedges.p; EP.poses = d_poses.p; EP.delta = d_delta.p; EP.out = d_out.p; EP.vmap = nullptr; EP.partial = d_partial.p;
    std::vector<float> hw((size_t)2 * CH);
    std::vector<double> hdelta, hout;
    double acct_gathers = 0; long acct_sweeps = 0;      // for the roofline: band samples x rotations gathered, summed over the sweeps (wedge-masked samples included)
    mark("set up");
    if (!volumes_on_device) {       // first chunk
        HIPCHK(hipMemcpyAsync(d_vols.p, volumes, (size_t)std::min(CH, n_vol) * n3 * sizeof(float), hipMemcpyHostToDevice, cur_copy()));
        HIPCHK(hipStreamSynchronize(cur_copy()));
    }
    struct Uploader {           // joins on every exit path
        std::thread t; hipError_t err = hipSuccess;
        void join() { if (t.joinable()) t.join(); }
        ~Uploader() { join(); }
    } up;
    for (int c0 = 0, ci = 0; c0 < n_vol; c0 += CH, ci++) {
        const int nb = std::min(CH, n_vol - c0);
        const float *dv = (const float *)volumes + (size_t)c0 * n3;
        if (!volumes_on_device) {
            up.join();
            if (up.err != hipSuccess) return fail(-5, std::string("HIP: ") + hipGetErrorString(up.err) + " while uploading sub-volumes");
            dv = d_vols.p + (size_t)(ci & 1) * (two_bufs ? (size_t)CH * n3 : 0);
            if (c0 + CH < n_vol) {      // the host drives the search of this chunk (a synchronisation per sweep): the copy of the next one gets its own thread and stream
                const int nn = std::min(CH, n_vol - (c0 + CH));
                float *dst = d_vols.p + (size_t)((ci + 1) & 1) * CH * n3;
                const float *src = (const float *)volumes + (size_t)(c0 + CH) * n3;
                const size_t bytes = (size_t)nn * n3 * sizeof(float);
                const int dev = g.device; hipStream_t cs = cur_copy();
                up.err = hipSuccess;
                up.t = std::thread([&up, dst, src, bytes, dev, cs] {
                    hipError_t e = hipSetDevice(dev);
                    if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs);
                    if (e == hipSuccess) e = hipStreamSynchronize(cs);
                    up.err = e;
                });
            }
        }
        for (int v = 0; v < nb; v++) { hw[2 * v] = wedges ? wedges[2 * (size_t)(c0 + v)] : -90.f; hw[2 * v + 1] = wedges ? wedges[2 * (size_t)(c0 + v) + 1] : 90.f; }
        HIPCHK(hipMemcpyAsync(d_wedges.p, hw.data(), (size_t)2 * nb * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipMemsetAsync(d_stats.p, 0, (size_t)2 * nb * sizeof(double), cur_stream()));
        {
            ProfScope ps(PPM_K_PREP);
            if (!fast16) hipLaunchKernelGGL(k_sva_stats, dim3(64, nb), dim3(256), 0, cur_stream(), dv, n3, d_stats.p);       // (the two-step x pass gathers the statistics itself)
            if (int rc = ensure_plan(N)) return rc;
            SvaXP XP; XP.stats = nullptr; XP.out = d_f.p; XP.plan = g.plans[N].plan; XP.n = N; XP.KX = KX; XP.nlines = (long)N * N; XP.W = W;
            XP.L = std::max(1, std::min(16, 7000 / N));
            while (((long)N * N) % XP.L) XP.L--;
            for (int v0 = 0; v0 < nb; v0 += NB) {
                // pruned transform of NB sub-volumes per launch (k_sva_xpass): x pass from the real volumes into [vol][z][y][KX], y pass on
                // that, z pass on |ky| <= R only
                const int m = std::min(NB, nb - v0);
                const long NN2 = (long)N * N;
                if (fast16) {
                    const int L16 = N <= 256 ? 16 : 8;
                    const size_t lds = (size_t)L16 * (N + 1) * sizeof(float2);
                    // mode 1 / 2 of k_sva_x16 for mv sub-volumes, the two k_sva_yz16 passes, the samples picked out of B
                    auto transform = [&](int mode, const float *vols_, double *stats_, int mv, float2 *F_, const double *gstats, const float2 *Fw_) {
                        SvaX16P X; X.vol = vols_; X.stats = stats_; X.A = d_f.p; X.tw = g.plans[N].plan.tw; X.n = N; X.L = L16; X.KX = KX; X.mode = mode;
                        X.nlines = (long)mv * NN2; X.W = W;
                        hipLaunchKernelGGL(k_sva_x16, dim3((unsigned)(X.nlines / L16)), dim3(256), lds, cur_stream(), X);
                        if (mode == 1) hipLaunchKernelGGL(k_sva_stats_sum, dim3(mv), dim3(64), 0, cur_stream(), stats_, (int)(NN2 / L16), (double *)gstats);
                        SvaYZ16P Y; Y.A = d_f.p; Y.B = ref->s_g.p; Y.tw = X.tw; Y.n = N; Y.L = L16; Y.KX = KX; Y.KY = KY; Y.R = R; Y.in_place = 0; Y.nlines = 0;
                        Y.pos = nullptr; Y.F = nullptr; Y.S = S; Y.stats = nullptr; Y.Fw = nullptr;
                        hipLaunchKernelGGL(k_sva_yz16, dim3((unsigned)((long)mv * KX * (N / L16))), dim3(256), lds, cur_stream(), Y);
                        Y.in_place = 1; Y.nlines = (long)mv * KX * KY;
                        if (sva_fold) {         // the z pass emits the band's samples itself
                            Y.pos = ref->s_plan.pos.p; Y.F = F_; Y.stats = gstats; Y.Fw = Fw_;
                            hipLaunchKernelGGL(k_sva_yz16, dim3((unsigned)((Y.nlines + L16 - 1) / L16)), dim3(256), lds, cur_stream(), Y);
                        } else {
                            hipLaunchKernelGGL(k_sva_yz16, dim3((unsigned)((Y.nlines + L16 - 1) / L16)), dim3(256), lds, cur_stream(), Y);
                            hipLaunchKernelGGL(k_sva_gather16, dim3((unsigned)((S + 255) / 256), mv), dim3(256), 0, cur_stream(), ref->s_g.p, d_samples.p, S, N, KX, KY, F_, gstats, Fw_);
                        }
                    };
                    const float wkey[4] = { W.w[0], W.w[1], W.w[2], W.sigma };
                    if (!ref->s_plan.fw_valid || std::memcmp(wkey, ref->s_plan.wkey, sizeof(wkey)) != 0) {     // the window's own transform, once per window
                        if (int rc = ref->s_plan.Fw.ensure(S)) return rc;
                        transform(2, dv, nullptr, 1, ref->s_plan.Fw.p, nullptr, nullptr);
                        std::memcpy(ref->s_plan.wkey, wkey, sizeof(wkey)); ref->s_plan.fw_valid = true;
                    }
                    transform(1, dv + (size_t)v0 * n3, d_spart.p, m, d_F.p + (size_t)v0 * S, d_stats.p + 2 * v0, ref->s_plan.Fw.p);
                    continue;
                }
                XP.vol = dv + (size_t)v0 * n3; XP.stats = d_stats.p + 2 * v0; XP.nlines = (long)m * NN2;
                hipLaunchKernelGGL(k_sva_xpass, dim3((unsigned)((XP.nlines + XP.L - 1) / XP.L)), dim3(256), (size_t)XP.L * N * sizeof(float2), cur_stream(), XP);
                if (int rc = fft_lines_pass(d_f.p, N, (long)m * N * KX, KX, 1, (long)N * KX, KX, 1, false)) return rc;
                if (2 * R + 1 >= N) {
                    if (int rc = fft_lines_pass(d_f.p, N, (long)m * N * KX, (long)N * KX, 1, NN2 * KX, (long)N * KX, 1, false)) return rc;
                } else {
                    if (int rc = fft_lines_pass(d_f.p, N, (long)m * (R + 1) * KX, (long)(R + 1) * KX, 1, NN2 * KX, (long)N * KX, 1, false)) return rc;
                    if (int rc = fft_lines_pass(d_f.p + (size_t)(N - R) * KX, N, (long)m * R * KX, (long)R * KX, 1, NN2 * KX, (long)N * KX, 1, false)) return rc;
                }
                hipLaunchKernelGGL(k_sva_gather, dim3((unsigned)((S + 255) / 256), m), dim3(256), 0, cur_stream(), d_f.p, d_samples.p, S, N, KX, d_F.p + (size_t)v0 * S);
            }
        }
        HIPCHK(hipGetLastError());
        mark("chunk pre-processed");
        std::vector<CUnit> st(nb);
        for (int v = 0; v < nb; v++) { std::memcpy(st[v].N, poses + (size_t)(c0 + v) * 12, 9 * sizeof(double)); std::memcpy(st[v].p, poses + (size_t)(c0 + v) * 12 + 9, 3 * sizeof(double)); }
        std::vector<double> hp;
        // states -> device (poses are per STATE; `vm` maps a state to its sub-volume, null = identity)
        auto upload_states = [&](const std::vector<CUnit> &S_, const std::vector<int> *vm) -> int {
            const int ns_ = (int)S_.size();
            hp.resize((size_t)12 * ns_);
            for (int v = 0; v < ns_; v++) { std::memcpy(&hp[(size_t)12 * v], S_[v].N, 9 * sizeof(double)); std::memcpy(&hp[(size_t)12 * v + 9], S_[v].p, 3 * sizeof(double)); }
            HIPCHK(hipMemcpyAsync(d_poses.p, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
            if (vm) HIPCHK(hipMemcpyAsync(d_vmap.p, vm->data(), vm->size() * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
            EP.vmap = vm ? d_vmap.p : nullptr;
            return 0;
        };
        auto sweep = [&](int ns_, int nc, int nr_, double rb) -> int {
            HIPCHK(hipMemcpyAsync(d_delta.p, hdelta.data(), (size_t)ns_ * nc * 6 * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
            EP.ncand = nc; EP.nrot = nr_; EP.S_used = prefix_of(rb); EP.rmax2 = (float)(rb * rb);
            acct_gathers += (double)ns_ * EP.S_used * (1 + nr_); acct_sweeps++;
            {
                ProfScope ps(PPM_K_LOCAL);
                // A compass sweep runs best at TWO blocks per CU (8 waves): the seven rotations of a sample patch touch almost the same lines of the
                // reference, and with 16-20 patches in flight per CU the 32 KB L1 keeps none of them (9.6 L2 requests per load instruction;
                // search 0.084 ms per sub-volume at 4-5 blocks, 0.080 at 3, 0.075 at 2, 0.113 at 1: scripts/ab_sva.sh).  The blocks per CU are
                // set through the size of the dynamic LDS request (PPM_SVA_BLOCKS_PER_CU overrides; 0 = whatever the registers allow).
                int bpc = nr_ == 6 ? 2 : 0;
                if (const char *e = getenv("PPM_SVA_BLOCKS_PER_CU")) bpc = atoi(e);
                size_t tab_lds = cube_tab_bytes(EP.tabR);
                if (bpc > 0 && bpc < 8) tab_lds = std::max(tab_lds, (size_t)(160 * 1024 / (bpc + 1) + 1024) & ~(size_t)1023);      // more than a (bpc + 1)-th of the CU's LDS
                if (tab_lds > (size_t)64 * 1024) tab_lds = (size_t)64 * 1024;
                if (nr_ == 0) hipLaunchKernelGGL(k_sva_eval<0>, dim3(ns_, kSvaParts), dim3(256), tab_lds, cur_stream(), EP);
                else if (nr_ == 6) hipLaunchKernelGGL(k_sva_eval<6>, dim3(ns_, kSvaParts), dim3(256), tab_lds, cur_stream(), EP);
                else return fail(-22, "ppm_sva_align: a sweep has 0 or 6 rotated candidates");
                hipLaunchKernelGGL(k_sva_finish, dim3((unsigned)((ns_ * nc + 255) / 256)), dim3(256), 0, cur_stream(), EP.partial, ns_, nc, nr_, d_out.p);
            }
            HIPCHK(hipGetLastError());
            hout.resize((size_t)ns_ * nc);
            HIPCHK(hipMemcpyAsync(hout.data(), d_out.p, hout.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
            HIPCHK(hipStreamSynchronize(cur_stream()));
            return 0;
        };
        // `Tn` compass iterations of all states at once, steps halved after each: state and decisions on the device (k_csp_step_*), six
        // launches per iteration enqueued back to back, the poses come back once at the end
        auto launch_eval = [&](int ns_, int nc, int nr_, double rb, const double *delta, double *out) -> int {
            EP.delta = delta; EP.ncand = nc; EP.nrot = nr_; EP.S_used = prefix_of(rb); EP.rmax2 = (float)(rb * rb);
            acct_gathers += (double)ns_ * EP.S_used * (1 + nr_); acct_sweeps++;
            ProfScope ps(PPM_K_LOCAL);
            int bpc = nr_ == 6 ? 2 : 0;                                    // two blocks per CU for a compass sweep (see `sweep`)
            if (const char *e = getenv("PPM_SVA_BLOCKS_PER_CU")) bpc = atoi(e);
            size_t tab_lds = cube_tab_bytes(EP.tabR);
            if (bpc > 0 && bpc < 8) tab_lds = std::max(tab_lds, (size_t)(160 * 1024 / (bpc + 1) + 1024) & ~(size_t)1023);
            if (tab_lds > (size_t)64 * 1024) tab_lds = (size_t)64 * 1024;
            if (nr_ == 0) hipLaunchKernelGGL(k_sva_eval<0>, dim3(ns_, kSvaParts), dim3(256), tab_lds, cur_stream(), EP);
            else if (nr_ == 6) hipLaunchKernelGGL(k_sva_eval<6>, dim3(ns_, kSvaParts), dim3(256), tab_lds, cur_stream(), EP);
            else return fail(-22, "ppm_sva_align: a sweep has 0 or 6 rotated candidates");
            hipLaunchKernelGGL(k_sva_finish, dim3((unsigned)((ns_ * nc + 255) / 256)), dim3(256), 0, cur_stream(), EP.partial, ns_, nc, nr_, out);
            return 0;
        };
        auto compass = [&](std::vector<CUnit> &S_, const std::vector<int> *vm, const int *en_, const double *tol_, double &ha, double &hs, int Tn) -> int {
            const int ns_ = (int)S_.size();
            const int nrot_ = en_[0] ? 6 : 0, nsh_ = en_[3] ? 6 : 0, nc_ = 1 + nrot_ + nsh_;
            if (nc_ == 1 || ns_ == 0 || Tn <= 0) return 0;
            if (int rc = upload_states(S_, vm)) return rc;
            std::vector<double> hacc((size_t)ns_ * 6);
            for (int v = 0; v < ns_; v++) std::memcpy(&hacc[(size_t)v * 6], S_[v].acc, 6 * sizeof(double));
            HIPCHK(hipMemcpyAsync(d_acc.p, hacc.data(), hacc.size() * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
            auto band_of = [&](double ha_, double hs_) {
                if (bf < 0) return rband;
                double d = 0;
                if (en_[0]) d = rm_px * ha_ * kPi / 180.0;
                if (en_[3] && hs_ > d) d = hs_;
                if (!(d > 0)) return rband;
                double rit = bf * N / (2.0 * kPi * d);
                if (rit < 4.0) rit = 4.0;
                return rit < rband ? rit : rband;
            };
            CspStepP SP;
            SP.kind = PPM_CSP_PARTICLES; SP.n_active = ns_; SP.ncand = nc_; SP.active = nullptr; SP.unit_slot = nullptr;
            for (int i = 0; i < 6; i++) { SP.en[i] = en_[i]; SP.tol[i] = tol_[i]; }
            SP.mean = d_out.p; SP.tmean = d_tout.p; SP.acc = d_acc.p; SP.dtrial = d_dtrial.p; SP.fpm = d_fpm.p;
            SP.delta_c = d_delta.p; SP.delta_t = d_delta_t.p; SP.Nmat = d_poses.p; SP.pshift = d_poses.p + 9; SP.tl = nullptr; SP.nstride = 12; SP.pstride = 12;
            const unsigned gstep = (unsigned)((ns_ + 127) / 128);
            SP.ha = ha; SP.hs = hs; SP.ha_next = ha; SP.hs_next = hs;
            hipLaunchKernelGGL(k_csp_step_init, dim3(gstep), dim3(128), 0, cur_stream(), SP);
            for (int it = 0; it < Tn; it++) {
                const double rb = band_of(ha, hs);
                SP.ha = ha; SP.hs = hs; SP.ha_next = 0.5 * ha; SP.hs_next = 0.5 * hs;
                if (int rc = launch_eval(ns_, nc_, nrot_, rb, d_delta.p, d_out.p)) return rc;
                hipLaunchKernelGGL(k_csp_step_trial, dim3(gstep), dim3(128), 0, cur_stream(), SP);
                if (int rc = launch_eval(ns_, 1, 0, rb, d_delta_t.p, d_tout.p)) return rc;
                hipLaunchKernelGGL(k_csp_step_accept, dim3(gstep), dim3(128), 0, cur_stream(), SP);
                ha *= 0.5; hs *= 0.5;
            }
            HIPCHK(hipGetLastError());
            EP.delta = d_delta.p;
            hp.resize((size_t)12 * ns_);
            HIPCHK(hipMemcpyAsync(hp.data(), d_poses.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
            HIPCHK(hipMemcpyAsync(hacc.data(), d_acc.p, hacc.size() * sizeof(double), hipMemcpyDeviceToHost, cur_stream()));
            HIPCHK(hipStreamSynchronize(cur_stream()));
            for (int v = 0; v < ns_; v++) {
                std::memcpy(S_[v].N, &hp[(size_t)12 * v], 9 * sizeof(double)); std::memcpy(S_[v].p, &hp[(size_t)12 * v + 9], 3 * sizeof(double));
                std::memcpy(S_[v].acc, &hacc[(size_t)v * 6], 6 * sizeof(double));
            }
            return 0;
        };
        // scores of all states at the full band -> hout[state]
        auto final_scores = [&](std::vector<CUnit> &S_, const std::vector<int> *vm) -> int {
            if (int rc = upload_states(S_, vm)) return rc;
            hdelta.assign((size_t)S_.size() * 6, 0.0);
            return sweep((int)S_.size(), 1, 0, rband);
        };
        if (!global) {
            double ha = ha0, hs = hs0;
            if (int rc = compass(st, nullptr, en, tol, ha, hs, T)) return rc;
        } else {
            // ---- rotations ranked by the amplitude correlation on the coarse band (k_sva_global)
            hp.resize((size_t)12 * nb);
            for (int v = 0; v < nb; v++) { std::memcpy(&hp[(size_t)12 * v], st[v].N, 9 * sizeof(double)); std::memcpy(&hp[(size_t)12 * v + 9], st[v].p, 3 * sizeof(double)); }
            HIPCHK(hipMemcpyAsync(d_poses.p, hp.data(), (size_t)12 * nb * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
            SvaGlobalP GP;
            GP.cv = EP.cv; GP.samples = d_samples.p; GP.bandw = d_bandw.p; GP.F = d_F.p; GP.S = S; GP.N = N; GP.S_used = prefix_of(rg); GP.rmax2 = (float)(rg * rg);
            GP.use_wedge = EP.use_wedge; GP.wedges = d_wedges.p; GP.poses = d_poses.p; GP.grid = d_grid.p; GP.n_grid = n_grid; GP.RC = 8; GP.score = d_gscore.p;
            { ProfScope ps(PPM_K_GLOBAL); hipLaunchKernelGGL(k_sva_global, dim3((n_grid + GP.RC - 1) / GP.RC, nb), dim3(256), 0, cur_stream(), GP); }
            HIPCHK(hipGetLastError());
            std::vector<float> gsc((size_t)nb * n_grid);
            HIPCHK(hipMemcpyAsync(gsc.data(), d_gscore.p, gsc.size() * sizeof(float), hipMemcpyDeviceToHost, cur_stream()));
            HIPCHK(hipStreamSynchronize(cur_stream()));
            // ---- top-K per sub-volume (ties -> lower grid index) as states of their own, from the start shift
            std::vector<CUnit> cand; std::vector<int> vm; cand.reserve((size_t)nb * Kc); vm.reserve((size_t)nb * Kc);
            std::vector<int> order(n_grid);
            for (int v = 0; v < nb; v++) {
                const float *sc_ = &gsc[(size_t)v * n_grid];
                for (int q = 0; q < n_grid; q++) order[q] = q;
                std::partial_sort(order.begin(), order.begin() + Kc, order.end(), [&](int x, int y) { return sc_[x] > sc_[y] || (sc_[x] == sc_[y] && x < y); });
                for (int k = 0; k < Kc; k++) {
                    CUnit c = st[v];
                    double Nq[9]; mat_mul3h(st[v].N, &grid_d[(size_t)order[k] * 9], Nq); std::memcpy(c.N, Nq, sizeof(Nq));
                    cand.push_back(c); vm.push_back(v);
                }
            }
            double ha = 0.5 * gstep, hs = 0.5 * cfg->tol_shift;
            if (int rc = compass(cand, &vm, eng, tolg, ha, hs, 2)) return rc;
            if (int rc = final_scores(cand, &vm)) return rc;
            for (int v = 0; v < nb; v++) {
                int bk = 0;
                for (int k = 1; k < Kc; k++) if (hout[(size_t)v * Kc + k] > hout[(size_t)v * Kc + bk]) bk = k;
                st[v] = cand[(size_t)v * Kc + bk];
            }
            ha = 0.25 * gstep; hs = 0.25 * cfg->tol_shift;
            const double m = std::max(ha, hs);
            int Tf = m > steptol ? (int)std::ceil(std::log(m / steptol) / std::log(2.0)) : 0; Tf = std::min(12, Tf);
            if (int rc = compass(st, nullptr, eng, tolg, ha, hs, Tf)) return rc;
        }
        mark("chunk searched");
        if (int rc = final_scores(st, nullptr)) return rc;
        for (int v = 0; v < nb; v++) {
            std::memcpy(poses + (size_t)(c0 + v) * 12, st[v].N, 9 * sizeof(double)); std::memcpy(poses + (size_t)(c0 + v) * 12 + 9, st[v].p, 3 * sizeof(double));
            if (scores) scores[c0 + v] = hout[v];
        }
        if (avg) {
            if (int rc = sva_insert_device(avg, cfg, dv, nb, wedges ? wedges + 2 * (size_t)c0 : nullptr, poses + (size_t)c0 * 12, index ? index + c0 : nullptr, c0)) return rc;
            mark("chunk averaged");
        }
    }
    // ppm_refine_last_counts after an alignment: grid rotations of the global search, sweeps (k_sva_eval launches), samples of the band
    // (half space, before the wedge), band samples x gathered rotations per sub-volume summed over the sweeps
    ref->last_counts[0] = n_grid; ref->last_counts[1] = acct_sweeps; ref->last_counts[2] = S; ref->last_counts[3] = (long)(acct_gathers / n_vol);
    return 0;
}

extern "C" int ppm_sva_align(ppm_ref_t *ref, const ppm_sva_cfg *cfg, const void *volumes, int volumes_on_device, int n_vol, const float *wedges,
                             double *poses, double *scores) {
    return sva_align_impl(ref, nullptr, cfg, volumes, volumes_on_device, n_vol, wedges, poses, scores, nullptr);
}

extern "C" int ppm_sva_align_average(ppm_ref_t *ref, ppm_accum_t *acc, const ppm_sva_cfg *cfg, const void *volumes, int volumes_on_device, int n_vol,
                                     const float *wedges, double *poses, double *scores, const long *index) {
    if (!acc) return fail(-22, "null accumulator");
    if (cfg && acc->N != cfg->box) return fail(-22, "sub-volume box differs from the accumulator's box");
    return sva_align_impl(ref, acc, cfg, volumes, volumes_on_device, n_vol, wedges, poses, scores, index);
}

// ------------------------------------------------------------------------------ sub-tomogram average
// include/ppm.h: ppm_sva_insert.  Per batch of <= 32 sub-volumes: the FULL 3-D transforms (the pruned passes of the alignment with
// the band at Nyquist; normalisation (v - mean) / sigma applied through the statistics the x pass gathers), then one k_sva_insert
// launch that gathers them into the accumulator.
// one batch-wise pass over DEVICE-resident sub-volumes (d_vols: n_vol x N^3 floats); runs on the caller's current stream scope
static int sva_insert_device(ppm_accum_t *a, const ppm_sva_cfg *cfg, const float *d_vols, int n_vol, const float *wedges, const double *poses,
                             const long *index, long index_base) {
    const int N = cfg->box;
    if (!box_ok(N) || N != a->N) return fail(-22, "sub-volume box differs from the accumulator's box (even, 32..512, prime factors 2, 3, 5, 7)");
    if (a->nsym != 1) return fail(-22, "sub-tomogram averaging needs a C1 accumulator");
    const size_t n3 = (size_t)N * N * N;
    const int KX = N / 2 + 1, KY = N, NB = std::min(n_vol, kSvaInsBatch);
    const bool fast16 = N % 16 == 0 && getenv("PPM_SVA_GENERIC_FFT") == nullptr;
    if (int rc = ensure_plan(N)) return rc;
    if (int rc = a->s_f.ensure((size_t)NB * N * N * KX)) return rc;
    if (fast16) if (int rc = a->s_g.ensure((size_t)NB * KX * KY * N)) return rc;
    DevTmp<double> d_spart, d_stats, d_poses; DevTmp<float> d_wedges; DevTmp<int> d_half;
    const int L16 = N <= 256 ? 16 : 8;
    HIPCHK(d_spart.alloc((size_t)2 * NB * ((size_t)N * N / L16 + 1))); HIPCHK(d_stats.alloc((size_t)2 * NB)); HIPCHK(d_poses.alloc((size_t)12 * NB));
    HIPCHK(d_wedges.alloc((size_t)2 * NB)); HIPCHK(d_half.alloc(NB));
    SvaWin W; for (int k = 0; k < 3; k++) W.w[k] = 0.f;       // the average is made of the whole sub-volumes: no window, no band-pass
    W.sigma = 0.f;
    std::vector<float> hw((size_t)2 * NB); std::vector<int> hh(NB);
    long added[2] = { 0, 0 };
    for (int v0 = 0; v0 < n_vol; v0 += NB) {
        const int m = std::min(NB, n_vol - v0);
        const float *dv = d_vols + (size_t)v0 * n3;
        for (int v = 0; v < m; v++) {
            hw[2 * v] = wedges ? wedges[2 * (size_t)(v0 + v)] : -90.f; hw[2 * v + 1] = wedges ? wedges[2 * (size_t)(v0 + v) + 1] : 90.f;
            const long key = index ? index[v0 + v] : index_base + (long)(v0 + v);
            hh[v] = (int)(((key % 2) + 2) % 2);
            added[hh[v]]++;
        }
        HIPCHK(hipMemcpyAsync(d_wedges.p, hw.data(), (size_t)2 * m * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipMemcpyAsync(d_half.p, hh.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, cur_stream()));
        HIPCHK(hipMemcpyAsync(d_poses.p, poses + (size_t)v0 * 12, (size_t)12 * m * sizeof(double), hipMemcpyHostToDevice, cur_stream()));
        const long NN2 = (long)N * N;
        SvaInsP IP;
        {
            ProfScope ps(PPM_K_PREP);
            if (fast16) {
                const size_t lds = (size_t)L16 * (N + 1) * sizeof(float2);
                SvaX16P X; X.vol = dv; X.stats = d_spart.p; X.A = a->s_f.p; X.tw = g.plans[N].plan.tw; X.n = N; X.L = L16; X.KX = KX; X.mode = 1; X.nlines = (long)m * NN2; X.W = W;
                hipLaunchKernelGGL(k_sva_x16, dim3((unsigned)(X.nlines / L16)), dim3(256), lds, cur_stream(), X);
                hipLaunchKernelGGL(k_sva_stats_sum, dim3(m), dim3(64), 0, cur_stream(), d_spart.p, (int)(NN2 / L16), d_stats.p);
                SvaYZ16P Y; Y.A = a->s_f.p; Y.B = a->s_g.p; Y.tw = X.tw; Y.n = N; Y.L = L16; Y.KX = KX; Y.KY = KY; Y.R = N / 2; Y.in_place = 0; Y.nlines = 0;
                Y.pos = nullptr; Y.F = nullptr; Y.S = 0; Y.stats = nullptr; Y.Fw = nullptr;
                hipLaunchKernelGGL(k_sva_yz16, dim3((unsigned)((long)m * KX * (N / L16))), dim3(256), lds, cur_stream(), Y);
                Y.in_place = 1; Y.nlines = (long)m * KX * KY;
                hipLaunchKernelGGL(k_sva_yz16, dim3((unsigned)((Y.nlines + L16 - 1) / L16)), dim3(256), lds, cur_stream(), Y);
                IP.T = a->s_g.p; IP.layout = 1; IP.stats = d_stats.p;
            } else {
                HIPCHK(hipMemsetAsync(d_stats.p, 0, (size_t)2 * m * sizeof(double), cur_stream()));
                hipLaunchKernelGGL(k_sva_stats, dim3(64, m), dim3(256), 0, cur_stream(), dv, n3, d_stats.p);
                SvaXP XP; XP.vol = dv; XP.stats = d_stats.p; XP.out = a->s_f.p; XP.plan = g.plans[N].plan; XP.n = N; XP.KX = KX; XP.nlines = (long)m * NN2; XP.W = W;
                XP.L = std::max(1, std::min(16, 7000 / N));
                while (NN2 % XP.L) XP.L--;
                hipLaunchKernelGGL(k_sva_xpass, dim3((unsigned)((XP.nlines + XP.L - 1) / XP.L)), dim3(256), (size_t)XP.L * N * sizeof(float2), cur_stream(), XP);
                if (int rc = fft_lines_pass(a->s_f.p, N, (long)m * N * KX, KX, 1, (long)N * KX, KX, 1, false)) return rc;
                if (int rc = fft_lines_pass(a->s_f.p, N, (long)m * N * KX, (long)N * KX, 1, NN2 * KX, (long)N * KX, 1, false)) return rc;
                IP.T = a->s_f.p; IP.layout = 0; IP.stats = nullptr;
            }
        }
        IP.N = N; IP.KX = KX; IP.KY = KY; IP.nv = m; IP.poses = d_poses.p; IP.wedges = d_wedges.p; IP.half = d_half.p;
        IP.use_wedge = cfg->use_missing_wedge != 0; IP.scale = 1.0f / (float)N; IP.acc = a->acc;
        {
            ProfScope ps(PPM_K_INSERT);
            hipLaunchKernelGGL(k_sva_insert, dim3((unsigned)((NN2 * (N / 2 + 1) + 255) / 256)), dim3(256), 0, cur_stream(), IP);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(cur_stream()));          // the host tables of the batch are reused
        // the counters follow every completed batch: after an error in a later one they still say which sub-volumes are in the sums
        for (int h = 0; h < 2; h++) { ppm_accum_set_count(a, h, a->counts[h] + added[h]); added[h] = 0; }
    }
    return 0;
}

extern "C" int ppm_sva_insert(ppm_accum_t *a, const ppm_sva_cfg *cfg, const void *volumes, int volumes_on_device, int n_vol, const float *wedges,
                              const double *poses, const long *index) {
    if (!g.inited) return fail(-1, "ppm_init has not been called");
    if (!a || !cfg || !volumes || !poses) return fail(-22, "null argument");
    StreamScope ss_(a->stream, a->copy);
    if (n_vol <= 0) return 0;
    if (volumes_on_device) return sva_insert_device(a, cfg, (const float *)volumes, n_vol, wedges, poses, index, 0);
    const int N = cfg->box;
    if (!box_ok(N) || N != a->N) return fail(-22, "sub-volume box differs from the accumulator's box (even, 32..512, prime factors 2, 3, 5, 7)");
    const size_t n3 = (size_t)N * N * N;
    const int NB = std::min(n_vol, kSvaInsBatch);
    if (int rc = a->s_vols.ensure((size_t)NB * n3)) return rc;
    for (int v0 = 0; v0 < n_vol; v0 += NB) {            // host volumes: staged batch by batch
        const int m = std::min(NB, n_vol - v0);
        HIPCHK(hipMemcpyAsync(a->s_vols.p, (const float *)volumes + (size_t)v0 * n3, (size_t)m * n3 * sizeof(float), hipMemcpyHostToDevice, cur_stream()));
        if (int rc = sva_insert_device(a, cfg, a->s_vols.p, m, wedges ? wedges + 2 * (size_t)v0 : nullptr, poses + (size_t)v0 * 12, index ? index + v0 : nullptr, v0)) return rc;
    }
    return 0;
}
