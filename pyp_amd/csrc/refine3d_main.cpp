// refine3d — native drop-in for the program PYP scripts at src/pyp/refine/frealign/frealign.py:3918-3994
// ("<dir>/refine3d << eot >> log ... eot": 50 answers on stdin).
//
// The reference's refine3d is a compiled program; so is this one.  It covers the call PYP makes by default (a .cistem table in
// the standard column order, a float32 stack and reference, no statistics weighting, no priors, no matching projections, no 2-D
// focus mask) and does nothing but parse, prepare the reference (ppm_reference_create_padded), stream the particle range from
// the stack file into libpypmatch (include/ppm.h: ppm_host_read -> ppm_device_upload -> ppm_refine_batch) and write the output
// tables.  Everything else — the `.par` surface, the other answers, every input it would have to refuse — is handed to
// bin/refine3d.py (pyp_amd/surface/cli.py:refine3d_main) as a child process with the same stdin, BEFORE the GPU is touched, so
// that behaviour and messages have one definition.
//
// Built by pyp_amd/csrc/Makefile into bin/refine3d (g++, no HIP: the C ABI only).
#include "dropin_common.h"

using namespace dropin;

namespace {

enum { C_POS = 0, C_PSI = 1, C_THETA = 2, C_PHI = 3, C_SHX = 4, C_SHY = 5, C_SCORE = 14 };

[[noreturn]] void fall_back(const std::string &input) { hand_to_python("refine3d.py", input); }

// answer names in script order (pyp_amd/surface/prompts.py:REFINE3D_CISTEM)
const char *kNames[50] = { "stack", "input_params", "global_stats", "reference", "statistics", "use_statistics", "use_priors", "match_out", "output_params",
    "output_changes", "symmetry", "first", "last", "fraction", "pixel_size", "molecular_mass", "inner_radius", "outer_radius", "res_low", "res_high",
    "res_signed_cc", "res_classification", "search_mask_radius", "res_search", "angular_step", "top_hits", "search_range_x", "search_range_y", "focus_x",
    "focus_y", "focus_z", "focus_r", "defocus_range", "defocus_step", "padding", "global_search", "local_refine", "refine_psi", "refine_theta", "refine_phi",
    "refine_x", "refine_y", "calc_match", "mask_2d", "refine_defocus", "normalize", "invert", "exclude_edges", "normalize_reference", "threshold_reference" };

// a float32 MRC volume (mode 2, little-endian), nx = ny = nz = n
bool read_volume(const std::string &path, int n, std::vector<float> &vol) {
    MrcHead h;
    if (!read_mrc_head(path, h) || h.mode != 2 || h.nx != n || h.ny != n || h.nz != n) return false;
    vol.resize((size_t)n * n * n);
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    size_t done = 0, want = vol.size() * 4;
    while (done < want) {
        ssize_t r = pread(fd, (char *)vol.data() + done, want - done, (off_t)(h.offset + done));
        if (r <= 0) { close(fd); return false; }
        done += (size_t)r;
    }
    close(fd);
    return true;
}

}  // namespace

int main() {
    const auto t0 = Clock::now();
    const std::string input = read_all_stdin();
    if (const char *e = getenv("PPM_NATIVE")) if (!strcmp(e, "0")) fall_back(input);
    const std::vector<std::string> a = read_answers(input);
    if (a.size() < 50 || !ends_with(a[1], ".cistem")) fall_back(input);          // the 45-answer .par surface lives in Python
    for (int k = 0; k < 11; k++) if (a[k].empty()) fall_back(input);
    double num[50] = { 0 }; bool flag[50] = { false };
    static const int kNum[] = { 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34 };
    static const int kBool[] = { 5, 6, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49 };
    bool ok = true;
    for (int k : kNum) ok = ok && parse_num(a[k], num[k]);
    for (int k : kBool) ok = ok && parse_bool(a[k], flag[k]);
    if (!ok) fall_back(input);
    const std::string stack = a[0], params = a[1], reference = a[3], out_params = a[8], out_changes = a[9], symmetry = a[10];
    const double first = num[11], last = num[12], fraction = num[13], px = num[14], inner_radius = num[16], padding = num[34];
    const bool use_stats = flag[5], use_priors = flag[6], calc_match = flag[42], mask_2d = flag[43];
    const int pad = (int)std::lround(padding);
    // what the Python implementation owns (non-default answers, refusals and their messages)
    if (use_stats || use_priors || calc_match || mask_2d || flag[47] || flag[48] || flag[49] || inner_radius != 0.0 || fraction != 1.0 || num[21] < 0 ||
        std::fabs(padding - pad) > 1e-6 || (pad != 1 && pad != 2 && pad != 4) || first < 1 || last < first || px <= 0 || !ends_with(out_params, ".cistem") ||
        !exists(stack) || !exists(params) || !exists(reference))
        fall_back(input);
    MrcHead mh, rh;
    if (!cistem_is_standard(params) || !read_mrc_head(stack, mh) || mh.mode != 2 || mh.nx != mh.ny || !read_mrc_head(reference, rh) || rh.mode != 2 ||
        rh.nx != mh.nx || rh.ny != mh.nx || rh.nz != mh.nx || mh.nx * pad > 512)
        fall_back(input);
    const long ifirst = (long)first, ilast = (long)last;
    const int box = mh.nx;
    const size_t sec = (size_t)box * box * 4;

    // ---- from here on the GPU is in use: no more hand-overs.  The lock comes first (PYP may start several processes per node:
    // a waiting process holds neither a context nor page-locked memory); then the device, the page-locked staging buffers and the
    // reference are brought up by threads of their own while the parameter file is read.
    setenv("PPM_SYNC", "block", 0);
    const int dev = getenv("PPM_DEVICE") ? atoi(getenv("PPM_DEVICE")) : 0;
    const int lockfd = gpu_lock(dev);
    long chunk_mb = 64, call_mb = 512;         // compute-bound: 64 MB chunks, 512 MB per refinement call (scripts/dropin_ab.py)
    if (const char *e = getenv("PPM_IO_CHUNK_MB")) chunk_mb = std::max(1L, atol(e));
    const size_t pin_bytes = std::min(std::max((size_t)16, ((size_t)chunk_mb << 20) / sec), (size_t)(ilast - ifirst + 1)) * sec;
    Stream st;
    std::mutex up_m; std::condition_variable up_cv; int up_stage = 0; std::string up_err;      // 1 = device ready, 2 + k = pinned[k] ready, 99 = failed
    std::atomic<int> want_pinned{3};
    ppm_ref_t *ref = nullptr; bool ref_done = false; std::string ref_err;
    const auto t_dev = Clock::now();
    double init_s = 0, ref_s = 0;
    auto fail_up = [&](const char *m) { { std::lock_guard<std::mutex> lk(up_m); up_err = m && *m ? m : "ERROR: device start-up failed"; up_stage = 99; } up_cv.notify_all(); };
    std::thread starter([&] {
        if (ppm_init(dev) != 0) return fail_up(ppm_last_error());
        init_s = since(t_dev);
        { std::lock_guard<std::mutex> lk(up_m); up_stage = 1; } up_cv.notify_all();
        for (int k = 0; k < 3 && k < want_pinned.load(); k++) {
            st.pinned[k] = ppm_host_alloc(pin_bytes);
            if (!st.pinned[k]) return fail_up(ppm_last_error());
            { std::lock_guard<std::mutex> lk(up_m); up_stage = 2 + k; } up_cv.notify_all();
        }
    });
    auto wait_stage = [&](int s) {
        std::unique_lock<std::mutex> lk(up_m);
        up_cv.wait(lk, [&] { return up_stage >= s; });
        return up_stage != 99;
    };
    std::thread refmaker([&] {
        std::vector<float> vol;
        std::string err;
        if (!read_volume(reference, box, vol)) err = "ERROR: refine3d: cannot read the reference " + reference;
        else if (!wait_stage(1)) err = "";                       // the start-up's own message is reported
        else {
            ref = ppm_reference_create_padded(vol.data(), box, (float)(box / 2.0), pad);
            if (!ref) err = ppm_last_error();
        }
        ref_s = since(t_dev);
        { std::lock_guard<std::mutex> lk(up_m); ref_err = err; ref_done = true; } up_cv.notify_all();
    });
    auto wait_ref = [&] { std::unique_lock<std::mutex> lk(up_m); up_cv.wait(lk, [&] { return ref_done; }); };
    auto bail = [&](const std::string &msg) { wait_stage(1); starter.join(); wait_ref(); refmaker.join(); die(msg); };      // never exit in the middle of the start-up

    std::vector<double> rows; long nrows = 0;
    if (!read_cistem(params, rows, nrows)) bail("ERROR: " + params + ": binary file is broken");
    std::vector<double> rin;
    for (long i = 0; i < nrows; i++) {
        const double pos = rows[(size_t)i * 32 + C_POS];
        if (pos >= ifirst && pos <= ilast) rin.insert(rin.end(), rows.begin() + (size_t)i * 32, rows.begin() + (size_t)(i + 1) * 32);
    }
    const long n = (long)(rin.size() / 32);
    if (n == 0) bail("ERROR: no rows with POSITION_IN_STACK in " + std::to_string(ifirst) + ".." + std::to_string(ilast));
    bool contiguous = true;
    double pmax = 0, pmin = 1e300;
    for (long i = 0; i < n; i++) {
        const double pos = rin[(size_t)i * 32 + C_POS];
        pmax = std::max(pmax, pos); pmin = std::min(pmin, pos);
        if (i && pos != rin[(size_t)(i - 1) * 32 + C_POS] + 1) contiguous = false;
    }
    if (pmax > mh.nz || pmin < 1) bail("ERROR: " + stack + ": stack has " + std::to_string(mh.nz) + " images, rows ask for " + std::to_string((long)pmax));

    printf("\n        **   Welcome to Refine3D (MI355X / libpypmatch, native)   **\n\n");
    for (int k = 0; k < 50; k++) printf("%-28s: %s\n", kNames[k], a[k].c_str());
    // ---- ppm_refine_cfg from the answers (pyp_amd/surface/cli.py:refine_cfg_from_answers)
    ppm_refine_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.box = box; cfg.pixel_size = (float)px; cfg.molecular_mass_kda = (float)num[15]; cfg.mask_radius = (float)num[17];
    cfg.res_low = (float)num[18]; cfg.res_high = (float)num[19]; cfg.res_signed_cc = (float)num[20]; cfg.res_classification = (float)num[21];
    cfg.search_mask_radius = (float)num[22]; cfg.res_search = (float)(num[23] != 0 ? num[23] : num[19]); cfg.angular_step = (float)num[24];
    cfg.top_hits = (int)num[25]; cfg.search_range_x = (float)num[26]; cfg.search_range_y = (float)num[27];
    cfg.defocus_range = (float)num[32]; cfg.defocus_step = (float)num[33];
    cfg.global_search = flag[35]; cfg.local_refine = flag[36];
    cfg.refine_psi = flag[37]; cfg.refine_theta = flag[38]; cfg.refine_phi = flag[39]; cfg.refine_x = flag[40]; cfg.refine_y = flag[41];
    cfg.refine_defocus = flag[44]; cfg.normalize = flag[45]; cfg.invert = flag[46];
    snprintf(cfg.symmetry, sizeof cfg.symmetry, "%.7s", symmetry.c_str());
    const auto t1 = Clock::now();

    // ---- reader -> uploader -> refinement
    const long chunk = std::max(1L, std::min(n, (long)(pin_bytes / sec)));
    const long nchunks = (n + chunk - 1) / chunk;
    const long group = std::max(1L, std::min(nchunks, (long)(((size_t)call_mb << 20) / ((size_t)chunk * sec))));
    st.n = n; st.chunk = chunk; st.group = group; st.sec = sec; st.contiguous = contiguous;
    st.npin = (int)std::min(3L, nchunks); st.ndev = (int)std::min(2L, (nchunks + group - 1) / group);
    want_pinned = st.npin;
    st.nread = getenv("PPM_IO_THREADS") ? std::max(1, std::min(16, atoi(getenv("PPM_IO_THREADS")))) : 8;
    st.fd = open(stack.c_str(), O_RDONLY);
    if (st.fd < 0) bail("ERROR: refine3d: cannot open " + stack);
    st.img_off = [&](long i) { return mh.offset + (long long)((long)rin[(size_t)i * 32 + C_POS] - 1) * (long long)sec; };
    st.wait_pinned = [&](int slot) { return wait_stage(2 + slot); };
    if (!wait_stage(1)) { starter.join(); wait_ref(); refmaker.join(); die(up_err); }
    st.start();
    std::vector<double> rout((size_t)n * 32);
    double t_comp = 0, w_data = 0; long ncalls = 0;
    auto t2 = Clock::now();
    bool have_ref = false;
    for (long lo = 0; lo < n;) {
        auto ta = Clock::now();
        Stream::Item it;
        const bool got = st.next(it);
        if (!have_ref) {                       // the reference was prepared while the first images arrived
            wait_ref();
            have_ref = true;
            t2 = Clock::now();
            if (!ref) { st.abort(); starter.join(); refmaker.join(); die(!ref_err.empty() ? ref_err : up_err); }
        }
        if (!got) { starter.join(); refmaker.join(); die(!up_err.empty() ? up_err : (!st.err.empty() ? st.err : std::string("ERROR: refine3d: reading or uploading the particle stack failed"))); }
        auto tb = Clock::now();
        if (ppm_refine_batch(ref, &cfg, st.dbuf[it.slot], 1, (int)(it.hi - it.lo), rin.data() + (size_t)it.lo * 32, rout.data() + (size_t)it.lo * 32) != 0) {
            st.abort(); starter.join(); refmaker.join(); die(ppm_last_error());
        }
        st.release(it.slot);
        w_data += secs(ta, tb); t_comp += since(tb); ncalls++;
        lo = it.hi;
    }
    st.join(); starter.join(); refmaker.join();
    close(st.fd);
    const auto t3 = Clock::now();
    const std::string note = ppm_refine_note(ref);
    ppm_reference_destroy(ref);
    for (void *p : st.dbuf) if (p) ppm_device_free(p);
    gpu_unlock(lockfd);
    if (!note.empty()) printf("\n%s\n", note.c_str());
    std::vector<double> changes((size_t)n * 32);
    for (long i = 0; i < n; i++) {
        for (int c = 0; c < 32; c++) changes[(size_t)i * 32 + c] = rout[(size_t)i * 32 + c] - rin[(size_t)i * 32 + c];
        changes[(size_t)i * 32 + C_POS] = rin[(size_t)i * 32 + C_POS];
    }
    if (!write_cistem(out_params, rout.data(), n)) die("ERROR: refine3d: could not write " + out_params);
    if (out_changes != "/dev/null" && out_changes != "null") {
        if (!ends_with(out_changes, ".cistem")) { unlink(out_params.c_str()); die("ERROR: output " + out_changes + " must have .cistem extension"); }
        if (!write_cistem(out_changes, changes.data(), n)) { unlink(out_params.c_str()); die("ERROR: refine3d: could not write " + out_changes); }
    }
    printf("\n   NO     PSI   THETA     PHI       SHX       SHY     SCORE   CHANGE\n");
    long double ssum = 0;
    for (long i = 0; i < n; i++) {
        const double *r = &rout[(size_t)i * 32];
        ssum += r[C_SCORE];
        if (i < 50) printf("%7d%8.2f%8.2f%8.2f%10.2f%10.2f%10.4f%9.4f\n", (int)r[C_POS], r[C_PSI], r[C_THETA], r[C_PHI], r[C_SHX], r[C_SHY], r[C_SCORE], changes[(size_t)i * 32 + C_SCORE]);
    }
    printf("\nRefined %ld particles in %.1f s; mean score %.4f\n", n, since(t0), (double)(ssum / n));
    printf("Timing: inputs %.2f s, device + reference %.2f s, particles %.2f s, outputs %.2f s\n", secs(t0, t1), secs(t1, t2), secs(t2, t3), since(t3));
    printf("Start-up: device context %.2f s, reference ready %.2f s after the answers were read (threads of their own)\n", init_s, ref_s);
    printf("Pipeline: %ld chunks; reader: read %.2f s, waited for a buffer %.2f s; uploader: copied %.2f s, waited for a buffer %.2f s; main thread: computed %.2f s, "
           "waited for data %.2f s\n", ncalls, st.t_read, st.w_pin, st.t_up, st.w_dev, t_comp, w_data);
    printf("\nRefine3D: Normal termination\n\n");
    fflush(stdout);
    _exit(0);          // the library's reader pool is parked on purpose
}
