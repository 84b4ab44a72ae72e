// reconstruct3d — native drop-in for the program PYP scripts at src/pyp/refine/frealign/frealign.py:1780-1824
// ("<dir>/reconstruct3d << eot >> log ... eot": 39 answers on stdin, 43 with the dose-weighting block).
//
// The reference's reconstruct3d is a compiled program; so is this one.  It covers the call PYP makes by default (a .cistem table
// in the standard column order, a float32 stack, no dose weighting, no likelihood blurring) and does nothing but parse, stream
// the particle range from the stack file into libpypmatch (include/ppm.h: ppm_host_read -> ppm_device_upload -> ppm_insert_batch)
// and write the two dump files.  Everything else — the other answers, and every input it would have to refuse — is handed to
// bin/reconstruct3d.py (pyp_amd/surface/cli.py:reconstruct3d_main) with the same stdin, BEFORE the GPU is touched, so that
// behaviour and messages have one definition.  Start-up is what this buys: no interpreter and no numpy import in front of a
// run that moves 26 GB in half a second (bench.py, "dropin").
//
// Built by pyp_amd/csrc/Makefile into bin/reconstruct3d (g++, no HIP: the C ABI only).
#include "dropin_common.h"

using namespace dropin;

namespace {

enum { C_POS = 0, C_DF1 = 6, C_DF2 = 7, C_OCC = 11, C_SCORE = 14, C_PIND = 26 };

[[noreturn]] void fall_back(const std::string &input) { hand_to_python("reconstruct3d.py", input); }

int write_dump(const std::string &path, int box, float pixel, long long count, const float *data, size_t nfloat) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return -1;
    unsigned char head[24];
    memcpy(head, "PPMDUMP1", 8); memcpy(head + 8, &box, 4); memcpy(head + 12, &pixel, 4); memcpy(head + 16, &count, 8);
    const size_t bytes = nfloat * 4;
    std::atomic<int> bad{0};
    if (pwrite(fd, head, 24, 0) != 24 || ftruncate(fd, (off_t)(24 + bytes)) != 0) bad = 1;
    std::vector<std::thread> th;
    for (int k = 0; k < 4 && !bad; k++)
        th.emplace_back([&, k] {
            size_t a = bytes * k / 4, e = bytes * (k + 1) / 4;
            while (a < e) {
                ssize_t w = pwrite(fd, (const char *)data + a, std::min(e - a, (size_t)64 << 20), (off_t)(24 + a));
                if (w <= 0) { bad = 1; return; }
                a += (size_t)w;
            }
        });
    for (auto &t : th) t.join();
    close(fd);
    if (bad || rename(tmp.c_str(), path.c_str()) != 0) { unlink(tmp.c_str()); return -1; }
    return 0;
}

}  // namespace

int main() {
    const auto t0 = Clock::now();
    const std::string input = read_all_stdin();
    if (const char *e = getenv("PPM_NATIVE")) if (!strcmp(e, "0")) fall_back(input);
    // ---- the answers (pyp_amd/surface/prompts.py: read_answers, parse_reconstruct3d)
    const std::vector<std::string> a = read_answers(input);
    if (a.size() < 39) fall_back(input);
    const std::string stack = a[0], params = a[1], gstats = a[2], symmetry = a[8], res_file = a[7];
    double first, last, px, outer_radius, res_limit, bfac, thr, padding;
    bool score_weighting, dose, normalize, adjust, invert, excl, crop, split_eo, by_pind, center, blur, thrref, dump;
    bool ok = parse_num(a[9], first) && parse_num(a[10], last) && parse_num(a[11], px) && parse_num(a[14], outer_radius) && parse_num(a[15], res_limit) &&
              parse_num(a[17], bfac) && parse_bool(a[18], score_weighting) && parse_bool(a[21], dose);
    // answers this build refuses or treats specially unless they carry the value PYP always sends (frealign.py:1763-1770, :1796-1808):
    // the Python implementation owns the messages and the tilt window, so anything else goes there
    double mass, inner_radius, res_reference, tilt_lo, tilt_hi, smoothing, threads;
    ok = ok && parse_num(a[12], mass) && parse_num(a[13], inner_radius) && parse_num(a[16], res_reference) && parse_num(a[19], tilt_lo) && parse_num(a[20], tilt_hi);
    if (!ok || dose) fall_back(input);                 // dose weighting: five more answers, side files, a table over the whole file
    ok = parse_num(a[22], thr) && parse_num(a[23], smoothing) && parse_num(a[24], padding) && parse_bool(a[25], normalize) && parse_bool(a[26], adjust) &&
         parse_bool(a[27], invert) && parse_bool(a[28], excl) && parse_bool(a[29], crop) && parse_bool(a[30], split_eo) && parse_bool(a[31], by_pind) &&
         parse_bool(a[32], center) && parse_bool(a[33], blur) && parse_bool(a[34], thrref) && parse_bool(a[35], dump) && parse_num(a[38], threads);
    ok = ok && inner_radius == 0.0 && res_reference == 0.0 && smoothing == 1.0 && tilt_lo <= 0.0 && tilt_hi < 0.0;
    for (int k = 0; k < 9; k++) ok = ok && !a[k].empty();
    ok = ok && !a[36].empty() && !a[37].empty();
    const std::string dump1 = a[36], dump2 = a[37];
    if (!ok || center || thrref || excl || !split_eo || !dump || blur || std::fabs(padding - 1.0) > 1e-6 || !ends_with(params, ".cistem") ||
        !exists(stack) || !exists(params) || first < 1 || last < first || px <= 0)
        fall_back(input);
    const long ifirst = (long)first, ilast = (long)last;
    MrcHead mh;
    const bool have_gs = gstats != "null" && exists(gstats);
    if (!cistem_is_standard(params) || (have_gs && !cistem_is_standard(gstats)) || !read_mrc_head(stack, mh) || mh.mode != 2 || mh.nx != mh.ny) fall_back(input);

    // ---- from here on the GPU is in use: no more fall-backs.  Device start-up (context, code object, accumulators, page-locked
    // staging buffers) runs in a thread of its own while the parameter file is read.
    setenv("PPM_SYNC", "block", 0);
    const int dev = getenv("PPM_DEVICE") ? atoi(getenv("PPM_DEVICE")) : 0;
    // advisory per-GPU lock: PYP may start several processes per node (src/pyp/system/mpi.py:104)
    const int lockfd = gpu_lock(dev);
    const int box = mh.nx;
    const size_t sec = (size_t)box * box * 4;
    long chunk_mb = 256, call_mb = 2048;
    if (const char *e = getenv("PPM_IO_CHUNK_MB")) chunk_mb = std::max(1L, atol(e));
    // one staging buffer: whole images, no larger than the range can fill (page-locking costs ~0.2 s per GB)
    const size_t pin_bytes = std::min(std::max((size_t)16, ((size_t)chunk_mb << 20) / sec), (size_t)(ilast - ifirst + 1)) * sec;
    ppm_accum_t *acc = nullptr;
    Stream st;
    void **pinned = st.pinned;
    std::mutex up_m; std::condition_variable up_cv; int up_stage = 0; std::string up_err;      // up_stage: 1 = accumulator ready, 2 + k = pinned[k] ready
    std::atomic<int> want_pinned{3};
    auto t_dev = Clock::now();
    double dev_s = 0, init_s = 0;
    std::thread starter([&] {
        auto fail_ = [&](const char *m) { { std::lock_guard<std::mutex> lk(up_m); up_err = m && *m ? m : "ERROR: device start-up failed"; up_stage = 99; } up_cv.notify_all(); };
        if (ppm_init(dev) != 0) return fail_(ppm_last_error());
        init_s = since(t_dev);
        acc = ppm_accum_create(box, (float)px, symmetry.c_str(), nullptr);
        if (!acc) return fail_(ppm_last_error());
        dev_s = since(t_dev);
        { std::lock_guard<std::mutex> lk(up_m); up_stage = 1; } up_cv.notify_all();
        for (int k = 0; k < 3 && k < want_pinned.load(); k++) {
            pinned[k] = ppm_host_alloc(pin_bytes);
            if (!pinned[k]) return fail_(ppm_last_error());
            { std::lock_guard<std::mutex> lk(up_m); up_stage = 2 + k; } up_cv.notify_all();
        }
    });
    auto wait_stage = [&](int st) {
        std::unique_lock<std::mutex> lk(up_m);
        up_cv.wait(lk, [&] { return up_stage >= st; });
        return up_stage != 99;
    };
    auto bail = [&](const std::string &msg) { wait_stage(1); starter.join(); die(msg); };         // never exit in the middle of the start-up

    std::vector<double> rows; long nrows = 0;
    if (!read_cistem(params, rows, nrows)) bail("ERROR: " + params + ": binary file is broken");
    std::vector<double> gs; long ngs = 0;
    if (have_gs && !read_cistem(gstats, gs, ngs)) bail("ERROR: " + gstats + ": binary file is broken");
    // ---- the range
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
    // ---- scores (cli.py:reconstruct3d_main): defocus regression removed, average for the weighting
    long nused = 0;
    for (long i = 0; i < n; i++) nused += rin[(size_t)i * 32 + C_OCC] > 0;
    if (adjust && nused > 10) {
        long double sx = 0, sy = 0;
        for (long i = 0; i < n; i++) if (rin[(size_t)i * 32 + C_OCC] > 0) { sx += 0.5 * (rin[(size_t)i * 32 + C_DF1] + rin[(size_t)i * 32 + C_DF2]); sy += rin[(size_t)i * 32 + C_SCORE]; }
        const long double mx = sx / nused, my = sy / nused;
        long double sxx = 0, sxy = 0;
        for (long i = 0; i < n; i++) if (rin[(size_t)i * 32 + C_OCC] > 0) {
            const long double dx = 0.5 * (rin[(size_t)i * 32 + C_DF1] + rin[(size_t)i * 32 + C_DF2]) - mx;
            sxx += dx * dx; sxy += dx * (rin[(size_t)i * 32 + C_SCORE] - my);
        }
        if (sxx > 0) {
            const double slope = (double)(sxy / sxx);
            for (long i = 0; i < n; i++) rin[(size_t)i * 32 + C_SCORE] -= slope * (0.5 * (rin[(size_t)i * 32 + C_DF1] + rin[(size_t)i * 32 + C_DF2]) - (double)mx);
        }
    }
    double score_avg = 0;
    if (have_gs) score_avg = gs[C_SCORE];
    else if (nused) { long double s = 0; for (long i = 0; i < n; i++) if (rin[(size_t)i * 32 + C_OCC] > 0) s += rin[(size_t)i * 32 + C_SCORE]; score_avg = (double)(s / nused); }

    printf("\n        **   Welcome to Reconstruct3D (MI355X / libpypmatch, native)   **\n\n");
    static const char *names[39] = { "stack", "input_params", "global_stats", "reference", "map1", "map2", "output", "res_file", "symmetry", "first", "last",
        "pixel_size", "molecular_mass", "inner_radius", "outer_radius", "res_limit", "res_reference", "score_bfactor", "score_weighting", "min_tilt_score",
        "max_tilt_score", "dose_weighting", "score_threshold", "smoothing", "padding", "normalize", "adjust_scores", "invert", "exclude_edges", "crop",
        "split_even_odd", "per_particle_splitting", "center_mass", "likelihood_blurring", "threshold_reference", "dump", "dump_1", "dump_2", "threads" };
    for (int k = 0; k < 39; k++) printf("%-28s: %s\n", names[k], a[k].c_str());
    if (crop) printf("NOTE: crop = yes has no effect: the full box is transformed\n");
    const auto t1 = Clock::now();
    ppm_recon_cfg rc;
    memset(&rc, 0, sizeof rc);
    rc.box = box; rc.pixel_size = (float)px; rc.res_limit = (float)res_limit; rc.score_weight_bfactor = score_weighting ? (float)bfac : 0.f;
    rc.score_average = (float)score_avg; rc.score_threshold = (float)thr; rc.normalize = normalize; rc.invert = invert; rc.split_by_pind = by_pind;
    rc.mask_radius = (float)outer_radius;
    if (!wait_stage(1)) { starter.join(); die(up_err); }
    const auto t2 = Clock::now();

    // ---- reader -> uploader -> insertion (the stages of pyp_amd/surface/cli.py:_iter_image_chunks)
    const long chunk = std::max(1L, std::min(n, (long)(pin_bytes / sec)));
    const long nchunks = (n + chunk - 1) / chunk;
    const long group = std::max(1L, std::min(nchunks, (long)(((size_t)call_mb << 20) / ((size_t)chunk * sec))));
    st.n = n; st.chunk = chunk; st.group = group; st.sec = sec; st.contiguous = contiguous;
    st.npin = (int)std::min(3L, nchunks); st.ndev = (int)std::min(2L, (nchunks + group - 1) / group);
    want_pinned = st.npin;
    st.nread = getenv("PPM_IO_THREADS") ? std::max(1, std::min(16, atoi(getenv("PPM_IO_THREADS")))) : 8;
    st.fd = open(stack.c_str(), O_RDONLY);
    if (st.fd < 0) bail("ERROR: reconstruct3d: cannot open " + stack);
    st.img_off = [&](long i) { return mh.offset + (long long)((long)rin[(size_t)i * 32 + C_POS] - 1) * (long long)sec; };
    st.wait_pinned = [&](int slot) { return wait_stage(2 + slot); };          // page-locked by the start-up thread
    st.start();
    double t_comp = 0, w_data = 0; long ncalls = 0;
    for (long lo = 0; lo < n;) {
        auto ta = Clock::now();
        Stream::Item it;
        if (!st.next(it)) {
            // a stage failed: the stream is aborted (no thread waits for a buffer any more); only the start-up thread is joined - it
            // may be inside the runtime - and the process leaves through _exit
            starter.join();
            die(!up_err.empty() ? up_err : (!st.err.empty() ? st.err : std::string("ERROR: reconstruct3d: reading or uploading the particle stack failed")));
        }
        auto tb = Clock::now();
        if (ppm_insert_batch(acc, &rc, st.dbuf[it.slot], 1, (int)(it.hi - it.lo), rin.data() + (size_t)it.lo * 32) != 0) { st.abort(); starter.join(); die(ppm_last_error()); }
        st.release(it.slot);
        w_data += secs(ta, tb); t_comp += since(tb); ncalls++;
        lo = it.hi;
    }
    st.join(); starter.join();
    close(st.fd);
    const double t_read = st.t_read, t_up = st.t_up, w_pin = st.w_pin, w_dev = st.w_dev;
    const auto t3 = Clock::now();
    const size_t nf = ppm_accum_floats(box);
    const size_t half = nf / 2;
    // each half into a staging buffer that is already page-locked (a half map of 256^3 is 203 MB), else into plain memory
    float *h_even, *h_odd; void *plain = nullptr;
    if (pinned[0] && pinned[1] && pin_bytes >= half * sizeof(float)) { h_even = (float *)pinned[0]; h_odd = (float *)pinned[1]; }
    else {
        plain = malloc(nf * sizeof(float));
        if (!plain) die("ERROR: reconstruct3d: out of memory for the dump files");
        h_even = (float *)plain; h_odd = h_even + half;
    }
    if (ppm_accum_download_range(acc, h_even, 0, half) != 0 || ppm_accum_download_range(acc, h_odd, half, half) != 0) die(ppm_last_error());
    const long c0 = ppm_accum_count(acc, 0), c1 = ppm_accum_count(acc, 1);
    ppm_accum_destroy(acc);
    for (void *p : st.dbuf) if (p) ppm_device_free(p);
    gpu_unlock(lockfd);
    int e1 = 0, e2 = 0;
    std::thread w2([&] { e2 = write_dump(dump2, box, (float)px, c0, h_even, half); });                   // even keys -> map 2
    e1 = write_dump(dump1, box, (float)px, c1, h_odd, half);                                            // odd keys  -> map 1
    w2.join();
    if (e1 || e2) { unlink(dump1.c_str()); unlink(dump2.c_str()); die("ERROR: reconstruct3d: could not write " + (e1 ? dump1 : dump2)); }
    if (FILE *f = fopen(res_file.c_str(), "w")) {
        fprintf(f, "C Reconstruct3D (libpypmatch): particles %ld..%ld, inserted %ld + %ld\n", ifirst, ilast, c1, c0);
        fclose(f);
    }
    printf("\nInserted %ld of %ld particles in %.1f s\n", c0 + c1, n, since(t0));
    printf("Timing: inputs %.2f s, device %.2f s, particles %.2f s, dumps %.2f s\n", std::chrono::duration<double>(t1 - t0).count(),
           std::chrono::duration<double>(t2 - t1).count(), std::chrono::duration<double>(t3 - t2).count(), since(t3));
    printf("Start-up: device context %.2f s, accumulators %.2f s after the answers were read (in a thread of its own)\n", init_s, dev_s);
    printf("Pipeline: %ld chunks; reader: read %.2f s, waited for a buffer %.2f s; uploader: copied %.2f s, waited for a buffer %.2f s; main thread: computed %.2f s, "
           "waited for data %.2f s\n", ncalls, t_read, w_pin, t_up, w_dev, t_comp, w_data);
    printf("NOTE: the dump files are in libpypmatch's own format (PPMDUMP1): only this build's local_merge3d / merge3d read them "
           "(frealign.py:1852 consumers must be replaced together, INTEGRATION.md 1)\n");
    printf("\nNormal termination, intermediate files dumped\n");
    printf("\nReconstruct3D: Normal termination\n\n");
    fflush(stdout);
    _exit(0);          // the library's reader pool is parked on purpose
}
