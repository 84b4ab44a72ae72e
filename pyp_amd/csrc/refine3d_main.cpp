// refine3d — native drop-in for the program PYP scripts at src/pyp/refine/frealign/frealign.py:3918-3994
// ("<dir>/refine3d << eot >> log ... eot": 50 answers on stdin).
//
// The reference's refine3d is a compiled program; so is this one.  It covers the call PYP makes by default (a .cistem table in
// the standard column order, a float32 stack and reference, no statistics weighting, no priors, no matching projections, no 2-D
// focus mask) and does nothing but parse, prepare the reference (ppm_reference_create_padded), stream the particle range from
// the stack file into libpypmatch (include/ppm.h: ppm_host_read -> ppm_device_upload -> ppm_refine_batch) and write the output
// tables.  Everything else — the `.par` surface, the other answers, every input it would have to refuse — is handed to
// bin/refine3d.py (pyp_amd/surface/cli.py:refine3d_main) as a child process with the same stdin, BEFORE the GPU is touched, so
// that behaviour and messages have one definition.  With PPM_STACK_CACHE=1 the call is served by the resident per-GPU server
// (dropin_server.h): the context, the prepared reference and the particle range uploaded by an earlier call are already there.
//
// Built by pyp_amd/csrc/Makefile into bin/refine3d (g++, no HIP: the C ABI only).
#include "dropin_server.h"

using namespace dropin;

int main() {
    const auto t0 = Clock::now();
    const std::string input = read_all_stdin();
    auto fall_back = [&]() { hand_to_python("refine3d.py", input); };
    if (const char *e = getenv("PPM_NATIVE")) if (!strcmp(e, "0")) fall_back();
    RefineJob j;
    if (!refine_parse(input, j)) fall_back();
    {
        int status = 1; std::string text;
        if (run_through_server(kProgRefine, input, status, text)) {
            if (status == kHandOver) fall_back();
            fputs(text.c_str(), stdout); fflush(stdout);
            _exit(status);
        }
    }
    const int box = j.box;
    const size_t sec = j.sec;
    // ---- from here on the GPU is in use: no more hand-overs.  The lock comes first (PYP may start several processes per node:
    // a waiting process holds neither a context nor page-locked memory); then the device, the page-locked staging buffers and the
    // reference are brought up by threads of their own while the parameter file is read.
    setenv("PPM_SYNC", "block", 0);
    const int dev = getenv("PPM_DEVICE") ? atoi(getenv("PPM_DEVICE")) : 0;
    const int lockfd = gpu_lock(dev);
    long chunk_mb = 64, call_mb = 512;         // compute-bound: 64 MB chunks, 512 MB per refinement call (scripts/dropin_ab.py)
    if (const char *e = getenv("PPM_IO_CHUNK_MB")) chunk_mb = std::max(1L, atol(e));
    const size_t pin_bytes = std::min(std::max((size_t)16, ((size_t)chunk_mb << 20) / sec), (size_t)(j.ilast - j.ifirst + 1)) * sec;
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
        if (!read_volume(j.reference, box, vol)) err = "ERROR: refine3d: cannot read the reference " + j.reference;
        else if (!wait_stage(1)) err = "";                       // the start-up's own message is reported
        else {
            ref = ppm_reference_create_padded(vol.data(), box, (float)(box / 2.0), j.pad);
            if (!ref) err = ppm_last_error();
        }
        ref_s = since(t_dev);
        { std::lock_guard<std::mutex> lk(up_m); ref_err = err; ref_done = true; } up_cv.notify_all();
    });
    auto wait_ref = [&] { std::unique_lock<std::mutex> lk(up_m); up_cv.wait(lk, [&] { return ref_done; }); };
    auto bail = [&](const std::string &msg) { wait_stage(1); starter.join(); wait_ref(); refmaker.join(); die(msg); };      // never exit in the middle of the start-up
    try { refine_rows(j); } catch (const Fail &f) { bail(f.msg); }
    const long n = j.n;
    Out out;
    refine_banner(j, out, "native");
    const auto t1 = Clock::now();

    // ---- reader -> uploader -> refinement
    const long chunk = std::max(1L, std::min(n, (long)(pin_bytes / sec)));
    const long nchunks = (n + chunk - 1) / chunk;
    const long group = std::max(1L, std::min(nchunks, (long)(((size_t)call_mb << 20) / ((size_t)chunk * sec))));
    st.n = n; st.chunk = chunk; st.group = group; st.sec = sec; st.contiguous = j.contiguous;
    st.npin = (int)std::min(3L, nchunks); st.ndev = (int)std::min(2L, (nchunks + group - 1) / group);
    want_pinned = st.npin;
    st.nread = getenv("PPM_IO_THREADS") ? std::max(1, std::min(16, atoi(getenv("PPM_IO_THREADS")))) : 8;
    st.fd = open(j.stack.c_str(), O_RDONLY);
    if (st.fd < 0) bail("ERROR: refine3d: cannot open " + j.stack);
    st.img_off = [&](long i) { return j.mh.offset + (long long)((long)j.rin[(size_t)i * 32 + RF_POS] - 1) * (long long)sec; };
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
        if (ppm_refine_batch(ref, &j.cfg, st.group_ptr(it), 1, (int)(it.hi - it.lo), j.rin.data() + (size_t)it.lo * 32, rout.data() + (size_t)it.lo * 32) != 0) {
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
    double mean = 0;
    try { refine_outputs(j, rout, note, out, mean); } catch (const Fail &f) { die(f.msg); }
    printf("\nRefined %ld particles in %.1f s; mean score %.4f\n", n, since(t0), mean);
    printf("Timing: inputs %.2f s, device + reference %.2f s, particles %.2f s, outputs %.2f s\n", secs(t0, t1), secs(t1, t2), secs(t2, t3), since(t3));
    printf("Start-up: device context %.2f s, reference ready %.2f s after the answers were read (threads of their own)\n", init_s, ref_s);
    printf("Pipeline: %ld chunks; reader: read %.2f s, waited for a buffer %.2f s; uploader: copied %.2f s, waited for a buffer %.2f s; main thread: computed %.2f s, "
           "waited for data %.2f s\n", ncalls, st.t_read, st.w_pin, st.t_up, st.w_dev, t_comp, w_data);
    printf("\nRefine3D: Normal termination\n\n");
    fflush(stdout);
    _exit(0);          // the library's reader pool is parked on purpose
}
