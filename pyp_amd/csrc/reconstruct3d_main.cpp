// reconstruct3d — native drop-in for the program PYP scripts at src/pyp/refine/frealign/frealign.py:1780-1824
// ("<dir>/reconstruct3d << eot >> log ... eot": 39 answers on stdin, 43 with the dose-weighting block).
//
// The reference's reconstruct3d is a compiled program; so is this one.  It covers the call PYP makes by default (a .cistem table
// in the standard column order, a float32 stack, no dose weighting, no likelihood blurring) and does nothing but parse, stream
// the particle range from the stack file into libpypmatch (include/ppm.h: ppm_host_read -> ppm_device_upload -> ppm_insert_batch)
// and write the two dump files.  Everything else — the other answers, and every input it would have to refuse — is handed to
// bin/reconstruct3d.py (pyp_amd/surface/cli.py:reconstruct3d_main) as a child process with the same stdin, BEFORE the GPU is
// touched, so that behaviour and messages have one definition.  Start-up is what this buys: no interpreter and no numpy import in
// front of a run that moves 26 GB in half a second (bench.py, "dropin").  With PPM_STACK_CACHE=1 the call is served by the resident
// per-GPU server instead (dropin_server.h): no context creation and, when an earlier call uploaded the range, no PCIe pass.
//
// Built by pyp_amd/csrc/Makefile into bin/reconstruct3d (g++, no HIP: the C ABI only).
#include "dropin_server.h"

using namespace dropin;

int main() {
    const auto t0 = Clock::now();
    const std::string input = read_all_stdin();
    auto fall_back = [&]() { hand_to_python("reconstruct3d.py", input); };
    if (const char *e = getenv("PPM_NATIVE")) if (!strcmp(e, "0")) fall_back();
    ReconJob j;
    if (!recon_parse(input, j)) fall_back();
    {   // the resident server, if asked for (it answers "hand over" exactly where this program would)
        int status = 1; std::string text;
        if (run_through_server(kProgRecon, input, status, text)) {
            if (status == kHandOver) fall_back();
            fputs(text.c_str(), stdout); fflush(stdout);
            _exit(status);
        }
    }
    // ---- from here on the GPU is in use: no more hand-overs.  Device start-up (context, code object, accumulators, page-locked
    // staging buffers) runs in a thread of its own while the parameter file is read.
    setenv("PPM_SYNC", "block", 0);
    const int dev = getenv("PPM_DEVICE") ? atoi(getenv("PPM_DEVICE")) : 0;
    const int lockfd = gpu_lock(dev);         // advisory per-GPU lock: PYP may start several processes per node (src/pyp/system/mpi.py:104)
    const int box = j.box;
    const size_t sec = j.sec;
    long chunk_mb = 256, call_mb = 2048;
    if (const char *e = getenv("PPM_IO_CHUNK_MB")) chunk_mb = std::max(1L, atol(e));
    // one staging buffer: whole images, no larger than the range can fill (page-locking costs ~0.2 s per GB)
    const size_t pin_bytes = std::min(std::max((size_t)16, ((size_t)chunk_mb << 20) / sec), (size_t)(j.ilast - j.ifirst + 1)) * sec;
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
        acc = ppm_accum_create(box, (float)j.px, j.symmetry.c_str(), nullptr);
        if (!acc) return fail_(ppm_last_error());
        dev_s = since(t_dev);
        { std::lock_guard<std::mutex> lk(up_m); up_stage = 1; } up_cv.notify_all();
        for (int k = 0; k < 3 && k < want_pinned.load(); k++) {
            pinned[k] = ppm_host_alloc(pin_bytes);
            if (!pinned[k]) return fail_(ppm_last_error());
            { std::lock_guard<std::mutex> lk(up_m); up_stage = 2 + k; } up_cv.notify_all();
        }
    });
    auto wait_stage = [&](int s) {
        std::unique_lock<std::mutex> lk(up_m);
        up_cv.wait(lk, [&] { return up_stage >= s; });
        return up_stage != 99;
    };
    auto bail = [&](const std::string &msg) { wait_stage(1); starter.join(); die(msg); };         // never exit in the middle of the start-up
    try { recon_rows(j); } catch (const Fail &f) { bail(f.msg); }
    const long n = j.n;
    Out out;
    recon_banner(j, out, "native");
    const auto t1 = Clock::now();
    if (!wait_stage(1)) { starter.join(); die(up_err); }
    const auto t2 = Clock::now();

    // ---- reader -> uploader -> insertion (the stages of pyp_amd/surface/cli.py:_iter_image_chunks)
    const long chunk = std::max(1L, std::min(n, (long)(pin_bytes / sec)));
    const long nchunks = (n + chunk - 1) / chunk;
    const long group = std::max(1L, std::min(nchunks, (long)(((size_t)call_mb << 20) / ((size_t)chunk * sec))));
    st.n = n; st.chunk = chunk; st.group = group; st.sec = sec; st.contiguous = j.contiguous;
    st.npin = (int)std::min(3L, nchunks); st.ndev = (int)std::min(2L, (nchunks + group - 1) / group);
    want_pinned = st.npin;
    st.nread = getenv("PPM_IO_THREADS") ? std::max(1, std::min(16, atoi(getenv("PPM_IO_THREADS")))) : 8;
    st.fd = open(j.stack.c_str(), O_RDONLY);
    if (st.fd < 0) bail("ERROR: reconstruct3d: cannot open " + j.stack);
    st.img_off = [&](long i) { return j.mh.offset + (long long)((long)j.rin[(size_t)i * 32 + RC_POS] - 1) * (long long)sec; };
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
        if (ppm_insert_batch(acc, &j.rc, st.group_ptr(it), 1, (int)(it.hi - it.lo), j.rin.data() + (size_t)it.lo * 32) != 0) { st.abort(); starter.join(); die(ppm_last_error()); }
        st.release(it.slot);
        w_data += secs(ta, tb); t_comp += since(tb); ncalls++;
        lo = it.hi;
    }
    st.join(); starter.join();
    close(st.fd);
    const auto t3 = Clock::now();
    const size_t half = ppm_accum_floats(box) / 2;
    // each half into a staging buffer that is already page-locked (a half map of 256^3 is 203 MB), else into plain memory
    float *h_even, *h_odd; void *plain = nullptr;
    if (pinned[0] && pinned[1] && pin_bytes >= half * sizeof(float)) { h_even = (float *)pinned[0]; h_odd = (float *)pinned[1]; }
    else {
        plain = malloc(2 * half * sizeof(float));
        if (!plain) die("ERROR: reconstruct3d: out of memory for the dump files");
        h_even = (float *)plain; h_odd = h_even + half;
    }
    long c0 = 0, c1 = 0;
    try { recon_outputs(j, acc, h_even, h_odd, c0, c1); } catch (const Fail &f) { die(f.msg); }
    ppm_accum_destroy(acc);
    for (void *p : st.dbuf) if (p) ppm_device_free(p);
    gpu_unlock(lockfd);
    printf("\nInserted %ld of %ld particles in %.1f s\n", c0 + c1, n, since(t0));
    printf("Timing: inputs %.2f s, device %.2f s, particles %.2f s, dumps %.2f s\n", secs(t0, t1), secs(t1, t2), secs(t2, t3), since(t3));
    printf("Start-up: device context %.2f s, accumulators %.2f s after the answers were read (in a thread of its own)\n", init_s, dev_s);
    printf("Pipeline: %ld chunks; reader: read %.2f s, waited for a buffer %.2f s; uploader: copied %.2f s, waited for a buffer %.2f s; main thread: computed %.2f s, "
           "waited for data %.2f s\n", ncalls, st.t_read, st.w_pin, st.t_up, st.w_dev, t_comp, w_data);
    recon_footer(out);
    fflush(stdout);
    _exit(0);          // the library's reader pool is parked on purpose
}
