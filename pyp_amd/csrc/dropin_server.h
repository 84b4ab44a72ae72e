// dropin_server.h — the resident per-GPU server behind the compiled executables (SURVEY.md 8b "Threading": "multiplex via a daemon").
//
// Every PYP iteration runs reconstruct3d and then refine3d over the SAME particle stack (src/pyp/refine/frealign/frealign.py:1780-1824,
// :3918-3994): as one-shot processes each of them creates a GPU context, prepares its reference and pushes the whole stack
// (26 GB at 100 k x 256^2) through PCIe again.  With PPM_STACK_CACHE=1 the executables become thin clients of `bin/ppm_server`, one
// process per GPU that stays alive between calls and keeps
//   * the GPU context and the library's code object,
//   * the particle ranges it has uploaded, resident in HBM, keyed by the stack file's identity (device, inode, size, mtime) -
//     288 GB of HBM hold a whole data set,
//   * the prepared references (cube + slice bank), keyed by the map file's identity and the padding factor,
//   * the page-locked staging buffers.
// A client sends its working directory and the here-doc it received on stdin; the server runs the call (the same C-ABI sequence the
// one-shot executable runs), writes the output files itself and sends back the log text and the exit status.  Calls outside the
// compiled fast path are answered with "hand over" and the client starts the Python implementation as it would without a server.
// One request at a time (the server IS the per-GPU lock for its clients; it also takes the advisory file lock while it computes, so
// one-shot processes of the same node wait their turn).  It exits after PPM_STACK_CACHE_IDLE_S seconds without a request (default 600).
//
// Wire format (unix stream socket $PPM_LOCK_DIR/pyp_amd_gpu<dev>.sock, mode 0600), little-endian:
//   request : "PPMS" u32 program (1 reconstruct3d, 2 refine3d, 8 statistics, 9 stop) u32 len cwd u32 len stdin
//   reply   : i32 status (0 ok, 1 ERROR, -100 hand over to Python) u32 len text
#pragma once
#include <poll.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>

#include "dropin_common.h"

namespace dropin {

enum { kProgRecon = 1, kProgRefine = 2, kProgHello = 7, kProgStats = 8, kProgStop = 9, kHandOver = -100 };

// Where the server's socket, its guard lock and its log live: PPM_LOCK_DIR, else the user's runtime directory, else /tmp — the names carry
// the user id, so two users of a node run a server each (the GPU lock file itself, pyp_amd_gpu<N>.lock, stays shared between them).
inline std::string server_dir() {
    const char *ld = getenv("PPM_LOCK_DIR");
    if (ld && *ld) return ld;
    const char *xr = getenv("XDG_RUNTIME_DIR");
    if (xr && *xr && access(xr, W_OK) == 0) return xr;
    return "/tmp";
}
inline std::string server_socket_path(int dev) {
    return server_dir() + "/pyp_amd_gpu" + std::to_string(dev) + ".u" + std::to_string((long)getuid()) + ".sock";
}
inline bool write_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) { ssize_t w = send(fd, c, n, MSG_NOSIGNAL); if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; } c += w; n -= (size_t)w; }
    return true;
}
inline bool read_all(int fd, void *p, size_t n) {
    char *c = (char *)p;
    while (n) { ssize_t r = recv(fd, c, n, 0); if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; } c += r; n -= (size_t)r; }
    return true;
}
inline bool send_blob(int fd, const std::string &s) { const uint32_t n = (uint32_t)s.size(); return write_all(fd, &n, 4) && write_all(fd, s.data(), n); }
inline bool recv_blob(int fd, std::string &s) {
    uint32_t n = 0;
    if (!read_all(fd, &n, 4) || n > (64u << 20)) return false;
    s.resize(n);
    return n == 0 || read_all(fd, &s[0], n);
}
inline int connect_server(int dev) {
    const std::string path = server_socket_path(dev);
    int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    sockaddr_un ad; memset(&ad, 0, sizeof ad); ad.sun_family = AF_UNIX;
    if (path.size() >= sizeof ad.sun_path) { close(fd); return -1; }
    strcpy(ad.sun_path, path.c_str());
    if (connect(fd, (sockaddr *)&ad, sizeof ad) != 0) { close(fd); return -1; }
    return fd;
}
// What a request carries besides its input: the CLIENT's file-creation mask and its PPM_* settings (PPM_IO_THREADS, the kernels' knobs),
// one "name=value" per line — a call served by the resident process behaves like the one-shot run the client would have made, whoever
// started the server.  The server's own settings (cache size, idle limit, lock directory, device) are not the caller's to change.
inline bool server_owned_setting(const std::string &name) {
    return name.compare(0, 15, "PPM_STACK_CACHE") == 0 || name == "PPM_LOCK_DIR" || name == "PPM_DEVICE";
}
inline std::string client_settings() {
    const mode_t m = umask(0); umask(m);
    std::string s = "umask=" + std::to_string((unsigned)m) + "\n";
    for (char **e = environ; e && *e; e++) {
        const std::string kv = *e;
        const size_t eq = kv.find('=');
        if (eq == std::string::npos || kv.compare(0, 4, "PPM_") != 0 || kv.find('\n') != std::string::npos) continue;
        if (!server_owned_setting(kv.substr(0, eq))) s += kv + "\n";
    }
    return s;
}
// applied by the server for the duration of one request, undone afterwards (requests are served one at a time)
struct RequestSettings {
    mode_t old_mask = 0; bool mask_set = false;
    std::vector<std::pair<std::string, std::pair<bool, std::string>>> saved;       // name -> (was set, old value)
    explicit RequestSettings(const std::string &text) {
        // PPM_* settings of the server's own environment the client does not have are unset for this request
        std::vector<std::string> mine;
        for (char **e = environ; e && *e; e++) {
            const std::string kv = *e; const size_t eq = kv.find('=');
            if (eq != std::string::npos && kv.compare(0, 4, "PPM_") == 0 && !server_owned_setting(kv.substr(0, eq))) mine.push_back(kv.substr(0, eq));
        }
        std::vector<std::pair<std::string, std::string>> want;
        size_t p0 = 0;
        while (p0 < text.size()) {
            size_t p1 = text.find('\n', p0); if (p1 == std::string::npos) p1 = text.size();
            const std::string kv = text.substr(p0, p1 - p0); p0 = p1 + 1;
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) continue;
            const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
            if (k == "umask") { old_mask = umask((mode_t)(strtoul(v.c_str(), nullptr, 10) & 0777)); mask_set = true; }
            else if (k.compare(0, 4, "PPM_") == 0 && !server_owned_setting(k)) want.emplace_back(k, v);
        }
        for (const auto &k : mine) {
            bool keep = false; for (const auto &w : want) keep = keep || w.first == k;
            if (!keep) { saved.push_back({ k, { true, getenv(k.c_str()) } }); unsetenv(k.c_str()); }
        }
        for (const auto &w : want) {
            const char *o = getenv(w.first.c_str());
            saved.push_back({ w.first, { o != nullptr, o ? o : "" } });
            setenv(w.first.c_str(), w.second.c_str(), 1);
        }
    }
    ~RequestSettings() {
        for (auto it = saved.rbegin(); it != saved.rend(); ++it) {
            if (it->second.first) setenv(it->first.c_str(), it->second.second.c_str(), 1); else unsetenv(it->first.c_str());
        }
        if (mask_set) umask(old_mask);
    }
};
// one request / reply; false = no server could be reached (the caller runs the call itself)
inline bool server_call(int dev, int prog, const std::string &input, int &status, std::string &text) {
    int fd = connect_server(dev);
    if (fd < 0) return false;
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) { close(fd); return false; }
    const uint32_t p = (uint32_t)prog;
    bool ok = write_all(fd, "PPMS", 4) && write_all(fd, &p, 4) && send_blob(fd, cwd) && send_blob(fd, input) && send_blob(fd, client_settings());
    int32_t st = 1;
    ok = ok && read_all(fd, &st, 4) && recv_blob(fd, text);
    close(fd);
    status = st;
    return ok;
}
// start bin/ppm_server for `dev` as a detached child (never an exec of this process) and wait until its socket answers
inline bool start_server(int dev, double wait_s = 20.0) {
    const std::string exe = self_dir() + "/ppm_server";
    if (!exists(exe)) return false;
    const std::string d = std::to_string(dev);
    char *const argv[] = { (char *)exe.c_str(), (char *)"--device", (char *)d.c_str(), (char *)"--daemon", nullptr };
    pid_t pid = 0;
    if (posix_spawn(&pid, exe.c_str(), nullptr, nullptr, argv, environ) != 0) return false;
    int st = 0;
    while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}       // --daemon: the child forks the server and returns at once
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) return false;   // the launcher itself failed: nothing will ever listen
    const auto t0 = Clock::now();
    while (since(t0) < wait_s) {
        int fd = connect_server(dev);
        if (fd >= 0) { close(fd); return true; }
        usleep(20000);
    }
    return false;
}
// the client side of an executable: true = the call was served (status / text set); false = run it in this process
inline bool run_through_server(int prog, const std::string &input, int &status, std::string &text) {
    const char *e = getenv("PPM_STACK_CACHE");
    if (!e || !*e || !strcmp(e, "0")) return false;
    const int dev = getenv("PPM_DEVICE") ? atoi(getenv("PPM_DEVICE")) : 0;
    // a server left over from an older build of the library (it answers with its own build id) is stopped and replaced
    int hs = 0; std::string hello;
    if (server_call(dev, kProgHello, "", hs, hello) && hello != ppm_build_id()) {
        int st = 0; std::string t;
        (void)server_call(dev, kProgStop, "", st, t);
        for (int i = 0; i < 250; i++) { int fd = connect_server(dev); if (fd < 0) break; close(fd); usleep(20000); }
    }
    if (server_call(dev, prog, input, status, text)) return true;
    if (!start_server(dev)) return false;
    return server_call(dev, prog, input, status, text);
}

// ---------------------------------------------------------------------------------------------- server side
struct StackEntry { FileId id; long first = 0, count = 0; int box = 0; void *dptr = nullptr; size_t bytes = 0; double used = 0; };
struct RefEntry { FileId id; int pad = 1, box = 0; ppm_ref_t *ref = nullptr; double used = 0; };
struct Cache {
    int dev = 0;
    size_t budget = (size_t)160 << 30, used = 0;        // upper bound on resident ranges; what the device really has free decides (reserve)
    size_t headroom = (size_t)24 << 30;                 // left free on the device for work buffers, cached references and other programs (PPM_STACK_CACHE_HEADROOM_GB)
    std::vector<StackEntry> stacks; std::vector<RefEntry> refs;
    void *pinned[3] = { nullptr, nullptr, nullptr }; size_t pin_bytes = 0;
    long hits = 0, misses = 0, served = 0;
    Clock::time_point t0 = Clock::now();
    double now() const { return since(t0); }

    const StackEntry *find_stack(const FileId &id, long pmin, long pmax, int box) {
        for (auto &e : stacks) if (e.id == id && e.box == box && pmin >= e.first && pmax < e.first + e.count) { e.used = now(); return &e; }
        return nullptr;
    }
    void drop_stack(size_t i) { ppm_device_free(stacks[i].dptr); used -= stacks[i].bytes; stacks.erase(stacks.begin() + (long)i); }
    // a device buffer of `bytes` for a new resident range (older ranges of the same file and the least recently used ones make room); null = does not fit
    void *reserve(const FileId &id, size_t bytes) {
        if (bytes > budget) return nullptr;
        for (size_t i = 0; i < stacks.size();) if (stacks[i].id.dev == id.dev && stacks[i].id.ino == id.ino && !(stacks[i].id == id)) drop_stack(i); else i++;   // the file has changed
        for (int attempt = 0; attempt < 2; attempt++) {
            // least recently used ranges go until the new one fits the budget AND the device keeps `headroom` bytes free next to it
            // (hipMemGetInfo: references, work buffers and other processes' allocations are counted by the device, not by this table)
            for (;;) {
                size_t fr = 0, tot = 0;
                const bool tight = ppm_device_mem_info(&fr, &tot) == 0 && fr < bytes + headroom;
                if ((used + bytes <= budget && !tight) || stacks.empty()) break;
                size_t lru = 0;
                for (size_t i = 1; i < stacks.size(); i++) if (stacks[i].used < stacks[lru].used) lru = i;
                drop_stack(lru);
            }
            { size_t fr = 0, tot = 0; if (ppm_device_mem_info(&fr, &tot) == 0 && fr < bytes + headroom / 2) return nullptr; }      // does not fit next to what else lives on the device: stream it
            if (void *p = ppm_device_alloc(bytes)) return p;
            while (!stacks.empty()) drop_stack(0);                  // the device is fuller than the budget assumed: give everything back, try once more
        }
        return nullptr;
    }
    void keep(const FileId &id, long first, long count, int box, void *p, size_t bytes) {
        StackEntry e; e.id = id; e.first = first; e.count = count; e.box = box; e.dptr = p; e.bytes = bytes; e.used = now();
        stacks.push_back(e); used += bytes;
    }
    ppm_ref_t *find_ref(const FileId &id, int pad, int box) {
        for (auto &e : refs) if (e.id == id && e.pad == pad && e.box == box) { e.used = now(); return e.ref; }
        return nullptr;
    }
    void keep_ref(const FileId &id, int pad, int box, ppm_ref_t *r) {
        if (refs.size() >= 4) {                                     // a handful of class references at most
            size_t lru = 0;
            for (size_t i = 1; i < refs.size(); i++) if (refs[i].used < refs[lru].used) lru = i;
            ppm_reference_destroy(refs[lru].ref); refs.erase(refs.begin() + (long)lru);
        }
        RefEntry e; e.id = id; e.pad = pad; e.box = box; e.ref = r; e.used = now(); refs.push_back(e);
    }
    bool pin(size_t bytes) {                                        // three staging buffers of at least `bytes`
        if (pin_bytes >= bytes && pinned[0]) return true;
        for (void *&p : pinned) { if (p) ppm_host_free(p); p = nullptr; }
        pin_bytes = 0;
        for (void *&p : pinned) { p = ppm_host_alloc(bytes); if (!p) return false; }
        pin_bytes = bytes;
        return true;
    }
    // everything this process holds on the device: before a call is handed over to a child process that needs the memory itself
    void release_device() {
        while (!stacks.empty()) drop_stack(0);
        for (auto &e : refs) ppm_reference_destroy(e.ref);
        refs.clear();
    }
    void clear() {
        while (!stacks.empty()) drop_stack(0);
        for (auto &e : refs) ppm_reference_destroy(e.ref);
        refs.clear();
        for (void *&p : pinned) { if (p) ppm_host_free(p); p = nullptr; }
        pin_bytes = 0;
    }
};

// the particle range of a job in device memory: from the cache, or streamed from the file (and kept when it fits).  `use(ptr, lo, hi)`
// is called for consecutive pieces [lo, hi) of the range with their device pointer; throws Fail.
template <typename Job, typename Use>
inline void with_stack(Cache &c, const Job &j, Out &o, long call_mb, long chunk_mb, Use use, const char *prog) {
    FileId id;
    if (!file_id(j.stack, id)) throw Fail{ std::string("ERROR: ") + prog + ": cannot stat " + j.stack };
    const long pmin = (long)j.rin[RC_POS], pmax = (long)j.rin[(size_t)(j.n - 1) * 32 + RC_POS];
    const size_t sec = j.sec;
    const long per_call = std::max(1L, (long)(((size_t)call_mb << 20) / sec));
    if (j.contiguous) {
        if (const StackEntry *e = c.find_stack(id, pmin, pmax, j.box)) {
            c.hits++;
            o.print("Stack: particles %ld..%ld are resident in device memory (uploaded by an earlier call; %.1f GB of %.0f GB cached)\n", pmin, pmax, c.used / 1e9, c.budget / 1e9);
            char *base = (char *)e->dptr + (size_t)(pmin - e->first) * sec;
            // resident data: calls as large as the library likes them (it chunks a call by itself)
            const long big = std::max(per_call, 1L << 17);
            for (long lo = 0; lo < j.n; lo += big) { const long hi = std::min(j.n, lo + big); use(base + (size_t)lo * sec, lo, hi); }
            return;
        }
    }
    c.misses++;
    const size_t pin_bytes = std::min(std::max((size_t)16, ((size_t)chunk_mb << 20) / sec), (size_t)j.n) * sec;
    if (!c.pin(pin_bytes)) throw Fail{ std::string("ERROR: ") + ppm_last_error() };
    Stream st;
    for (int k = 0; k < 3; k++) st.pinned[k] = c.pinned[k];
    const long chunk = std::max(1L, std::min(j.n, (long)(pin_bytes / sec)));
    const long nchunks = (j.n + chunk - 1) / chunk;
    const long group = std::max(1L, std::min(nchunks, (long)(((size_t)call_mb << 20) / ((size_t)chunk * sec))));
    st.n = j.n; st.chunk = chunk; st.group = group; st.sec = sec; st.contiguous = j.contiguous;
    st.npin = (int)std::min(3L, nchunks); st.ndev = (int)std::min(2L, (nchunks + group - 1) / group);
    st.nread = getenv("PPM_IO_THREADS") ? std::max(1, std::min(16, atoi(getenv("PPM_IO_THREADS")))) : 8;
    void *resident = j.contiguous ? c.reserve(id, (size_t)j.n * sec) : nullptr;
    st.resident = resident;
    st.fd = open(j.stack.c_str(), O_RDONLY);
    if (st.fd < 0) { if (resident) ppm_device_free(resident); throw Fail{ std::string("ERROR: ") + prog + ": cannot open " + j.stack }; }
    const long long off0 = j.mh.offset;
    const std::vector<double> &rin = j.rin;
    st.img_off = [&rin, off0, sec](long i) { return off0 + (long long)((long)rin[(size_t)i * 32 + RC_POS] - 1) * (long long)sec; };
    st.start();
    std::string err;
    try {
        for (long lo = 0; lo < j.n;) {
            Stream::Item it;
            if (!st.next(it)) { err = !st.err.empty() ? st.err : std::string("ERROR: ") + prog + ": reading or uploading the particle stack failed"; break; }
            use(st.group_ptr(it), it.lo, it.hi);
            st.release(it.slot);
            lo = it.hi;
        }
    } catch (const Fail &f) { err = f.msg; }
    if (!err.empty()) st.abort();
    st.join();
    close(st.fd);
    for (void *p : st.dbuf) if (p) ppm_device_free(p);
    if (!err.empty()) { if (resident) ppm_device_free(resident); throw Fail{ err }; }
    if (resident) {
        c.keep(id, pmin, j.n, j.box, resident, (size_t)j.n * sec);
        o.print("Stack: particles %ld..%ld uploaded and kept resident for the next call (%.1f GB of %.0f GB cached)\n", pmin, pmax, c.used / 1e9, c.budget / 1e9);
    } else o.print("Stack: streamed (not kept: %s)\n", j.contiguous ? "it does not fit the cache" : "the rows are not one contiguous range");
    o.print("Pipeline: reader: read %.2f s, waited for a buffer %.2f s; uploader: copied %.2f s, waited for a buffer %.2f s\n", st.t_read, st.w_pin, st.t_up, st.w_dev);
}

inline int serve_reconstruct3d(Cache &c, const std::string &input, Out &o) {
    const auto t0 = Clock::now();
    ReconJob j;
    if (!recon_parse(input, j)) return kHandOver;
    recon_rows(j);
    recon_banner(j, o, "native, resident server");
    ppm_accum_t *acc = ppm_accum_create(j.box, (float)j.px, j.symmetry.c_str(), nullptr);
    if (!acc) throw Fail{ ppm_last_error() };
    struct Guard { ppm_accum_t *a; ~Guard() { if (a) ppm_accum_destroy(a); } } guard{ acc };
    const auto t1 = Clock::now();
    double t_comp = 0;
    with_stack(c, j, o, 2048, 256, [&](void *ptr, long lo, long hi) {
        const auto tb = Clock::now();
        if (ppm_insert_batch(acc, &j.rc, ptr, 1, (int)(hi - lo), j.rin.data() + (size_t)lo * 32) != 0) throw Fail{ ppm_last_error() };
        t_comp += since(tb);
    }, "reconstruct3d");
    const auto t2 = Clock::now();
    const size_t half = ppm_accum_floats(j.box) / 2;
    std::vector<float> plain;
    float *h_even, *h_odd;
    if (c.pinned[0] && c.pinned[1] && c.pin_bytes >= half * sizeof(float)) { h_even = (float *)c.pinned[0]; h_odd = (float *)c.pinned[1]; }
    else { plain.resize(2 * half); h_even = plain.data(); h_odd = h_even + half; }
    long c0 = 0, c1 = 0;
    recon_outputs(j, acc, h_even, h_odd, c0, c1);
    o.print("\nInserted %ld of %ld particles in %.2f s\n", c0 + c1, j.n, since(t0));
    o.print("Timing: inputs %.2f s, particles %.2f s (insertion calls %.2f s), dumps %.2f s\n", secs(t0, t1), secs(t1, t2), t_comp, since(t2));
    recon_footer(o);
    return 0;
}

inline int serve_refine3d(Cache &c, const std::string &input, Out &o) {
    const auto t0 = Clock::now();
    RefineJob j;
    if (!refine_parse(input, j)) return kHandOver;
    refine_rows(j);
    refine_banner(j, o, "native, resident server");
    FileId rid;
    if (!file_id(j.reference, rid)) throw Fail{ "ERROR: refine3d: cannot stat " + j.reference };
    ppm_ref_t *ref = c.find_ref(rid, j.pad, j.box);
    if (ref) o.print("Reference: %s is prepared already (kept from an earlier call)\n", j.reference.c_str());
    else {
        std::vector<float> vol;
        if (!read_volume(j.reference, j.box, vol)) throw Fail{ "ERROR: refine3d: cannot read the reference " + j.reference };
        ref = ppm_reference_create_padded(vol.data(), j.box, (float)(j.box / 2.0), j.pad);
        if (!ref) throw Fail{ ppm_last_error() };
        c.keep_ref(rid, j.pad, j.box, ref);
    }
    const auto t1 = Clock::now();
    std::vector<double> rout((size_t)j.n * 32);
    double t_comp = 0;
    with_stack(c, j, o, 512, 64, [&](void *ptr, long lo, long hi) {
        const auto tb = Clock::now();
        if (ppm_refine_batch(ref, &j.cfg, ptr, 1, (int)(hi - lo), j.rin.data() + (size_t)lo * 32, rout.data() + (size_t)lo * 32) != 0) throw Fail{ ppm_last_error() };
        t_comp += since(tb);
    }, "refine3d");
    const auto t2 = Clock::now();
    double mean = 0;
    refine_outputs(j, rout, ppm_refine_note(ref), o, mean);
    o.print("\nRefined %ld particles in %.2f s; mean score %.4f\n", j.n, since(t0), mean);
    o.print("Timing: inputs + reference %.2f s, particles %.2f s (refinement calls %.2f s), outputs %.2f s\n", secs(t0, t1), secs(t1, t2), t_comp, since(t2));
    o.print("\nRefine3D: Normal termination\n\n");
    return 0;
}

}  // namespace dropin
