// dropin_common.h — what the compiled drop-in executables (bin/refine3d, bin/reconstruct3d) share: the here-doc answers,
// the `.cistem` table codec (src/pyp/inout/metadata/cistem_star_file.py:596-628, :694-776), the MRC header
// (src/pyp/inout/image/mrc.py:74-116), the hand-over to the Python implementation, the per-GPU lock and the three-stage
// stream (reader -> uploader -> compute) that moves a particle range from the stack file into device memory.
//
// No HIP in here: the executables use the C ABI only (include/ppm.h).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <functional>
#include <mutex>
#include <spawn.h>
#include <string>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/ppm.h"

extern char **environ;

namespace dropin {

using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t) { return std::chrono::duration<double>(Clock::now() - t).count(); }
inline double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

inline std::string self_dir() {
    char buf[4096];
    ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    std::string s(buf);
    size_t p = s.rfind('/');
    return p == std::string::npos ? "." : s.substr(0, p);
}

[[noreturn]] inline void die(const std::string &msg) {
    printf("%s\n", msg.find("ERROR") != std::string::npos ? msg.c_str() : ("ERROR: " + msg).c_str());
    fflush(stdout);
    _exit(1);          // no destructors: helper threads may still be inside the library
}

// Hand the call to the Python implementation (bin/<prog>.py) with the same stdin, as a CHILD process whose exit status becomes
// ours.  Never an exec of this process: a preloaded tool (rocprofv3 --pmc ...) may have initialised the GPU before main(), and
// replacing a GPU-initialised process is what the pool's machines do not survive.  The interpreter: $PPM_PYTHON, else `python3`
// found on PATH (so a venv / conda interpreter that has numpy is the one that runs), else /usr/bin/python3.
[[noreturn]] inline void hand_to_python(const char *script_name, const std::string &input) {
    fflush(stdout);
    int fd = memfd_create("dropin_stdin", 0);
    if (fd >= 0) {
        size_t done = 0;
        while (done < input.size()) {
            ssize_t w = write(fd, input.data() + done, input.size() - done);
            if (w <= 0) break;
            done += (size_t)w;
        }
        lseek(fd, 0, SEEK_SET);
    }
    const std::string script = self_dir() + "/" + script_name;
    const char *py = getenv("PPM_PYTHON");
    const bool by_path = !(py && *py);
    if (by_path) py = "python3";
    char *const argv[] = { (char *)py, (char *)script.c_str(), nullptr };
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    if (fd >= 0) posix_spawn_file_actions_adddup2(&fa, fd, 0);
    pid_t pid = 0;
    int rc = by_path ? posix_spawnp(&pid, py, &fa, nullptr, argv, environ) : posix_spawn(&pid, py, &fa, nullptr, argv, environ);
    if (rc != 0 && by_path) { char *const argv2[] = { (char *)"/usr/bin/python3", (char *)script.c_str(), nullptr }; rc = posix_spawn(&pid, "/usr/bin/python3", &fa, nullptr, argv2, environ); }
    posix_spawn_file_actions_destroy(&fa);
    if (fd >= 0) close(fd);
    if (rc != 0) die(std::string("ERROR: cannot start ") + py + " " + script + ": " + strerror(rc));
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
    _exit(WIFEXITED(status) ? WEXITSTATUS(status) : 1);
}

inline std::string strip(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}
inline bool parse_bool(const std::string &s, bool &v) {
    std::string t;
    for (char c : s) t += (char)tolower((unsigned char)c);
    if (t == "yes" || t == "y" || t == "true" || t == "1") { v = true; return true; }
    if (t == "no" || t == "n" || t == "false" || t == "0") { v = false; return true; }
    return false;
}
inline bool parse_num(const std::string &s, double &v) {
    if (s.empty()) return false;
    char *end = nullptr;
    v = strtod(s.c_str(), &end);
    return end && *end == 0 && end != s.c_str();
}
inline bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
inline bool ends_with(const std::string &s, const char *suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

inline std::string read_all_stdin() {
    std::string input;
    char buf[65536];
    ssize_t r;
    while ((r = read(0, buf, sizeof buf)) > 0) input.append(buf, (size_t)r);
    return input;
}
// the lines of the here-doc, stripped, up to 'eot' (pyp_amd/surface/prompts.py:read_answers)
inline std::vector<std::string> read_answers(const std::string &input) {
    std::vector<std::string> a;
    size_t p = 0;
    while (p <= input.size()) {
        size_t q = input.find('\n', p);
        if (q == std::string::npos) q = input.size();
        std::string s = strip(input.substr(p, q - p));
        if (s == "eot") break;
        a.push_back(s);
        p = q + 1;
    }
    while (!a.empty() && a.back().empty()) a.pop_back();
    return a;
}

// the 32 standard columns of a .cistem table in file order (cistem_star_file.py:596-628): code, type (2 = int32, 3 = float32, 9 = uint32)
static const long long kCodes[32] = { 1, 4, 4194304, 8388608, 8, 16, 32, 64, 128, 256, 2, 512, 1024, 2048, 4096, 16384, 32768, 65536, 131072, 262144,
                                      524288, 1048576, 2097152, 8589934592LL, 17179869184LL, 20, 15, 35, 70, 55, 11, 121 };
static const int kTypes[32] = { 9, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 3, 3 };

// rows of a .cistem file as doubles (what Parameters.get_data() holds); false = not the plain standard layout (-> Python)
inline bool read_cistem(const std::string &path, std::vector<double> &rows, long &n) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { close(fd); return false; }
    std::vector<unsigned char> buf((size_t)st.st_size);
    size_t done = 0;
    while (done < buf.size()) {
        ssize_t r = pread(fd, buf.data() + done, buf.size() - done, (off_t)done);
        if (r <= 0) { close(fd); return false; }
        done += (size_t)r;
    }
    close(fd);
    int32_t ncols, nrows;
    memcpy(&ncols, buf.data(), 4); memcpy(&nrows, buf.data() + 4, 4);
    if (ncols != 32 || nrows <= 0) return false;
    size_t pos = 8;
    if (buf.size() < pos + 9u * 32u) return false;
    for (int c = 0; c < 32; c++) {
        int64_t code; int8_t ty;
        memcpy(&code, buf.data() + pos, 8); ty = (int8_t)buf[pos + 8]; pos += 9;
        if (code != kCodes[c] || ty != kTypes[c]) return false;
    }
    if (buf.size() - pos < (size_t)nrows * 128u) return false;
    n = nrows;
    rows.resize((size_t)nrows * 32);
    const unsigned char *p = buf.data() + pos;
    for (long i = 0; i < nrows; i++)
        for (int c = 0; c < 32; c++, p += 4) {
            double v;
            if (kTypes[c] == 3) { float f; memcpy(&f, p, 4); v = f; }
            else if (kTypes[c] == 2) { int32_t q; memcpy(&q, p, 4); v = q; }
            else { uint32_t q; memcpy(&q, p, 4); v = q; }
            rows[(size_t)i * 32 + c] = v;
        }
    return true;
}

// the same table written back (pyp_amd/formats/cistem.py:write_parameters; the reference's writer, cistem_star_file.py:694-776):
// integers are the double cast the way numpy's astype does it (truncation toward zero); under a temporary name, renamed when complete
inline bool write_cistem(const std::string &path, const double *rows, long n) {
    std::vector<unsigned char> buf(8 + 9 * 32 + (size_t)n * 128);
    const int32_t ncols = 32, nrows = (int32_t)n;
    memcpy(buf.data(), &ncols, 4); memcpy(buf.data() + 4, &nrows, 4);
    size_t pos = 8;
    for (int c = 0; c < 32; c++) { const int64_t code = kCodes[c]; memcpy(buf.data() + pos, &code, 8); buf[pos + 8] = (unsigned char)kTypes[c]; pos += 9; }
    unsigned char *p = buf.data() + pos;
    for (long i = 0; i < n; i++)
        for (int c = 0; c < 32; c++, p += 4) {
            const double v = rows[(size_t)i * 32 + c];
            if (kTypes[c] == 3) { const float f = (float)v; memcpy(p, &f, 4); }
            else if (kTypes[c] == 2) { const int32_t q = (int32_t)v; memcpy(p, &q, 4); }
            else { const uint32_t q = (uint32_t)(long long)v; memcpy(p, &q, 4); }
        }
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return false;
    size_t done = 0;
    while (done < buf.size()) {
        ssize_t w = write(fd, buf.data() + done, buf.size() - done);
        if (w <= 0) { close(fd); unlink(tmp.c_str()); return false; }
        done += (size_t)w;
    }
    close(fd);
    if (rename(tmp.c_str(), path.c_str()) != 0) { unlink(tmp.c_str()); return false; }
    return true;
}

// the cheap part of read_cistem's test: 32 standard columns in file order
inline bool cistem_is_standard(const std::string &path) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    unsigned char b[8 + 9 * 32];
    const bool got = pread(fd, b, sizeof b, 0) == (ssize_t)sizeof b;
    close(fd);
    if (!got) return false;
    int32_t ncols, nrows;
    memcpy(&ncols, b, 4); memcpy(&nrows, b + 4, 4);
    if (ncols != 32 || nrows <= 0) return false;
    for (int c = 0; c < 32; c++) {
        int64_t code;
        memcpy(&code, b + 8 + 9 * c, 8);
        if (code != kCodes[c] || (int8_t)b[8 + 9 * c + 8] != kTypes[c]) return false;
    }
    return true;
}

struct MrcHead { int nx, ny, nz, mode; long offset; };
inline bool read_mrc_head(const std::string &path, MrcHead &h) {
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    unsigned char b[1024];
    bool ok = pread(fd, b, 1024, 0) == 1024;
    struct stat st;
    ok = ok && fstat(fd, &st) == 0;
    close(fd);
    if (!ok) return false;
    int32_t w[56];
    memcpy(w, b, sizeof w);
    h.nx = w[0]; h.ny = w[1]; h.nz = w[2]; h.mode = w[3];
    const int nsymbt = w[23];
    if (h.nx <= 0 || h.ny <= 0 || h.nz <= 0 || nsymbt < 0 || h.nx > 65536 || h.ny > 65536) return false;
    if (!(b[212] == 0x44 && (b[213] == 0x44 || b[213] == 0x41)) && !(b[212] == 0 && b[213] == 0)) return false;     // little-endian stamp (or none)
    h.offset = 1024 + nsymbt;
    return (long long)st.st_size >= h.offset + (long long)h.nx * h.ny * h.nz * 4;
}

// advisory per-GPU lock: PYP may start several processes per node (src/pyp/system/mpi.py:104); the same file as cli.gpu_lock
inline int gpu_lock(int dev) {
    const char *ld = getenv("PPM_LOCK_DIR");
    const std::string lp = std::string(ld && *ld ? ld : "/tmp") + "/pyp_amd_gpu" + std::to_string(dev) + ".lock";
    mode_t old = umask(0);
    int fd = open(lp.c_str(), O_RDWR | O_CREAT, 0666);
    umask(old);
    if (fd >= 0) flock(fd, LOCK_EX);
    return fd;
}
inline void gpu_unlock(int fd) { if (fd >= 0) { flock(fd, LOCK_UN); close(fd); } }

// ---- the three-stage stream: reader (ppm_host_read into page-locked buffers) -> uploader (ppm_device_upload into one of two
// device buffers) -> the caller's compute on the other.  `stop` ends every wait: a stage that fails posts an error item, the
// consumer calls abort(), and no thread is left blocked on a buffer that will never come back (a failed upload used to leave the
// reader waiting for its page-locked buffer for ever).
struct Stream {
    struct Item { long lo, hi; int slot; int err; };
    std::mutex m; std::condition_variable cv;
    std::atomic<bool> stop{false};
    std::deque<Item> filled, ready;
    bool pin_free[3] = { true, true, true }, dev_free[2] = { true, true };
    void *pinned[3] = { nullptr, nullptr, nullptr }, *dbuf[2] = { nullptr, nullptr };
    void *resident = nullptr;                           // set: one device buffer holds the whole range (n x sec bytes; the resident server's
                                                        // stack cache) - chunks are uploaded to their final place, nothing is recycled
    long n = 0, chunk = 1, group = 1; size_t sec = 0; int npin = 1, ndev = 1, nread = 8, fd = -1;
    bool contiguous = true;
    std::function<long long(long)> img_off;             // byte offset of image i of the range in the stack file
    std::function<bool(int)> wait_pinned;               // blocks until pinned[slot] has been page-locked (start-up thread); false = start-up failed
    double t_read = 0, t_up = 0, w_pin = 0, w_dev = 0;
    std::string err;
    std::thread reader_t, uploader_t;

    template <typename Pred> bool wait_for(Pred p) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return stop.load() || p(); }); return !stop.load(); }
    void post(std::deque<Item> &q, const Item &it) { { std::lock_guard<std::mutex> lk(m); q.push_back(it); } cv.notify_all(); }
    bool take(std::deque<Item> &q, Item &it) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop.load() || !q.empty(); });
        if (q.empty()) return false;
        it = q.front(); q.pop_front();
        return true;
    }
    void fail(const std::string &msg) { { std::lock_guard<std::mutex> lk(m); if (err.empty()) err = msg; } post(ready, { 0, 0, 0, 1 }); }
    void abort() { stop = true; { std::lock_guard<std::mutex> lk(m); } cv.notify_all(); }

    void start() {
        reader_t = std::thread([this] {
            for (long k = 0, lo = 0; lo < n; k++, lo += chunk) {
                const long hi = std::min(lo + chunk, n); const int slot = (int)(k % npin);
                auto ta = Clock::now();
                if (!wait_for([&] { return pin_free[slot]; })) return;
                { std::lock_guard<std::mutex> lk(m); pin_free[slot] = false; }
                if (wait_pinned && !wait_pinned(slot)) { fail("ERROR: page-locked staging memory could not be prepared"); return; }
                auto tb = Clock::now();
                if (contiguous) {
                    if (ppm_host_read(fd, img_off(lo), pinned[slot], (size_t)(hi - lo) * sec, nread) != 0) { fail(std::string("ERROR: reading the particle stack failed: ") + ppm_last_error()); return; }
                } else {
                    for (long i = lo; i < hi; i++)                                                 // scattered rows: image by image
                        if (ppm_host_read(fd, img_off(i), (char *)pinned[slot] + (size_t)(i - lo) * sec, sec, 1) != 0) { fail(std::string("ERROR: reading the particle stack failed: ") + ppm_last_error()); return; }
                }
                w_pin += secs(ta, tb); t_read += since(tb);
                post(filled, { lo, hi, slot, 0 });
            }
            post(filled, { -1, -1, 0, 0 });
        });
        uploader_t = std::thread([this] {
            long k = 0, glo = 0;
            const char *tf = getenv("PPM_TEST_FAIL_UPLOAD");           // test hook: the k-th upload (1-based) fails like a device error would
            const long fail_at = tf ? atol(tf) : 0;
            for (;;) {
                Item it;
                if (!take(filled, it) || it.lo < 0) return;
                const int dslot = (int)((k / group) % ndev); const long part = k % group;
                auto ta = Clock::now();
                if (part == 0) {
                    if (!resident) {
                        if (!wait_for([&] { return dev_free[dslot]; })) return;
                        { std::lock_guard<std::mutex> lk(m); dev_free[dslot] = false; }
                    }
                    glo = it.lo;
                }
                if (!resident && !dbuf[dslot]) { dbuf[dslot] = ppm_device_alloc((size_t)group * chunk * sec); if (!dbuf[dslot]) { fail(std::string("ERROR: ") + ppm_last_error()); return; } }
                auto tb = Clock::now();
                char *dst = resident ? (char *)resident + (size_t)it.lo * sec : (char *)dbuf[dslot] + (size_t)part * chunk * sec;
                if ((fail_at > 0 && k + 1 == fail_at) ||
                    ppm_device_upload(dst, pinned[it.slot], (size_t)(it.hi - it.lo) * sec) != 0) {
                    fail(fail_at > 0 && k + 1 == fail_at ? std::string("ERROR: upload failed (PPM_TEST_FAIL_UPLOAD)") : std::string("ERROR: ") + ppm_last_error());
                    return;
                }
                { std::lock_guard<std::mutex> lk(m); pin_free[it.slot] = true; } cv.notify_all();
                w_dev += secs(ta, tb); t_up += since(tb);
                if (part == group - 1 || it.hi == n) post(ready, { glo, it.hi, dslot, 0 });
                k++;
            }
        });
    }
    // next group of uploaded images: [lo, hi) of the range in dbuf[slot]; false = a stage failed (err says why; the stream is aborted)
    bool next(Item &it) {
        if (!take(ready, it) || it.err) { abort(); return false; }
        return true;
    }
    void *group_ptr(const Item &it) const { return resident ? (void *)((char *)resident + (size_t)it.lo * sec) : dbuf[it.slot]; }
    void release(int dslot) { if (resident) return; { std::lock_guard<std::mutex> lk(m); dev_free[dslot] = true; } cv.notify_all(); }
    void join() { if (reader_t.joinable()) reader_t.join(); if (uploader_t.joinable()) uploader_t.join(); }
};


// ---- text the executable prints: straight to stdout in a one-shot process, collected and sent back by the resident server
struct Out {
    std::string *sink = nullptr;            // null = stdout
    void print(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        va_list ap; va_start(ap, fmt);
        if (!sink) { vprintf(fmt, ap); va_end(ap); return; }
        char buf[4096];
        const int n = vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        if (n > 0) sink->append(buf, std::min((size_t)n, sizeof buf - 1));
    }
};
struct Fail { std::string msg; };           // an ERROR the caller must see: one-shot -> die(), server -> status 1 + the text

// identity of a file's contents for the resident caches: device, inode, size, modification time (ns)
struct FileId {
    unsigned long long dev = 0, ino = 0, size = 0; long long mtime_ns = 0;
    bool operator==(const FileId &o) const { return dev == o.dev && ino == o.ino && size == o.size && mtime_ns == o.mtime_ns; }
};
inline bool file_id(const std::string &path, FileId &id) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return false;
    id.dev = (unsigned long long)st.st_dev; id.ino = (unsigned long long)st.st_ino; id.size = (unsigned long long)st.st_size;
    id.mtime_ns = (long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec;
    return true;
}

// ---- the reconstruct3d call (39 answers; frealign.py:1780-1824), parsed and prepared up to the point where the GPU is needed
enum { RC_POS = 0, RC_DF1 = 6, RC_DF2 = 7, RC_OCC = 11, RC_SCORE = 14 };
struct ReconJob {
    std::vector<std::string> a;
    std::string stack, params, gstats, symmetry, res_file, dump1, dump2;
    long ifirst = 0, ilast = 0, n = 0; double px = 0; bool crop = false, contiguous = true, have_gs = false;
    MrcHead mh{}; int box = 0; size_t sec = 0;
    std::vector<double> rin; ppm_recon_cfg rc{};
};
// false = outside the compiled fast path (the Python implementation owns everything else, its refusals included)
inline bool recon_parse(const std::string &input, ReconJob &j) {
    j.a = read_answers(input);
    const std::vector<std::string> &a = j.a;
    if (a.size() < 39) return false;
    j.stack = a[0]; j.params = a[1]; j.gstats = a[2]; j.symmetry = a[8]; j.res_file = a[7];
    double first, last, outer_radius, res_limit, bfac, thr, padding;
    bool score_weighting, dose, normalize, adjust, invert, excl, split_eo, by_pind, center, blur, thrref, dump;
    bool ok = parse_num(a[9], first) && parse_num(a[10], last) && parse_num(a[11], j.px) && parse_num(a[14], outer_radius) && parse_num(a[15], res_limit) &&
              parse_num(a[17], bfac) && parse_bool(a[18], score_weighting) && parse_bool(a[21], dose);
    // answers this build refuses or treats specially unless they carry the value PYP always sends (frealign.py:1763-1770, :1796-1808):
    // the Python implementation owns the messages and the tilt window, so anything else goes there
    double mass, inner_radius, res_reference, tilt_lo, tilt_hi, smoothing, threads;
    ok = ok && parse_num(a[12], mass) && parse_num(a[13], inner_radius) && parse_num(a[16], res_reference) && parse_num(a[19], tilt_lo) && parse_num(a[20], tilt_hi);
    if (!ok || dose) return false;                     // dose weighting: five more answers, side files, a table over the whole file
    ok = parse_num(a[22], thr) && parse_num(a[23], smoothing) && parse_num(a[24], padding) && parse_bool(a[25], normalize) && parse_bool(a[26], adjust) &&
         parse_bool(a[27], invert) && parse_bool(a[28], excl) && parse_bool(a[29], j.crop) && parse_bool(a[30], split_eo) && parse_bool(a[31], by_pind) &&
         parse_bool(a[32], center) && parse_bool(a[33], blur) && parse_bool(a[34], thrref) && parse_bool(a[35], dump) && parse_num(a[38], threads);
    ok = ok && inner_radius == 0.0 && res_reference == 0.0 && smoothing == 1.0 && tilt_lo <= 0.0 && tilt_hi < 0.0;
    for (int k = 0; k < 9; k++) ok = ok && !a[k].empty();
    ok = ok && !a[36].empty() && !a[37].empty();
    if (!ok) return false;
    j.dump1 = a[36]; j.dump2 = a[37];
    if (center || thrref || excl || !split_eo || !dump || blur || std::fabs(padding - 1.0) > 1e-6 || !ends_with(j.params, ".cistem") ||
        !exists(j.stack) || !exists(j.params) || first < 1 || last < first || j.px <= 0)
        return false;
    j.ifirst = (long)first; j.ilast = (long)last;
    j.have_gs = j.gstats != "null" && exists(j.gstats);
    if (!cistem_is_standard(j.params) || (j.have_gs && !cistem_is_standard(j.gstats)) || !read_mrc_head(j.stack, j.mh) || j.mh.mode != 2 || j.mh.nx != j.mh.ny) return false;
    j.box = j.mh.nx; j.sec = (size_t)j.box * j.box * 4;
    memset(&j.rc, 0, sizeof j.rc);
    j.rc.box = j.box; j.rc.pixel_size = (float)j.px; j.rc.res_limit = (float)res_limit; j.rc.score_weight_bfactor = score_weighting ? (float)bfac : 0.f;
    j.rc.score_threshold = (float)thr; j.rc.normalize = normalize; j.rc.invert = invert; j.rc.split_by_pind = by_pind; j.rc.mask_radius = (float)outer_radius;
    j.rc.score_average = adjust ? 1.f : 0.f;           // carries "adjust scores" to recon_rows, which replaces it by the mean score
    return true;
}
// the rows of the range, the score regression and the mean score (pyp_amd/surface/cli.py:reconstruct3d_main); throws Fail
inline void recon_rows(ReconJob &j) {
    const bool adjust = j.rc.score_average != 0.f;
    std::vector<double> rows; long nrows = 0;
    if (!read_cistem(j.params, rows, nrows)) throw Fail{ "ERROR: " + j.params + ": binary file is broken" };
    std::vector<double> gs; long ngs = 0;
    if (j.have_gs && !read_cistem(j.gstats, gs, ngs)) throw Fail{ "ERROR: " + j.gstats + ": binary file is broken" };
    std::vector<double> &rin = j.rin;
    rin.clear();
    for (long i = 0; i < nrows; i++) {
        const double pos = rows[(size_t)i * 32 + RC_POS];
        if (pos >= j.ifirst && pos <= j.ilast) rin.insert(rin.end(), rows.begin() + (size_t)i * 32, rows.begin() + (size_t)(i + 1) * 32);
    }
    const long n = j.n = (long)(rin.size() / 32);
    if (n == 0) throw Fail{ "ERROR: no rows with POSITION_IN_STACK in " + std::to_string(j.ifirst) + ".." + std::to_string(j.ilast) };
    j.contiguous = true;
    double pmax = 0, pmin = 1e300;
    for (long i = 0; i < n; i++) {
        const double pos = rin[(size_t)i * 32 + RC_POS];
        pmax = std::max(pmax, pos); pmin = std::min(pmin, pos);
        if (i && pos != rin[(size_t)(i - 1) * 32 + RC_POS] + 1) j.contiguous = false;
    }
    if (pmax > j.mh.nz || pmin < 1) throw Fail{ "ERROR: " + j.stack + ": stack has " + std::to_string(j.mh.nz) + " images, rows ask for " + std::to_string((long)pmax) };
    long nused = 0;
    for (long i = 0; i < n; i++) nused += rin[(size_t)i * 32 + RC_OCC] > 0;
    if (adjust && nused > 10) {
        long double sx = 0, sy = 0;
        for (long i = 0; i < n; i++) if (rin[(size_t)i * 32 + RC_OCC] > 0) { sx += 0.5 * (rin[(size_t)i * 32 + RC_DF1] + rin[(size_t)i * 32 + RC_DF2]); sy += rin[(size_t)i * 32 + RC_SCORE]; }
        const long double mx = sx / nused, my = sy / nused;
        long double sxx = 0, sxy = 0;
        for (long i = 0; i < n; i++) if (rin[(size_t)i * 32 + RC_OCC] > 0) {
            const long double dx = 0.5 * (rin[(size_t)i * 32 + RC_DF1] + rin[(size_t)i * 32 + RC_DF2]) - mx;
            sxx += dx * dx; sxy += dx * (rin[(size_t)i * 32 + RC_SCORE] - my);
        }
        if (sxx > 0) {
            const double slope = (double)(sxy / sxx);
            for (long i = 0; i < n; i++) rin[(size_t)i * 32 + RC_SCORE] -= slope * (0.5 * (rin[(size_t)i * 32 + RC_DF1] + rin[(size_t)i * 32 + RC_DF2]) - (double)mx);
        }
    }
    double score_avg = 0;
    if (j.have_gs) score_avg = gs[RC_SCORE];
    else if (nused) { long double s = 0; for (long i = 0; i < n; i++) if (rin[(size_t)i * 32 + RC_OCC] > 0) s += rin[(size_t)i * 32 + RC_SCORE]; score_avg = (double)(s / nused); }
    j.rc.score_average = (float)score_avg;
}
inline void recon_banner(const ReconJob &j, Out &o, const char *how) {
    o.print("\n        **   Welcome to Reconstruct3D (MI355X / libpypmatch, %s)   **\n\n", how);
    static const char *names[39] = { "stack", "input_params", "global_stats", "reference", "map1", "map2", "output", "res_file", "symmetry", "first", "last",
        "pixel_size", "molecular_mass", "inner_radius", "outer_radius", "res_limit", "res_reference", "score_bfactor", "score_weighting", "min_tilt_score",
        "max_tilt_score", "dose_weighting", "score_threshold", "smoothing", "padding", "normalize", "adjust_scores", "invert", "exclude_edges", "crop",
        "split_even_odd", "per_particle_splitting", "center_mass", "likelihood_blurring", "threshold_reference", "dump", "dump_1", "dump_2", "threads" };
    for (int k = 0; k < 39; k++) o.print("%-28s: %s\n", names[k], j.a[k].c_str());
    if (j.crop) o.print("NOTE: crop = yes has no effect: the full box is transformed\n");
}

inline int write_dump_file(const std::string &path, int box, float pixel, long long count, const float *data, size_t nfloat) {
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
// downloads the two half maps (into h_even / h_odd, each ppm_accum_floats / 2 floats) and writes dumps + the .res file; throws Fail
inline void recon_outputs(const ReconJob &j, ppm_accum_t *acc, float *h_even, float *h_odd, long &c0, long &c1) {
    const size_t half = ppm_accum_floats(j.box) / 2;
    if (ppm_accum_download_range(acc, h_even, 0, half) != 0 || ppm_accum_download_range(acc, h_odd, half, half) != 0) throw Fail{ ppm_last_error() };
    c0 = ppm_accum_count(acc, 0); c1 = ppm_accum_count(acc, 1);
    int e1 = 0, e2 = 0;
    std::thread w2([&] { e2 = write_dump_file(j.dump2, j.box, (float)j.px, c0, h_even, half); });                   // even keys -> map 2
    e1 = write_dump_file(j.dump1, j.box, (float)j.px, c1, h_odd, half);                                            // odd keys  -> map 1
    w2.join();
    if (e1 || e2) { unlink(j.dump1.c_str()); unlink(j.dump2.c_str()); throw Fail{ "ERROR: reconstruct3d: could not write " + (e1 ? j.dump1 : j.dump2) }; }
    if (FILE *f = fopen(j.res_file.c_str(), "w")) {
        fprintf(f, "C Reconstruct3D (libpypmatch): particles %ld..%ld, inserted %ld + %ld\n", j.ifirst, j.ilast, c1, c0);
        fclose(f);
    }
}
inline void recon_footer(Out &o) {
    o.print("NOTE: the dump files are in libpypmatch's own format (PPMDUMP1): only this build's local_merge3d / merge3d read them "
            "(frealign.py:1852 consumers must be replaced together, INTEGRATION.md 1)\n");
    o.print("\nNormal termination, intermediate files dumped\n");
    o.print("\nReconstruct3D: Normal termination\n\n");
}

// ---- the refine3d call (50 answers; frealign.py:3918-3994)
enum { RF_POS = 0, RF_PSI = 1, RF_THETA = 2, RF_PHI = 3, RF_SHX = 4, RF_SHY = 5, RF_SCORE = 14 };
struct RefineJob {
    std::vector<std::string> a;
    std::string stack, params, reference, out_params, out_changes, symmetry;
    long ifirst = 0, ilast = 0, n = 0; double px = 0; int pad = 1; bool contiguous = true;
    MrcHead mh{}; int box = 0; size_t sec = 0;
    std::vector<double> rin; ppm_refine_cfg cfg{};
};
inline bool refine_parse(const std::string &input, RefineJob &j) {
    j.a = read_answers(input);
    const std::vector<std::string> &a = j.a;
    if (a.size() < 50 || !ends_with(a[1], ".cistem")) return false;          // the 45-answer .par surface lives in Python
    for (int k = 0; k < 11; k++) if (a[k].empty()) return false;
    double num[50] = { 0 }; bool flag[50] = { false };
    static const int kNum[] = { 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34 };
    static const int kBool[] = { 5, 6, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49 };
    bool ok = true;
    for (int k : kNum) ok = ok && parse_num(a[k], num[k]);
    for (int k : kBool) ok = ok && parse_bool(a[k], flag[k]);
    if (!ok) return false;
    j.stack = a[0]; j.params = a[1]; j.reference = a[3]; j.out_params = a[8]; j.out_changes = a[9]; j.symmetry = a[10];
    const double first = num[11], last = num[12], fraction = num[13], inner_radius = num[16], padding = num[34];
    j.px = num[14];
    const bool use_stats = flag[5], use_priors = flag[6], calc_match = flag[42], mask_2d = flag[43];
    j.pad = (int)std::lround(padding);
    // what the Python implementation owns (non-default answers, refusals and their messages)
    if (use_stats || use_priors || calc_match || mask_2d || flag[47] || flag[48] || flag[49] || inner_radius != 0.0 || fraction != 1.0 || num[21] < 0 ||
        std::fabs(padding - j.pad) > 1e-6 || (j.pad != 1 && j.pad != 2 && j.pad != 4) || first < 1 || last < first || j.px <= 0 || !ends_with(j.out_params, ".cistem") ||
        !exists(j.stack) || !exists(j.params) || !exists(j.reference))
        return false;
    MrcHead rh;
    if (!cistem_is_standard(j.params) || !read_mrc_head(j.stack, j.mh) || j.mh.mode != 2 || j.mh.nx != j.mh.ny || !read_mrc_head(j.reference, rh) || rh.mode != 2 ||
        rh.nx != j.mh.nx || rh.ny != j.mh.nx || rh.nz != j.mh.nx || j.mh.nx * j.pad > 512)
        return false;
    j.ifirst = (long)first; j.ilast = (long)last; j.box = j.mh.nx; j.sec = (size_t)j.box * j.box * 4;
    ppm_refine_cfg &cfg = j.cfg;                       // pyp_amd/surface/cli.py:refine_cfg_from_answers
    memset(&cfg, 0, sizeof cfg);
    cfg.box = j.box; cfg.pixel_size = (float)j.px; cfg.molecular_mass_kda = (float)num[15]; cfg.mask_radius = (float)num[17];
    cfg.res_low = (float)num[18]; cfg.res_high = (float)num[19]; cfg.res_signed_cc = (float)num[20]; cfg.res_classification = (float)num[21];
    cfg.search_mask_radius = (float)num[22]; cfg.res_search = (float)(num[23] != 0 ? num[23] : num[19]); cfg.angular_step = (float)num[24];
    cfg.top_hits = (int)num[25]; cfg.search_range_x = (float)num[26]; cfg.search_range_y = (float)num[27];
    cfg.defocus_range = (float)num[32]; cfg.defocus_step = (float)num[33];
    cfg.global_search = flag[35]; cfg.local_refine = flag[36];
    cfg.refine_psi = flag[37]; cfg.refine_theta = flag[38]; cfg.refine_phi = flag[39]; cfg.refine_x = flag[40]; cfg.refine_y = flag[41];
    cfg.refine_defocus = flag[44]; cfg.normalize = flag[45]; cfg.invert = flag[46];
    snprintf(cfg.symmetry, sizeof cfg.symmetry, "%.7s", j.symmetry.c_str());
    return true;
}
inline void refine_rows(RefineJob &j) {
    std::vector<double> rows; long nrows = 0;
    if (!read_cistem(j.params, rows, nrows)) throw Fail{ "ERROR: " + j.params + ": binary file is broken" };
    j.rin.clear();
    for (long i = 0; i < nrows; i++) {
        const double pos = rows[(size_t)i * 32 + RF_POS];
        if (pos >= j.ifirst && pos <= j.ilast) j.rin.insert(j.rin.end(), rows.begin() + (size_t)i * 32, rows.begin() + (size_t)(i + 1) * 32);
    }
    const long n = j.n = (long)(j.rin.size() / 32);
    if (n == 0) throw Fail{ "ERROR: no rows with POSITION_IN_STACK in " + std::to_string(j.ifirst) + ".." + std::to_string(j.ilast) };
    j.contiguous = true;
    double pmax = 0, pmin = 1e300;
    for (long i = 0; i < n; i++) {
        const double pos = j.rin[(size_t)i * 32 + RF_POS];
        pmax = std::max(pmax, pos); pmin = std::min(pmin, pos);
        if (i && pos != j.rin[(size_t)(i - 1) * 32 + RF_POS] + 1) j.contiguous = false;
    }
    if (pmax > j.mh.nz || pmin < 1) throw Fail{ "ERROR: " + j.stack + ": stack has " + std::to_string(j.mh.nz) + " images, rows ask for " + std::to_string((long)pmax) };
}
inline void refine_banner(const RefineJob &j, Out &o, const char *how) {
    static const char *kNames[50] = { "stack", "input_params", "global_stats", "reference", "statistics", "use_statistics", "use_priors", "match_out", "output_params",
        "output_changes", "symmetry", "first", "last", "fraction", "pixel_size", "molecular_mass", "inner_radius", "outer_radius", "res_low", "res_high",
        "res_signed_cc", "res_classification", "search_mask_radius", "res_search", "angular_step", "top_hits", "search_range_x", "search_range_y", "focus_x",
        "focus_y", "focus_z", "focus_r", "defocus_range", "defocus_step", "padding", "global_search", "local_refine", "refine_psi", "refine_theta", "refine_phi",
        "refine_x", "refine_y", "calc_match", "mask_2d", "refine_defocus", "normalize", "invert", "exclude_edges", "normalize_reference", "threshold_reference" };
    o.print("\n        **   Welcome to Refine3D (MI355X / libpypmatch, %s)   **\n\n", how);
    for (int k = 0; k < 50; k++) o.print("%-28s: %s\n", kNames[k], j.a[k].c_str());
}
// a float32 MRC volume (mode 2, little-endian), nx = ny = nz = n
inline bool read_volume(const std::string &path, int n, std::vector<float> &vol) {
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
// output tables + the log's table of the first rows; throws Fail
inline void refine_outputs(const RefineJob &j, const std::vector<double> &rout, const std::string &note, Out &o, double &mean_score) {
    const long n = j.n;
    if (!note.empty()) o.print("\n%s\n", note.c_str());
    std::vector<double> changes((size_t)n * 32);
    for (long i = 0; i < n; i++) {
        for (int c = 0; c < 32; c++) changes[(size_t)i * 32 + c] = rout[(size_t)i * 32 + c] - j.rin[(size_t)i * 32 + c];
        changes[(size_t)i * 32 + RF_POS] = j.rin[(size_t)i * 32 + RF_POS];
    }
    if (!write_cistem(j.out_params, rout.data(), n)) throw Fail{ "ERROR: refine3d: could not write " + j.out_params };
    if (j.out_changes != "/dev/null" && j.out_changes != "null") {
        if (!ends_with(j.out_changes, ".cistem")) { unlink(j.out_params.c_str()); throw Fail{ "ERROR: output " + j.out_changes + " must have .cistem extension" }; }
        if (!write_cistem(j.out_changes, changes.data(), n)) { unlink(j.out_params.c_str()); throw Fail{ "ERROR: refine3d: could not write " + j.out_changes }; }
    }
    o.print("\n   NO     PSI   THETA     PHI       SHX       SHY     SCORE   CHANGE\n");
    long double ssum = 0;
    for (long i = 0; i < n; i++) {
        const double *r = &rout[(size_t)i * 32];
        ssum += r[RF_SCORE];
        if (i < 50) o.print("%7d%8.2f%8.2f%8.2f%10.2f%10.2f%10.4f%9.4f\n", (int)r[RF_POS], r[RF_PSI], r[RF_THETA], r[RF_PHI], r[RF_SHX], r[RF_SHY], r[RF_SCORE], changes[(size_t)i * 32 + RF_SCORE]);
    }
    mean_score = (double)(ssum / n);
}

}  // namespace dropin
