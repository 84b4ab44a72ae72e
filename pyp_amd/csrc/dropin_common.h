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
                    if (!wait_for([&] { return dev_free[dslot]; })) return;
                    { std::lock_guard<std::mutex> lk(m); dev_free[dslot] = false; }
                    glo = it.lo;
                }
                if (!dbuf[dslot]) { dbuf[dslot] = ppm_device_alloc((size_t)group * chunk * sec); if (!dbuf[dslot]) { fail(std::string("ERROR: ") + ppm_last_error()); return; } }
                auto tb = Clock::now();
                if ((fail_at > 0 && k + 1 == fail_at) ||
                    ppm_device_upload((char *)dbuf[dslot] + (size_t)part * chunk * sec, pinned[it.slot], (size_t)(it.hi - it.lo) * sec) != 0) {
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
    void release(int dslot) { { std::lock_guard<std::mutex> lk(m); dev_free[dslot] = true; } cv.notify_all(); }
    void join() { if (reader_t.joinable()) reader_t.join(); if (uploader_t.joinable()) uploader_t.join(); }
};

}  // namespace dropin
