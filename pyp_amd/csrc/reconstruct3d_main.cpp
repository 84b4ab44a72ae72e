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
#include <mutex>
#include <string>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/ppm.h"

namespace {

using Clock = std::chrono::steady_clock;
double since(Clock::time_point t) { return std::chrono::duration<double>(Clock::now() - t).count(); }

std::string self_dir() {
    char buf[4096];
    ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return ".";
    buf[n] = 0;
    std::string s(buf);
    size_t p = s.rfind('/');
    return p == std::string::npos ? "." : s.substr(0, p);
}

// hand the call to the Python implementation with the same stdin (nothing has touched the GPU yet)
[[noreturn]] void fall_back(const std::string &input) {
    int fd = memfd_create("reconstruct3d_stdin", 0);
    if (fd >= 0) {
        size_t done = 0;
        while (done < input.size()) {
            ssize_t w = write(fd, input.data() + done, input.size() - done);
            if (w <= 0) break;
            done += (size_t)w;
        }
        lseek(fd, 0, SEEK_SET);
        dup2(fd, 0);
        close(fd);
    }
    const std::string script = self_dir() + "/reconstruct3d.py";
    const char *py = getenv("PPM_PYTHON");
    if (!py || !*py) py = "/usr/bin/python3";
    char *const argv[] = { (char *)py, (char *)script.c_str(), nullptr };
    execv(py, argv);
    printf("ERROR: reconstruct3d: cannot start %s %s\n", py, script.c_str());
    fflush(stdout);
    _exit(1);
}

[[noreturn]] void die(const std::string &msg) {
    printf("%s\n", msg.rfind("ERROR", 0) == 0 || msg.find("ERROR") != std::string::npos ? msg.c_str() : ("ERROR: " + msg).c_str());
    fflush(stdout);
    _exit(1);          // no destructors: helper threads may still be inside the library
}

std::string strip(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

bool parse_bool(const std::string &s, bool &v) {
    std::string t;
    for (char c : s) t += (char)tolower((unsigned char)c);
    if (t == "yes" || t == "y" || t == "true" || t == "1") { v = true; return true; }
    if (t == "no" || t == "n" || t == "false" || t == "0") { v = false; return true; }
    return false;
}
bool parse_num(const std::string &s, double &v) {
    if (s.empty()) return false;
    char *end = nullptr;
    v = strtod(s.c_str(), &end);
    return end && *end == 0 && end != s.c_str();
}
bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool ends_with(const std::string &s, const char *suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

// the 32 standard columns of a .cistem table in file order (src/pyp/inout/metadata/cistem_star_file.py:596-628): code, type
// (2 = int32, 3 = float32, 9 = uint32)
const long long kCodes[32] = { 1, 4, 4194304, 8388608, 8, 16, 32, 64, 128, 256, 2, 512, 1024, 2048, 4096, 16384, 32768, 65536, 131072, 262144,
                               524288, 1048576, 2097152, 8589934592LL, 17179869184LL, 20, 15, 35, 70, 55, 11, 121 };
const int kTypes[32] = { 9, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 3, 3 };
enum { C_POS = 0, C_DF1 = 6, C_DF2 = 7, C_OCC = 11, C_SCORE = 14, C_PIND = 26 };

// rows of a .cistem file as doubles (what Parameters.get_data() holds); false = not the plain standard layout (-> Python)
bool read_cistem(const std::string &path, std::vector<double> &rows, long &n) {
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

// the cheap part of the same test: 32 standard columns in file order
bool cistem_is_standard(const std::string &path) {
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
bool read_mrc_head(const std::string &path, MrcHead &h) {
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

// ---- a tiny blocking queue / flag set for the three pipeline stages
template <typename T> struct Queue {
    std::mutex m; std::condition_variable cv; std::deque<T> q;
    void put(const T &v) { { std::lock_guard<std::mutex> lk(m); q.push_back(v); } cv.notify_one(); }
    T get() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !q.empty(); }); T v = q.front(); q.pop_front(); return v; }
};
struct Flag {
    std::mutex m; std::condition_variable cv; bool on = true;
    void set() { { std::lock_guard<std::mutex> lk(m); on = true; } cv.notify_all(); }
    void wait_clear() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return on; }); on = false; }
};
struct Item { long lo, hi; int slot; int err; };

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
    std::string input;
    {
        char buf[65536];
        ssize_t r;
        while ((r = read(0, buf, sizeof buf)) > 0) input.append(buf, (size_t)r);
    }
    if (const char *e = getenv("PPM_NATIVE")) if (!strcmp(e, "0")) fall_back(input);
    // ---- the answers (pyp_amd/surface/prompts.py: read_answers, parse_reconstruct3d)
    std::vector<std::string> a;
    {
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
    }
    if (a.size() < 39) fall_back(input);
    const std::string stack = a[0], params = a[1], gstats = a[2], symmetry = a[8], res_file = a[7];
    double first, last, px, outer_radius, res_limit, bfac, thr, padding;
    bool score_weighting, dose, normalize, adjust, invert, excl, crop, split_eo, by_pind, center, blur, thrref, dump;
    bool ok = parse_num(a[9], first) && parse_num(a[10], last) && parse_num(a[11], px) && parse_num(a[14], outer_radius) && parse_num(a[15], res_limit) &&
              parse_num(a[17], bfac) && parse_bool(a[18], score_weighting) && parse_bool(a[21], dose);
    double dummy;
    ok = ok && parse_num(a[12], dummy) && parse_num(a[13], dummy) && parse_num(a[16], dummy) && parse_num(a[19], dummy) && parse_num(a[20], dummy);
    if (!ok || dose) fall_back(input);                 // dose weighting: five more answers, side files, a table over the whole file
    ok = parse_num(a[22], thr) && parse_num(a[23], dummy) && parse_num(a[24], padding) && parse_bool(a[25], normalize) && parse_bool(a[26], adjust) &&
         parse_bool(a[27], invert) && parse_bool(a[28], excl) && parse_bool(a[29], crop) && parse_bool(a[30], split_eo) && parse_bool(a[31], by_pind) &&
         parse_bool(a[32], center) && parse_bool(a[33], blur) && parse_bool(a[34], thrref) && parse_bool(a[35], dump) && parse_num(a[38], dummy);
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
    int lockfd = -1;
    {
        const char *ld = getenv("PPM_LOCK_DIR");
        const std::string lp = std::string(ld && *ld ? ld : "/tmp") + "/pyp_amd_gpu" + std::to_string(dev) + ".lock";
        mode_t old = umask(0);
        lockfd = open(lp.c_str(), O_RDWR | O_CREAT, 0666);
        umask(old);
        if (lockfd >= 0) flock(lockfd, LOCK_EX);
    }
    const int box = mh.nx;
    const size_t sec = (size_t)box * box * 4;
    long chunk_mb = 256, call_mb = 2048;
    if (const char *e = getenv("PPM_IO_CHUNK_MB")) chunk_mb = std::max(1L, atol(e));
    // one staging buffer: whole images, no larger than the range can fill (page-locking costs ~0.2 s per GB)
    const size_t pin_bytes = std::min(std::max((size_t)16, ((size_t)chunk_mb << 20) / sec), (size_t)(ilast - ifirst + 1)) * sec;
    ppm_accum_t *acc = nullptr;
    void *pinned[3] = { nullptr, nullptr, nullptr };
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
    const int npin = (int)std::min(3L, nchunks), ndev = (int)std::min(2L, (nchunks + group - 1) / group);
    want_pinned = npin;
    const int nread = getenv("PPM_IO_THREADS") ? std::max(1, std::min(16, atoi(getenv("PPM_IO_THREADS")))) : 8;
    const int fd = open(stack.c_str(), O_RDONLY);
    if (fd < 0) bail("ERROR: reconstruct3d: cannot open " + stack);
    auto img_off = [&](long i) { return mh.offset + (long long)((long)rin[(size_t)i * 32 + C_POS] - 1) * (long long)sec; };
    void *dbuf[2] = { nullptr, nullptr };
    Flag pin_free[3], dev_free[2];
    Queue<Item> filled, ready;
    double t_read = 0, t_up = 0, w_pin = 0, w_dev = 0;
    std::thread reader([&] {
        for (long k = 0, lo = 0; lo < n; k++, lo += chunk) {
            const long hi = std::min(lo + chunk, n); const int slot = (int)(k % npin);
            auto ta = Clock::now();
            pin_free[slot].wait_clear();
            if (!wait_stage(2 + slot)) { filled.put({ 0, 0, 0, 1 }); return; }          // page-locked by the start-up thread
            auto tb = Clock::now();
            if (contiguous) {
                if (ppm_host_read(fd, img_off(lo), pinned[slot], (size_t)(hi - lo) * sec, nread) != 0) { filled.put({ 0, 0, 0, 1 }); return; }
            } else {
                for (long i = lo; i < hi; i++)                                                 // scattered rows: image by image
                    if (ppm_host_read(fd, img_off(i), (char *)pinned[slot] + (size_t)(i - lo) * sec, sec, 1) != 0) { filled.put({ 0, 0, 0, 1 }); return; }
            }
            w_pin += std::chrono::duration<double>(tb - ta).count(); t_read += since(tb);
            filled.put({ lo, hi, slot, 0 });
        }
        filled.put({ -1, -1, 0, 0 });
    });
    std::thread uploader([&] {
        long k = 0, glo = 0;
        for (;;) {
            Item it = filled.get();
            if (it.err) { ready.put(it); return; }
            if (it.lo < 0) return;
            const int dslot = (int)((k / group) % ndev); const long part = k % group;
            auto ta = Clock::now();
            if (part == 0) { dev_free[dslot].wait_clear(); glo = it.lo; }
            if (!dbuf[dslot]) { dbuf[dslot] = ppm_device_alloc((size_t)group * chunk * sec); if (!dbuf[dslot]) { ready.put({ 0, 0, 0, 1 }); return; } }
            auto tb = Clock::now();
            if (ppm_device_upload((char *)dbuf[dslot] + (size_t)part * chunk * sec, pinned[it.slot], (size_t)(it.hi - it.lo) * sec) != 0) { ready.put({ 0, 0, 0, 1 }); return; }
            pin_free[it.slot].set();
            w_dev += std::chrono::duration<double>(tb - ta).count(); t_up += since(tb);
            if (part == group - 1 || it.hi == n) ready.put({ glo, it.hi, dslot, 0 });
            k++;
        }
    });
    double t_comp = 0, w_data = 0; long ncalls = 0;
    for (long lo = 0; lo < n;) {
        auto ta = Clock::now();
        Item it = ready.get();
        if (it.err) { reader.join(); uploader.join(); starter.join(); die(!up_err.empty() ? up_err : "ERROR: reconstruct3d: reading or uploading the particle stack failed"); }
        auto tb = Clock::now();
        if (ppm_insert_batch(acc, &rc, dbuf[it.slot], 1, (int)(it.hi - it.lo), rin.data() + (size_t)it.lo * 32) != 0) die(ppm_last_error());
        dev_free[it.slot].set();
        w_data += std::chrono::duration<double>(tb - ta).count(); t_comp += since(tb); ncalls++;
        lo = it.hi;
    }
    reader.join(); uploader.join(); starter.join();
    close(fd);
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
    for (void *p : dbuf) if (p) ppm_device_free(p);
    if (lockfd >= 0) { flock(lockfd, LOCK_UN); close(lockfd); }
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
