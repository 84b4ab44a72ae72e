// ppm_server — the resident per-GPU server of the compiled drop-in executables (dropin_server.h): keeps the GPU context, the uploaded
// particle ranges, the prepared references and the page-locked staging buffers between the calls PYP makes
// (src/pyp/refine/frealign/frealign.py:1780-1824 reconstruct3d, :3918-3994 refine3d; SURVEY.md 8b "multiplex via a daemon").
//
//   ppm_server [--device N] [--daemon]     serve ($PPM_LOCK_DIR/pyp_amd_gpu<N>.sock); --daemon: detach and return at once
//   ppm_server [--device N] --stop         ask the running server to free the device and exit
//   ppm_server [--device N] --stats        print what it holds
// Started on demand by bin/refine3d / bin/reconstruct3d when PPM_STACK_CACHE=1.  Settings: PPM_STACK_CACHE_GB (resident particle
// ranges, default 160), PPM_STACK_CACHE_IDLE_S (exit after so many seconds without a request, default 600).
//
// Built by pyp_amd/csrc/Makefile into bin/ppm_server (g++, no HIP: the C ABI only).
#include "dropin_server.h"

using namespace dropin;

int main(int argc, char **argv) {
    int dev = getenv("PPM_DEVICE") ? atoi(getenv("PPM_DEVICE")) : 0;
    bool daemon = false, stop = false, stats = false;
    for (int i = 1; i < argc; i++) {
        const std::string s = argv[i];
        if (s == "--device" && i + 1 < argc) dev = atoi(argv[++i]);
        else if (s == "--daemon") daemon = true;
        else if (s == "--stop") stop = true;
        else if (s == "--stats") stats = true;
        else { fprintf(stderr, "usage: ppm_server [--device N] [--daemon | --stop | --stats]\n"); return 2; }
    }
    if (stop || stats) {
        int st = 1; std::string text;
        if (!server_call(dev, stop ? kProgStop : kProgStats, "", st, text)) { printf("no server is running for device %d\n", dev); return stop ? 0 : 1; }
        fputs(text.c_str(), stdout);
        return st;
    }
    if (daemon) {                                   // nothing has touched the GPU yet: a plain fork is safe
        const pid_t pid = fork();
        if (pid < 0) return 1;
        if (pid > 0) return 0;
        setsid();
        int nfd = open("/dev/null", O_RDWR);
        if (nfd >= 0) { dup2(nfd, 0); dup2(nfd, 1); dup2(nfd, 2); close(nfd); }      // the log is opened once this process holds the guard lock
    }
    setenv("PPM_SYNC", "block", 0);
    // ---- one server per device: the server holds an exclusive lock on <socket>.lock for its lifetime (two clients may start servers at the
    // same moment: without it the second would unlink the first one's socket and leave it serving nobody); a live socket means the same
    const std::string path = server_socket_path(dev);
    const int guard_fd = open((path + ".lock").c_str(), O_RDWR | O_CREAT, 0600);
    if (guard_fd < 0 || flock(guard_fd, LOCK_EX | LOCK_NB) != 0) { printf("a server is already running (or starting) for device %d\n", dev); return 0; }
    { int fd = connect_server(dev); if (fd >= 0) { close(fd); printf("a server is already running for device %d\n", dev); return 0; } }
    if (daemon) {                                   // this process is THE server of the device: only now may it touch the log (appended, never followed through a link)
        const std::string log = server_dir() + "/pyp_amd_gpu" + std::to_string(dev) + ".u" + std::to_string((long)getuid()) + ".server.log";
        const int lfd = open(log.c_str(), O_WRONLY | O_CREAT | O_APPEND | O_NOFOLLOW, 0600);
        if (lfd >= 0) { dup2(lfd, 1); dup2(lfd, 2); close(lfd); }
    }
    unlink(path.c_str());
    int ls = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un ad; memset(&ad, 0, sizeof ad); ad.sun_family = AF_UNIX;
    if (ls < 0 || path.size() >= sizeof ad.sun_path) { printf("ERROR: ppm_server: cannot make the socket %s\n", path.c_str()); return 1; }
    strcpy(ad.sun_path, path.c_str());
    const mode_t old = umask(0077);
    const int brc = bind(ls, (sockaddr *)&ad, sizeof ad);
    umask(old);
    if (brc != 0 || listen(ls, 16) != 0) { printf("ERROR: ppm_server: cannot listen on %s: %s\n", path.c_str(), strerror(errno)); return 1; }
    Cache cache;
    cache.dev = dev;
    if (const char *e = getenv("PPM_STACK_CACHE_GB")) { const double gb = atof(e); if (gb > 0) cache.budget = (size_t)(gb * (double)(1ull << 30)); }
    if (const char *e = getenv("PPM_STACK_CACHE_HEADROOM_GB")) { const double gb = atof(e); if (gb >= 0) cache.headroom = (size_t)(gb * (double)(1ull << 30)); }
    const double idle_s = getenv("PPM_STACK_CACHE_IDLE_S") ? atof(getenv("PPM_STACK_CACHE_IDLE_S")) : 600.0;
    bool inited = false;
    printf("ppm_server: device %d, socket %s, cache %.0f GB, idle limit %.0f s\n", dev, path.c_str(), cache.budget / 1e9, idle_s);
    fflush(stdout);
    for (;;) {
        pollfd pf; pf.fd = ls; pf.events = POLLIN; pf.revents = 0;
        const int pr = poll(&pf, 1, (int)std::min(idle_s * 1000.0, 2.0e9));
        if (pr < 0 && errno == EINTR) continue;
        if (pr <= 0) { printf("ppm_server: idle for %.0f s, leaving\n", idle_s); break; }
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        { timeval tv; tv.tv_sec = 10; tv.tv_usec = 0; setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv); }      // a client that connects and stays silent does not block the others
        char magic[4]; uint32_t prog = 0; std::string cwd, input, settings;
        if (!read_all(fd, magic, 4) || memcmp(magic, "PPMS", 4) != 0 || !read_all(fd, &prog, 4) || !recv_blob(fd, cwd) || !recv_blob(fd, input) || !recv_blob(fd, settings)) { close(fd); continue; }
        std::string text; Out out; out.sink = &text;
        int32_t status = 0;
        bool leave = false;
        if (prog == kProgHello) out.print("%s", ppm_build_id());
        else if (prog == kProgStop) { out.print("ppm_server: stopping (served %ld calls, %ld resident hits, %ld uploads)\n", cache.served, cache.hits, cache.misses); leave = true; }
        else if (prog == kProgStats) {
            out.print("ppm_server: device %d, served %ld calls, %ld resident hits, %ld uploads, %.2f of %.0f GB cached\n", dev, cache.served, cache.hits, cache.misses, cache.used / 1e9, cache.budget / 1e9);
            for (const auto &e : cache.stacks) out.print("  stack inode %llu: particles %ld..%ld, box %d, %.2f GB\n", e.id.ino, e.first, e.first + e.count - 1, e.box, e.bytes / 1e9);
            for (const auto &e : cache.refs) out.print("  reference inode %llu: box %d, padding %d\n", e.id.ino, e.box, e.pad);
        } else if (prog == kProgRecon || prog == kProgRefine) {
            const auto t0 = Clock::now();
            if (chdir(cwd.c_str()) != 0) { status = 1; out.print("ERROR: ppm_server: cannot enter %s\n", cwd.c_str()); }
            else {
                const int lockfd = gpu_lock(dev);               // one-shot processes of the node wait their turn on the same lock
                RequestSettings as_the_client(settings);        // the caller's umask and PPM_* settings, for this request only
                try {
                    if (!inited) { if (ppm_init(dev) != 0) throw Fail{ ppm_last_error() }; inited = true; }
                    status = prog == kProgRecon ? serve_reconstruct3d(cache, input, out) : serve_refine3d(cache, input, out);
                } catch (const Fail &f) {
                    status = 1;
                    out.print("%s\n", f.msg.find("ERROR") != std::string::npos ? f.msg.c_str() : ("ERROR: " + f.msg).c_str());
                } catch (const std::exception &e) {             // out of host memory, a bad table ...: the call fails, the server stays
                    status = 1;
                    out.print("ERROR: ppm_server: %s\n", e.what());
                }
                gpu_unlock(lockfd);
            }
            cache.served++;
            printf("ppm_server: %s in %s -> status %d, %.2f s\n", prog == kProgRecon ? "reconstruct3d" : "refine3d", cwd.c_str(), status, since(t0));
            fflush(stdout);
        } else { status = 1; out.print("ERROR: ppm_server: unknown request %u\n", prog); }
        if (status == kHandOver) { text.clear(); cache.release_device(); }      // the client runs the call in a child process of its own, on this GPU: it gets the memory
        (void)(write_all(fd, &status, 4) && send_blob(fd, text));
        close(fd);
        if (leave) break;
    }
    close(ls);
    unlink(path.c_str());
    cache.clear();
    fflush(stdout);
    _exit(0);          // the library's reader pool is parked on purpose
}
