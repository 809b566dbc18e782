// Test-only LD_PRELOAD shim (tools/reftest_soak.sh): when std::terminate runs -- "terminate called without an active
// exception" is what a joinable std::thread's destructor, or a second exception during unwinding, ends in -- print the
// native stack of the calling thread and the names of the process' threads before aborting, so that a child of a
// multi-process test that dies at interpreter exit says WHO called terminate.  Not part of the product.
#include <dirent.h>
#include <execinfo.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>

namespace {
void on_terminate() {
    static const char head[] = "\n[terminate_trace] std::terminate called; native stack of the calling thread:\n";
    (void)!write(2, head, sizeof(head) - 1);
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    static const char mid[] = "[terminate_trace] threads of this process (tid comm):\n";
    (void)!write(2, mid, sizeof(mid) - 1);
    if (DIR *d = opendir("/proc/self/task")) {
        while (dirent *e = readdir(d)) {
            if (e->d_name[0] < '0' || e->d_name[0] > '9') continue;
            char path[300], comm[64] = {0};
            snprintf(path, sizeof(path), "/proc/self/task/%s/comm", e->d_name);
            if (FILE *f = fopen(path, "r")) {
                if (fgets(comm, sizeof(comm), f)) comm[strcspn(comm, "\n")] = 0;
                fclose(f);
            }
            dprintf(2, "  %s %s\n", e->d_name, comm);
        }
        closedir(d);
    }
    abort();
}
struct Install {
    Install() { std::set_terminate(on_terminate); }
} g_install;
}  // namespace
