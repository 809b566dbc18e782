// LD_PRELOAD shim: who issues the large read()s of the first HIP call when a rocprofiler-sdk tool is attached?
// Prints a dladdr-resolved backtrace for the first few reads of >= 1 MiB.   gcc -O1 -g -shared -fPIC -o readtrace.so readtrace.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <stdatomic.h>
static ssize_t (*real_read)(int, void *, size_t);
static ssize_t (*real_pread)(int, void *, size_t, off_t);
static atomic_int shown;
static atomic_ulong total;
static void show(const char *what, int fd, size_t n) {
    int k = atomic_fetch_add(&shown, 1);
    if (k >= 4) return;
    char path[256], link[64];
    snprintf(link, sizeof link, "/proc/self/fd/%d", fd);
    ssize_t l = readlink(link, path, sizeof path - 1);
    path[l > 0 ? l : 0] = 0;
    fprintf(stderr, "[readtrace] %s(fd=%d %s, %zu bytes)\n", what, fd, path, n);
    void *bt[48];
    int d = backtrace(bt, 48);
    for (int i = 1; i < d; i++) {
        Dl_info info;
        if (dladdr(bt[i], &info) && info.dli_fname) {
            const char *base = strrchr(info.dli_fname, '/');
            fprintf(stderr, "[readtrace]   #%d %s!%s+0x%lx (lib+0x%lx)\n", i, base ? base + 1 : info.dli_fname, info.dli_sname ? info.dli_sname : "?",
                    info.dli_saddr ? (unsigned long)((char *)bt[i] - (char *)info.dli_saddr) : 0ul, (unsigned long)((char *)bt[i] - (char *)info.dli_fbase));
        } else
            fprintf(stderr, "[readtrace]   #%d %p\n", i, bt[i]);
    }
}
ssize_t read(int fd, void *buf, size_t n) {
    if (!real_read) real_read = dlsym(RTLD_NEXT, "read");
    if (n >= (1u << 20)) { atomic_fetch_add(&total, n); show("read", fd, n); }
    return real_read(fd, buf, n);
}
ssize_t pread64(int fd, void *buf, size_t n, off_t off) {
    if (!real_pread) real_pread = dlsym(RTLD_NEXT, "pread64");
    if (n >= (1u << 20)) { atomic_fetch_add(&total, n); show("pread64", fd, n); }
    return real_pread(fd, buf, n, off);
}
__attribute__((destructor)) static void fini(void) { fprintf(stderr, "[readtrace] large reads total %.2f GB\n", total / 1e9); }
