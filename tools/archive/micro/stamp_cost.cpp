// What a GPU-timed section entry costs the USER's stream: N iterations of a fixed work kernel (spins ~30 us on the
// constant-rate clock) bracketed by (0) nothing, (1) the two one-thread stamp kernels of k_stamp_begin / k_stamp_end,
// (2) two hipEventRecord (default flags), (3) two hipEventRecord on events created with hipEventDisableSystemFence,
// (4) ONE stamp kernel, (5) two hipStreamWriteValue64 (a command-processor packet without a dispatch), (6) the work kernel
// itself launched through hipExtLaunchKernel with start/stop events (no extra packet at all).
// Prints GPU time per iteration (outer event pair / N) and host enqueue time per iteration.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_work(unsigned long long ticks, unsigned long long *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (ticks == 1) *sink = t0;
}
__global__ void k_stamp_begin(unsigned long long *slot) { *slot = wall_clock64(); }
__global__ void k_stamp_end(const unsigned long long *slot, float us_per_tick, float *dst) { *dst = (float)(wall_clock64() - *slot) * us_per_tick; }
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 400;
    const unsigned long long ticks = argc > 2 ? atoll(argv[2]) : 3000;
    hipStream_t st; hipStreamCreate(&st);
    unsigned long long *slot, *sink; float *dst; hipMalloc(&slot, 8); hipMalloc(&sink, 8); hipMalloc(&dst, 4 * 4096);
    unsigned long long *wv; hipMalloc(&wv, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<hipEvent_t> evd(2 * N), evf(2 * N);
    for (auto &e : evd) hipEventCreate(&e);
    for (auto &e : evf) hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    const char *names[] = {"none", "two stamp kernels", "two hipEventRecord", "two hipEventRecord (DisableSystemFence)", "one stamp kernel",
                           "two hipStreamWriteValue64", "work via hipExtLaunchKernel(start, stop)"};
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 7; v++) {
            hipStreamSynchronize(st);
            hipEventRecord(e0, st);
            auto h0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (v == 1) hipLaunchKernelGGL(k_stamp_begin, dim3(1), dim3(1), 0, st, slot);
                if (v == 2) hipEventRecord(evd[2 * i], st);
                if (v == 3) hipEventRecord(evf[2 * i], st);
                if (v == 5) hipStreamWriteValue64(st, wv, 1, 0);
                if (v == 6) hipExtLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, evf[2 * i], evf[2 * i + 1], 0, ticks, sink);
                else hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, ticks, sink);
                if (v == 1 || v == 4) hipLaunchKernelGGL(k_stamp_end, dim3(1), dim3(1), 0, st, slot, 0.01f, dst + (i & 4095));
                if (v == 2) hipEventRecord(evd[2 * i + 1], st);
                if (v == 3) hipEventRecord(evf[2 * i + 1], st);
                if (v == 5) hipStreamWriteValue64(st, wv + 1, 2, 0);
            }
            auto h1 = std::chrono::steady_clock::now();
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            float inner = 0;
            if (v == 2) hipEventElapsedTime(&inner, evd[2 * (N / 2)], evd[2 * (N / 2) + 1]);
            if (v == 3 || v == 6) hipEventElapsedTime(&inner, evf[2 * (N / 2)], evf[2 * (N / 2) + 1]);
            if (rep == 1)
                printf("%-44s gpu %7.2f us/iter   host enqueue %6.2f us/iter   %s%.2f us\n", names[v], ms * 1e3 / N,
                       std::chrono::duration<double, std::micro>(h1 - h0).count() / N, inner > 0 ? "inner pair reads " : "", inner * 1e3);
        }
    return 0;
}
