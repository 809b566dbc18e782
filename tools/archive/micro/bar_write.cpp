// Can the host store straight into device memory (large BAR), and does the GPU see it -- also for lines it has cached?
// (1) host stores through the device pointer of fine-grained (hipExtMallocWithFlags) and plain hipMalloc memory;
// (2) the ring-reuse pattern: a kernel READS the buffer (lines now in L2), the host overwrites it through the BAR, the
//     kernel reads again -- 200 rounds, every value checked; (3) what a streaming read of 20 MB costs from either type.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void k_sum(const float *p, int n, float *out) {
    __shared__ float s[256];
    float a = 0;
    for (int i = threadIdx.x; i < n; i += 256) a += p[i];
    s[threadIdx.x] = a; __syncthreads();
    if (threadIdx.x == 0) { float t = 0; for (int i = 0; i < 256; i++) t += s[i]; *out = t; }
}
__global__ void k_stream(const float4 *p, size_t n4, float *sink) {
    float a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 123.f) *sink = a;
}
int main() {
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    const int n = 10000;
    const size_t big = 512ull * 10000 * 4;
    float *fine = nullptr, *plain = nullptr, *out = nullptr, h = 0;
    hipExtMallocWithFlags((void **)&fine, big, hipDeviceMallocFinegrained);
    hipMalloc((void **)&plain, big); hipMalloc((void **)&out, 4);
    hipMemset(fine, 0, big); hipMemset(plain, 0, big); hipDeviceSynchronize();
    const char *names[] = {"fine-grained device memory", "plain hipMalloc memory"};
    float *ptrs[] = {fine, plain};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; v++) {
        printf("%s\n", names[v]);
        if (sigsetjmp(jb, 1) != 0) { printf("  host store faulted: not CPU-accessible\n"); continue; }
        volatile float *q = ptrs[v];
        int bad = 0; double ns = 0;
        for (int round = 1; round <= 200; round++) {
            hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, ptrs[v], n, out);   // the GPU reads the row: its lines are cached
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) q[i] = (float)round;                      // the host overwrites it through the BAR
            ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
            hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, ptrs[v], n, out);
            hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
            if (h != (float)round * n) bad++;
        }
        printf("  read / host-overwrite / read, 200 rounds of %d floats: %d stale rounds; %.1f ns per host store\n", n, bad, ns / 200 / n);
        float best = 1e9f;
        for (int rep = 0; rep < 20; rep++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const float4 *)ptrs[v], big / 16, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        {   // (4) bulk: memcpy of 20.48 MB of pageable host memory through the BAR against hipMemcpy of the same buffer
            std::vector<float> src(big / 4, 1.5f);
            double best_bar = 1e9, best_cpy = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                auto t0 = std::chrono::steady_clock::now();
                memcpy(ptrs[v], src.data(), big);
                __builtin_ia32_sfence();
                hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, 0, ptrs[v] + big / 4 - n, n, out);
                hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (h != 1.5f * n) bad++;
                if (us < best_bar) best_bar = us;
                t0 = std::chrono::steady_clock::now();
                hipMemcpy(ptrs[v], src.data(), big, hipMemcpyHostToDevice);
                us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (us < best_cpy) best_cpy = us;
            }
            printf("  20.48 MB pageable -> device: CPU memcpy through the BAR %.0f us (%.1f GB/s), hipMemcpy %.0f us (%.1f GB/s); bad %d\n",
                   best_bar, big / best_bar / 1e3, best_cpy, big / best_cpy / 1e3, bad);
        }
        printf("  streaming read of %.1f MB: %.2f us best of 20 (%.0f GB/s)\n", big / 1e6, best * 1e3, big / (best * 1e-3) / 1e9);
    }
    return 0;
}
