// What clock does a short kernel actually run at?  One wave per CU runs a dependent chain of N v_fma (4 cycles each
// when nothing else is on the SIMD); wall time comes from the constant 100 MHz counter (wall_clock64) and the shader
// counter (clock64).  Variants: launched after an idle pause, back-to-back, and right after a 20 ms busy kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void chain(float *out, unsigned long long *t, int n) {
    float a = threadIdx.x * 1e-9f, b = 1.0000001f;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) a = __builtin_fmaf(a, b, 1e-7f);
    }
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) {
        t[blockIdx.x * 2] = w1 - w0;
        t[blockIdx.x * 2 + 1] = c1 - c0;
    }
    if (a == 123.f) *out = a;
}
__global__ void busy(float *out, int iters) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = __builtin_fmaf(a, 1.0000001f, 1e-7f);
    if (a == 123.f) *out = a;
}
int main() {
    float *d; unsigned long long *t, h[512];
    hipMalloc(&d, 4); hipMalloc(&t, 512 * 8);
    const int n = 16384;
    auto run = [&](const char *what) {
        hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, d, t, n);
        hipDeviceSynchronize();
        hipMemcpy(h, t, 512 * 8, hipMemcpyDeviceToHost);
        double w = 0, c = 0;
        for (int i = 0; i < 256; i++) { w += h[2 * i]; c += h[2 * i + 1]; }
        w /= 256; c /= 256;
        printf("%-34s wall %.2f us  clock64 %.0f  -> clock64 rate %.0f MHz ; %.2f ns per dependent FMA = %.0f MHz if 4 clk each (8 clk: %.0f)\n", what, w * 0.01, c,
               c / (w * 0.01), w * 10.0 / n, 4.0 / (w * 10.0 / n) * 1e3, 8.0 / (w * 10.0 / n) * 1e3);
    };
    run("first launch");
    for (int i = 0; i < 3; i++) run("back-to-back");
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    run("after 200 ms idle");
    run("next");
    for (int ms : {1, 5, 20, 100}) {
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, 0, d, ms * 60000);
        char b[64]; snprintf(b, 64, "after busy kernel (~%d units)", ms);
        run(b);
    }
    for (int gap_us : {10, 30, 100, 1000}) {
        for (int i = 0; i < 20; i++) {
            hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, d, t, 256);
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(gap_us)) {}
        }
        char b[64]; snprintf(b, 64, "after 20 short kernels, %d us gaps", gap_us);
        run(b);
    }
    return 0;
}
