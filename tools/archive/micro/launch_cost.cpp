// Host cost of a kernel launch by argument count (device-memory kernargs are this platform's default): does a kernel
// WITHOUT arguments skip the kernarg write, i.e. would an argument-free timestamp kernel make a section entry cheaper?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__device__ unsigned long long g_slot[4096];
__device__ unsigned g_head;
__global__ void k0() { g_slot[atomicAdd(&g_head, 1u) & 4095u] = wall_clock64(); }
__global__ void k1(unsigned long long *p) { *p = wall_clock64(); }
__global__ void k5(const unsigned long long *a, float f, float *b, float *c, float g) {
    if (b) *b = (float)(wall_clock64() - *a) * f + g;
    if (c) *c = g;
}
template <class F>
static double per_launch_us(F launch, int n) {
    for (int i = 0; i < 200; i++) launch();
    hipDeviceSynchronize();
    double total = 0;
    for (int blk = 0; blk < n / 200; blk++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; i++) launch();
        total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        hipDeviceSynchronize();  // keep the queue short: the host's cost per launch, not back-pressure
    }
    return total / (n / 200 * 200);
}
int main() {
    unsigned long long *d; float *f;
    hipMalloc(&d, 64); hipMalloc(&f, 64);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int round = 0; round < 3; round++) {
        double a = per_launch_us([&] { hipLaunchKernelGGL(k0, dim3(1), dim3(1), 0, s); }, 4000);
        double b = per_launch_us([&] { hipLaunchKernelGGL(k1, dim3(1), dim3(1), 0, s, d); }, 4000);
        double c = per_launch_us([&] { hipLaunchKernelGGL(k5, dim3(1), dim3(1), 0, s, d, 0.01f, f, f + 1, 1.f); }, 4000);
        printf("host us per launch: no arguments %.2f | one pointer %.2f | five arguments %.2f\n", a, b, c);
    }
    return 0;
}
