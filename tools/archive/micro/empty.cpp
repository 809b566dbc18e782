// Event-timed duration (hipExtLaunchKernel start/stop events = the dispatch's own timestamps, what rocprofv3 reports)
// of kernels that do nothing, at the launch shapes of k_row_stats: the fixed cost every launch pays.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_touch(const float4 *__restrict__ src, float *sink, int vpt, int n4) {  // pure load: vpt float4 per lane, one add each
    float acc = 0.f;
    for (int i = 0; i < vpt; i++) {
        const int v = blockIdx.x * (blockDim.x * vpt) + i * blockDim.x + threadIdx.x;
        if (v < n4) { const float4 x = src[v]; acc += x.x + x.y + x.z + x.w; }
    }
    if (acc == 123.456f) *sink = acc;
}
int main() {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float *d; hipMalloc(&d, 512ull * 10000 * 4 + 64); hipMemset(d, 0, 512ull * 10000 * 4);
    auto timeit = [&](const char *what, auto launch) {
        std::vector<float> t;
        for (int i = 0; i < 60; i++) { launch(); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (i >= 10) t.push_back(ms * 1e3f); }
        std::sort(t.begin(), t.end());
        printf("%-44s median %.2f us  min %.2f us\n", what, t[t.size() / 2], t[0]);
    };
    for (int threads : {256, 512, 1024})
        for (int blocks : {1, 64, 512}) {
            char w[64]; snprintf(w, 64, "empty kernel %4d blocks x %4d threads", blocks, threads);
            timeit(w, [&] { hipExtLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, nullptr, a, b, 0, (int *)nullptr); });
        }
    timeit("pure load 512 rows x 10000 f32 (512x512x5)", [&] { hipExtLaunchKernelGGL(k_touch, dim3(512), dim3(512), 0, nullptr, a, b, 0, (const float4 *)d, d, 5, 512 * 2500); });
    timeit("pure load 64 rows x 10000 f32 (64x512x5)", [&] { hipExtLaunchKernelGGL(k_touch, dim3(64), dim3(512), 0, nullptr, a, b, 0, (const float4 *)d, d, 5, 64 * 2500); });
    return 0;
}
