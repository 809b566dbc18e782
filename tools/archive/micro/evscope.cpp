// What the event-timed duration of a launch contains: the same kernels timed with hipExtLaunchKernel start/stop events
// created with different release-scope flags, next to the span the kernel measures on itself (first workgroup's start ->
// last workgroup's end on the constant-rate wall clock, 100 MHz).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k_empty(unsigned long long *span) {
    if (threadIdx.x == 0) {
        const unsigned long long t = wall_clock64();
        atomicMin(&span[0], t);
        atomicMax(&span[1], t);
    }
}
__global__ void k_touch(const float4 *__restrict__ src, float *sink, int vpt, int n4, unsigned long long *span) {
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = wall_clock64();
    float acc = 0.f;
    for (int i = 0; i < vpt; i++) {
        const int v = blockIdx.x * (blockDim.x * vpt) + i * blockDim.x + threadIdx.x;
        if (v < n4) { const float4 x = src[v]; acc += x.x + x.y + x.z + x.w; }
    }
    if (acc == 123.456f) *sink = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&span[0], t0);
        atomicMax(&span[1], wall_clock64());
    }
}
__global__ void k_write(float4 *__restrict__ dst, int vpt, int n4, unsigned long long *span) {  // leaves n4*16 bytes dirty
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = wall_clock64();
    for (int i = 0; i < vpt; i++) {
        const int v = blockIdx.x * (blockDim.x * vpt) + i * blockDim.x + threadIdx.x;
        if (v < n4) dst[v] = make_float4(1.f, 2.f, 3.f, (float)v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&span[0], t0);
        atomicMax(&span[1], wall_clock64());
    }
}
int main() {
    float *d; hipMalloc(&d, 512ull * 10000 * 4 + 64); hipMemset(d, 0, 512ull * 10000 * 4);
    float *w; hipMalloc(&w, 512ull * 10000 * 4 + 64);
    unsigned long long *span; hipMalloc(&span, 16);
    const unsigned flagv[] = {0u, hipEventDisableSystemFence, hipEventReleaseToDevice, hipEventReleaseToSystem};
    const char *flagn[] = {"default", "DisableSystemFence", "ReleaseToDevice", "ReleaseToSystem"};
    for (int f = 0; f < 4; f++) {
        hipEvent_t a, b;
        if (hipEventCreateWithFlags(&a, flagv[f]) != hipSuccess || hipEventCreateWithFlags(&b, flagv[f]) != hipSuccess) {
            printf("%s: event creation failed\n", flagn[f]);
            continue;
        }
        auto timeit = [&](const char *what, auto launch) {
            std::vector<float> t, sp;
            for (int i = 0; i < 60; i++) {
                const unsigned long long init[2] = {~0ull, 0ull};
                hipMemcpy(span, init, 16, hipMemcpyHostToDevice);
                hipDeviceSynchronize();
                launch(a, b);
                hipEventSynchronize(b);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, a, b);
                unsigned long long got[2];
                hipMemcpy(got, span, 16, hipMemcpyDeviceToHost);
                if (i >= 10) { t.push_back(ms * 1e3f); sp.push_back((float)(got[1] - got[0]) * 0.01f); }
            }
            std::sort(t.begin(), t.end()); std::sort(sp.begin(), sp.end());
            printf("%-20s %-44s events: median %.2f us min %.2f | in-kernel span: median %.2f us\n", flagn[f], what, t[t.size() / 2], t[0], sp[sp.size() / 2]);
        };
        timeit("empty 512 x 512", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, nullptr, a, b, 0, span); });
        timeit("empty 64 x 512", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(64), dim3(512), 0, nullptr, a, b, 0, span); });
        timeit("load 20.48 MB (512x512x5)", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_touch, dim3(512), dim3(512), 0, nullptr, a, b, 0, (const float4 *)d, d, 5, 512 * 2500, span); });
        timeit("load 2.56 MB (64x512x5)", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_touch, dim3(64), dim3(512), 0, nullptr, a, b, 0, (const float4 *)d, d, 5, 64 * 2500, span); });
        timeit("store 20.48 MB (512x512x5)", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_write, dim3(512), dim3(512), 0, nullptr, a, b, 0, (float4 *)w, 5, 512 * 2500, span); });
        // two back-to-back launches, events around the pair: what a dependent boundary costs under each scope
        timeit("2 x empty 512 x 512 (one pair)", [&](hipEvent_t a, hipEvent_t b) {
            hipExtLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, nullptr, a, nullptr, 0, span);
            hipExtLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, nullptr, nullptr, b, 0, span);
        });
        hipEventDestroy(a); hipEventDestroy(b);
    }
    // plain hipEventRecord pair around a normal launch, for comparison
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        std::vector<float> t;
        for (int i = 0; i < 60; i++) {
            hipDeviceSynchronize();
            hipEventRecord(a, nullptr);
            hipLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, nullptr, span);
            hipEventRecord(b, nullptr);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (i >= 10) t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("hipEventRecord pair around hipLaunchKernelGGL(empty 512 x 512): median %.2f us min %.2f\n", t[t.size() / 2], t[0]);
    }
    return 0;
}
