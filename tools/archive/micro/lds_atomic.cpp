// LDS atomic throughput on gfx950 under different address patterns (one 512-thread workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
__global__ __launch_bounds__(512) void k(const uint32_t *__restrict__ idx, unsigned long long *out, uint32_t *sink, int mode) {
    __shared__ uint32_t h[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) h[i] = 0;
    uint32_t a[20];
#pragma unroll
    for (int j = 0; j < 20; j++) a[j] = idx[(size_t)j * 512 + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t0 = clock64();
    if (mode == 0) {
#pragma unroll
        for (int j = 0; j < 20; j++) atomicAdd(&h[a[j]], 1u);
    } else if (mode == 1) {
#pragma unroll
        for (int j = 0; j < 20; j++) h[a[j]] = j;
    } else {
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < 20; j++) s += h[a[j]];
        if (s == 0xdeadbeef) sink[1] = s;
    }
    __syncthreads();
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (h[threadIdx.x] == 0xffffffffu) sink[0] = 1;
}
int main() {
    std::mt19937 rng(3);
    std::normal_distribution<float> nd(1024.f, 256.f);
    uint32_t *d_idx, *d_sink; unsigned long long *d_out;
    hipMalloc(&d_idx, 20 * 512 * 4); hipMalloc(&d_out, 256 * 8); hipMalloc(&d_sink, 8);
    const char *names[] = {"uniform random 2048", "gaussian(1024,256)", "conflict-free (lane->own bank, 64 banks)", "conflict-free mod 32", "all same address", "same per wave", "random over 64 bins", "sorted-ish (lane-major monotone)"};
    for (int pat = 0; pat < 8; pat++) {
        std::vector<uint32_t> h(20 * 512);
        for (int j = 0; j < 20; j++) for (int t = 0; t < 512; t++) {
            uint32_t v;
            switch (pat) {
                case 0: v = rng() % 2048; break;
                case 1: { int x = (int)nd(rng); v = (uint32_t)std::min(2047, std::max(0, x)); } break;
                case 2: v = (uint32_t)((t % 64) + 64 * ((j * 7 + t / 64) % 32)); break;
                case 3: v = (uint32_t)((t % 32) + 32 * ((j * 7 + t / 32) % 64)); break;
                case 4: v = 5; break;
                case 5: v = (uint32_t)(t / 64 * 33 + j); break;
                case 6: v = rng() % 64; break;
                default: v = (uint32_t)(((t * 20 + j) * 2048ull) / (512 * 20)); break;
            }
            h[(size_t)j * 512 + t] = v;
        }
        hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; mode++) {
            for (int blocks : {1, 256, 512}) {
                unsigned long long o[512];
                hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d_idx, d_out, d_sink, mode);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d_idx, d_out, d_sink, mode);
                hipDeviceSynchronize();
                hipMemcpy(o, d_out, std::min(blocks, 256) * 8, hipMemcpyDeviceToHost);
                double s = 0; for (int i = 0; i < std::min(blocks, 256); i++) s += o[i];
                printf("%-45s mode=%s blocks=%3d : %.0f clk for 20 ops/lane x 8 waves -> %.1f clk per wave-instr\n", names[pat],
                       mode == 0 ? "atomic" : mode == 1 ? "write " : "read  ", blocks, s / std::min(blocks, 256), s / std::min(blocks, 256) / 160.0);
            }
        }
    }
    return 0;
}
