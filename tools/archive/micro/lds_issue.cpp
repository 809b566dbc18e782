// how LDS instruction cost scales with waves / ops per wave / active lanes (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
template <int OPS>
__global__ __launch_bounds__(1024) void k(const uint32_t *__restrict__ idx, unsigned long long *out, uint32_t *sink, int waves_active, int lanes_active, int mode) {
    __shared__ uint32_t h[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) h[i] = 0;
    uint32_t a[OPS];
#pragma unroll
    for (int j = 0; j < OPS; j++) a[j] = idx[(size_t)j * 1024 + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t0 = clock64();
    if ((int)(threadIdx.x >> 6) < waves_active && (int)(threadIdx.x & 63) < lanes_active) {
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < OPS; j++) atomicAdd(&h[a[j]], 1u);
        } else {
#pragma unroll
            for (int j = 0; j < OPS; j++) h[a[j]] = j;
        }
    }
    __syncthreads();
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (h[threadIdx.x] == 0xffffffffu) sink[0] = 1;
}
int main() {
    std::mt19937 rng(3);
    uint32_t *d_idx, *d_sink; unsigned long long *d_out;
    hipMalloc(&d_idx, 40 * 1024 * 4); hipMalloc(&d_out, 256 * 8); hipMalloc(&d_sink, 8);
    std::vector<uint32_t> h(40 * 1024);
    for (auto &v : h) v = rng() % 2048;
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++)
        for (int threads : {512, 1024})
            for (int wa : {1, 2, 4, 8, 16}) {
                if (wa * 64 > threads) continue;
                for (int la : {64, 4}) {
                    unsigned long long o20, o40;
                    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<20>, dim3(1), dim3(threads), 0, 0, d_idx, d_out, d_sink, wa, la, mode);
                    hipDeviceSynchronize(); hipMemcpy(&o20, d_out, 8, hipMemcpyDeviceToHost);
                    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<40>, dim3(1), dim3(threads), 0, 0, d_idx, d_out, d_sink, wa, la, mode);
                    hipDeviceSynchronize(); hipMemcpy(&o40, d_out, 8, hipMemcpyDeviceToHost);
                    printf("%s threads=%4d waves_active=%2d lanes=%2d : 20 ops %5llu clk (%.1f/op/wave, %.1f per wave-instr)   40 ops %5llu clk (%.1f/op/wave, %.1f per wave-instr)\n",
                           mode ? "write " : "atomic", threads, wa, la, o20, o20 / 20.0, o20 / 20.0 / wa, o40, o40 / 40.0, o40 / 40.0 / wa);
                }
            }
    return 0;
}
