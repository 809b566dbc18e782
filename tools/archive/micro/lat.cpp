// latencies of the primitives the selection chain is built from (gfx950, one 512-thread workgroup)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 64
__global__ __launch_bounds__(512) void k(unsigned long long *out, uint32_t *sink, int mode, int waves_active) {
    __shared__ uint32_t h[2048];
    __shared__ uint32_t cur;
    for (int i = threadIdx.x; i < 2048; i += 512) h[i] = (i * 7 + 1) & 2047;
    if (threadIdx.x == 0) cur = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    uint32_t v = threadIdx.x;
    unsigned long long t0 = clock64();
    if (wave < waves_active) {
        if (mode == 0) {  // dependent ds_read chain
#pragma unroll
            for (int i = 0; i < N; i++) v = h[v & 2047];
        } else if (mode == 1) {  // dependent ds_add_rtn chain (same address per wave, one lane)
#pragma unroll
            for (int i = 0; i < N; i++) { if ((threadIdx.x & 63) == 0) v = atomicAdd(&cur, v & 1); }
        } else if (mode == 2) {  // dependent VALU chain (v_add)
#pragma unroll
            for (int i = 0; i < N; i++) v = v * 3 + 1;
        } else if (mode == 3) {  // readlane + VALU chain
#pragma unroll
            for (int i = 0; i < N; i++) v = v + (uint32_t)__builtin_amdgcn_readlane((int)v, 5);
        } else if (mode == 4) {  // DPP chain
#pragma unroll
            for (int i = 0; i < N; i++) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
        } else if (mode == 5) {  // ballot + scalar chain
#pragma unroll
            for (int i = 0; i < N; i++) v += (uint32_t)__popcll(__ballot(v & 1));
        } else if (mode == 6) {  // f64 fma chain
            double d = (double)v;
#pragma unroll
            for (int i = 0; i < N; i++) d = d * 1.0000001 + 0.5;
            v = (uint32_t)d;
        } else if (mode == 7) {  // f64 divide chain
            double d = (double)v + 3.0;
#pragma unroll
            for (int i = 0; i < 8; i++) d = 1.0e6 / d;
            v = (uint32_t)d;
        } else if (mode == 9) {  // LDS write then read back (same thread)
#pragma unroll
            for (int i = 0; i < N; i++) { h[threadIdx.x] = v; v = h[threadIdx.x] + 1; }
        }
    }
    if (mode == 8) {  // barriers back to back
#pragma unroll
        for (int i = 0; i < N; i++) __syncthreads();
    }
    if (mode == 10) {  // LDS write, barrier, read neighbour wave's value
#pragma unroll
        for (int i = 0; i < N; i++) { h[threadIdx.x] = v; __syncthreads(); v = h[(threadIdx.x + 64) & 511] + 1; __syncthreads(); }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (v == 0xdeadbeef) sink[0] = v;
}
int main() {
    unsigned long long *d_out; uint32_t *d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 8);
    const char *names[] = {"ds_read dependent", "ds_add_rtn dependent (1 lane)", "v_mad dependent", "readlane+add", "dpp row_shr+add", "ballot+popc+add",
                           "f64 fma dependent", "f64 divide (x8)", "s_barrier x64 (8 waves)", "ds_write+ds_read same thread", "write,barrier,read,barrier"};
    for (int mode = 0; mode <= 10; mode++)
        for (int wa : {1, 8}) {
            unsigned long long o;
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d_out, d_sink, mode, wa);
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d_out, d_sink, mode, wa);
            hipDeviceSynchronize();
            hipMemcpy(&o, d_out, 8, hipMemcpyDeviceToHost);
            const int cnt = mode == 7 ? 8 : N;
            printf("%-34s waves=%d : %6llu clk total -> %.1f clk per step\n", names[mode], wa, o, (double)o / cnt);
        }
    return 0;
}
