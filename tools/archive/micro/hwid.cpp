// Which workgroups of a 512 x 512-thread launch share a CU (HW_REG_HW_ID / HW_REG_XCC_ID): on MI355X block b and block b + 256 do,
// on all 256 CUs; block b runs on XCD b % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned *out, float *sink) {
    __shared__ float s[5000];  // ~20 KB like k_row_stats
    s[threadIdx.x] = 1.f;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the block alive a little so that all 512 are co-resident
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 500) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    if (s[threadIdx.x] == 2.f) *sink = 0;
}
int main() {
    unsigned *d; float *sink; hipMalloc(&d, 512 * 8); hipMalloc(&sink, 4);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, d, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(1024); hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> cu;
    for (int b = 0; b < 512; b++) {
        unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("distinct CUs used: %zu\n", cu.size());
    int shown = 0; std::map<int,int> delta;
    for (auto &kv : cu) {
        if (shown++ < 12) { printf("xcc %llu se %llu cu %llu:", kv.first >> 16, (kv.first >> 8) & 0xff, kv.first & 0xff); for (int b : kv.second) printf(" %d", b); printf("\n"); }
        if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
    }
    for (auto &d : delta) printf("pair distance %d: %d CUs\n", d.first, d.second);
    printf("first 16 blocks xcc:"); for (int b = 0; b < 16; b++) printf(" %u", h[b*2+1] & 0xf); printf("\n");
    return 0;
}
