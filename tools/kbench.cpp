// tools/kbench.cpp -- standalone timing of k_row_stats variants on the folded N=1 shape
// (512 rows x 10000 f32).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude [-DNVRX_ABLATE=k]
//   tools/kbench.cpp -o /tmp/kbench ; run: /tmp/kbench [rows] [n] [threads]
// Includes the library source directly so ablation builds need no second copy of the kernel.
#include "../nvidia-resiliency-ext_amd/csrc/nvrx_straggler.hip"

#include <random>

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 512;
    const int n = argc > 2 ? atoi(argv[2]) : 10000;
    const int stride = (n + 3) & ~3;
    std::vector<float> h((size_t)rows * stride);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(10.f, 0.3f);
    for (auto &v : h) v = nd(rng);
    float *d_s, *d_stats;
    uint32_t *d_c;
    hipMalloc(&d_s, h.size() * 4);
    hipMalloc(&d_stats, (size_t)rows * 8 * 4);
    hipMalloc(&d_c, rows * 4);
    hipMemcpy(d_s, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> c(rows, (uint32_t)n);
    hipMemcpy(d_c, c.data(), rows * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    Epilogue ep{};
    for (int threads : {256, 512, 1024}) {
        if (argc > 3 && atoi(argv[3]) != threads) continue;
        const StatsVariant *best = nullptr;
        for (const StatsVariant &v : kVariants)
            if (v.threads == threads && v.threads * v.vpt * 4 >= stride && (!best || v.vpt < best->vpt)) best = &v;
        if (!best) continue;
        double tot = 0;
        float mn = 1e9f;
        const int reps = 50;
        for (int i = 0; i < reps + 5; i++) {
            hipExtLaunchKernelGGL(best->fn, dim3(rows), dim3(best->threads), 0, nullptr, a, b, 0, (const float *)d_s,
                                  (const uint32_t *)d_c, (const uint8_t *)nullptr, stride, d_stats, ep);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (i >= 5) {
                tot += ms;
                mn = std::min(mn, ms);
            }
        }
        float st[8];
        hipMemcpy(st, d_stats, sizeof(st), hipMemcpyDeviceToHost);
        printf("ablate=%d threads=%4d vpt=%2d rows=%d n=%d : avg %.2f us  min %.2f us  -> %.0f GB/s   (row0 med %.5f)\n",
               NVRX_ABLATE, best->threads, best->vpt, rows, n, tot / reps * 1e3, mn * 1e3,
               (double)rows * n * 4 / (tot / reps * 1e-3) / 1e9, st[2]);
    }
    return 0;
}
