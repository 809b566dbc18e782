// tools/kbench.cpp -- standalone timing of k_row_stats variants on the folded N=1 shape
// (512 rows x 10000 f32), with a CPU check of every row's median/min/max/mean/std.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude [-DNVRX_ABLATE=k] [-DNVRX_PHASE_CLOCKS]
//   tools/kbench.cpp -o tools/bin/kbench ; run: kbench [rows] [n] [threads] [dist]
// dist: 0 = N(10,0.3) (bench shape), 1 = lognormal heavy tail, 2 = few distinct values, 3 = drifting ramp + noise
// Includes the library source directly so ablation builds need no second copy of the kernel.
#ifndef NVRX_SRC
#define NVRX_SRC "../nvidia-resiliency-ext_amd/csrc/nvrx_straggler.hip"
#endif
#include NVRX_SRC

#include <random>

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

static std::mt19937 g_rng(1);

// what runs between two timed launches (argv[6]): 0 nothing, 1 a one-workgroup kernel that stores to pinned host
// memory behind a system-scope fence (what k_score does between two reports), 2 a 1 GiB read sweep that evicts L2 and
// the Infinity Cache (true cold-HBM reads), 3 = 1 + a 30 us host pause
__global__ void k_between_fence(uint32_t *host_word, uint32_t v) {
    if (threadIdx.x == 0) host_word[1] = v;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_between_release(uint32_t *host_word, uint32_t v) {  // between=4: release-only publication
    if (threadIdx.x == 0) host_word[1] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_between_plain(float *dev_word, float v) {  // between=6: a different kernel, nothing special in it
    if (threadIdx.x == 0) dev_word[0] = v;
}
__global__ void k_between_hoststore(uint32_t *host_word, uint32_t v) {  // between=7: stores to pinned host memory, drained, no fence
    if (threadIdx.x < 64) __hip_atomic_store(host_word + threadIdx.x % 8, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__global__ void k_spin(unsigned long long ticks, uint32_t *sink) {  // between=11: resident poller on ANOTHER stream while the rows run
    const unsigned long long t0 = wall_clock64();
    uint32_t spins = 0;
    while (wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(4);
        spins += __hip_atomic_load(sink, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (spins == 0xFFFFFFFFu) *sink = spins;
}
__global__ void k_between_sweep(const float4 *p, size_t n4, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) *sink = acc;
}

static void gen(std::vector<float> &h, int rows, int n, int stride, int dist, float scale) {
    std::normal_distribution<float> nd(10.f, 0.3f);
    std::lognormal_distribution<float> ln(1.0f, 1.5f);
    for (int r = 0; r < rows; r++)
        for (int i = 0; i < n; i++) {
            float v;
            switch (dist) {
                case 1: v = ln(g_rng); break;
                case 2: v = 5.0f + (float)(g_rng() % 7); break;
                case 3: v = 10.f + 5.f * (float)i / (float)n + 0.1f * nd(g_rng); break;
                case 4: v = (g_rng() % 100 == 0) ? 1000.f * nd(g_rng) : nd(g_rng); break;  // outliers
                case 5: v = -nd(g_rng); break;                                               // negative values
                case 6: v = 10.f + 0.01f * (float)(g_rng() % 100); break;                   // 100 distinct values: ~100 equal members per bin
                case 7: v = 10.f + 0.001f * (float)(g_rng() % 400); break;                  // 400 distinct values: ~25 equal members per bin
                default: v = nd(g_rng);
            }
            h[(size_t)r * stride + i] = v * scale;
        }
}

struct Expect {
    std::vector<float> med, mn, mx;
    std::vector<double> avg, sd;
};

static Expect expect(const std::vector<float> &h, int rows, int n, int stride) {
    Expect e;
    e.med.resize(rows); e.mn.resize(rows); e.mx.resize(rows); e.avg.resize(rows); e.sd.resize(rows);
    for (int r = 0; r < rows; r++) {
        std::vector<float> v(h.begin() + (size_t)r * stride, h.begin() + (size_t)r * stride + n);
        double s = 0;
        for (float x : v) s += x;
        const double m = s / n;
        double ss = 0;
        for (float x : v) ss += (x - m) * (x - m);
        e.avg[r] = m;
        e.sd[r] = n > 1 ? sqrt(ss / (n - 1)) : NAN;
        std::nth_element(v.begin(), v.begin() + (n - 1) / 2, v.end());
        e.med[r] = v[(n - 1) / 2];
        e.mn[r] = *std::min_element(v.begin(), v.end());
        e.mx[r] = *std::max_element(v.begin(), v.end());
    }
    return e;
}

#ifdef NVRX_ORACLE_SPLIT
// tools/experiments/oracle_split_patch.py: for every row the bin [base, base + 2^sh) that holds the lower median and at
// most 48 samples, straight from the data (what a splitter-based selection would have to FIND)
static void set_oracle(const std::vector<float> &h, int rows, int n, int stride) {
    static std::vector<uint32_t> o(8192 * 2);
    for (int r = 0; r < rows; r++) {
        std::vector<uint32_t> k(n);
        memcpy(k.data(), h.data() + (size_t)r * stride, (size_t)n * 4);   // non-negative floats: the bits order like the values
        std::sort(k.begin(), k.end());
        const uint32_t med = k[(n - 1) / 2];
        uint32_t sh = 0, base = med;
        for (uint32_t s2 = 1; s2 < 31; s2++) {
            const uint32_t b = med & ~((1u << s2) - 1u);
            const size_t cnt = std::lower_bound(k.begin(), k.end(), b + (1u << s2)) - std::lower_bound(k.begin(), k.end(), b);
            if (cnt > 48) break;
            sh = s2;
            base = b;
        }
        o[(size_t)r * 2] = base;
        o[(size_t)r * 2 + 1] = sh;
    }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_oracle), o.data(), (size_t)rows * 8));
}
#endif

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 512;
    const int n = argc > 2 ? atoi(argv[2]) : 10000;
    const int only_threads = argc > 3 ? atoi(argv[3]) : 0;
    const int dist = argc > 4 ? atoi(argv[4]) : 0;
    const int use_win = argc > 5 ? atoi(argv[5]) : 1;
    const int between = argc > 6 ? atoi(argv[6]) : 0;
    uint32_t *h_word = nullptr;
    CK(hipHostMalloc(&h_word, 64, hipHostMallocMapped));
    float4 *d_sweep = nullptr;
    const size_t sweep_n4 = (size_t)1 << 26;  // 1 GiB
    if (between == 2) CK(hipMalloc(&d_sweep, sweep_n4 * 16));
    const int stride = (n + 3) & ~3;
    std::vector<float> h((size_t)rows * stride, 0.f);
    float *d_s, *d_stats;
    uint32_t *d_c, *d_wlo, *d_wsh;
    CK(hipMalloc(&d_s, h.size() * 4));
    CK(hipMalloc(&d_stats, (size_t)rows * 8 * 4));
    CK(hipMalloc(&d_c, rows * 4));
    CK(hipMalloc(&d_wlo, rows * 4));
    CK(hipMalloc(&d_wsh, rows * 4));
    std::vector<uint32_t> c(rows, (uint32_t)n);
    CK(hipMemcpy(d_c, c.data(), rows * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    // KB_EVFLAGS: flags of the timing events (0x20000000 hipEventDisableSystemFence, 0x40000000 hipEventReleaseToDevice)
    const unsigned evflags = getenv("KB_EVFLAGS") ? (unsigned)strtoul(getenv("KB_EVFLAGS"), nullptr, 0) : 0u;
    CK(hipEventCreateWithFlags(&a, evflags));
    CK(hipEventCreateWithFlags(&b, evflags));
    for (int threads : {256, 512, 1024}) {
        if (only_threads && only_threads != threads) continue;
        const StatsVariant *best = nullptr;
        for (const StatsVariant &v : kVariants)
            if (v.threads == threads && v.threads * v.vpt * 4 >= stride && (!best || v.vpt < best->vpt)) best = &v;
        if (!best) continue;
        Epilogue ep{};
        (void)use_win;
        // KB_EP=1: the report path's epilogue (gid / history loads, exchange-row stores); KB_STREAM=1|2: launch on a created
        // stream (non-blocking / high priority) instead of the null stream
        static float *d_send = nullptr, *d_hmin = nullptr;
        static int32_t *d_gidv = nullptr;
        if (getenv("KB_EP")) {
            if (!d_send) {
                CK(hipMalloc(&d_send, (size_t)(2 * rows + 1) * 4 * 8));
                CK(hipMalloc(&d_hmin, rows * 4));
                CK(hipMalloc(&d_gidv, rows * 4));
                std::vector<int32_t> g(rows);
                for (int r = 0; r < rows; r++) g[r] = r % 64;
                CK(hipMemcpy(d_gidv, g.data(), rows * 4, hipMemcpyHostToDevice));
                CK(hipMemset(d_hmin, 0x7f, rows * 4));
            }
            ep.gid = d_gidv;
            ep.hist_min = d_hmin;
            ep.send = d_send;
            ep.rows_per_rank = 64;
            ep.rows_active = 64;
            ep.K = 0;
            ep.KS = 64;
            ep.L = 129;
            ep.names_ok = 1.0f;
        }
#ifdef NVRX_HAS_RANGE_HINT
        // KB_HINT=1: the rows keep their range hints from launch to launch, as inside the report path
        static uint32_t *d_hint = nullptr;
        if (getenv("KB_HINT")) {
            if (!d_hint) CK(hipMalloc(&d_hint, (size_t)rows * 8));
            CK(hipMemset(d_hint, 0xFF, (size_t)rows * 8));
            ep.hint = d_hint;
        }
#endif
        static hipStream_t kb_stream = nullptr;
        if (getenv("KB_STREAM") && !kb_stream) {
            if (atoi(getenv("KB_STREAM")) == 2) {
                int lo, hi;
                CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                CK(hipStreamCreateWithPriority(&kb_stream, hipStreamNonBlocking, hi));
            } else {
                CK(hipStreamCreateWithFlags(&kb_stream, hipStreamNonBlocking));
            }
        }
        CK(hipMemset(d_wlo, 0, rows * 4));
        CK(hipMemset(d_wsh, 0xFF, rows * 4));
        static hipStream_t spin_stream = nullptr;
        if (between == 11 && !spin_stream) CK(hipStreamCreateWithFlags(&spin_stream, hipStreamNonBlocking));
        auto launch = [&]() -> float {
            hipExtLaunchKernelGGL(best->fn, dim3(rows), dim3(best->threads), 0, kb_stream, a, b, 0, (const float *)d_s,
                                  (const uint32_t *)d_c, (const uint8_t *)nullptr, stride, d_stats, ep, getenv("KB_UNIFORM") ? n : -1);
            if (between == 11) {  // 256-thread poller for ~15 us (100 MHz wall clock), launched right behind the rows
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, spin_stream, 1500ull, (uint32_t *)d_wlo);
                CK(hipStreamSynchronize(spin_stream));
            }
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            return ms;
        };
        std::vector<float> st((size_t)rows * 8);
        int total_bad = 0;
        auto check = [&](const Expect &e, const char *what) {
            CK(hipMemcpy(st.data(), d_stats, st.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0, hits = 0, radix = 0, rebinned = 0, hinted_rows = 0, hint_misses = 0;
            double worst_avg = 0, worst_std = 0;
            for (int r = 0; r < rows; r++) {
                const float *o = &st[(size_t)r * 8];
                if (NVRX_ABLATE == 0 && o[2] != e.med[r]) bad++;
                if (o[0] != e.mn[r] || o[1] != e.mx[r]) bad++;
                hits += o[7] == 1.0f || o[7] == 33.0f;  // estimate (or range hint) held, no refinement
                hinted_rows += o[7] == 33.0f;
                hint_misses += o[7] == 66.0f;
                radix += ((int)o[7] & 8) != 0;  // integer radix fallback
                rebinned += ((int)o[7] & 6) != 0 && ((int)o[7] & 8) == 0;  // re-binned in the float domain, then ranked
                worst_avg = std::max(worst_avg, fabs(o[3] - e.avg[r]) / fabs(e.avg[r]));
                worst_std = std::max(worst_std, fabs(o[4] - e.sd[r]) / fabs(e.sd[r]));
            }
            total_bad += bad;
            printf("  %-28s mismatches %d  fast path %d/%d (hint held %d, missed %d) rebinned %d radix %d  avg_err %.1e std_err %.1e\n", what, bad, hits, rows, hinted_rows, hint_misses, rebinned, radix, worst_avg, worst_std);
        };
        // fresh draws of the same distribution: report 0 is cold, later ones should hit the window
        float ms_seq[6];
        for (int step = 0; step < 6; step++) {
            gen(h, rows, n, stride, dist, 1.0f);
            CK(hipMemcpy(d_s, h.data(), h.size() * 4, hipMemcpyHostToDevice));
#ifdef NVRX_ORACLE_SPLIT
            set_oracle(h, rows, n, stride);
#endif
            ms_seq[step] = launch();
            char what[64];
            snprintf(what, sizeof(what), "fresh draw %d (%.2f us)", step, ms_seq[step] * 1e3);
            check(expect(h, rows, n, stride), what);
        }
        Expect e = expect(h, rows, n, stride);
        double tot = 0;
        float mn = 1e9f;
        const int reps = 50;
        for (int i = 0; i < reps + 5; i++) {
            if (between == 1 || between == 3) hipLaunchKernelGGL(k_between_fence, dim3(1), dim3(256), 0, nullptr, h_word, (uint32_t)i);
            if (between == 4) hipLaunchKernelGGL(k_between_release, dim3(8), dim3(256), 0, nullptr, h_word, (uint32_t)i);
            if (between == 5) hipLaunchKernelGGL(k_between_fence, dim3(8), dim3(256), 0, nullptr, h_word, (uint32_t)i);
            if (between == 6) hipLaunchKernelGGL(k_between_plain, dim3(1), dim3(256), 0, nullptr, d_stats + (size_t)rows * 8 - 1, (float)i);
            if (between == 7) hipLaunchKernelGGL(k_between_hoststore, dim3(1), dim3(256), 0, nullptr, h_word, (uint32_t)i);
            if (between == 9)  // the SAME kernel function on one row in between
                hipLaunchKernelGGL(best->fn, dim3(1), dim3(best->threads), 0, nullptr, (const float *)d_s, (const uint32_t *)d_c,
                                   (const uint8_t *)nullptr, stride, d_stats, ep, -1);
            if (between == 10)  // another instantiation of the same template on one row in between
                hipLaunchKernelGGL(kVariants[0].fn, dim3(1), dim3(kVariants[0].threads), 0, nullptr, (const float *)d_s,
                                   (const uint32_t *)d_c, (const uint8_t *)nullptr, 1024, d_stats, ep, 777);
            if (between == 8) {  // host-side gap only: the queue sits idle for 30 us before the launch
                const auto t0 = std::chrono::steady_clock::now();
                while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(30)) {}
            }
            if (between == 2) hipLaunchKernelGGL(k_between_sweep, dim3(2048), dim3(256), 0, nullptr, (const float4 *)d_sweep, sweep_n4, d_stats);
            if (between == 3) {
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(30)) {}
            }
            const float ms = launch();
            if (i >= 5) {
                tot += ms;
                mn = std::min(mn, ms);
            }
        }
        check(e, "steady state (same data)");
        printf("between=%d ", between);
        printf("ablate=%d dist=%d win=%d threads=%4d vpt=%2d rows=%d n=%d : avg %.2f us  min %.2f us  -> %.0f GB/s\n", NVRX_ABLATE, dist,
               use_win, best->threads, best->vpt, rows, n, tot / reps * 1e3, mn * 1e3, (double)rows * n * 4 / (tot / reps * 1e-3) / 1e9);
#ifdef NVRX_PHASE_CLOCKS
        {
            static unsigned long long ph[4096][12];
            CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph)));
            const int nb = std::min(rows, 4096);
            unsigned long long w0 = ~0ull, w1 = 0;
            for (int r = 0; r < nb; r++) {
                w0 = std::min(w0, ph[r][10]);
                w1 = std::max(w1, ph[r][9]);
            }
            // clock64 bases differ per XCD: report per-block deltas between consecutive phase marks
            for (int i = 1; i <= 8; i++) {
                double s = 0;
                unsigned long long lo = ~0ull, hi = 0;
                for (int r = 0; r < nb; r++) {
                    const unsigned long long v = ph[r][i] - ph[r][i - 1];
                    s += (double)v;
                    lo = std::min(lo, v);
                    hi = std::max(hi, v);
                }
                printf("  phase %d-%d: mean %7.0f  min %6llu  max %6llu clk\n", i - 1, i, s / nb, lo, hi);
            }
            double s = 0, stg = 0;
            unsigned long long hi = 0;
            for (int r = 0; r < nb; r++) {
                s += (double)(ph[r][8] - ph[r][0]);
                stg += (double)(ph[r][10] - w0);
                hi = std::max(hi, ph[r][10] - w0);
            }
            {
                static unsigned long long sb[4096][2][16];
                CK(hipMemcpyFromSymbol(sb, HIP_SYMBOL(g_sub), sizeof(sb)));
                // sub-marks relative to phase mark 3 (all waves released from barrier (2)): first wave / last wave
                printf("  sub-marks after the barrier (2) release, mean clk [first wave | last wave]:");
                for (int i = 0; i < 10; i++) {
                    double a0 = 0, a1 = 0;
                    for (int r = 0; r < nb; r++) {
                        a0 += (double)(long long)(sb[r][0][i] - ph[r][3]);
                        a1 += (double)(long long)(sb[r][1][i] - ph[r][3]);
                    }
                    printf(" s%d %.0f|%.0f", i, a0 / nb, a1 / nb);
                }
                printf("\n");
            }
            printf("  block total: mean %.0f clk; start stagger (wall, 10 ns ticks): mean %.1f max %llu; kernel wall span %llu ticks\n",
                   s / nb, stg / nb, hi, w1 - w0);
        }
#endif
        // the distribution moves: x1.5 (every window misses once), then a mild x1.01 drift
        for (float scale : {1.5f, 1.5f, 1.515f, 0.2f, 0.2f}) {
            gen(h, rows, n, stride, dist, scale);
            CK(hipMemcpy(d_s, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            const float ms = launch();
            char what[64];
            snprintf(what, sizeof(what), "scale x%.3f (%.2f us)", scale, ms * 1e3);
            check(expect(h, rows, n, stride), what);
        }
        printf("  TOTAL mismatches %d\n", total_bad);
    }
    return 0;
}
