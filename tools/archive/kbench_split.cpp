// tools/kbench_split.cpp -- PROTOTYPE of the row split for the 64-row production shape ("next" item 4):
// SPLIT workgroups per row, each histograms its share of the samples; partial histograms meet in the row's
// last-arriving workgroup (agent-scope release / ticket / acquire), which locates the median's bin, re-reads the other
// shares from L2 to collect the bin's members and selects.  Medians are exact (checked against nth_element); min / max /
// moments are NOT merged here -- leaving that work out can only flatter the split.  Timed like kbench (hipExtLaunchKernel
// start/stop events), next to the shipped k_row_stats on the same rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/kbench_split.cpp -o tools/kb/kb_split ; kb_split [rows] [n]
#define NVRX_SRC "../nvidia-resiliency-ext_amd/csrc/nvrx_straggler.hip"
#include NVRX_SRC

#include <random>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

namespace {

constexpr int SP_THREADS = 512;
constexpr int SP_BINS = 4096;
constexpr int SP_PER = SP_BINS / SP_THREADS;

// blockIdx -> (row, part) with all parts of a row on ONE XCD (workgroup b runs on XCD b % 8): the friendliest placement
// for the hand-off
__device__ __forceinline__ void split_place(int b, int split, int &row, int &part) {
    const int xcd = b & 7, q = b >> 3;
    part = q % split;
    row = (q / split) * 8 + xcd;
}

template <int SPLIT, int VPT>
__global__ __launch_bounds__(SP_THREADS) void k_split(const float *__restrict__ samples, int n, int row_stride,
                                                      uint32_t *__restrict__ g_hist, uint32_t *__restrict__ g_ticket,
                                                      float *__restrict__ med_out, int rows) {
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[SP_BINS];
    __shared__ __attribute__((aligned(16))) uint32_t s_sum[SP_THREADS];
    __shared__ uint32_t s_cand[64];
    __shared__ uint32_t s_mm[2], s_cur, s_ticket;
    int row, part;
    split_place((int)blockIdx.x, SPLIT, row, part);
    if (row >= rows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 *src = reinterpret_cast<const float4 *>(samples + (size_t)row * row_stride);
    const int n4 = n / 4;                                  // n is a multiple of 4 here
    const int share = (n4 + SPLIT - 1) / SPLIT;            // float4s per part
    const int lo4 = part * share, hi4 = min(n4, lo4 + share);
    float4 x[VPT];
#pragma unroll
    for (int i = 0; i < VPT; i++) {
        const int v = lo4 + i * SP_THREADS + tid;
        x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < hi4) x[i] = src[v];
    }
    float4 e4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave == 0) e4 = src[lane];                         // the row's first 256 samples: every part estimates from THEM
#pragma unroll
    for (int j = 0; j < SP_PER; j++) s_hist[tid * SP_PER + j] = 0u;
    if (tid < 64) s_cand[tid] = 0xFFFFFFFFu;
    if (tid == 0) s_cur = 0u;
    if (wave == 0) {
        uint32_t a = min(min(__float_as_uint(e4.x), __float_as_uint(e4.y)), min(__float_as_uint(e4.z), __float_as_uint(e4.w)));
        uint32_t b = max(max(__float_as_uint(e4.x), __float_as_uint(e4.y)), max(__float_as_uint(e4.z), __float_as_uint(e4.w)));
        wave_minmax_u32(a, b);
        if (lane == 0) {
            s_mm[0] = a;
            s_mm[1] = b;
        }
    }
    __syncthreads();
    const uint32_t mn0 = s_mm[0], mx0 = s_mm[1], R = mx0 - mn0;
    const uint32_t lo0 = mn0 > R ? mn0 - R : 0u;
    const uint32_t hi0 = mx0 < 0xFFFFFFFFu - R ? mx0 + R : 0xFFFFFFFFu;
    const int top = 32 - __clz((int)(hi0 - lo0));
    const uint32_t sh = (uint32_t)(top > 12 ? top - 12 : 0);
    uint32_t key[VPT * 4];
#pragma unroll
    for (int i = 0; i < VPT; i++) {
        const bool valid = lo4 + i * SP_THREADS + tid < hi4;
        const float xs[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            key[i * 4 + c] = valid ? __float_as_uint(xs[c]) : 0xFFFFFFFFu;
            if (valid) atomicAdd(&s_hist[min(__builtin_elementwise_sub_sat(key[i * 4 + c], lo0) >> sh, (uint32_t)(SP_BINS - 1))], 1u);
        }
    }
    __syncthreads();
    // ---- hand-off: partial histogram -> global, release, ticket -------------------------------------------------
    uint4 *gh = reinterpret_cast<uint4 *>(g_hist + ((size_t)row * SPLIT + part) * SP_BINS);
    const uint4 *sh4 = reinterpret_cast<const uint4 *>(s_hist);
    gh[tid * 2] = sh4[tid * 2];
    gh[tid * 2 + 1] = sh4[tid * 2 + 1];
    __syncthreads();  // the workgroup's stores happen-before thread 0's release (ONE L2 write-back per workgroup, not per wave)
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_ticket = __hip_atomic_fetch_add(&g_ticket[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_ticket != (uint32_t)(SPLIT - 1)) return;         // not the last one of this row
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) g_ticket[row] = 0u;                       // ready for the next launch
    // ---- the finisher merges the other parts' histograms ----------------------------------------------------------
    for (int p = 0; p < SPLIT; p++) {
        if (p == part) continue;
        const uint4 *oh = reinterpret_cast<const uint4 *>(g_hist + ((size_t)row * SPLIT + p) * SP_BINS);
        const uint4 a = oh[tid * 2], b = oh[tid * 2 + 1];
        uint32_t *h = s_hist + tid * SP_PER;
        h[0] += a.x; h[1] += a.y; h[2] += a.z; h[3] += a.w;
        h[4] += b.x; h[5] += b.y; h[6] += b.z; h[7] += b.w;
    }
    // ---- locate (every wave on its own, as the round-2 kernel did) --------------------------------------------------
    {
        uint32_t local = 0u;
#pragma unroll
        for (int j = 0; j < SP_PER; j++) local += s_hist[tid * SP_PER + j];
        s_sum[tid] = local;
    }
    __syncthreads();
    uint32_t k = (uint32_t)(n - 1) >> 1, pop = 0u, bin;
    {
        constexpr int G = SP_THREADS / 64;
        uint32_t local = 0u;
#pragma unroll
        for (int g = 0; g < G; g++) local += s_sum[lane * G + g];
        const uint32_t incl = wave_scan_u32(local), excl = incl - local;
        const int L = __builtin_ctzll(__ballot(k >= excl && k < incl));
        uint32_t krem = k - (uint32_t)__builtin_amdgcn_readlane((int)excl, L);
        uint32_t tsel = 0u;
        for (int g = 0; g < G; g++) {   // (scalar walk: prototype)
            const uint32_t v = s_sum[L * G + g];
            if (krem < v) { tsel = (uint32_t)(L * G + g); break; }
            krem -= v;
        }
        bin = tsel * SP_PER;
        for (int j = 0; j < SP_PER; j++) {
            const uint32_t v = s_hist[tsel * SP_PER + j];
            if (krem < v) { bin = tsel * SP_PER + j; pop = v; break; }
            krem -= v;
        }
        k = krem;
    }
    const uint32_t base = lo0 + (bin << sh), width = sh ? (1u << sh) : 1u;
    // ---- members: own keys from registers, the other parts re-read from L2 -------------------------------------------
    auto offer = [&](uint32_t kk) {
        const uint32_t r = kk - base;
        if (r < width) {
            const uint32_t pos = atomicAdd(&s_cur, 1u);
            if (pos < 64u) s_cand[pos] = r;
        }
    };
#pragma unroll
    for (int j = 0; j < VPT * 4; j++) offer(key[j]);
    for (int p = 0; p < SPLIT; p++) {
        if (p == part) continue;
        const int plo = p * share, phi = min(n4, plo + share);
#pragma unroll
        for (int i = 0; i < VPT; i++) {
            const int v = plo + i * SP_THREADS + tid;
            if (v < phi) {
                const float4 y = src[v];
                offer(__float_as_uint(y.x));
                offer(__float_as_uint(y.y));
                offer(__float_as_uint(y.z));
                offer(__float_as_uint(y.w));
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    float med = __builtin_nanf("");
    if (pop <= 64u && s_cur == pop) {
        const uint32_t own = s_cand[lane];
        unsigned long long live = pop >= 64u ? ~0ull : ((1ull << pop) - 1ull);
        uint32_t kk = k;
        for (uint32_t bit = sh ? 1u << (sh - 1u) : 0u; bit != 0u && (live & (live - 1ull)) != 0ull; bit >>= 1) {
            const unsigned long long ones = __ballot((own & bit) != 0u) & live, zeros = live & ~ones;
            const uint32_t c0 = (uint32_t)__popcll(zeros);
            const bool low = kk < c0;
            live = low ? zeros : ones;
            kk -= low ? 0u : c0;
        }
        med = __uint_as_float(base + (uint32_t)__builtin_amdgcn_readlane((int)own, __builtin_ctzll(live)));
    }
    if (lane == 0) med_out[row] = med;
}

}  // namespace

template <int SPLIT, int VPT>
static void run_split(const char *what, const float *d_s, int rows, int n, int stride, uint32_t *d_hist, uint32_t *d_ticket,
                      float *d_med, const std::vector<float> &expect) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int blocks = ((rows + 7) / 8) * 8 * SPLIT;
    double tot = 0;
    float mn = 1e9f;
    const int reps = 50;
    for (int i = 0; i < reps + 5; i++) {
        hipExtLaunchKernelGGL((k_split<SPLIT, VPT>), dim3(blocks), dim3(SP_THREADS), 0, nullptr, a, b, 0, d_s, n, stride, d_hist,
                              d_ticket, d_med, rows);
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 5) {
            tot += ms;
            mn = std::min(mn, ms);
        }
    }
    std::vector<float> got(rows);
    CK(hipMemcpy(got.data(), d_med, rows * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < rows; r++) bad += got[r] != expect[r];
    printf("%-46s rows=%d n=%d : avg %.2f us  min %.2f us   median mismatches %d\n", what, rows, n, tot / reps * 1e3, mn * 1e3, bad);
}

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 64;
    const int n = argc > 2 ? atoi(argv[2]) : 10000;
    const int stride = (n + 3) & ~3;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(10.f, 0.3f);
    std::vector<float> h((size_t)rows * stride, 0.f), expect(rows);
    for (int r = 0; r < rows; r++) {
        for (int i = 0; i < n; i++) h[(size_t)r * stride + i] = nd(rng);
        std::vector<float> v(h.begin() + (size_t)r * stride, h.begin() + (size_t)r * stride + n);
        std::nth_element(v.begin(), v.begin() + (n - 1) / 2, v.end());
        expect[r] = v[(n - 1) / 2];
    }
    float *d_s, *d_med, *d_stats;
    uint32_t *d_hist, *d_ticket, *d_c;
    CK(hipMalloc(&d_s, h.size() * 4));
    CK(hipMemcpy(d_s, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_med, rows * 4));
    CK(hipMalloc(&d_stats, (size_t)rows * 8 * 4));
    CK(hipMalloc(&d_hist, (size_t)rows * 4 * SP_BINS * 4));
    CK(hipMalloc(&d_ticket, rows * 4));
    CK(hipMemset(d_ticket, 0, rows * 4));
    CK(hipMalloc(&d_c, rows * 4));
    std::vector<uint32_t> c(rows, (uint32_t)n);
    CK(hipMemcpy(d_c, c.data(), rows * 4, hipMemcpyHostToDevice));
    // the shipped kernel on the same rows
    {
        const StatsVariant *best = nullptr;
        for (const StatsVariant &v : kVariants)
            if (v.threads == 512 && v.threads * v.vpt * 4 >= stride && (!best || v.vpt < best->vpt)) best = &v;
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        Epilogue ep{};
        double tot = 0;
        float mn = 1e9f;
        for (int i = 0; i < 55; i++) {
            hipExtLaunchKernelGGL(best->fn, dim3(rows), dim3(512), 0, nullptr, a, b, 0, (const float *)d_s, (const uint32_t *)d_c,
                                  (const uint8_t *)nullptr, stride, d_stats, ep, -1);
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (i >= 5) {
                tot += ms;
                mn = std::min(mn, ms);
            }
        }
        std::vector<float> st((size_t)rows * 8);
        CK(hipMemcpy(st.data(), d_stats, st.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int r = 0; r < rows; r++) bad += st[(size_t)r * 8 + 2] != expect[r];
        printf("%-46s rows=%d n=%d : avg %.2f us  min %.2f us   median mismatches %d\n", "shipped k_row_stats<512,5>, one workgroup per row",
               rows, n, tot / 50 * 1e3, mn * 1e3, bad);
    }
    run_split<1, 5>("split prototype, 1 part (no hand-off partner)", d_s, rows, n, stride, d_hist, d_ticket, d_med, expect);
    run_split<2, 3>("split prototype, 2 workgroups per row", d_s, rows, n, stride, d_hist, d_ticket, d_med, expect);
    run_split<4, 2>("split prototype, 4 workgroups per row", d_s, rows, n, stride, d_hist, d_ticket, d_med, expect);
    return 0;
}
