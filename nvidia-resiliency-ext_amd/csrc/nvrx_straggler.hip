// nvrx_straggler.hip -- gfx950 (MI355X / CDNA4) implementation of include/nvrx_straggler.h.
//
// The straggler-scoring hot path is a streaming reduction + an exact selection, not a contraction:
// no MFMA anywhere.  What matters on CDNA4 is (1) each timing row is read from HBM exactly once with
// coalesced 16-byte loads, (2) the row then stays in VGPRs (10 000 f32 = 40 KB spread over 512-1024
// lanes) while min / max / mean / std are reduced with wave64 shuffles and the median is found by an
// LDS-histogram radix select on order-preserving integer keys, (3) one workgroup per row so that a
// folded 8-rank x 64-section job (512 rows) puts two workgroups on every one of the 256 CUs.
//
// Reference semantics reproduced here (paths relative to
// /root/reference/src/nvidia_resiliency_ext/attribution/straggler/):
//   section rows : straggler.py:185-195   torch.min/max/median(LOWER)/mean/std(unbiased), f64
//   kernel rows  : cupti_src/CuptiProfiler.cpp:44-74  sort, mean-of-middles median, population std
//   rings        : straggler.py:80-83 deque(maxlen) / cupti_src/CircularBuffer.h:53-69
//   scoring      : reporting.py:196-296,338-380   (see k_score)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "nvrx_straggler.h"

namespace {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(NVRX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Host clocks of the last nvrx_report of this thread (microseconds on the monotonic clock): a handful of reads per report,
// kept on so that a report at production cadence can be taken apart where it runs (nvrx_report_clocks; tools/
// cadence_detector_breakdown.py).  [0] entry, [1] stream ordering decided / events enqueued, [2] staged samples flushed,
// [3] statistics kernel launched, [4] exchange enqueued, [5] score kernel launched, [6] completion word seen, [7] spare.
thread_local double g_report_clk[8];
inline void report_clk(int i) {
    g_report_clk[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// Order-preserving float <-> uint32 map: a < b  <=>  f2key(a) < f2key(b) (with -0.0 < +0.0).
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    return __uint_as_float(u);
}

// ---- wave64 reductions on the DPP cross-lane path (no LDS traffic) ------------------------------
// quad_perm -> row_shr:4/8 -> row_bcast:15/31 leaves the wave total in lane 63; readlane broadcasts it
// through an SGPR.  `old` supplies the value for lanes a DPP step has no source for.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
// lanes a DPP step has no source for read 0 (bound_ctrl): no `old` register has to be prepared
template <int CTRL>
__device__ __forceinline__ uint32_t dpp0(uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)src, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_QUAD_1032 = 0xb1, DPP_QUAD_2301 = 0x4e, DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112,
              DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;

// min / max steps hand the operation's identity to lanes without a source, which lets the compiler fold the DPP
// move into the min / max itself (one instruction per step instead of three)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp<DPP_QUAD_1032>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_QUAD_2301>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_ROW_SHR4>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_ROW_SHR8>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_BCAST15>(0xFFFFFFFFu, v));
    v = min(v, dpp<DPP_BCAST31>(0xFFFFFFFFu, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp0<DPP_QUAD_1032>(v));
    v = max(v, dpp0<DPP_QUAD_2301>(v));
    v = max(v, dpp0<DPP_ROW_SHR4>(v));
    v = max(v, dpp0<DPP_ROW_SHR8>(v));
    v = max(v, dpp0<DPP_BCAST15>(v));
    v = max(v, dpp0<DPP_BCAST31>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += dpp0<DPP_QUAD_1032>(v);
    v += dpp0<DPP_QUAD_2301>(v);
    v += dpp0<DPP_ROW_SHR4>(v);
    v += dpp0<DPP_ROW_SHR8>(v);
    v += dpp0<DPP_BCAST15>(v);
    v += dpp0<DPP_BCAST31>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double src) {
    const uint64_t u = (uint64_t)__double_as_longlong(src);
    const uint32_t lo = dpp0<CTRL>((uint32_t)u), hi = dpp0<CTRL>((uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));  // +0.0 where no source lane
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<DPP_QUAD_1032>(v);
    v += dpp_f64<DPP_QUAD_2301>(v);
    v += dpp_f64<DPP_ROW_SHR4>(v);
    v += dpp_f64<DPP_ROW_SHR8>(v);
    v += dpp_f64<DPP_BCAST15>(v);
    v += dpp_f64<DPP_BCAST31>(v);
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// inclusive prefix sum across the wave (row_shr:1,2,4,8 inside a row, then the two row broadcasts)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {
    v += dpp<DPP_ROW_SHR1>(0u, v);
    v += dpp<DPP_ROW_SHR2>(0u, v);
    v += dpp<DPP_ROW_SHR4>(0u, v);
    v += dpp<DPP_ROW_SHR8>(0u, v);
    v += dpp<DPP_BCAST15, 0xa>(0u, v);
    v += dpp<DPP_BCAST31, 0xc>(0u, v);
    return v;
}

struct Epilogue {
    const int32_t *gid;  // [rows] position in the exchange row, -1 = not exchanged; may be null
    float *hist_min;     // [rows] running minimum of MED; may be null
    float *send;         // [local_ranks][L]; may be null
    int rows_per_rank;
    int rows_active;  // blocks per logical rank (0: blockIdx.x is the row)
    int K;
    int KS;
    int L;
    float names_ok;
    // resident-scorer reports: instead of the plain exchange-row stores, thread 0 publishes the row's results as
    // ROW_GRANULES 8-byte {epoch, f32} granules (agent-scope atomic stores: the data is its own flag) that the score
    // kernel, already resident on another stream, polls for; null = plain stores, ordered by the stream
    unsigned long long *rowg;
    uint32_t epoch;
};
constexpr int ROW_GRANULES = 11;  // med, history minimum, weight | the 8 words of the statistics row

// Ablation switch for tools/kbench.cpp only (always 0 in the shipped library):
// 1 = load + min/max/mean/std only (no selection), 0 = everything.
#ifndef NVRX_ABLATE
#define NVRX_ABLATE 0
#endif

// Phase clocks for tools/kbench.cpp only (never defined in the shipped library): thread 0 of every
// workgroup stores the shader clock at phase boundaries.
#ifdef NVRX_PHASE_CLOCKS
__device__ unsigned long long g_phase[4096][12];
#define NVRX_PHASE(i)                                                                     \
    do {                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_phase[blockIdx.x][i] = clock64(); \
    } while (0)
// finer marks inside the selection, taken by the first and the last wave of the workgroup
__device__ unsigned long long g_sub[4096][2][16];
#define NVRX_SUB(i)                                                                                              \
    do {                                                                                                         \
        if ((threadIdx.x == 0 || threadIdx.x == blockDim.x - 64) && blockIdx.x < 4096)                           \
            g_sub[blockIdx.x][threadIdx.x ? 1 : 0][i] = clock64();                                               \
    } while (0)
#else
#define NVRX_PHASE(i) \
    do {              \
    } while (0)
#define NVRX_SUB(i) \
    do {            \
    } while (0)
#endif

// Histogram size: 2^12 bins for the 512- and 1024-thread launches (rows beyond 2048 samples: the selected bin then holds
// 20-40 of 10 000 samples, few enough for one wave to rank), 2^11 for the 256-thread ones.  NVRX_HIST_BITS_WIDE is a
// tools/kbench.cpp switch.
#ifndef NVRX_HIST_BITS_WIDE
#define NVRX_HIST_BITS_WIDE 12
#endif
// the score kernel loads one dword of every 64-byte line of its argument segment up front (A/B switch for experiments)
#ifndef NVRX_TOUCH_KERNARGS
#define NVRX_TOUCH_KERNARGS 1
#endif

// ------------------------------------------------------------------------------------------------
// Cross-rank scoring on the exchanged table (layout in nvrx_straggler.h).  (Fusing this into the tail
// of k_row_stats behind a last-workgroup ticket was measured and dropped: 512 device-scope atomics on
// one word serialise at the memory side of the 8 XCDs, ~40 ns each = 20 us.)
// ------------------------------------------------------------------------------------------------
struct ScoreArgs {
    const float *table;
    const float *minmed_pre;  // null: compute column minima in LDS
    int R, K, S;
    int do_indiv, do_rel;
    double thr[4];  // gpu_rel, section_rel, gpu_indiv, section_indiv
    float *scores;
    uint8_t *flags;
    uint32_t *meta;
    uint32_t *done_counter;  // device word, zero between launches; null = no completion word
    uint32_t seq;
    const float4 *stats_src;  // optional: statistics rows to forward (device -> pinned host)
    float4 *stats_dst;
    int stats_n4;
    int tab_in_lds;   // k_score1 with a prologue (row gather / peer exchange): keep the table it assembles in LDS
    int send_floats;  // local_ranks * L
};

// all_reduce(MIN) of the f32 MED tensor with -1 sentinels (reporting.py:273-295) into s_min[KS]
// `tab` is the table ([R][L]) wherever it lives: the callers pass a pointer whose address space the compiler can see (LDS
// in the kernels that assemble the table themselves: ds_read instead of flat loads), and eight ranks are loaded at a
// time so that the loads are in flight together instead of one round trip per rank.
__device__ __forceinline__ void score_colmin(const ScoreArgs &a, const float *tab, int tid, int nthr, float *s_min) {
    const int KS = a.K + a.S;
    const int L = NVRX_TABLE_LEN(a.K, a.S);
    for (int j = tid; j < KS; j += nthr) {
        float m = INFINITY;
        for (int q = 0; q < a.R; q += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = q + u < a.R ? tab[(size_t)(q + u) * L + j] : INFINITY;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (v[u] < m) m = v[u];
        }
        s_min[j] = m >= 0.0f ? m : __builtin_nanf("");  // reporting.py:289,295
    }
}

// Scores and flags of rank r, by all NTHR threads of the workgroup (contains one barrier).
template <int NTHR>
__device__ __forceinline__ void score_rank(const ScoreArgs &a, const float *tab, int r, int tid, const float *minmed,
                                           double (*s_red)[NTHR / 64], uint32_t (*s_cnt)[NTHR / 64],
                                           float *__restrict__ out, uint8_t *__restrict__ fl) {
    const int K = a.K, S = a.S, KS = K + S;
    const int L = NVRX_TABLE_LEN(K, S);
    const int lane = tid & 63, wave = tid >> 6;
    const float *__restrict__ row = tab + (size_t)r * L;
    const float NaN = __builtin_nanf("");

    // section scores: reference / MED (reporting.py:196-217), rounded to f32 (reporting.py:352)
    for (int s = tid; s < S; s += NTHR) {
        const float med = row[K + s];
        float si = NaN, sr = NaN;
        if (med >= 0.0f) {
            if (a.do_indiv) si = (float)((double)row[KS + K + s] / (double)med);
            if (a.do_rel) sr = (float)((double)minmed[K + s] / (double)med);
        }
        out[2 + s] = si;
        out[2 + S + s] = sr;
        if (fl) {
            fl[2 + s] = ((double)si < a.thr[3]) ? 1 : 0;
            fl[2 + S + s] = ((double)sr < a.thr[1]) ? 1 : 0;
        }
    }

    // GPU score: weighted mean of per-kernel ratios (reporting.py:219-253)
    double wi = 0.0, si = 0.0, wr = 0.0, sr = 0.0;
    uint32_t nk = 0, ncommon = 0;
    for (int k = tid; k < K; k += NTHR) {
        const float medf = row[k];
        if (!(medf >= 0.0f)) continue;
        const double med = (double)medf;
        const double w = (double)row[2 * KS + k];
        nk++;
        si += ((double)row[KS + k] / med) * w;
        wi += w;
        const float mm = minmed[k];
        if (mm == mm) {
            ncommon++;
            sr += ((double)mm / med) * w;
            wr += w;
        }
    }
    if (K > 0) {  // block-uniform
        wi = wave_sum_f64(wi);
        si = wave_sum_f64(si);
        wr = wave_sum_f64(wr);
        sr = wave_sum_f64(sr);
        nk = wave_sum_u32(nk);
        ncommon = wave_sum_u32(ncommon);
    }
    if (lane == 0) {
        s_red[0][wave] = wi;
        s_red[1][wave] = si;
        s_red[2][wave] = wr;
        s_red[3][wave] = sr;
        s_cnt[0][wave] = nk;
        s_cnt[1][wave] = ncommon;
    }
    __syncthreads();
    if (tid == 0) {
        wi = si = wr = sr = 0.0;
        nk = ncommon = 0;
        for (int w = 0; w < NTHR / 64; w++) {
            wi += s_red[0][w];
            si += s_red[1][w];
            wr += s_red[2][w];
            sr += s_red[3][w];
            nk += s_cnt[0][w];
            ncommon += s_cnt[1][w];
        }
        const float gi = (a.do_indiv && nk > 0) ? (float)(si / wi) : NaN;
        const float gr = (a.do_rel && ncommon > 0) ? (float)(sr / wr) : NaN;
        out[0] = gi;
        out[1] = gr;
        if (fl) {
            fl[0] = ((double)gi < a.thr[2]) ? 1 : 0;
            fl[1] = ((double)gr < a.thr[0]) ? 1 : 0;
        }
    }
}

// is_all_true(has_all_names) (name_mapper.py:68-69, dist_utils.py:107-115) folded into the table; one wave
__device__ __forceinline__ void score_meta(const ScoreArgs &a, const float *tab, int lane) {
    const int L = NVRX_TABLE_LEN(a.K, a.S);
    uint32_t bad = 0;
    for (int q = lane; q < a.R; q += 64) bad += (tab[(size_t)q * L + (L - 1)] > 0.0f) ? 0u : 1u;
    bad = wave_sum_u32(bad);
    if (lane == 0) {
        a.meta[0] = bad ? 0u : 1u;
        a.meta[1] = (uint32_t)a.R;
        a.meta[2] = (uint32_t)a.K;
        a.meta[3] = (uint32_t)a.S;
    }
}

// ------------------------------------------------------------------------------------------------
// k_row_stats: one workgroup per timing row.
//   HBM: the row is read once, 16 B per lane per load, VPT independent loads in flight per lane.
//   Registers: the row lives in VPT*4 order-preserving keys per lane for the rest of the kernel (the raw bit patterns:
//   non-negative floats order like their bits; a row with a set sign bit is re-keyed on a cold path).
//   LDS: 16 KB histogram (4096 bins; 2048 for the 256-thread launches) + thread sums + a 1 KB candidate list + a
//   few words of reduction scratch.
//
// What bounds it (tools/micro/*.cpp, phase clocks and PMC passes of tools/kbench.cpp): with one row per CU the kernel
// is a chain of dependent steps (a dependent VALU instruction issues every ~8 clk per wave, an LDS round trip is ~75-130
// clk, a returning LDS atomic ~250, a 6-step DPP reduction ~100, an exchange through a barrier ~175): chain length
// matters, instruction counts do not.  With two rows per CU (the folded bench shape) the CU's VALU issue and its LDS
// pipe (~10 clk per histogram ds_add under random bank conflicts, n/64 of them per row) are shared by 16 waves and
// the instruction count per wave matters as well (871 -> 673 VALU per wave was worth 0.7 us).  So the kernel (1) keeps
// the per-key instruction count low (6.5 VALU + one ds_add per key in the hot loop), (2) starts the histogram before
// the row's exact range is known, so that its DS traffic runs under the HBM load instead of after it, (3) keeps the
// exchanges after the last tile few and every chain between them short, and (4) lets work that only one wave needs
// be done by one wave:
//
//   tile 0 (the first THREADS*4 samples) gives a range estimate [mn0 - R, mx0 + R], R = mx0 - mn0
//   -> every key of every tile goes into the LDS histogram over that range as its tile arrives (keys outside are
//      clamped into the two edge bins), while min / max / packed f32 moments accumulate
//   -> ONE exchange (thread sums); every wave scans them on its own (DPP prefix scan), then the lanes of one DPP row
//      resolve the thread and the bin holding rank k = (n-1)/2 in parallel
//   -> the bin's 20-40 members are collected without a branch per key (per lane: member count, last member, XOR of
//      members) into a compact list (one returning LDS add per wave reserves its stretch), and wave 0 ranks them
//      directly: lane L counts the candidates below candidate L.
//
// The result is always exact: if the median lands in an edge bin (the estimate missed: drifting or heavy-tailed rows)
// the histogram is rebuilt over the exact [kmin, kmax]; a bin too heavy to rank directly (> CAND_MAX members, e.g.
// many equal samples) is refined by further passes until it is a single key value; bins of 65-256 members are
// ranked by all waves together.  Rows that fit in one tile skip the estimate: tile 0 is the whole row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// min and max reductions interleaved so that the two dependent DPP chains hide each other's latency
__device__ __forceinline__ void wave_minmax_u32(uint32_t &mn, uint32_t &mx) {
    mn = min(mn, dpp<DPP_QUAD_1032>(0xFFFFFFFFu, mn));
    mx = max(mx, dpp0<DPP_QUAD_1032>(mx));
    mn = min(mn, dpp<DPP_QUAD_2301>(0xFFFFFFFFu, mn));
    mx = max(mx, dpp0<DPP_QUAD_2301>(mx));
    mn = min(mn, dpp<DPP_ROW_SHR4>(0xFFFFFFFFu, mn));
    mx = max(mx, dpp0<DPP_ROW_SHR4>(mx));
    mn = min(mn, dpp<DPP_ROW_SHR8>(0xFFFFFFFFu, mn));
    mx = max(mx, dpp0<DPP_ROW_SHR8>(mx));
    mn = min(mn, dpp<DPP_BCAST15>(0xFFFFFFFFu, mn));
    mx = max(mx, dpp0<DPP_BCAST15>(mx));
    mn = min(mn, dpp<DPP_BCAST31>(0xFFFFFFFFu, mn));
    mx = max(mx, dpp0<DPP_BCAST31>(mx));
    mn = (uint32_t)__builtin_amdgcn_readlane((int)mn, 63);
    mx = (uint32_t)__builtin_amdgcn_readlane((int)mx, 63);
}

// 1 / c for a sample count (an exactly representable integer, no special cases to patch up): hardware estimate + two
// Newton steps instead of the full division sequence; 1/0 = +inf (a one-sample section: its deviation is NaN anyway)
__device__ __forceinline__ double recip_count(uint32_t c) {
    const double d = (double)c;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return c ? r : __builtin_inf();
}

template <int THREADS, int VPT>
__global__ __launch_bounds__(THREADS) void k_row_stats(const float *__restrict__ samples,
                                                       const uint32_t *__restrict__ counts,
                                                       const uint8_t *__restrict__ kinds, int row_stride,
                                                       float *__restrict__ stats, Epilogue ep, int uniform_n) {
    constexpr int WAVES = THREADS / 64;
    constexpr int HIST_BITS = THREADS >= 512 ? NVRX_HIST_BITS_WIDE : 11;
    constexpr int HIST_BINS = 1 << HIST_BITS;
    constexpr int PER = HIST_BINS / THREADS;  // histogram bins summed per thread
    constexpr int G = THREADS / 64;           // thread sums per lane in the wave-redundant scan
    constexpr int NKEY = VPT * 4;
    constexpr int CAND_MAX = 256;  // a selected bin this small is finished by direct ranking
    constexpr int CAND_ONE = 64;   // ... and one this small by every wave on its own, without a further exchange
    static_assert(HIST_BINS % THREADS == 0, "THREADS must divide the histogram size");
    static_assert(THREADS >= CAND_MAX, "candidate list is initialised one word per thread");

    __shared__ __attribute__((aligned(16))) uint32_t s_hist[HIST_BINS];
    __shared__ __attribute__((aligned(16))) uint32_t s_sum[THREADS];
    __shared__ __attribute__((aligned(16))) uint32_t s_cand[CAND_MAX];  // compact candidate list (unused slots: 0xFFFFFFFF)
    __shared__ __attribute__((aligned(16))) uint32_t s_lt[CAND_MAX];    // per slot: how many candidates are smaller
    __shared__ __attribute__((aligned(16))) double s_d[2 * WAVES];  // [0,W) partial sums, [W,2W) squared deviations
    __shared__ __attribute__((aligned(16))) uint32_t s_mm[4];       // {tile-0 min, tile-0 max, row min, row max}
    __shared__ uint32_t s_cur[1];                                   // append cursor of the candidate list

    const int row = ep.rows_active ? (int)(blockIdx.x / ep.rows_active) * ep.rows_per_rank + (int)(blockIdx.x % ep.rows_active)
                                   : (int)blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    NVRX_PHASE(0);
#ifdef NVRX_PHASE_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_phase[blockIdx.x][10] = wall_clock64();
#endif
    // uniform_n >= 0: every launched row holds that many samples (kernel argument: no counts upload)
    uint32_t n = uniform_n >= 0 ? (uint32_t)uniform_n : counts[row];
    if (n > (uint32_t)row_stride) n = (uint32_t)row_stride;
    const int kind = kinds ? kinds[row] : NVRX_KIND_SECTION;
    // epilogue inputs are fetched now so their latency hides under the row load
    const int row_gid = (ep.send && ep.gid) ? ep.gid[row] : -1;
    const float row_hmin = ep.hist_min ? ep.hist_min[row] : __builtin_nanf("");

    float r_min, r_max, r_med, r_avg, r_std;
    int path = 0;  // diagnostic: 1 = estimate held, 2 = histogram rebuilt over the exact range, +4 = refined

    if (n == 0) {
        r_min = r_max = r_med = r_avg = r_std = __builtin_nanf("");
    } else {
        const float4 *__restrict__ src = reinterpret_cast<const float4 *>(samples + (size_t)row * (size_t)row_stride);
        uint32_t key[NKEY];

        // ---- load (HBM -> VGPR): VPT independent 16-byte loads per lane ---------------------------
        float4 x[VPT];
#pragma unroll
        for (int i = 0; i < VPT; i++) {
            const int v = i * THREADS + tid;
            x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((uint32_t)(v * 4) < n) x[i] = src[v];
        }
        // LDS scratch is initialised while the loads are in flight
        if constexpr (PER % 4 == 0) {
#pragma unroll
            for (int j = 0; j < PER; j += 4) reinterpret_cast<uint4 *>(s_hist)[(tid * PER + j) >> 2] = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int j = 0; j < PER; j++) s_hist[tid * PER + j] = 0u;
        }
        if (tid < CAND_MAX) {
            s_cand[tid] = 0xFFFFFFFFu;
            s_lt[tid] = 0u;
        }
        if (tid < 4) s_mm[tid] = (tid & 1) ? 0u : 0xFFFFFFFFu;
        if (tid == 0) s_cur[0] = 0u;
        __syncthreads();  // (0)

        // Tiles that are valid for every lane (i < full_tiles, block-uniform) skip the tail masking;
        // slots past the row's end get the key 0xFFFFFFFF and never take part in anything.
        const int full_tiles = (int)(n / (uint32_t)(THREADS * 4));
        const uint32_t k_rank = (n - 1u) >> 1;
        // reciprocals off the critical path (computed while the loads are in flight): 1/n for the mean, and the
        // variance's denominator (n-1 for sections, n for kernel rows: CuptiProfiler.cpp:66-72)
        const double inv_n = recip_count(n);
        const double inv_den = kind == NVRX_KIND_KERNEL ? inv_n : recip_count(n - 1u);

        // ---- tile 0: keys, and the range estimate the histogram is laid over ---------------------------
        uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
        {
            const float xs[4] = {x[0].x, x[0].y, x[0].z, x[0].w};
            const uint32_t e = (uint32_t)tid * 4u;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool valid = full_tiles > 0 || e + c < n;
                const uint32_t kk = __float_as_uint(xs[c]);  // raw bits: see "keys" below
                key[c] = valid ? kk : 0xFFFFFFFFu;
                kmn = min(kmn, key[c]);
                kmx = max(kmx, valid ? kk : 0u);
            }
            uint32_t a = kmn, b = kmx;
            wave_minmax_u32(a, b);
            if (lane == 0) {
                atomicMin(&s_mm[0], a);
                atomicMax(&s_mm[1], b);
            }
        }
        __syncthreads();  // (1) tile-0 range
        const bool speculative = VPT > 1 && n > (uint32_t)(THREADS * 4);
        uint32_t lo0, sh;  // histogram origin (a key) and log2 of the bin width
        {
            const uint32_t mn0 = uni(s_mm[0]), mx0 = uni(s_mm[1]);
            const uint32_t R = speculative ? mx0 - mn0 : 0u;
            lo0 = mn0 > R ? mn0 - R : 0u;
            const uint32_t hi0 = mx0 < 0xFFFFFFFFu - R ? mx0 + R : 0xFFFFFFFFu;
            const int top = 32 - __clz((int)(hi0 - lo0));  // (hi0 - lo0) >> sh < 2048
            sh = (uint32_t)(top > HIST_BITS ? top - HIST_BITS : 0);
        }
        NVRX_PHASE(1);

        // ---- every tile: histogram + min / max + moments in ONE pass over the registers ------------------
        // f32 partials per lane around the lane's own pivot p (its first sample): s = sum(x-p),
        // q = sum((x-p)^2).  Once the row mean m is known each lane turns them into its exact share of
        // sum((x-m)^2) = q - 2(m-p)s + cnt(m-p)^2 in f64; cross-lane sums are f64 throughout.
        const float pivot = x[0].x;
        const f32x2 pivot2 = {pivot, pivot};
        uint32_t cnt = 0u;
        float psum = 0.f, psq = 0.f;  // partial tiles (lane-masked), scalar
        f32x2 psum2 = {0.f, 0.f}, psq2 = {0.f, 0.f};  // full tiles, packed
#pragma unroll
        for (int i = 0; i < VPT; i++) {
            const float xs[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
            if (i < full_tiles) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    uint32_t kk;
                    if (i == 0) {
                        kk = key[c];
                    } else {
                        kk = __float_as_uint(xs[c]);
                        key[i * 4 + c] = kk;
                        kmn = min(kmn, kk);
                        kmx = max(kmx, kk);
                    }
                    if (NVRX_ABLATE == 0) atomicAdd(&s_hist[min(__builtin_elementwise_sub_sat(kk, lo0) >> sh, (uint32_t)(HIST_BINS - 1))], 1u);
                }
                // moments of the tile as two packed pairs (v_pk_add_f32 / v_pk_fma_f32)
                const f32x2 d01 = f32x2{xs[0], xs[1]} - pivot2, d23 = f32x2{xs[2], xs[3]} - pivot2;
                psum2 += d01;
                psq2 = __builtin_elementwise_fma(d01, d01, psq2);
                psum2 += d23;
                psq2 = __builtin_elementwise_fma(d23, d23, psq2);
                cnt += 4u;
            } else {
                const uint32_t e = (uint32_t)(i * THREADS + tid) * 4u;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const bool valid = e + c < n;
                    uint32_t kk;
                    if (i == 0) {
                        kk = key[c];
                    } else {
                        kk = valid ? __float_as_uint(xs[c]) : 0xFFFFFFFFu;
                        key[i * 4 + c] = kk;
                        kmn = min(kmn, kk);
                        kmx = max(kmx, valid ? kk : 0u);
                    }
                    const float d = valid ? xs[c] - pivot : 0.f;
                    psum += d;
                    psq = fmaf(d, d, psq);
                    cnt += valid ? 1u : 0u;
                    if (NVRX_ABLATE == 0 && valid)
                        atomicAdd(&s_hist[min(__builtin_elementwise_sub_sat(kk, lo0) >> sh, (uint32_t)(HIST_BINS - 1))], 1u);
                }
            }
        }
        psum += psum2.x + psum2.y;
        psq += psq2.x + psq2.y;
        wave_minmax_u32(kmn, kmx);
        double sum = wave_sum_f64((double)psum + (double)cnt * (double)pivot);
        if (lane == 0) {
            atomicMin(&s_mm[2], kmn);
            atomicMax(&s_mm[3], kmx);
            s_d[wave] = sum;
        }
        NVRX_PHASE(2);
        __syncthreads();  // (2) histogram complete, row min / max / partial sums published
        kmn = uni(s_mm[2]);
        kmx = uni(s_mm[3]);
        NVRX_PHASE(3);

        // Locate rank k in the finished histogram: every thread sums its PER consecutive bins, ONE
        // exchange, then every wave scans all THREADS sums on its own (lane L holds sums [L*G, L*G+G)).
        // Returns the bin; k becomes the rank inside it, pop its population (all wave-uniform).
        auto locate = [&](uint32_t &k, uint32_t &pop) -> uint32_t {
            {
                uint32_t local = 0u;
#pragma unroll
                for (int j = 0; j < PER; j++) local += s_hist[tid * PER + j];
                s_sum[tid] = local;
            }
            NVRX_SUB(0);
            __syncthreads();  // thread sums published
            NVRX_SUB(1);
            uint32_t ts[G];
            uint32_t local = 0u;
#pragma unroll
            for (int g = 0; g < G; g++) {
                ts[g] = s_sum[lane * G + g];
                local += ts[g];
            }
            const uint32_t incl = wave_scan_u32(local);
            const uint32_t excl = incl - local;
            NVRX_SUB(2);
            const int L = __builtin_ctzll(__ballot(k >= excl && k < incl));  // exactly one lane owns rank k
            uint32_t krem = k - (uint32_t)__builtin_amdgcn_readlane((int)excl, L);
            // Two more levels, each resolved by the lanes of the first DPP row in parallel (one value per lane, a prefix
            // scan inside the row, a ballot): which of that lane's G thread sums holds the rank, then which of that
            // thread's PER bins.  `cnt` <= 16 values at `src`; returns the index, leaves the rank inside it in krem
            // and the value itself in `picked`.
            auto row_pick = [&](const uint32_t *src, int cnt, uint32_t &picked) -> uint32_t {
                const uint32_t val = lane < cnt ? src[lane & 15] : 0u;
                uint32_t inc2 = val;
                if (cnt > 1) inc2 += dpp0<DPP_ROW_SHR1>(inc2);
                if (cnt > 2) inc2 += dpp0<DPP_ROW_SHR2>(inc2);
                if (cnt > 4) inc2 += dpp0<DPP_ROW_SHR4>(inc2);
                if (cnt > 8) inc2 += dpp0<DPP_ROW_SHR8>(inc2);
                // first lane whose inclusive count exceeds the remaining rank (the last one always does)
                const int B = __builtin_ctzll(__ballot(lane < cnt && krem < inc2) | (1ull << (cnt - 1)));
                picked = (uint32_t)__builtin_amdgcn_readlane((int)val, B);
                krem -= (uint32_t)__builtin_amdgcn_readlane((int)(inc2 - val), B);
                return (uint32_t)B;
            };
            static_assert(PER <= 16 && G <= 16, "one DPP row resolves a level");
            uint32_t tsum;
            const uint32_t tsel = (uint32_t)L * G + row_pick(s_sum + L * G, G, tsum);  // thread whose PER bins hold rank k
            const uint32_t bsel = row_pick(s_hist + tsel * PER, PER, pop);
            k = krem;
            NVRX_SUB(3);
            return tsel * PER + bsel;
        };

        // pairwise sum of WAVES per-wave partials (a 2- or 3-level tree instead of a chain of dependent f64 adds)
        auto sum_partials = [&](const double *q) -> double {
            double t[WAVES];
#pragma unroll
            for (int w = 0; w < WAVES; w++) t[w] = q[w];
#pragma unroll
            for (int step = 1; step < WAVES; step *= 2)
#pragma unroll
                for (int w = 0; w + step < WAVES; w += 2 * step) t[w] += t[w + step];
            return t[0];
        };
        // mean and this wave's share of the squared deviations (needs the partial sums of barrier (2))
        double mean = 0.0;
        auto finish_moments = [&]() {
            mean = sum_partials(s_d) * inv_n;  // row sum: the same fixed pairing in every thread
            const double dm = mean - (double)pivot;
            const double lane_ss = (double)psq - 2.0 * dm * (double)psum + (double)cnt * dm * dm;
            const double wss = wave_sum_f64(lane_ss);
            if (lane == 0) s_d[WAVES + wave] = wss;
        };

        // Keys.  Up to here a sample's key is its raw bit pattern: for non-negative floats -- every timing row there
        // is -- the bits order like the values, so the hot loop spends nothing on a key transform.  A set sign bit
        // anywhere (a negative sample, -0.0, a signed NaN) shows up as a row maximum >= 0x80000000: such a row is
        // re-keyed in its registers with the order-preserving map f2key, its range taken again, and it continues
        // through the exact-range rebuild below (block-uniform; costs that row a few extra exchanges).
        const bool conv = kmx >= 0x80000000u;
        if (conv) {
            uint32_t a = 0xFFFFFFFFu, b = 0u;
#pragma unroll
            for (int j = 0; j < NKEY; j++) {
                const bool valid = (j >> 2) < full_tiles || (uint32_t)(((j >> 2) * THREADS + tid) * 4 + (j & 3)) < n;
                const uint32_t kk = f2key(__uint_as_float(key[j]));
                key[j] = valid ? kk : 0xFFFFFFFFu;
                a = min(a, key[j]);
                b = max(b, valid ? kk : 0u);
            }
            wave_minmax_u32(a, b);
            __syncthreads();  // every thread holds the raw range
            if (tid == 0) {
                s_mm[2] = 0xFFFFFFFFu;
                s_mm[3] = 0u;
            }
            __syncthreads();
            if (lane == 0) {
                atomicMin(&s_mm[2], a);
                atomicMax(&s_mm[3], b);
            }
            __syncthreads();
            kmn = uni(s_mm[2]);
            kmx = uni(s_mm[3]);
        }
        auto key_value = [&](uint32_t kk) -> float { return conv ? key2f(kk) : __uint_as_float(kk); };

        uint32_t med_key;
        if (NVRX_ABLATE != 0 || kmn == kmx) {
            med_key = kmn;  // all samples equal (or selection ablated)
            finish_moments();
            __syncthreads();
        } else {
            uint32_t k = k_rank, pop = 0u, bin = 0u;
            path = 1;
            bool rebuild = conv;
            if (!conv) {
                bin = locate(k, pop);
                rebuild = speculative && (bin == 0u || bin == (uint32_t)(HIST_BINS - 1));
            }
            if (rebuild) {
                // The estimate missed (the edge bins also hold everything that was clamped), or the row was re-keyed:
                // rebuild the histogram over the exact range (nothing is clamped any more) and locate again.
                path = conv ? 18 : 2;
                lo0 = kmn;
                const int top = 32 - __clz((int)(kmx - kmn));
                sh = (uint32_t)(top > HIST_BITS ? top - HIST_BITS : 0);
                // every wave must be done with the first locate (it reads s_hist after its barrier) before any wave
                // clears the histogram: a wave that read cleared bins picked another bin, did not rebuild, and the row's
                // median came out wrong (seen with 16 waves on rows whose first tile is the row's maximum)
                __syncthreads();
#pragma unroll
                for (int j = 0; j < PER; j++) s_hist[tid * PER + j] = 0u;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NKEY; j++)
                    if (key[j] != 0xFFFFFFFFu || (j < full_tiles * 4)) atomicAdd(&s_hist[(key[j] - lo0) >> sh], 1u);
                __syncthreads();
                k = k_rank;
                bin = locate(k, pop);
            }
            uint32_t base = lo0 + (bin << sh);  // first key of the bin that holds the median
            NVRX_PHASE(4);
            bool moments_done = false;
            const bool all_waves_finish = kind == NVRX_KIND_KERNEL && (n & 1u) == 0u;
            while (sh > 0u) {
                if (pop <= (uint32_t)CAND_MAX) {
                    // ---- finish by ranking the bin's few members directly ----------------------------------
                    // One branch-free stream over the keys: membership compare, the wave's member count from the
                    // compare masks on the scalar unit, and per lane its own member count, its last member and the XOR
                    // of its members.  One returning LDS add reserves the wave's stretch of the compact candidate list
                    // (issued by hand: the compiler's uniform-atomic rewrite would wait for the result on the spot); its
                    // latency hides under the moments.  Lanes holding one or two members -- all of them, bar one
                    // workgroup in thirty -- place them from a prefix scan of the counts (two members: last, and
                    // XOR ^ last); a wave in which some lane holds three or more revisits its keys one by one instead.
                    uint32_t wcnt = 0u, wbase = 0u;
                    if constexpr (NKEY <= 24) {
                        uint32_t mcnt = 0u, mcnt_hi = 0u, last = 0u, last_hi = 0u, mxor = 0u, mxor_hi = 0u;
#pragma unroll
                        for (int j = 0; j < NKEY; j++) {
                            const uint32_t r = key[j] - base;  // wraps to a huge value below the bin
                            const bool member = (r >> sh) == 0u;
                            wcnt += (uint32_t)__popcll(__ballot(member));
                            if (j < NKEY / 2) {  // two half-length chains per quantity
                                mcnt += member ? 1u : 0u;
                                last = member ? r : last;
                                mxor ^= member ? r : 0u;
                            } else {
                                mcnt_hi += member ? 1u : 0u;
                                last_hi = member ? r : last_hi;
                                mxor_hi ^= member ? r : 0u;
                            }
                        }
                        last = mcnt_hi ? last_hi : last;
                        mcnt += mcnt_hi;
                        mxor ^= mxor_hi;
                        NVRX_SUB(4);
                        if (lane == 0 && wcnt) {
                            const uint32_t cur_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)s_cur;
                            asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(wbase) : "v"(cur_addr), "v"(wcnt) : "memory");
                        }
                        if (!moments_done) {
                            finish_moments();
                            moments_done = true;
                        }
                        NVRX_SUB(5);
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wbase) : : "memory");
                        wbase = uni(wbase);
                        if (wcnt) {  // wave-uniform
                            if (__ballot(mcnt > 2u) == 0ull) {
                                const uint32_t pos = wbase + wave_scan_u32(mcnt) - mcnt;
                                if (mcnt >= 1u && pos < (uint32_t)CAND_MAX) s_cand[pos] = last;
                                if (mcnt >= 2u && pos + 1u < (uint32_t)CAND_MAX) s_cand[pos + 1u] = mxor ^ last;
                            } else {
                                uint32_t run = wbase;
#pragma unroll
                                for (int j = 0; j < NKEY; j++) {
                                    const uint32_t r = key[j] - base;
                                    const bool member = (r >> sh) == 0u;
                                    const unsigned long long b = __ballot(member);
                                    if (b) {  // wave-uniform
                                        const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                                        if (member && pos < (uint32_t)CAND_MAX) s_cand[pos] = r;
                                        run += (uint32_t)__popcll(b);
                                    }
                                }
                            }
                        }
                    } else {
                        // long register tiles (more masks than SGPRs): count first, then revisit mask by mask
#pragma unroll
                        for (int j = 0; j < NKEY; j++) wcnt += (uint32_t)__popcll(__ballot(((key[j] - base) >> sh) == 0u));
                        if (lane == 0 && wcnt) wbase = atomicAdd(&s_cur[0], wcnt);
                        wbase = uni(wbase);
                        if (!moments_done) {
                            finish_moments();
                            moments_done = true;
                        }
                        if (wcnt) {  // wave-uniform
                            uint32_t run = wbase;
#pragma unroll
                            for (int j = 0; j < NKEY; j++) {
                                const uint32_t r = key[j] - base;
                                const bool member = (r >> sh) == 0u;
                                const unsigned long long b = __ballot(member);
                                if (b) {  // wave-uniform
                                    const uint32_t pos = run + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                                    if (member && pos < (uint32_t)CAND_MAX) s_cand[pos] = r;
                                    run += (uint32_t)__popcll(b);
                                }
                            }
                        }
                    }
                    NVRX_SUB(6);
                    __syncthreads();  // candidates listed (unused slots hold 0xFFFFFFFF), deviation partials published
                    NVRX_SUB(7);
                    NVRX_PHASE(5);
                    if (pop <= (uint32_t)CAND_ONE) {
                        // Few enough for one wave: lane L owns candidate L and counts the candidates below it (the list is
                        // read as LDS broadcasts, 32 slots per batch of loads); the value at rank k is the largest
                        // candidate with lt <= k (empty slots compare as +inf: lt = pop > k).  Nothing is exchanged
                        // any more: section rows are finished by wave 0 alone (only thread 0 stores results; the other
                        // waves are done and leave the SIMDs to it), kernel rows with an even count need the median in
                        // every wave for their second middle element, so every wave ranks.
                        if (wave != 0 && !all_waves_finish) {
                            sh = 0u;
                            break;
                        }
                        const uint32_t own = s_cand[lane];
                        uint32_t lt0 = 0u, lt1 = 0u, lt2 = 0u, lt3 = 0u;  // four short add chains instead of one long one
#pragma unroll
                        for (int half = 0; half < CAND_ONE / 32; half++) {
                            if (half == 0 || pop > (uint32_t)(half * 32)) {  // wave-uniform
                                uint4 v[8];
#pragma unroll
                                for (int q = 0; q < 8; q++) v[q] = reinterpret_cast<const uint4 *>(s_cand)[half * 8 + q];
#pragma unroll
                                for (int q = 0; q < 8; q++) {
                                    lt0 += (v[q].x < own) ? 1u : 0u;
                                    lt1 += (v[q].y < own) ? 1u : 0u;
                                    lt2 += (v[q].z < own) ? 1u : 0u;
                                    lt3 += (v[q].w < own) ? 1u : 0u;
                                }
                            }
                        }
                        base += wave_max_u32((lt0 + lt1) + (lt2 + lt3) <= k ? own : 0u);
                    } else {
                        // All-pairs ranking split over the waves: every lane owns 4 of the CAND_MAX slots, each
                        // wave compares all owners with ITS OWN members only and adds its partial "smaller than"
                        // counts into s_lt; after one more exchange every wave picks the value at rank k on its
                        // own -- the largest candidate with lt <= k.
                        const uint4 own = reinterpret_cast<const uint4 *>(s_cand)[lane];
                        const uint32_t cv[4] = {own.x, own.y, own.z, own.w};
                        uint32_t lt[4] = {0u, 0u, 0u, 0u};
                        const uint32_t wend = min(wbase + wcnt, (uint32_t)CAND_MAX);
                        for (uint32_t j = wbase; j < wend; j++) {
                            const uint32_t v = s_cand[j];  // same address in every lane: LDS broadcast
#pragma unroll
                            for (int q = 0; q < 4; q++) lt[q] += (v < cv[q]) ? 1u : 0u;
                        }
                        if (wcnt) {  // wave-uniform
#pragma unroll
                            for (int q = 0; q < 4; q++) atomicAdd(&s_lt[lane * 4 + q], lt[q]);
                        }
                        __syncthreads();  // partial counts added up
                        const uint4 tot = reinterpret_cast<const uint4 *>(s_lt)[lane];
                        uint32_t best = tot.x <= k ? cv[0] : 0u;
                        best = max(best, tot.y <= k ? cv[1] : 0u);
                        best = max(best, tot.z <= k ? cv[2] : 0u);
                        best = max(best, tot.w <= k ? cv[3] : 0u);
                        base += wave_max_u32(best);
                    }
                    sh = 0u;
                    break;
                }
                // ---- too heavy to rank directly: spread the bin's members over up to HIST_BINS finer bins -------
                path |= 4;
                const uint32_t sh2 = sh > (uint32_t)HIST_BITS ? sh - (uint32_t)HIST_BITS : 0u;
                __syncthreads();  // every wave is done reading s_hist / s_sum
#pragma unroll
                for (int j = 0; j < PER; j++) s_hist[tid * PER + j] = 0u;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NKEY; j++) {
                    const uint32_t r = key[j] - base;
                    if ((r >> sh) == 0u) atomicAdd(&s_hist[r >> sh2], 1u);
                }
                __syncthreads();
                bin = locate(k, pop);
                base += bin << sh2;
                sh = sh2;
            }
            if (!moments_done) {
                finish_moments();
                __syncthreads();
            }
            med_key = base;
            NVRX_SUB(8);
            NVRX_PHASE(6);
        }
        const uint32_t dsel = med_key - kmn;
        float med = key_value(med_key);

        if (kind == NVRX_KIND_KERNEL && (n & 1u) == 0u) {
            // mean of the two middle order statistics (CuptiProfiler.cpp:57-59): also need rank k+1.
            // Slots past the row's end hold 0xFFFFFFFF and sort after every real sample.
            uint32_t cnt_le = 0u, mn_gt = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < NKEY; j++) {
                const uint32_t d = key[j] - kmn;
                cnt_le += (d <= dsel) ? 1u : 0u;
                mn_gt = min(mn_gt, d > dsel ? d : 0xFFFFFFFFu);
            }
            cnt_le = wave_sum_u32(cnt_le);
            mn_gt = wave_min_u32(mn_gt);
            __syncthreads();  // s_sum is free again
            if (lane == 0) {
                s_sum[wave] = cnt_le;
                s_sum[WAVES + wave] = mn_gt;
            }
            __syncthreads();
            cnt_le = s_sum[0];
            mn_gt = s_sum[WAVES];
#pragma unroll
            for (int w = 1; w < WAVES; w++) {
                cnt_le += s_sum[w];
                mn_gt = min(mn_gt, s_sum[WAVES + w]);
            }
            const uint32_t dnext = (cnt_le >= k_rank + 2u) ? dsel : mn_gt;
            med = (med + key_value(kmn + dnext)) / 2.0f;
        }

        if (wave != 0) return;  // thread 0 stores the row's results: the other waves are done
        NVRX_PHASE(7);
        // squared-deviation partials were published before the last barrier each path went through
        const double ss = sum_partials(s_d + WAVES);
        r_min = key_value(kmn);
        r_max = key_value(kmx);
        r_med = med;
        r_avg = (float)mean;
        r_std = (kind == NVRX_KIND_KERNEL || n > 1u) ? sqrtf((float)(ss * inv_den)) : __builtin_nanf("");
        NVRX_SUB(9);
    }

    if (tid == 0) {
        float *o = stats + (size_t)row * NVRX_STATS_STRIDE;
        const float weight = n ? (float)n * r_avg : 0.0f;
        o[NVRX_STAT_MIN] = r_min;
        o[NVRX_STAT_MAX] = r_max;
        o[NVRX_STAT_MED] = r_med;
        o[NVRX_STAT_AVG] = r_avg;
        o[NVRX_STAT_STD] = r_std;
        o[NVRX_STAT_NUM] = (float)n;
        o[NVRX_STAT_WEIGHT] = weight;
        o[7] = (float)path;  // diagnostic only (selection path taken)
        float h = row_hmin;
        if (ep.hist_min) {
            // _update_local_min_times (reporting.py:298-314): history is never reset by a report
            if (n && r_med < h) {
                h = r_med;
                ep.hist_min[row] = h;
            }
        }
        if (ep.rowg) {
            unsigned long long *g8 = ep.rowg + (size_t)blockIdx.x * ROW_GRANULES;
            const unsigned long long tag = (unsigned long long)ep.epoch << 32;
            const float vals[ROW_GRANULES] = {n ? r_med : -1.0f, n ? h : __builtin_nanf(""), weight,
                                              r_min, r_max, r_med, r_avg, r_std, (float)n, weight, (float)path};
#pragma unroll
            for (int q = 0; q < ROW_GRANULES; q++)
                __hip_atomic_store(g8 + q, tag | (unsigned long long)__float_as_uint(vals[q]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        } else if (ep.send) {
            const int lr = row / ep.rows_per_rank;
            float *s = ep.send + (size_t)lr * ep.L;
            const int g = row_gid;
            if (g >= 0 && g < ep.KS) {
                s[g] = n ? r_med : -1.0f;  // -1 = "no stats on this rank" (reporting.py:273)
                s[ep.KS + g] = n ? h : __builtin_nanf("");
                if (g < ep.K) s[2 * ep.KS + g] = weight;
            }
            if (row - lr * ep.rows_per_rank == 0) s[ep.L - 1] = ep.names_ok;
        }
    }
    NVRX_PHASE(8);
#ifdef NVRX_PHASE_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_phase[blockIdx.x][9] = wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------
// k_scatter: staged (row, slot, value) samples + row metadata from pinned host memory -> device.
// ------------------------------------------------------------------------------------------------
struct StagedSample {
    uint32_t row_slot;  // row << 16 | slot
    float value;
};

// Completion ticket: the launch's last workgroup stores `ticket` into `done_word` (pinned host memory) -- the host knows
// from that word alone that the pinned staging buffers have been read and may be filled again, so a flush records no
// event (a cold hipEventRecord was ~20 us of a report at production cadence, profiles/r04q).
__global__ void k_scatter(const uint32_t *__restrict__ h_counts, const StagedSample *__restrict__ h_entries,
                          int n_entries, float *__restrict__ samples, int row_stride,
                          uint32_t *__restrict__ d_counts, int rows, const uint8_t *__restrict__ h_kinds,
                          uint8_t *__restrict__ d_kinds, const int32_t *__restrict__ h_gid,
                          int32_t *__restrict__ d_gid, uint32_t *__restrict__ done_cnt, uint32_t *done_word,
                          uint32_t ticket) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_entries) {
        const StagedSample e = h_entries[i];
        samples[(size_t)(e.row_slot >> 16) * (size_t)row_stride + (e.row_slot & 0xFFFFu)] = e.value;
    }
    if (i < rows) {
        d_counts[i] = h_counts[i];
        if (h_kinds) {
            d_kinds[i] = h_kinds[i];
            d_gid[i] = h_gid[i];
        }
    }
    if (done_word) {
        __syncthreads();  // every load of this workgroup from the staging buffers has returned (its value was stored)
        if (threadIdx.x == 0) {
            const uint32_t arrived = __hip_atomic_fetch_add(done_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (arrived == gridDim.x - 1u) {
                __hip_atomic_store(done_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the buffer's next launch comes after the host saw the ticket
                __hip_atomic_store(done_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_stamp_begin / k_stamp_end: GPU time of a code region measured ON the device.  One-thread kernels
// on the user's stream read the constant-rate wall clock (s_memrealtime, 100 MHz on MI355X) when the
// region's work starts and ends in stream order; the end kernel writes the elapsed microseconds straight
// into the region's ring slot (and, optionally, the section's host-measured CPU sample into its slot),
// so no hipEvent pair has to be waited for and read back and no sample is staged through pinned memory.
// ------------------------------------------------------------------------------------------------
__global__ void k_stamp_begin(unsigned long long *slot) { *slot = wall_clock64(); }

// The same without a kernel argument, one instantiation per stamp slot.  This platform places kernel arguments in device
// memory (HIP_FORCE_DEV_KERNARG defaults to 1): a launch WITH arguments writes them through the PCIe BAR and waits for the
// write before it rings the doorbell -- 3.4-3.8 us of host time per launch against 1.3-2.3 us for a kernel that has none
// (tools/archive/micro/launch_cost.cpp).  The slot is a compile-time constant, its storage a __device__ array (one per
// device, shared by the contexts of a process; slots are handed out process-wide).
constexpr int NVRX_NSTAMP = 64;  // (the Python layer times the OUTERMOST region only -- cupti.py -- so two are open at most; 64 kernel names is what a user's own profile of the job shows at worst)
__device__ unsigned long long g_stamp_slots[NVRX_NSTAMP];
template <int SLOT>
__global__ void k_stamp_begin_at() {
    g_stamp_slots[SLOT] = wall_clock64();
}
using StampBeginFn = void (*)();
template <int... I>
static const StampBeginFn *stamp_begin_table(std::integer_sequence<int, I...>) {
    static const StampBeginFn table[] = {k_stamp_begin_at<I>...};
    return table;
}
static const StampBeginFn *const g_stamp_begin_fn = stamp_begin_table(std::make_integer_sequence<int, NVRX_NSTAMP>{});

__global__ void k_stamp_end(const unsigned long long *slot, float us_per_tick, float *dst_gpu, float *dst_cpu,
                            float cpu_value) {
    const unsigned long long t = wall_clock64();
    *dst_gpu = (float)(t - *slot) * us_per_tick;  // microseconds, as CuptiProfiler.cpp:191
    if (dst_cpu) *dst_cpu = cpu_value;
}

// nvrx_ring_push_device_rows: `width` floats of every row of a [rows][ld] matrix into ring rows `dpitch` floats apart.
// blockIdx.y = row, the x dimension strides over the row: consecutive lanes, consecutive floats (coalesced both ways).
__global__ void k_append_rows(float *__restrict__ dst, size_t dpitch, const float *__restrict__ src, size_t ld, int width, int rows) {
    for (int row = blockIdx.y; row < rows; row += gridDim.y) {
        const float *s = src + (size_t)row * ld;
        float *d = dst + (size_t)row * dpitch;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < width; i += gridDim.x * blockDim.x) d[i] = s[i];
    }
}

__global__ void k_fill_f32(float *p, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_send_init(float *send, int rows, int K, int S) {
    const int KS = K + S;
    const int L = NVRX_TABLE_LEN(K, S);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * L) return;
    const int j = (int)(i % L);
    float v;
    if (j < KS)
        v = -1.0f;
    else if (j < 2 * KS)
        v = __builtin_nanf("");
    else
        v = 0.0f;
    send[i] = v;
}

typedef float nvrx_f4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// k_colmin / k_score: cross-rank scoring on the exchanged table (layout in nvrx_straggler.h).
// ------------------------------------------------------------------------------------------------
__global__ void k_colmin(const float *__restrict__ table, int R, int KS, int L, float *__restrict__ minmed) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= KS) return;
    float m = INFINITY;
    for (int r = 0; r < R; r++) {
        const float v = table[(size_t)r * L + j];
        if (v < m) m = v;
    }
    minmed[j] = m >= 0.0f ? m : __builtin_nanf("");  // reporting.py:289,295
}

constexpr int SCORE_THREADS = 256;

__global__ __launch_bounds__(SCORE_THREADS) void k_score(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_min[];  // [KS] when minmed_pre == null
    __shared__ double s_red[4][SCORE_THREADS / 64];
    __shared__ uint32_t s_cnt[2][SCORE_THREADS / 64];

    const int r = blockIdx.x;
    const int tid = threadIdx.x;

    // forward the local statistics rows next to the scores (keeps PCIe stores out of k_row_stats)
    for (int i = r * SCORE_THREADS + tid; i < a.stats_n4; i += a.R * SCORE_THREADS) a.stats_dst[i] = a.stats_src[i];

    const float *minmed = a.minmed_pre;
    if (!minmed) {
        score_colmin(a, a.table, tid, SCORE_THREADS, s_min);
        __syncthreads();
        minmed = s_min;
    }
    score_rank<SCORE_THREADS>(a, a.table, r, tid, minmed, s_red, s_cnt, a.scores + (size_t)r * NVRX_SCORE_LEN(a.S),
                              a.flags ? a.flags + (size_t)r * NVRX_SCORE_LEN(a.S) : nullptr);
    if (r == 0 && a.meta && (tid >> 6) == 1) score_meta(a, a.table, tid & 63);

    if (a.done_counter) {
        // Completion word for a polling host (results may live in pinned host memory): every block
        // makes its stores visible system-wide, then takes a ticket; the last one publishes `seq`.
        __threadfence_system();  // every wave: its stores into the pinned result block have landed
        __syncthreads();
        if (tid == 0 && a.R == 1) {
            // one workgroup: nobody to wait for
            a.meta[5] = a.seq;
            __hip_atomic_store(&a.meta[4], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (tid == 0) {
            const uint32_t ticket = atomicAdd(a.done_counter, 1u);
            if (ticket == (uint32_t)a.R - 1u) {
                *a.done_counter = 0u;  // ready for the next launch (launches on one stream are ordered)
                a.meta[5] = a.seq;     // statistics were forwarded by every block before its ticket
                __threadfence_system();
                __hip_atomic_store(&a.meta[4], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Jobs beyond the single-workgroup scorer (R > 64 ranks): k_colmin_part / k_colmin_finish + k_score_tile.
//
// k_colmin walks the R rows of a column with ONE lane and k_score spends a workgroup, a system-scope fence and a
// device-scope ticket on every rank (the ticket alone: ~40 ns x R on one word), so both grow linearly with the job:
// 189 us at 1024 ranks x 64 sections, 728 us at 4096 (tools/score_scale.py).  Here the column minima are taken in two
// levels (row chunks x 64-column tiles, four waves per workgroup striding the chunk's rows with eight loads in flight
// each; a second, tiny pass folds the <= 32 chunk minima and applies the -1 -> NaN rule), and one workgroup scores a TILE
// of 8 or 16 ranks: results staged in LDS, stored to the (pinned) result block as 16-byte units, one fence and one
// ticket per tile.  Same arithmetic as score_rank (section scores: the f64 quotient rounded to f32; GPU scores: f64
// sums, one wave per rank), so the oracle comparisons of k_score apply unchanged.
// ------------------------------------------------------------------------------------------------
constexpr int COLMIN_MAX_CHUNKS = 32;

__global__ __launch_bounds__(256) void k_colmin_part(const float *__restrict__ table, int R, int KS, int L,
                                                     int rows_per_chunk, float *__restrict__ part) {
    __shared__ float s_m[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int r_end = min(R, ((int)blockIdx.y + 1) * rows_per_chunk);
    float m = INFINITY;
    if (j < KS) {
        for (int r = (int)blockIdx.y * rows_per_chunk + wave; r < r_end; r += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = r + 4 * u < r_end ? table[(size_t)(r + 4 * u) * L + j] : INFINITY;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (v[u] < m) m = v[u];
        }
    }
    s_m[wave][lane] = m;
    __syncthreads();
    if (wave == 0 && j < KS) {
#pragma unroll
        for (int w = 1; w < 4; w++)
            if (s_m[w][lane] < m) m = s_m[w][lane];
        part[(size_t)blockIdx.y * KS + j] = m;
    }
}

__global__ void k_colmin_finish(const float *__restrict__ part, int chunks, int KS, float *__restrict__ minmed) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= KS) return;
    float m = INFINITY;
    for (int c = 0; c < chunks; c++) {
        const float v = part[(size_t)c * KS + j];
        if (v < m) m = v;
    }
    minmed[j] = m >= 0.0f ? m : __builtin_nanf("");  // reporting.py:289,295
}

constexpr int TILE_THREADS = 256;
constexpr size_t TILE_MAX_LDS = 48 * 1024;

__host__ __device__ inline size_t score_tile_lds_bytes(int ranks, int S) {
    const size_t nout = (size_t)ranks * NVRX_SCORE_LEN(S);
    return ((nout + 3) & ~(size_t)3) * 4 + ((nout + 15) & ~(size_t)15);
}

__global__ __launch_bounds__(TILE_THREADS) void k_score_tile(ScoreArgs a, int tile_ranks) {
    extern __shared__ __attribute__((aligned(16))) float s_tile[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, S = a.S, KS = K + S, W = NVRX_SCORE_LEN(S);
    const int L = NVRX_TABLE_LEN(K, S);
    const int r0 = (int)blockIdx.x * tile_ranks;
    const int n = min(tile_ranks, a.R - r0);
    const int nout = n * W, nout4 = (nout + 3) & ~3, nfl16 = (nout + 15) & ~15;
    float *s_out = s_tile;
    uint8_t *s_fl = reinterpret_cast<uint8_t *>(s_tile + nout4);
    const float *__restrict__ minmed = a.minmed_pre;
    const float NaN = __builtin_nanf("");

    // forward the local statistics rows next to the scores (keeps PCIe stores out of k_row_stats)
    for (int i = (int)blockIdx.x * TILE_THREADS + tid; i < a.stats_n4; i += (int)gridDim.x * TILE_THREADS)
        a.stats_dst[i] = a.stats_src[i];
    // the padding of the staged arrays is stored too
    if (tid < nout4 - nout) s_out[nout + tid] = 0.f;
    if (tid < nfl16 - nout) s_fl[nout + tid] = 0;

    // section scores: reference / MED (reporting.py:196-217), rounded to f32 (reporting.py:352); every (rank, section)
    // pair of the tile is independent
    for (int idx = tid; idx < n * S; idx += TILE_THREADS) {
        const int q = idx / S, sct = idx - q * S;
        const float *__restrict__ row = a.table + (size_t)(r0 + q) * L;
        const float med = row[K + sct];
        float si = NaN, sr = NaN;
        if (med >= 0.0f) {
            if (a.do_indiv) si = (float)((double)row[KS + K + sct] / (double)med);
            if (a.do_rel) sr = (float)((double)minmed[K + sct] / (double)med);
        }
        s_out[q * W + 2 + sct] = si;
        s_out[q * W + 2 + S + sct] = sr;
        s_fl[q * W + 2 + sct] = ((double)si < a.thr[3]) ? 1 : 0;
        s_fl[q * W + 2 + S + sct] = ((double)sr < a.thr[1]) ? 1 : 0;
    }
    // GPU scores: weighted mean of per-kernel ratios (reporting.py:219-253), one wave per rank
    for (int q = wave; q < n; q += TILE_THREADS / 64) {
        const float *__restrict__ row = a.table + (size_t)(r0 + q) * L;
        double wi = 0.0, si = 0.0, wr = 0.0, sr = 0.0;
        uint32_t nk = 0, ncommon = 0;
        for (int k = lane; k < K; k += 64) {
            const float medf = row[k];
            if (!(medf >= 0.0f)) continue;
            const double med = (double)medf;
            const double w = (double)row[2 * KS + k];
            nk++;
            si += ((double)row[KS + k] / med) * w;
            wi += w;
            const float mm = minmed[k];
            if (mm == mm) {
                ncommon++;
                sr += ((double)mm / med) * w;
                wr += w;
            }
        }
        if (K > 0) {  // wave-uniform
            wi = wave_sum_f64(wi);
            si = wave_sum_f64(si);
            wr = wave_sum_f64(wr);
            sr = wave_sum_f64(sr);
            nk = wave_sum_u32(nk);
            ncommon = wave_sum_u32(ncommon);
        }
        if (lane == 0) {
            const float gi = (a.do_indiv && nk > 0) ? (float)(si / wi) : NaN;  // no kernels -> NaN, never flagged
            const float gr = (a.do_rel && ncommon > 0) ? (float)(sr / wr) : NaN;
            s_out[q * W] = gi;
            s_out[q * W + 1] = gr;
            s_fl[q * W] = ((double)gi < a.thr[2]) ? 1 : 0;
            s_fl[q * W + 1] = ((double)gr < a.thr[0]) ? 1 : 0;
        }
    }
    if (blockIdx.x == 0 && a.meta && wave == 1) score_meta(a, a.table, lane);
    __syncthreads();

    // staged results -> result block, 16 bytes per lane (the tile's rows are one contiguous, 16-byte aligned span)
    {
        const nvrx_f4 *src = reinterpret_cast<const nvrx_f4 *>(s_out);
        nvrx_f4 *dst = reinterpret_cast<nvrx_f4 *>(a.scores + (size_t)r0 * W);
        for (int i = tid; i < nout4 / 4; i += TILE_THREADS) dst[i] = src[i];
        if (a.flags) {
            const nvrx_f4 *fsrc = reinterpret_cast<const nvrx_f4 *>(s_fl);
            nvrx_f4 *fdst = reinterpret_cast<nvrx_f4 *>(a.flags + (size_t)r0 * W);
            for (int i = tid; i < nfl16 / 16; i += TILE_THREADS) fdst[i] = fsrc[i];
        }
    }
    if (a.done_counter) {
        // completion word for a polling host, as in k_score: every tile makes its stores visible system-wide, then
        // takes a ticket; the last one publishes `seq`
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            const uint32_t ticket = atomicAdd(a.done_counter, 1u);
            if (ticket == gridDim.x - 1u) {
                *a.done_counter = 0u;  // ready for the next launch (launches on one stream are ordered)
                a.meta[5] = a.seq;     // statistics were forwarded by every tile before its ticket
                __threadfence_system();
                __hip_atomic_store(&a.meta[4], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_peer_allgather: the report's exchange as direct xGMI peer stores (no RCCL kernel, no proxy thread).
//
// Every process owns a WINDOW in fine-grained device memory that all processes of the node have mapped through HIP
// IPC: [2 parities][world][stride] 8-byte granules {epoch << 32 | f32 bits}.  One workgroup per process:
//   publish  this process' `count` floats go to slot [epoch & 1][rank] of EVERY window (its own included) as
//            single 8-byte system-scope stores -- the data carries its own flag, so no fence and no ordering
//            between granules is needed (an aligned 8-byte store is never torn);
//   sweep    the same threads poll the granules of every rank's slot in their OWN window until the tag equals
//            this report's epoch and write the values to `recv` ([world][count] f32, plain device memory for the
//            score kernel that follows on the stream).
// Two parities, because a fast process may publish report n+1 while a slow one still sweeps report n; it cannot
// reach report n+2 before every process has published n+1, i.e. has finished sweeping n.
// The exchange carries ~0.5 KB per rank: latency-bound (one xGMI hop + one poll pass), link bandwidth irrelevant.
// A peer that never arrives turns into a bounded spin: after `timeout_ticks` of the constant-rate wall clock the
// thread gives up, stores NaN and raises the window's error word (host-visible), and the host reports it.
// The same code runs as the prologue of k_score1 (nvrx_report with the peer route and a table that fits the
// single-workgroup score kernel): the exchange then costs no kernel of its own.
// ------------------------------------------------------------------------------------------------
constexpr int PEER_THREADS = 1024;

struct PeerArgs {
    unsigned long long *const *windows;  // [world] device array: every process' window base, as mapped HERE
    const float *send;
    float *recv;
    uint32_t *err;  // host-visible error word
    int world, rank, count, stride;
    uint32_t epoch;
    unsigned long long timeout_ticks;
};

// The exchange, by every thread of the calling workgroup (no barrier inside; the caller synchronises before reading recv).
// `lds_send` (optional): this process' rows as the calling workgroup holds them in LDS (read instead of a.send);
// `lds_tab` (optional): LDS copy of the gathered table, filled next to a.recv, so the caller can score without
// reading back what it has just written to memory.
__device__ __forceinline__ void peer_exchange_block(const PeerArgs &a, const float *lds_send = nullptr, float *lds_tab = nullptr) {
    const int nthr = (int)blockDim.x;
    const int total = a.world * a.count;
    const size_t slot = (size_t)(a.epoch & 1u) * (size_t)a.world;
    const float *src = lds_send ? lds_send : a.send;
    for (int idx = threadIdx.x; idx < total; idx += nthr) {
        const int p = idx / a.count, j = idx - p * a.count;
        const unsigned long long g = ((unsigned long long)a.epoch << 32) | (unsigned long long)__float_as_uint(src[j]);
        __hip_atomic_store(a.windows[p] + (slot + (size_t)a.rank) * (size_t)a.stride + j, g, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // Sweep: every pass issues the loads of ALL granules this thread still waits for, so that one pass (one memory
    // round trip) after the last peer has published the thread is done -- polling them one after the other would put
    // a round trip per granule behind the last arrival.
    const unsigned long long *mine = a.windows[a.rank];
    const unsigned long long t0 = wall_clock64();
    constexpr int CH = 8;
    for (int base = threadIdx.x; base < total; base += nthr * CH) {
        unsigned long long x[CH];
        uint32_t pending = 0;
#pragma unroll
        for (int c = 0; c < CH; c++)
            if (base + c * nthr < total) pending |= 1u << c;
        uint32_t spins = 0;
        while (pending) {
#pragma unroll
            for (int c = 0; c < CH; c++)
                if (pending & (1u << c)) {
                    const int idx = base + c * nthr;
                    const int r = idx / a.count, j = idx - r * a.count;
                    x[c] = __hip_atomic_load(mine + (slot + (size_t)r) * (size_t)a.stride + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
#pragma unroll
            for (int c = 0; c < CH; c++)
                if ((pending & (1u << c)) && (uint32_t)(x[c] >> 32) == a.epoch) {
                    const float v = __uint_as_float((uint32_t)x[c]);
                    a.recv[base + c * nthr] = v;
                    if (lds_tab) lds_tab[base + c * nthr] = v;
                    pending &= ~(1u << c);
                }
            if (pending) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 0xFFu) == 0u && wall_clock64() - t0 > a.timeout_ticks) {
#pragma unroll
                    for (int c = 0; c < CH; c++)
                        if (pending & (1u << c)) {
                            a.recv[base + c * nthr] = __builtin_nanf("");
                            if (lds_tab) lds_tab[base + c * nthr] = __builtin_nanf("");
                        }
                    __hip_atomic_store(a.err, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    pending = 0;
                }
            }
        }
    }
}

__global__ __launch_bounds__(PEER_THREADS) void k_peer_allgather(PeerArgs a) { peer_exchange_block(a); }

// ------------------------------------------------------------------------------------------------
// k_score1: the whole table scored by ONE workgroup (R <= 64 ranks, results staged in LDS).
//
// The report's second kernel is pure latency (a few KB in, a few KB out over PCIe), so what counts is the
// number of dependent steps between "table visible" and "host sees the completion word":
//   * one workgroup -> no inter-workgroup ticket, no device-scope atomics;
//   * scores and flag bytes are staged in LDS and leave as 16-byte stores (a byte store to host memory is one
//     fabric write each; 1040 of them cost more than the arithmetic);
//   * the result block lives in pinned host memory, which the GPU maps uncached: its stores go straight to
//     the fabric, so instead of a system-scope release fence (an L2 write-back sweep per wave, ~1.7 us each
//     on gfx950) every wave drains its own stores with s_waitcnt vmcnt(0), the workgroup meets at a barrier
//     and lane 0 then stores the sequence word -- PCIe keeps posted writes of one requester in order.
//     NVRX_DEBUG_SCORE_FENCE=1 (read by the host library) adds the release fence back in front of that store.
// ------------------------------------------------------------------------------------------------
constexpr int SCORE1_THREADS = 1024;
constexpr int SCORE1_RESIDENT_THREADS = 256;
constexpr int SCORE1_MAX_RANKS = 64;
constexpr size_t SCORE1_MAX_LDS = 60 * 1024;

// 16-byte write-through store at system scope (sc0 sc1): not tracked by the compiler's waitcnt insertion,
// the caller drains with s_waitcnt vmcnt(0) before publishing.
__device__ __forceinline__ void store16_sys(void *dst, nvrx_f4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(v) : "memory");
}

__host__ __device__ inline size_t score1_lds_bytes(int R, int K, int S) {
    const size_t ks4 = (size_t)((K + S + 3) & ~3);
    const size_t nout = (size_t)R * NVRX_SCORE_LEN(S);
    return ks4 * 4 + ((nout + 3) & ~(size_t)3) * 4 + ((nout + 15) & ~(size_t)15);
}
// extra LDS of the prologue variants: the gathered table [R][L] and this process' rows [send_floats]
__host__ __device__ inline size_t score1_table_lds_bytes(int R, int K, int S, int send_floats) {
    return (((size_t)R * NVRX_TABLE_LEN(K, S) + 3) & ~(size_t)3) * 4 + (((size_t)send_floats + 3) & ~(size_t)3) * 4;
}

// Resident-scorer reports: where the score kernel finds the rows' results while k_row_stats is still running.
struct GatherArgs {
    const unsigned long long *g;  // this report's granules [n_blocks][ROW_GRANULES]; null = rows arrived by stream order
    const int32_t *gid;           // [rows] ring row -> slot in the exchange row (-1 = not exchanged)
    float *send;                  // [local_ranks][L] exchange rows, assembled here
    nvrx_f4 *stats_out;           // [rows][2] statistics rows in the result block (forwarded after the scores), or null
    uint32_t *err;                // host-visible word: epoch of a wait that gave up
    int n_blocks, rows_active, rows_per_rank, local_ranks;
    float names_ok;
    uint32_t epoch;
    unsigned long long timeout_ticks;
    int poll_naps;  // s_sleep(1) units (64 clk each) between two polling passes
};

// One granule of this epoch (bounded wait; NaN + error word when the row never shows up).
__device__ __forceinline__ float wait_granule(const unsigned long long *p, const GatherArgs &g, unsigned long long t0) {
    uint32_t spins = 0;
    for (;;) {
        const unsigned long long x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(x >> 32) == g.epoch) return __uint_as_float((uint32_t)x);
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 0xFFu) == 0u && wall_clock64() - t0 > g.timeout_ticks) {
            __hip_atomic_store(g.err, g.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return __builtin_nanf("");
        }
    }
}

// NTHR = 1024: launched behind k_row_stats (or behind the exchange) on the report's stream.
// NTHR = 256 : the RESIDENT scorer -- launched on a second stream next to k_row_stats, small enough (one wave per SIMD)
//              to sit on a CU beside two row workgroups; it polls the rows' granules, so k_row_stats has no successor in
//              its own queue (a queued successor adds ~1.9 us to a kernel's measured duration and the dependent launch
//              ~1.5-2 us more to the report) and this kernel's dispatch is off the critical path.
template <int NTHR>
__global__ __launch_bounds__(NTHR) void k_score1(ScoreArgs a, int fence, PeerArgs pa, GatherArgs ga) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];

    const int K = a.K, S = a.S, KS = K + S, R = a.R;
    const int L = NVRX_TABLE_LEN(K, S);
    const int W = NVRX_SCORE_LEN(S);
    const int nout = R * W;
    const int nout4 = (nout + 3) & ~3;
    float *s_min = reinterpret_cast<float *>(s_raw);
    float *s_out = s_min + ((KS + 3) & ~3);
    uint8_t *s_fl = reinterpret_cast<uint8_t *>(s_out + nout4);
    // prologue variants keep what they assemble in LDS: scoring then never reads back what was just written to memory
    float *s_tab = a.tab_in_lds ? reinterpret_cast<float *>(s_fl + ((nout + 15) & ~15)) : nullptr;  // [R][L]
    float *s_send = a.tab_in_lds ? s_tab + ((R * L + 3) & ~3) : nullptr;                             // [send_floats]
    const int tid = threadIdx.x;
    const float NaN = __builtin_nanf("");
    const unsigned long long t_begin = wall_clock64();
#if NVRX_TOUCH_KERNARGS
    {   // The ~300-byte argument segment is read lazily by scalar loads, a 64-byte line at a time where the code first
        // needs it -- one of them inside the scoring pass, i.e. behind the last row.  One dword of every line is loaded
        // here instead, while the kernel has nothing to do but wait for the rows (r02: with the arguments in host memory
        // the resident tail was 5.8 us instead of 2.8).
        constexpr int KARG_DWORDS = (int)((sizeof(ScoreArgs) + sizeof(int) + sizeof(PeerArgs) + sizeof(GatherArgs)) / 4);
        const uint32_t __attribute__((address_space(4))) *ka =
            (const uint32_t __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
        uint32_t touched = 0u;
#pragma unroll
        for (int d = 0; d < KARG_DWORDS; d += 16) touched ^= ka[d];
        asm volatile("" ::"s"(touched));
    }
#endif
    // (rank, section) of this thread's first pair in the flat scoring pass, and the step to its next one: the integer
    // divisions happen here, before the wait for the rows, not behind it
    const int pair_r0 = tid / max(S, 1), pair_s0 = tid - pair_r0 * S;
    const int pair_dr = NTHR / max(S, 1), pair_ds = NTHR - pair_dr * S;
    // Resident scorer: everything the waves hand each other travels through LDS, so its barriers only wait for the LDS
    // queue.  A full __syncthreads() also waits for every global store in flight -- the exchange rows written to
    // device memory and, worse, the meta words stored to pinned HOST memory: a PCIe round trip in the middle of the tail.
    const bool lds_only = ga.g && s_send;
    auto sync = [&]() {
        if (lds_only)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            __syncthreads();
    };

    if (ga.g && s_send) {
        // sentinels of the exchange row (nvrx_send_init): -1 = no statistics, NaN history, zero weight
        for (int j = tid; j < a.send_floats; j += NTHR) {
            const int col = j % L;
            s_send[j] = col < KS ? -1.0f : (col < 2 * KS ? NaN : 0.0f);
        }
        __syncthreads();
    }
    if (ga.g) {
        // the rows' exchange values, as they are published: med / history minimum / weight of every launched row go
        // to the slots k_row_stats itself would have written (packing loops of reporting.py:273-279)
        // (all granules a thread still waits for are loaded in every pass: one round trip behind the last arrival)
        constexpr int CH = 8;
        const int total = ga.n_blocks * 3;
        for (int base = tid; base < total; base += NTHR * CH) {
            unsigned long long x[CH];
            int slot[CH];  // where a granule's value goes in the exchange rows (-1: nowhere) -- looked up BEFORE the wait,
                           // so that the ring-row -> slot loads are not a dependent step behind the last arrival
            uint32_t pending = 0;
#pragma unroll
            for (int c = 0; c < CH; c++) {
                slot[c] = -1;
                const int idx = base + c * NTHR;
                if (idx < total) {
                    pending |= 1u << c;
                    const int b = idx / 3, q = idx - 3 * b;
                    const int lr = b / ga.rows_active;
                    const int gidv = ga.gid[lr * ga.rows_per_rank + (b - lr * ga.rows_active)];
                    if (gidv >= 0 && gidv < KS && (q < 2 || gidv < K)) slot[c] = lr * L + q * KS + gidv;
                }
            }
            uint32_t spins = 0;
            while (pending) {
#pragma unroll
                for (int c = 0; c < CH; c++)
                    if (pending & (1u << c)) {
                        const int idx = base + c * NTHR;
                        const int b = idx / 3, q = idx - 3 * b;
                        x[c] = __hip_atomic_load(ga.g + (size_t)b * ROW_GRANULES + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int c = 0; c < CH; c++) {
                    if (!(pending & (1u << c))) continue;
                    bool got = (uint32_t)(x[c] >> 32) == ga.epoch;
                    float v = __uint_as_float((uint32_t)x[c]);
                    if (!got && (spins & 0xFFu) == 0xFFu && wall_clock64() - t_begin > ga.timeout_ticks) {
                        __hip_atomic_store(ga.err, ga.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        v = NaN;
                        got = true;
                    }
                    if (got) {
                        if (slot[c] >= 0) {
                            ga.send[slot[c]] = v;
                            if (s_send) s_send[slot[c]] = v;
                        }
                        pending &= ~(1u << c);
                    }
                }
                if (pending) {
                    for (int nap = 0; nap < ga.poll_naps; nap++) __builtin_amdgcn_s_sleep(1);
                    spins++;
                }
            }
        }
        for (int lr = tid; lr < ga.local_ranks; lr += NTHR) {
            ga.send[(size_t)lr * L + (L - 1)] = ga.names_ok;
            if (s_send) s_send[lr * L + (L - 1)] = ga.names_ok;
        }
        sync();
    }
    const unsigned long long t_rows = wall_clock64();  // resident scorer: every row's exchange values have arrived
    if (pa.windows) {
        // the report's exchange as this kernel's prologue: publish this process' rows into every window, sweep ours
        // into a.table (plain device memory, written and read by this one workgroup) and into LDS
        peer_exchange_block(pa, ga.g ? s_send : nullptr, s_tab);
        __syncthreads();
        if (s_tab) a.table = s_tab;
    } else if (ga.g && s_send) {
        a.table = s_send;  // one process: its rows are the whole table
    }
    // local statistics rows -> result block; issued first so the loads overlap everything below
    for (int i = tid; i < a.stats_n4; i += NTHR) {
        const float4 v = a.stats_src[i];
        store16_sys(a.stats_dst + i, nvrx_f4{v.x, v.y, v.z, v.w});
    }
    // The scoring proper, instantiated twice: over the table in LDS (the variants that assemble it themselves; the
    // pointer's address space is visible to the compiler there, so every access is a ds_read) or over the table in
    // device memory.  Through one generic pointer the same loads are FLAT loads with the latency of a memory access:
    // eight dependent ones per column made up most of the resident scorer's tail.
    auto score_all = [&](const float *tab) {
        score_colmin(a, tab, tid, NTHR, s_min);
        if (a.meta && (tid >> 6) == NTHR / 64 - 1) score_meta(a, tab, tid & 63);
        // zero the padding of the staged arrays (it is stored too)
        if (tid < nout4 - nout) s_out[nout + tid] = 0.f;
        if (tid < ((nout + 15) & ~15) - nout) s_fl[nout + tid] = 0;
        sync();

        // Section scores of every rank in one flat pass: each (rank, section) pair is independent, no barriers.  (reference
        // / MED, reporting.py:196-217, rounded to f32 as in reporting.py:352.  The reference divides in f64 and rounds to
        // f32; the correctly rounded f32 quotient is the same number -- 53 >= 2 * 24 + 2 significand bits: rounding
        // twice is innocuous for a quotient -- at a third of the dependent chain.)
        {
            int r = pair_r0, sct = pair_s0;  // (rank, section) of pair `tid`, worked out before the wait for the rows
#pragma unroll 2
            for (int idx = tid; idx < R * S; idx += NTHR) {
                const float *__restrict__ row = tab + (size_t)r * L;
                const float med = row[K + sct];
                float si = NaN, sr = NaN;
                if (med >= 0.0f) {
                    if (a.do_indiv) si = __fdiv_rn(row[KS + K + sct], med);
                    if (a.do_rel) sr = __fdiv_rn(s_min[K + sct], med);
                }
                s_out[r * W + 2 + sct] = si;
                s_out[r * W + 2 + S + sct] = sr;
                s_fl[r * W + 2 + sct] = ((double)si < a.thr[3]) ? 1 : 0;
                s_fl[r * W + 2 + S + sct] = ((double)sr < a.thr[1]) ? 1 : 0;
                r += pair_dr;
                sct += pair_ds;
                if (sct >= S) {
                    sct -= S;
                    r++;
                }
            }
        }
        // GPU scores: weighted mean of per-kernel ratios (reporting.py:219-253).  One WAVE per rank (ranks w, w + waves,
        // ...): a rank's few GPU-timed rows fit the lanes of a wave, its sums are wave reductions, and nothing is
        // exchanged between waves -- R ranks cost one pass instead of R barrier-separated ones.
        if (K == 0) {
            for (int r = tid; r < R; r += NTHR) {  // reporting.py:226-228: no kernels -> NaN, never flagged
                s_out[r * W] = NaN;
                s_out[r * W + 1] = NaN;
                s_fl[r * W] = 0;
                s_fl[r * W + 1] = 0;
            }
        } else {
            const int lane = tid & 63;
            for (int r = tid >> 6; r < R; r += NTHR / 64) {
                const float *__restrict__ row = tab + (size_t)r * L;
                double wi = 0.0, si = 0.0, wr = 0.0, sr = 0.0;
                uint32_t nk = 0, ncommon = 0;
                for (int k = lane; k < K; k += 64) {
                    const float medf = row[k];
                    if (!(medf >= 0.0f)) continue;
                    const double med = (double)medf;
                    const double w = (double)row[2 * KS + k];
                    nk++;
                    si += ((double)row[KS + k] / med) * w;
                    wi += w;
                    const float mm = s_min[k];
                    if (mm == mm) {
                        ncommon++;
                        sr += ((double)mm / med) * w;
                        wr += w;
                    }
                }
                wi = wave_sum_f64(wi);
                si = wave_sum_f64(si);
                wr = wave_sum_f64(wr);
                sr = wave_sum_f64(sr);
                nk = wave_sum_u32(nk);
                ncommon = wave_sum_u32(ncommon);
                if (lane == 0) {
                    const float gi = (a.do_indiv && nk > 0) ? (float)(si / wi) : NaN;
                    const float gr = (a.do_rel && ncommon > 0) ? (float)(sr / wr) : NaN;
                    s_out[r * W] = gi;
                    s_out[r * W + 1] = gr;
                    s_fl[r * W] = ((double)gi < a.thr[2]) ? 1 : 0;
                    s_fl[r * W + 1] = ((double)gr < a.thr[0]) ? 1 : 0;
                }
            }
        }
        sync();
    };
    if (s_tab && pa.windows)
        score_all(s_tab);
    else if (ga.g && s_send && !pa.windows)
        score_all(s_send);
    else
        score_all(a.table);

    // staged results -> result block, 16 bytes per lane
    const unsigned long long t_staged = wall_clock64();
    {
        const nvrx_f4 *src = reinterpret_cast<const nvrx_f4 *>(s_out);
        nvrx_f4 *dst = reinterpret_cast<nvrx_f4 *>(a.scores);
        for (int i = tid; i < nout4 / 4; i += NTHR) store16_sys(dst + i, src[i]);
        if (a.flags) {
            const nvrx_f4 *fsrc = reinterpret_cast<const nvrx_f4 *>(s_fl);
            nvrx_f4 *fdst = reinterpret_cast<nvrx_f4 *>(a.flags);
            for (int i = tid; i < (nout + 15) / 16; i += NTHR) store16_sys(fdst + i, fsrc[i]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores (asm and plain) have been accepted
    __syncthreads();
    const bool publish = a.meta && a.done_counter;
    if (tid == 0 && publish) {
        if (fence) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(ga.g && ga.stats_out))  // statistics already forwarded above: both words at once
            __hip_atomic_store(&a.meta[5], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // diagnostics (constant-rate wall clock, 10 ns ticks): how long this kernel waited for the rows, and how long
        // it took from the last row to this store
        __hip_atomic_store(&a.meta[6], (uint32_t)(t_rows - t_begin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (low half: last row -> this store; high half: last row -> scores staged in LDS, i.e. before the host stores)
        __hip_atomic_store(&a.meta[7], (uint32_t)min(wall_clock64() - t_rows, 0xFFFFull) | ((uint32_t)min(t_staged - t_rows, 0xFFFFull) << 16),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&a.meta[4], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (ga.g && ga.stats_out) {
        // the statistics rows travel as granules too and are forwarded AFTER the scores were published: meta[5] is
        // their own completion word (the host reads statistics lazily)
        for (int idx = tid; idx < ga.n_blocks * 2; idx += NTHR) {
            const int b = idx >> 1, half = idx & 1;
            const unsigned long long *p = ga.g + (size_t)b * ROW_GRANULES + 3 + 4 * half;
            nvrx_f4 v;
            v.x = wait_granule(p + 0, ga, t_begin);
            v.y = wait_granule(p + 1, ga, t_begin);
            v.z = wait_granule(p + 2, ga, t_begin);
            v.w = wait_granule(p + 3, ga, t_begin);
            const int lr = b / ga.rows_active;
            const int row = lr * ga.rows_per_rank + (b - lr * ga.rows_active);
            store16_sys(ga.stats_out + (size_t)row * 2 + half, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0 && publish) __hip_atomic_store(&a.meta[5], a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// launch-shape selection for k_row_stats
// ------------------------------------------------------------------------------------------------
using StatsKernel = void (*)(const float *, const uint32_t *, const uint8_t *, int, float *, Epilogue, int);

struct StatsVariant {
    int threads;
    int vpt;
    StatsKernel fn;
};

#define SV(T, V) \
    { T, V, k_row_stats<T, V> }
const StatsVariant kVariants[] = {
    SV(256, 1),  SV(256, 2),  SV(256, 3),  SV(256, 4),  SV(256, 5),   SV(256, 6),  SV(256, 8),  SV(256, 10),
    SV(256, 12), SV(256, 16), SV(512, 1),  SV(512, 2),  SV(512, 3),   SV(512, 4),  SV(512, 5),  SV(512, 6),
    SV(512, 8),  SV(512, 10), SV(512, 12), SV(512, 16), SV(1024, 1),  SV(1024, 2), SV(1024, 3), SV(1024, 4),
    SV(1024, 5), SV(1024, 6), SV(1024, 8), SV(1024, 10), SV(1024, 12), SV(1024, 16),
};
#undef SV

// Smallest variant of the preferred width that holds a whole row in registers.
const StatsVariant *pick_variant(int row_stride) {
    // (the width A/B of rounds 2-3 is answered -- 512 threads, 256 for rows of at most 2048 samples -- and its switch is gone)
    int prefs[3] = {512, 1024, 256};
    if (row_stride <= 256 * 4 * 2) {
        prefs[0] = 256;
        prefs[1] = 512;
        prefs[2] = 1024;
    }
    for (int p = 0; p < 3; p++) {
        const StatsVariant *best = nullptr;
        for (const StatsVariant &v : kVariants) {
            if (v.threads != prefs[p]) continue;
            if (v.threads * v.vpt * 4 >= row_stride && (!best || v.vpt < best->vpt)) best = &v;
        }
        if (best) return best;
    }
    return nullptr;
}

// start/stop (optional): events that receive the kernel's own begin/end timestamps
// (hipExtLaunchKernel: the dispatch's profiling timestamps, the same clock rocprofv3 reports).
int launch_row_stats(const float *d_samples, const uint32_t *d_counts, const uint8_t *d_kinds, int rows,
                     int row_stride, float *d_stats, const Epilogue &ep, hipStream_t stream,
                     hipEvent_t start = nullptr, hipEvent_t stop = nullptr, int uniform_n = -1) {
    if (rows == 0) return NVRX_OK;
    const StatsVariant *v = pick_variant(row_stride);
    if (!v)
        return fail(NVRX_ERR_RANGE, "row_stride %d exceeds the register-resident limit %d", row_stride,
                    NVRX_MAX_RING_CAP);
    if (start || stop) {
        hipExtLaunchKernelGGL(v->fn, dim3(rows), dim3(v->threads), 0, stream, start, stop, 0, d_samples, d_counts,
                              d_kinds, row_stride, d_stats, ep, uniform_n);
    } else {
        hipLaunchKernelGGL(v->fn, dim3(rows), dim3(v->threads), 0, stream, d_samples, d_counts, d_kinds, row_stride,
                           d_stats, ep, uniform_n);
    }
    HIP_TRY(hipGetLastError());
    return NVRX_OK;
}

// The single-workgroup score kernel serves every shape that fits it (the A/B switch of round 2 is gone: answered).
constexpr int score_single_wg_enabled() { return 1; }
// The fence-free publication of k_score1 (write-through stores drained per wave, a barrier, then the sequence word)
// leans on how gfx942 / gfx950 map pinned host memory (uncached, posted PCIe writes of one requester kept in order),
// not on the HIP memory model: it is used on exactly those two architectures, every other device gets the
// system-scope release fence in front of the completion word.  NVRX_DEBUG_SCORE_FENCE=0|1 overrides the detection.
int score_fence_enabled() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("NVRX_DEBUG_SCORE_FENCE");
        if (e && *e) {
            v = atoi(e) != 0 ? 1 : 0;
        } else {
            int dev = 0;
            hipDeviceProp_t prop;
            v = 1;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                (strncmp(prop.gcnArchName, "gfx950", 6) == 0 || strncmp(prop.gcnArchName, "gfx942", 6) == 0))
                v = 0;
        }
    }
    return v;
}

// NVRX_DEBUG_RESIDENT_SCORER: 0 = the score kernel always behind the statistics kernel on the report's stream, 2 = resident
// whenever the shape allows it (A/B measurements), 1 / unset = the library decides (see nvrx_report).  Read once per
// context (nvrx_ctx_create): a report does not call getenv.
int resident_scorer_mode_from_env() {
    const char *e = getenv("NVRX_DEBUG_RESIDENT_SCORER");
    const int v = e ? atoi(e) : 1;
    return (v < 0 || v > 2) ? 1 : v;
}
constexpr int poll_naps_from_env() {
    return 4;  // 4 x 64 clk between passes: same latency as 1, less issue pressure on the CU's rows (round-3 A/B, switch gone)
}

// The peer-window exchange runs as the score kernel's prologue (its own kernel was the slower side of round 3's A/B).
constexpr int peer_prologue_enabled() { return 1; }

// scratch for the multi-kernel scoring path (large R or K+S): column minima and their per-chunk partials.  One buffer
// per launch stream -- launches on one stream are ordered, so a buffer is reused safely there, while two contexts
// scoring on their own streams at the same time must not share one.
struct ScoreScratch {
    float *ptr = nullptr;
    size_t elems = 0;
};
std::mutex g_scratch_mu;
std::unordered_map<void *, ScoreScratch> g_scratch_by_stream;

}  // namespace

// ================================================================================================
// context
// ================================================================================================
struct EventPair {
    hipEvent_t start = nullptr;
    hipEvent_t end = nullptr;
};

struct StageBuf {
    uint32_t *h_counts = nullptr;     // [rows]
    StagedSample *h_entries = nullptr;  // [stage_cap]
    // released by the scatter's completion ticket (k_scatter): *h_done == ticket once the launch that read this buffer is
    // through -- no event is recorded for a flush, whatever stream it ran on
    volatile uint32_t *h_done = nullptr;  // pinned word
    uint32_t *d_done = nullptr;           // ... as the kernel addresses it
    uint32_t *d_cnt = nullptr;            // device word: workgroups of the launch that have finished
    uint32_t ticket = 0;
    bool in_flight = false;
};

struct nvrx_ctx {
    int device = 0;
    int local_ranks = 1;
    int rows_per_rank = 0;
    int rows = 0;
    int ring_cap = 0;
    int row_stride = 0;
    int stage_cap = 0;

    float *d_samples = nullptr;
    uint32_t *d_counts = nullptr;
    uint8_t *d_kinds = nullptr;
    int32_t *d_gid = nullptr;
    float *d_hist_min = nullptr;

    std::vector<uint64_t> total;  // samples ever pushed per row since the last reset
    int rows_hi = 0;  // rows [rows_hi, rows) have never held a sample or been configured: a flush uploads counts / metadata below it only
    std::vector<uint8_t> occupied_seen;  // nvrx_ring_occupancy_changed: which rows held samples at the previous call
    int occupied_rows = -1;              // ... over how many rows (-1: never asked)
    int rows_used = 0;            // rows of a logical rank handed out so far (nvrx_row_alloc)
    uint8_t *h_kinds = nullptr;   // pinned
    int32_t *h_gid = nullptr;     // pinned
    bool meta_dirty = true;
    bool counts_dirty = true;

    static constexpr int NBUF = 4;
    StageBuf buf[NBUF];
    int cur = 0;
    int n_staged = 0;
    // One scatter launch has no order inside it, so two staged samples must never aim at the same (row, slot): a row
    // may hold at most ring_cap samples of the current staging buffer.  Counted per row, valid for the flush generation
    // it was written in (nothing is cleared at a flush) -- so the buffer is 4096 deep whatever the ring capacity, and a
    // tracer that feeds thousands of short rings launches one scatter per 4096 samples, not one per ring_cap.
    std::vector<uint32_t> stage_cnt, stage_gen;
    uint32_t flush_gen = 1;

    hipStream_t default_stream = nullptr;

    // hipEvent timing of code regions
    std::vector<EventPair> pool;
    std::vector<int> free_pairs;
    struct Open {
        int row;
        int pair;
    };
    std::vector<Open> open;
    std::deque<Open> pending;

    // benchmark instrumentation
    bool timing = false;
    std::vector<EventPair> timing_pairs;
    std::vector<int> timing_free;
    std::vector<int> timing_used;
    double timing_total_us = 0.0;
    int timing_launches = 0;

    hipEvent_t copy_done = nullptr;
    bool copy_pending = false;

    // bulk appends (nvrx_ring_push_pairs): a pinned entry buffer of its own, grown on demand
    StagedSample *h_bulk = nullptr;
    size_t bulk_cap = 0;
    volatile uint32_t *bulk_word = nullptr;  // the ticket word / value that releases h_bulk (see StageBuf)
    uint32_t bulk_ticket = 0;
    bool bulk_in_flight = false;
    uint32_t *h_stage_done = nullptr, *d_stage_done = nullptr;  // pinned [NBUF] ticket words (host / device address)
    uint32_t *d_stage_cnt = nullptr;                            // device [NBUF] arrival counters of the scatters
    std::vector<uint32_t> bulk_cnt, bulk_seen;

    // device-side region timing (k_stamp_begin / k_stamp_end)
    static constexpr int NSTAMP = NVRX_NSTAMP;
    unsigned long long *d_stamps = nullptr;  // [NSTAMP] begin timestamps, handed out round-robin: the device's g_stamp_slots
                                             // (argument-free begin kernels, the default) or an allocation of this context
    bool stamps_argfree = true;              // (false: the one-argument begin kernel -- kept for a context that cannot take the device's slots)
    int stamp_next = 0;
    float us_per_tick = 0.01f;
    struct OpenStamp {
        int row;
        int slot;
        bool own;  // the slot is one of this context's own (d_stamps_own), not one of the device's argument-free ones
    };
    unsigned long long *d_stamps_own = nullptr;  // [NSTAMP], allocated the first time every argument-free slot is open elsewhere
    int stamp_next_own = 0;
    std::vector<OpenStamp> open_stamps;
    std::vector<hipStream_t> stamp_streams;  // user streams with stamp kernels the rings have not been ordered after
    std::vector<int> skipped_regions;        // rows of regions opened on a capturing stream: closed without a sample
    unsigned regions_skipped = 0;
    std::vector<hipStream_t> report_streams;  // every stream a report of this context launched its score kernel on (scratch ownership)
    hipEvent_t stamp_ev = nullptr;
    hipEvent_t order_ev = nullptr;  // nvrx_report: report stream ordered after the caller's stream
    // asynchronous reports: the statistics kernel of a report the host did not wait for may still be reading the
    // rings; device-side writers on other streams (k_stamp_end) are ordered after it with this event
    // resident-scorer reports: the score kernel's own stream, the rows' granules (two parities) and their epoch
    hipStream_t score_stream = nullptr;
    unsigned long long *d_rowg = nullptr;
    uint32_t *h_gather_err = nullptr, *d_gather_err = nullptr;  // pinned: epoch of a granule wait that gave up
    uint32_t gran_epoch = 0;
    int wall_khz = 100000;
    int resident_mode = 1;  // NVRX_DEBUG_RESIDENT_SCORER as it stood when the context was created
    int poll_naps = 4;
    int rehome_mode = 1;    // NVRX_DEBUG_REPORT_REHOME: synchronous reports may run ON the one stream they must follow
    // work of ours that may still be running on a stream a re-homed report would not be ordered after (a staging flush
    // forced by a full buffer, device-side appends, bulk appends, history resets, asynchronous reports); cleared when
    // a synchronous report on the context's own stream has completed
    bool side_work = true;
    // Side work is counted.  An asynchronous report on the context's own stream remembers the count it was enqueued behind;
    // when the caller tells the next report that it has SEEN that report complete (desc.prev_settled), everything up to
    // that count is known to be done -- the way a job that only ever reports asynchronously gets to re-home.
    uint64_t reports_rehomed = 0;
    int last_rehome_verdict = 0;  // 1 re-homed; 0 not a candidate (mode / cadence); -1 streams; -2 side work; -3 async in flight; -4 timing
    uint64_t side_gen = 1, async_gen = 0;
    bool async_on_ctx = false;            // an asynchronous report may still be running on the context's own stream
    double last_report_us = 0.0;          // monotonic clock of the previous nvrx_report's entry
    double async_rehome_gap_us = 5000.0;  // NVRX_DEBUG_ASYNC_REHOME_GAP_US: asynchronous reports at least this far apart are re-homed (0 = never)
    hipEvent_t report_ev = nullptr;
    uint64_t report_epoch = 0;  // bumped by every guarded report
    // the stream the last guarded (asynchronous) report's statistics kernel runs on; whether report_ev has been recorded
    // behind it yet (a re-homed report records it only if a ring writer on ANOTHER stream turns up); and the epoch up to
    // which the caller has seen the guarded reports complete (nothing has to wait for those any more)
    hipStream_t guard_stream = nullptr;
    bool guard_recorded = true;
    bool stamps_used = false;  // a stamp kernel has written this context's rings (region timing): see nvrx_report, guard_rings
    uint64_t guard_done_epoch = 0;
    struct StreamEpoch {
        hipStream_t stream;
        uint64_t epoch;
    };
    std::vector<StreamEpoch> stream_epochs;

    std::mutex mu;
};

static bool stream_is_capturing(hipStream_t st);
static void stamp_slot_give(int slot);

namespace {

int ctx_set_device(const nvrx_ctx *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return NVRX_OK;
}

inline void mark_side_work(nvrx_ctx *ctx) {
    ctx->side_work = true;
    ctx->side_gen++;
}

// A device-side ring writer on stream `st` (a stamp kernel, a scatter, a device append) must not overtake the statistics
// kernel of an asynchronous report that may still be reading the rings.  Nothing to do when the caller has seen that
// report complete, or when the writer is on the very stream the report runs on (stream order); else the writer's stream
// waits, on the device, for report_ev -- which a re-homed report records only now, behind whatever its stream has
// been given since (conservative, and rare: it takes a writer on a second stream).
int guard_ring_writer(nvrx_ctx *ctx, hipStream_t st) {
    if (!ctx->report_epoch) return NVRX_OK;
    nvrx_ctx::StreamEpoch *se = nullptr;
    for (auto &e : ctx->stream_epochs)
        if (e.stream == st) se = &e;
    if (!se) {
        ctx->stream_epochs.push_back({st, 0});
        se = &ctx->stream_epochs.back();
    }
    if (se->epoch < ctx->report_epoch) {
        if (ctx->guard_done_epoch < ctx->report_epoch && st != ctx->guard_stream) {
            if (!ctx->guard_recorded) {
                if (hipEventRecord(ctx->report_ev, ctx->guard_stream) != hipSuccess) {
                    // the stream the report was enqueued on is gone (a user stream destroyed since): wait for the device
                    // once instead -- whatever was enqueued on that stream is then done
                    (void)hipGetLastError();
                    HIP_TRY(hipDeviceSynchronize());
                    ctx->guard_done_epoch = ctx->report_epoch;
                    se->epoch = ctx->report_epoch;
                    return NVRX_OK;
                }
                ctx->guard_recorded = true;
            }
            HIP_TRY(hipStreamWaitEvent(st, ctx->report_ev, 0));
        }
        se->epoch = ctx->report_epoch;
    }
    return NVRX_OK;
}

// Samples written by stamp kernels on user streams must be in the rings before anything on `stream`
// reads them: one event per such stream, waited for on the device (the host does not block).
// `also` (optional): a second stream to order the same way -- the resident scorer's.  A score kernel that became
// resident while the report still waits for the user's work would hold a CU for all that time (measured: a report
// every step behind 10 x matmul(4096^2) cost +560 us per step, the matmuls' workgroups no longer fit one per CU).
int order_after_stamps(nvrx_ctx *ctx, hipStream_t stream, hipStream_t also = nullptr) {
    for (hipStream_t s : ctx->stamp_streams) {
        if (s == stream) continue;
        HIP_TRY(hipEventRecord(ctx->stamp_ev, s));
        HIP_TRY(hipStreamWaitEvent(stream, ctx->stamp_ev, 0));
        if (also) HIP_TRY(hipStreamWaitEvent(also, ctx->stamp_ev, 0));
    }
    ctx->stamp_streams.clear();
    return NVRX_OK;
}

// `uniform_n` (optional, report path only): when nothing but the counts changed and every one of the
// `rows_active` rows per rank about to be launched holds the same number of samples, that number is
// returned through it and NOTHING is launched -- k_row_stats takes it as a kernel argument instead of
// reading d_counts (which stays marked dirty until a later flush uploads it).
// A staging buffer is free again when the scatter that read it has stored its ticket (k_scatter): usually long ago; else
// the host spins on the pinned word (cold paths only: new row metadata, or a pusher a whole rotation ahead of the GPU).
// The wait is bounded (NVRX_DEBUG_STAGE_WAIT_S, default 30 s: the scatter is a microsecond kernel on a stream of our own) and
// looks at the device between slices: a scatter that never runs (device fault, a destroyed stream) becomes a HIP error or
// a timeout after seconds, not a host thread spinning for half an hour with the context locked.
double stage_wait_s() {
    static const double v = [] {
        const char *e = getenv("NVRX_DEBUG_STAGE_WAIT_S");
        const double d = e ? atof(e) : 0.0;
        return d > 0.0 ? d : 30.0;
    }();
    return v;
}

int wait_stage_buffer(StageBuf &b) {
    if (!b.in_flight) return NVRX_OK;
    const double total = stage_wait_s();
    double waited = 0.0;
    int rc = NVRX_ERR_TIMEOUT;
    while (waited < total) {
        const double slice = std::min(0.25, total - waited);
        rc = nvrx_poll_u32(const_cast<const uint32_t *>(b.h_done), b.ticket, slice);
        if (rc != NVRX_ERR_TIMEOUT) break;
        waited += slice;
        const hipError_t e = hipGetLastError();  // a faulted device says so here
        if (e != hipSuccess) {
            b.in_flight = false;  // (nothing will ever read the buffer again)
            return fail(NVRX_ERR_HIP, "staging flush did not complete: %s", hipGetErrorString(e));
        }
    }
    if (rc == NVRX_ERR_TIMEOUT) {
        // give the buffer up for lost rather than relaunch on a half-counted ticket: its samples are gone with it
        b.in_flight = false;
        return fail(NVRX_ERR_TIMEOUT, "staging flush (ticket %u) did not complete within %.0f s", b.ticket, total);
    }
    if (rc) return rc;
    b.in_flight = false;
    return NVRX_OK;
}

// Every writer of total[] / h_kinds[] / h_gid[] says which row it touched: the per-kernel rings are 4096 rows of which a
// job uses tens, and a flush that fills and uploads 4096 counts per report was a third of the report's C call at cadence
// (profiles/r06a_kernels_mode_breakdown.txt: 14-21 us "staged samples flushed").
inline void touch_row(nvrx_ctx *ctx, int row) {
    if (row >= ctx->rows_hi) ctx->rows_hi = row + 1;
}

int flush_locked(nvrx_ctx *ctx, hipStream_t stream, int *uniform_n = nullptr, int rows_active = 0) {
    if (uniform_n) *uniform_n = -1;
    if (!ctx->stamp_streams.empty()) {
        int rc = order_after_stamps(ctx, stream);
        if (rc) return rc;
    }
    if (ctx->n_staged == 0 && !ctx->meta_dirty && !ctx->counts_dirty) return NVRX_OK;
    // a scatter captured into a hipGraph would never store its ticket (and would replay stale staging entries): refuse
    if (stream_is_capturing(stream))
        return fail(NVRX_ERR_STATE, "staged samples cannot be flushed on a stream that is being captured into a hipGraph");
    if (uniform_n && ctx->n_staged == 0 && !ctx->meta_dirty) {
        const uint64_t cap = (uint64_t)ctx->ring_cap;
        const uint64_t first = std::min<uint64_t>(ctx->total[0], cap);
        bool same = true;
        for (int lr = 0; lr < ctx->local_ranks && same; lr++)
            for (int r = 0; r < rows_active; r++)
                if (std::min<uint64_t>(ctx->total[(size_t)lr * ctx->rows_per_rank + r], cap) != first) {
                    same = false;
                    break;
                }
        if (same) {
            *uniform_n = (int)first;
            return NVRX_OK;
        }
    }
    {
        int grc = guard_ring_writer(ctx, stream);  // the scatter writes ring slots
        if (grc) return grc;
    }
    StageBuf &b = ctx->buf[ctx->cur];
    const int hi = ctx->rows_hi;  // (d_counts / d_kinds / d_gid above it still hold their initial values)
    for (int r = 0; r < hi; r++)
        b.h_counts[r] = (uint32_t)std::min<uint64_t>(ctx->total[r], (uint64_t)ctx->ring_cap);
    const int work = std::max(std::max(ctx->n_staged, hi), 1);  // (nothing touched yet, e.g. a reset of fresh rings: the ticket is still stored)
    const int threads = 256;
    const int blocks = (work + threads - 1) / threads;
    b.ticket = (b.ticket % 0x7FFFFFFFu) + 1u;
    hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(threads), 0, stream, b.h_counts, b.h_entries, ctx->n_staged,
                       ctx->d_samples, ctx->row_stride, ctx->d_counts, hi,
                       ctx->meta_dirty ? ctx->h_kinds : nullptr, ctx->d_kinds,
                       ctx->meta_dirty ? ctx->h_gid : nullptr, ctx->d_gid, b.d_cnt, b.d_done, b.ticket);
    HIP_TRY(hipGetLastError());
    b.in_flight = true;
    if (ctx->meta_dirty) {
        // h_kinds/h_gid are read by the kernel just launched: do not let the host modify them until
        // it has run.  Metadata changes are cold (new names only), so a blocking wait is fine.
        int wrc = wait_stage_buffer(b);
        if (wrc) return wrc;
    }
    ctx->meta_dirty = false;
    ctx->counts_dirty = false;
    ctx->n_staged = 0;
    ctx->flush_gen++;
    ctx->cur = (ctx->cur + 1) % nvrx_ctx::NBUF;
    return wait_stage_buffer(ctx->buf[ctx->cur]);  // (a pusher a whole rotation ahead of the GPU waits here)
}

inline int push_locked(nvrx_ctx *ctx, int row, float value) {
    if (ctx->stage_gen[(size_t)row] != ctx->flush_gen) {
        ctx->stage_gen[(size_t)row] = ctx->flush_gen;
        ctx->stage_cnt[(size_t)row] = 0;
    }
    if (ctx->n_staged == ctx->stage_cap || ctx->stage_cnt[(size_t)row] >= (uint32_t)ctx->ring_cap) {
        int rc = flush_locked(ctx, ctx->default_stream);
        if (rc) return rc;
        mark_side_work(ctx);
        ctx->stage_gen[(size_t)row] = ctx->flush_gen;
        ctx->stage_cnt[(size_t)row] = 0;
    }
    ctx->stage_cnt[(size_t)row]++;
    touch_row(ctx, row);
    const uint32_t slot = (uint32_t)(ctx->total[row] % (uint64_t)ctx->ring_cap);
    StagedSample &e = ctx->buf[ctx->cur].h_entries[ctx->n_staged++];
    e.row_slot = ((uint32_t)row << 16) | slot;
    e.value = value;
    ctx->total[row]++;
    ctx->counts_dirty = true;
    return NVRX_OK;
}

int get_pair(std::vector<EventPair> &pool, std::vector<int> &free_list, int *out) {
    if (free_list.empty()) {
        EventPair p;
        HIP_TRY(hipEventCreate(&p.start));
        HIP_TRY(hipEventCreate(&p.end));
        pool.push_back(p);
        free_list.push_back((int)pool.size() - 1);
    }
    *out = free_list.back();
    free_list.pop_back();
    return NVRX_OK;
}

}  // namespace

extern "C" {

int nvrx_abi_version(void) { return NVRX_ABI_VERSION; }

int nvrx_report_desc_size(void) { return (int)sizeof(nvrx_report_desc); }

const char *nvrx_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------
// stateless operators
// ------------------------------------------------------------------------------------------------
int nvrx_row_stats(const float *d_samples, const uint32_t *d_counts, const uint8_t *d_kinds, int rows,
                   int row_stride, float *d_stats, void *stream) {
    if (rows < 0 || row_stride <= 0 || (row_stride & 3)) return fail(NVRX_ERR_INVALID, "row_stride must be a positive multiple of 4 (got %d), rows >= 0", row_stride);
    if (rows > 0 && (!d_samples || !d_counts || !d_stats)) return fail(NVRX_ERR_INVALID, "null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_samples) & 15u) != 0) return fail(NVRX_ERR_INVALID, "d_samples must be 16-byte aligned");
    Epilogue ep{};
    return launch_row_stats(d_samples, d_counts, d_kinds, rows, row_stride, d_stats, ep, as_stream(stream));
}

// `pa` (optional): the peer-window exchange to run as the score kernel's prologue.  Only honoured by the
// single-workgroup kernel; *pa_used tells the caller whether it was (else the caller enqueues the exchange itself).
static bool score_fits_single_wg(int R, int K, int S, const float *d_scores, const uint8_t *d_flags) {
    return R <= SCORE1_MAX_RANKS && score1_lds_bytes(R, K, S) <= SCORE1_MAX_LDS &&
           (reinterpret_cast<uintptr_t>(d_scores) & 15u) == 0 && (reinterpret_cast<uintptr_t>(d_flags) & 15u) == 0 &&
           score_single_wg_enabled();
}

static int score_launch(const float *d_table, int R, int K, int S, int do_indiv, int do_rel, const double *thresholds,
                        float *d_scores, uint8_t *d_flags, uint32_t *d_meta, uint32_t *d_done_counter, uint32_t seq,
                        const float *d_stats_src, float *d_stats_dst, int stats_rows, void *stream, const PeerArgs *pa,
                        const GatherArgs *ga = nullptr) {
    if (R <= 0 || K < 0 || S < 0) return fail(NVRX_ERR_INVALID, "bad table shape R=%d K=%d S=%d", R, K, S);
    if (!d_table || !d_scores) return fail(NVRX_ERR_INVALID, "null device pointer");
    hipStream_t st = as_stream(stream);
    ScoreArgs a{};
    a.table = d_table;
    a.R = R;
    a.K = K;
    a.S = S;
    a.do_indiv = do_indiv;
    a.do_rel = do_rel;
    for (int i = 0; i < 4; i++) a.thr[i] = thresholds ? thresholds[i] : 0.75;
    a.scores = d_scores;
    a.flags = d_flags;
    a.meta = d_meta;
    a.done_counter = d_meta ? d_done_counter : nullptr;
    a.seq = seq;
    if (d_stats_src && d_stats_dst && stats_rows > 0) {
        a.stats_src = reinterpret_cast<const float4 *>(d_stats_src);
        a.stats_dst = reinterpret_cast<float4 *>(d_stats_dst);
        a.stats_n4 = stats_rows * (NVRX_STATS_STRIDE / 4);
    }
    const int KS = K + S;
    // One workgroup does it all when the staged results fit in LDS.  The result arrays are then written in 16-byte
    // units: the caller's buffers must be 16-byte aligned and padded to a multiple of 16 bytes (the workspace is).
    if (score_fits_single_wg(R, K, S, d_scores, d_flags)) {
        PeerArgs none{};
        GatherArgs nog{};
        size_t lds1 = score1_lds_bytes(R, K, S);
        if (ga || pa) {
            const int send_floats = pa ? pa->count : R * NVRX_TABLE_LEN(K, S);
            const size_t extra = score1_table_lds_bytes(R, K, S, send_floats);
            if (lds1 + extra <= SCORE1_MAX_LDS) {
                a.tab_in_lds = 1;
                a.send_floats = send_floats;
                lds1 += extra;
            }
        }
        if (ga) {
            a.stats_n4 = 0;  // statistics come through the granules
            hipLaunchKernelGGL(k_score1<SCORE1_RESIDENT_THREADS>, dim3(1), dim3(SCORE1_RESIDENT_THREADS), lds1, st, a,
                               score_fence_enabled(), pa ? *pa : none, *ga);
        } else {
            report_clk(7);  // (diagnostics: argument set-up ends / the launch call begins)
            hipLaunchKernelGGL(k_score1<SCORE1_THREADS>, dim3(1), dim3(SCORE1_THREADS), lds1, st, a, score_fence_enabled(),
                               pa ? *pa : none, nog);
        }
        HIP_TRY(hipGetLastError());
        return NVRX_OK;
    }
    if (ga) return fail(NVRX_ERR_STATE, "the resident scorer needs the single-workgroup score kernel");
    if (pa) return fail(NVRX_ERR_STATE, "the exchange prologue needs the single-workgroup score kernel");
    size_t lds = (size_t)KS * sizeof(float);
    int tile_ranks = 0;
    if (R > 64 || lds > 48 * 1024) {
        // large jobs: column minima in their own two-level pass over a coalesced grid
        const int chunks = std::max(1, std::min(COLMIN_MAX_CHUNKS, (R + 63) / 64));
        const int rows_per_chunk = (R + chunks - 1) / chunks;
        const size_t need = (size_t)KS * (size_t)(chunks + 1);
        float *scratch = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_scratch_mu);
            ScoreScratch &sc = g_scratch_by_stream[stream];
            if (sc.elems < need) {
                if (sc.ptr) HIP_TRY(hipFree(sc.ptr));  // (waits for the device: nothing still reads the old buffer)
                sc = ScoreScratch{};
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&sc.ptr), std::max<size_t>(need, 1024) * sizeof(float)));
                sc.elems = std::max<size_t>(need, 1024);
            }
            scratch = sc.ptr;
        }
        if (KS > 0) {
            float *part = scratch + KS;
            hipLaunchKernelGGL(k_colmin_part, dim3((KS + 63) / 64, chunks), dim3(256), 0, st, d_table, R, KS,
                               NVRX_TABLE_LEN(K, S), rows_per_chunk, part);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_colmin_finish, dim3((KS + 255) / 256), dim3(256), 0, st, part, chunks, KS, scratch);
            HIP_TRY(hipGetLastError());
        }
        a.minmed_pre = scratch;
        lds = 0;
        // a tile of 16 (or 8) ranks per workgroup when its staged results fit in LDS and the result arrays can be
        // written in 16-byte units (every tile then starts on a 16-byte boundary of both arrays: the score row is an
        // even number of floats / bytes)
        const bool aligned = ((reinterpret_cast<uintptr_t>(d_scores) | reinterpret_cast<uintptr_t>(d_flags)) & 15u) == 0;
        if (aligned && R > 64)
            for (int t : {16, 8})
                if (!tile_ranks && score_tile_lds_bytes(t, S) <= TILE_MAX_LDS) tile_ranks = t;
    }
    if (tile_ranks) {
        hipLaunchKernelGGL(k_score_tile, dim3((R + tile_ranks - 1) / tile_ranks), dim3(TILE_THREADS),
                           score_tile_lds_bytes(tile_ranks, S), st, a, tile_ranks);
    } else {
        hipLaunchKernelGGL(k_score, dim3(R), dim3(SCORE_THREADS), lds, st, a);
    }
    HIP_TRY(hipGetLastError());
    return NVRX_OK;
}

int nvrx_score(const float *d_table, int R, int K, int S, int do_indiv, int do_rel, const double *thresholds,
               float *d_scores, uint8_t *d_flags, uint32_t *d_meta, uint32_t *d_done_counter, uint32_t seq,
               const float *d_stats_src, float *d_stats_dst, int stats_rows, void *stream) {
    return score_launch(d_table, R, K, S, do_indiv, do_rel, thresholds, d_scores, d_flags, d_meta, d_done_counter, seq,
                        d_stats_src, d_stats_dst, stats_rows, stream, nullptr);
}

int nvrx_send_init(float *d_send, int rows, int K, int S, void *stream) {
    if (!d_send || rows <= 0 || K < 0 || S < 0) return fail(NVRX_ERR_INVALID, "bad arguments");
    const size_t n = (size_t)rows * NVRX_TABLE_LEN(K, S);
    hipLaunchKernelGGL(k_send_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), d_send, rows, K, S);
    HIP_TRY(hipGetLastError());
    return NVRX_OK;
}

// ------------------------------------------------------------------------------------------------
// context lifecycle
// ------------------------------------------------------------------------------------------------
int nvrx_ctx_create(int device, int local_ranks, int rows_per_rank, int ring_cap, nvrx_ctx **out) {
    if (!out) return fail(NVRX_ERR_INVALID, "out is null");
    *out = nullptr;
    if (local_ranks <= 0 || rows_per_rank <= 0 || ring_cap <= 0) return fail(NVRX_ERR_INVALID, "sizes must be positive");
    if (ring_cap > NVRX_MAX_RING_CAP) return fail(NVRX_ERR_RANGE, "ring_cap %d > %d", ring_cap, NVRX_MAX_RING_CAP);
    if ((long long)local_ranks * rows_per_rank > NVRX_MAX_ROWS) return fail(NVRX_ERR_RANGE, "too many rows");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(NVRX_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    nvrx_ctx *ctx = new (std::nothrow) nvrx_ctx();
    if (!ctx) return fail(NVRX_ERR_NOMEM, "host allocation failed");
    ctx->device = device;
    ctx->local_ranks = local_ranks;
    ctx->rows_per_rank = rows_per_rank;
    ctx->rows = local_ranks * rows_per_rank;
    ctx->ring_cap = ring_cap;
    ctx->row_stride = (ring_cap + 3) & ~3;
    ctx->stage_cap = 4096;
    ctx->total.assign((size_t)ctx->rows, 0);
    ctx->stage_cnt.assign((size_t)ctx->rows, 0);
    ctx->stage_gen.assign((size_t)ctx->rows, 0);
    ctx->resident_mode = resident_scorer_mode_from_env();
    ctx->poll_naps = poll_naps_from_env();
    {
        const char *e = getenv("NVRX_DEBUG_REPORT_REHOME");
        ctx->rehome_mode = (e && atoi(e) == 0) ? 0 : 1;
        const char *g = getenv("NVRX_DEBUG_ASYNC_REHOME_GAP_US");
        if (g && *g) ctx->async_rehome_gap_us = atof(g);
        const char *eg = getenv("NVRX_DEBUG_EAGER_GUARD");  // (A/B: the guard event of every asynchronous report recorded at once)
        ctx->stamps_used = eg && atoi(eg) != 0;
    }

#define CTX_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            int rc_ = fail(NVRX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
            nvrx_ctx_destroy(ctx);                                                            \
            return rc_;                                                                       \
        }                                                                                     \
    } while (0)

    const size_t nsamp = (size_t)ctx->rows * (size_t)ctx->row_stride;
    CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_samples), nsamp * sizeof(float)));
    CTX_TRY(hipMemset(ctx->d_samples, 0, nsamp * sizeof(float)));
    CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_counts), (size_t)ctx->rows * sizeof(uint32_t)));
    CTX_TRY(hipMemset(ctx->d_counts, 0, (size_t)ctx->rows * sizeof(uint32_t)));
    CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_kinds), (size_t)ctx->rows));
    CTX_TRY(hipMemset(ctx->d_kinds, 0, (size_t)ctx->rows));
    CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_gid), (size_t)ctx->rows * sizeof(int32_t)));
    CTX_TRY(hipMemset(ctx->d_gid, 0xFF, (size_t)ctx->rows * sizeof(int32_t)));
    CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_hist_min), (size_t)ctx->rows * sizeof(float)));
    hipLaunchKernelGGL(k_fill_f32, dim3((ctx->rows + 255) / 256), dim3(256), 0, nullptr, ctx->d_hist_min, (size_t)ctx->rows, INFINITY);
    CTX_TRY(hipGetLastError());

    CTX_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_kinds), (size_t)ctx->rows, hipHostMallocDefault));
    CTX_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_gid), (size_t)ctx->rows * sizeof(int32_t), hipHostMallocDefault));
    memset(ctx->h_kinds, 0, (size_t)ctx->rows);
    for (int r = 0; r < ctx->rows; r++) ctx->h_gid[r] = -1;
    // completion tickets of the scatters: one pinned word (64 bytes apart) and one device counter per staging buffer
    CTX_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage_done), (size_t)nvrx_ctx::NBUF * 64, hipHostMallocMapped));
    memset(ctx->h_stage_done, 0, (size_t)nvrx_ctx::NBUF * 64);
    CTX_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->d_stage_done), ctx->h_stage_done, 0));
    CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_stage_cnt), (size_t)nvrx_ctx::NBUF * sizeof(uint32_t)));
    CTX_TRY(hipMemset(ctx->d_stage_cnt, 0, (size_t)nvrx_ctx::NBUF * sizeof(uint32_t)));
    {
        int bi = 0;
        for (StageBuf &b : ctx->buf) {
            CTX_TRY(hipHostMalloc(reinterpret_cast<void **>(&b.h_counts), (size_t)ctx->rows * sizeof(uint32_t), hipHostMallocDefault));
            CTX_TRY(hipHostMalloc(reinterpret_cast<void **>(&b.h_entries), (size_t)ctx->stage_cap * sizeof(StagedSample), hipHostMallocDefault));
            b.h_done = ctx->h_stage_done + (size_t)bi * 16;
            b.d_done = ctx->d_stage_done + (size_t)bi * 16;
            b.d_cnt = ctx->d_stage_cnt + bi;
            bi++;
        }
    }
    CTX_TRY(hipEventCreateWithFlags(&ctx->copy_done, hipEventDisableTiming));
    CTX_TRY(hipEventCreateWithFlags(&ctx->stamp_ev, hipEventDisableTiming));
    CTX_TRY(hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming));
    CTX_TRY(hipEventCreateWithFlags(&ctx->report_ev, hipEventDisableTiming));
    {   // (the resident scorer's own stream is created with the first resident report: a process that never runs one
        // should not hold another hardware queue)
        const size_t gbytes = 2ull * (size_t)ctx->rows * ROW_GRANULES * sizeof(unsigned long long);
        CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_rowg), gbytes));
        CTX_TRY(hipMemset(ctx->d_rowg, 0, gbytes));  // tag 0 is never a valid epoch
        CTX_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_gather_err), 64, hipHostMallocMapped));
        ctx->h_gather_err[0] = 0;
        CTX_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->d_gather_err), ctx->h_gather_err, 0));
    }
    ctx->stamps_argfree = true;  // (the one-argument begin kernel was the slower side of round 4's A/B; its switch is gone)
    if (ctx->stamps_argfree) {
        CTX_TRY(hipGetSymbolAddress(reinterpret_cast<void **>(&ctx->d_stamps), HIP_SYMBOL(g_stamp_slots)));
    } else {
        CTX_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_stamps), nvrx_ctx::NSTAMP * sizeof(unsigned long long)));
        CTX_TRY(hipMemset(ctx->d_stamps, 0, nvrx_ctx::NSTAMP * sizeof(unsigned long long)));
    }
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) {
            ctx->us_per_tick = 1000.0f / (float)khz;
            ctx->wall_khz = khz;
        }
    }
    CTX_TRY(hipDeviceSynchronize());
#undef CTX_TRY
    *out = ctx;
    return NVRX_OK;
}

int nvrx_ctx_destroy(nvrx_ctx *ctx) {
    if (!ctx) return NVRX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->d_samples) (void)hipFree(ctx->d_samples);
    if (ctx->d_counts) (void)hipFree(ctx->d_counts);
    if (ctx->d_kinds) (void)hipFree(ctx->d_kinds);
    if (ctx->d_gid) (void)hipFree(ctx->d_gid);
    if (ctx->d_hist_min) (void)hipFree(ctx->d_hist_min);
    if (ctx->h_bulk) (void)hipHostFree(ctx->h_bulk);
    if (ctx->h_stage_done) (void)hipHostFree(ctx->h_stage_done);
    if (ctx->d_stage_cnt) (void)hipFree(ctx->d_stage_cnt);
    if (ctx->h_kinds) (void)hipHostFree(ctx->h_kinds);
    if (ctx->h_gid) (void)hipHostFree(ctx->h_gid);
    for (StageBuf &b : ctx->buf) {
        if (b.h_counts) (void)hipHostFree(b.h_counts);
        if (b.h_entries) (void)hipHostFree(b.h_entries);
    }
    for (EventPair &p : ctx->pool) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.end);
    }
    for (EventPair &p : ctx->timing_pairs) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.end);
    }
    if (ctx->copy_done) (void)hipEventDestroy(ctx->copy_done);
    if (ctx->stamp_ev) (void)hipEventDestroy(ctx->stamp_ev);
    if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
    if (ctx->report_ev) (void)hipEventDestroy(ctx->report_ev);
    {
        // score scratch of the streams this context's reports ran on (its own, the resident scorer's, every user stream
        // a report was re-homed onto): the device is idle, nothing reads it any more
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        ctx->report_streams.push_back(ctx->score_stream);
        for (hipStream_t rs : ctx->report_streams) {
            auto it = g_scratch_by_stream.find(reinterpret_cast<void *>(rs));
            if (it == g_scratch_by_stream.end()) continue;
            if (it->second.ptr) (void)hipFree(it->second.ptr);
            g_scratch_by_stream.erase(it);
        }
    }
    if (ctx->score_stream) (void)hipStreamDestroy(ctx->score_stream);
    if (ctx->d_rowg) (void)hipFree(ctx->d_rowg);
    if (ctx->h_gather_err) (void)hipHostFree(ctx->h_gather_err);
    if (ctx->d_stamps && !ctx->stamps_argfree) (void)hipFree(ctx->d_stamps);
    if (ctx->d_stamps_own) (void)hipFree(ctx->d_stamps_own);
    for (const auto &o : ctx->open_stamps)  // regions still open when the context goes: their device slots are free again
        if (!o.own) stamp_slot_give(o.slot);
    delete ctx;
    return NVRX_OK;
}

int nvrx_ctx_set_stream(nvrx_ctx *ctx, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->default_stream = as_stream(stream);
    return NVRX_OK;
}

int nvrx_ctx_info(const nvrx_ctx *ctx, int what) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    switch (what) {
        case 0: return ctx->local_ranks;
        case 1: return ctx->rows_per_rank;
        case 2: return ctx->ring_cap;
        case 3: return ctx->row_stride;
        case 4: return ctx->device;
        case 5: return (int)ctx->reports_rehomed;   // reports that were enqueued on the stream they had to follow (diagnostics)
        case 6: return ctx->last_rehome_verdict;    // why the last report was / was not re-homed: see nvrx_report
        case 7: return (int)ctx->regions_skipped;   // GPU-timed regions that got no sample because their stream was being captured
        case 8: {                                   // rows of a logical rank handed out so far (nvrx_row_alloc)
            std::lock_guard<std::mutex> lk(const_cast<nvrx_ctx *>(ctx)->mu);
            return ctx->rows_used;
        }
        default: return fail(NVRX_ERR_INVALID, "unknown info selector %d", what);
    }
}

int nvrx_row_configure(nvrx_ctx *ctx, int row, int kind, int gid) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    if (kind != NVRX_KIND_SECTION && kind != NVRX_KIND_KERNEL) return fail(NVRX_ERR_INVALID, "bad kind %d", kind);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->h_kinds[row] != (uint8_t)kind || ctx->h_gid[row] != gid) {
        ctx->h_kinds[row] = (uint8_t)kind;
        ctx->h_gid[row] = gid;
        touch_row(ctx, row);
        ctx->meta_dirty = true;
    }
    return NVRX_OK;
}

int nvrx_row_alloc(nvrx_ctx *ctx, int kind) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (kind != NVRX_KIND_SECTION && kind != NVRX_KIND_KERNEL) return fail(NVRX_ERR_INVALID, "bad kind %d", kind);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->rows_used >= ctx->rows_per_rank)
        return fail(NVRX_ERR_RANGE, "straggler rings are full: %d timing rows per rank", ctx->rows_per_rank);
    const int row = ctx->rows_used++;
    for (int lr = 0; lr < ctx->local_ranks; lr++) {
        const size_t g = (size_t)lr * ctx->rows_per_rank + row;
        if (ctx->h_kinds[g] != (uint8_t)kind || ctx->h_gid[g] != -1) {
            ctx->h_kinds[g] = (uint8_t)kind;
            ctx->h_gid[g] = -1;
            ctx->meta_dirty = true;
            touch_row(ctx, (int)g);
        }
    }
    return row;
}

// ------------------------------------------------------------------------------------------------
// rings
// ------------------------------------------------------------------------------------------------
int nvrx_ring_push(nvrx_ctx *ctx, int row, float value) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return push_locked(ctx, row, value);
}

int nvrx_ring_push_many(nvrx_ctx *ctx, int row, const float *values, int n) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    if (n < 0 || (n > 0 && !values)) return fail(NVRX_ERR_INVALID, "bad values/n");
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        rc = push_locked(ctx, row, values[i]);
        if (rc) return rc;
    }
    return NVRX_OK;
}

// The per-kernel tracer's path into the rings (nvrx_ktrace_sink.push): n (row, value) pairs appended one by one under
// ONE lock, staged in pinned memory like nvrx_ring_push -- nothing is launched unless a staging buffer fills up (one
// scatter per 4096 staged samples, or as soon as ONE row has gathered ring_cap of them; on the context's stream), and what is still staged when a report comes is
// flushed by the report.  Called on the tracer's thread, so the device is selected first.
int nvrx_ring_push_staged(nvrx_ctx *ctx, const int32_t *rows, const float *values, int n) {
    if (!ctx || (n > 0 && (!rows || !values)) || n < 0) return fail(NVRX_ERR_INVALID, "bad arguments");
    if (n == 0) return NVRX_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        const int32_t r = rows[i];
        if (r < 0) continue;
        if (r >= ctx->rows) return fail(NVRX_ERR_RANGE, "row %d out of range (%d rows)", (int)r, ctx->rows);
        rc = push_locked(ctx, r, values[i]);
        if (rc) return rc;
    }
    return NVRX_OK;
}

// The two functions a kernel tracer's sink (nvrx_ktrace_sink, include/nvrx_ktrace.h) is given: nvrx_ring_push_staged and
// nvrx_row_alloc with the sink's EXACT C types (a `void *` context) -- calling a function through a pointer of another
// function type is undefined behaviour even where the ABI is the same, and UBSan's function check says so.
int nvrx_sink_push(void *ctx, const int32_t *rows, const float *values, int n) {
    return nvrx_ring_push_staged(static_cast<nvrx_ctx *>(ctx), rows, values, n);
}

int nvrx_sink_row_alloc(void *ctx, int kind) { return nvrx_row_alloc(static_cast<nvrx_ctx *>(ctx), kind); }

// Bulk append of (row, value) pairs in arrival order: ONE scatter launch however many rows the pairs touch.  This is how
// the per-kernel tracer's drained records reach the rings (hundreds to thousands of kernel keys per report,
// CuptiProfiler.cpp:186-207 appends the same records to one CircularBuffer per key on the host).  Ring semantics are
// those of nvrx_ring_push applied pair by pair: slot = pushes % ring_cap, and a pair that a LATER pair of the same call
// would overwrite is not written at all (the scatter has no order inside one launch).  rows[i] < 0 skips the pair.
int nvrx_ring_push_pairs(nvrx_ctx *ctx, const int32_t *rows, const float *values, int n) {
    if (!ctx || (n > 0 && (!rows || !values)) || n < 0) return fail(NVRX_ERR_INVALID, "bad arguments");
    if (n == 0) return NVRX_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    mark_side_work(ctx);  // (a later report is only re-homed onto another stream once this is known to be done)
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->default_stream;
    rc = flush_locked(ctx, st);  // samples staged before this call keep their place in the order
    if (rc) return rc;
    ctx->bulk_cnt.assign((size_t)ctx->rows, 0u);
    for (int i = 0; i < n; i++) {
        const int32_t r = rows[i];
        if (r < 0) continue;
        if (r >= ctx->rows) return fail(NVRX_ERR_RANGE, "row %d out of range (%d rows)", (int)r, ctx->rows);
        ctx->bulk_cnt[(size_t)r]++;
        touch_row(ctx, (int)r);
    }
    if (ctx->bulk_in_flight) {  // the previous call's scatter still reads the entry buffer
        int wrc = nvrx_poll_u32(const_cast<const uint32_t *>(ctx->bulk_word), ctx->bulk_ticket, stage_wait_s());
        if (wrc) return wrc;
        ctx->bulk_in_flight = false;
    }
    if (ctx->bulk_cap < (size_t)n) {
        if (ctx->h_bulk) HIP_TRY(hipHostFree(ctx->h_bulk));
        ctx->h_bulk = nullptr;
        ctx->bulk_cap = 0;
        const size_t cap = std::max<size_t>((size_t)n + (size_t)n / 2, 1u << 16);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_bulk), cap * sizeof(StagedSample), hipHostMallocDefault));
        ctx->bulk_cap = cap;
    }
    ctx->bulk_seen.assign((size_t)ctx->rows, 0u);
    const uint64_t cap = (uint64_t)ctx->ring_cap;
    int m = 0;
    for (int i = 0; i < n; i++) {
        const int32_t r = rows[i];
        if (r < 0) continue;
        const uint32_t k = ctx->bulk_seen[(size_t)r]++;
        if ((uint64_t)(ctx->bulk_cnt[(size_t)r] - k) > cap) continue;  // overwritten later in this very call
        StagedSample &e = ctx->h_bulk[m++];
        e.row_slot = ((uint32_t)r << 16) | (uint32_t)((ctx->total[(size_t)r] + k) % cap);
        e.value = values[i];
    }
    for (int r = 0; r < ctx->rows; r++) ctx->total[(size_t)r] += ctx->bulk_cnt[(size_t)r];
    // the launch of flush_locked with this call's entries: counts (and pending row metadata) travel with it
    StageBuf &b = ctx->buf[ctx->cur];
    for (int r = 0; r < ctx->rows; r++)
        b.h_counts[r] = (uint32_t)std::min<uint64_t>(ctx->total[(size_t)r], cap);
    const int work = std::max(m, ctx->rows);
    {
        int grc = guard_ring_writer(ctx, st);  // the scatter writes ring slots
        if (grc) return grc;
    }
    b.ticket = (b.ticket % 0x7FFFFFFFu) + 1u;
    hipLaunchKernelGGL(k_scatter, dim3((work + 255) / 256), dim3(256), 0, st, b.h_counts, ctx->h_bulk, m, ctx->d_samples,
                       ctx->row_stride, ctx->d_counts, ctx->rows, ctx->meta_dirty ? ctx->h_kinds : nullptr, ctx->d_kinds,
                       ctx->meta_dirty ? ctx->h_gid : nullptr, ctx->d_gid, b.d_cnt, b.d_done, b.ticket);
    HIP_TRY(hipGetLastError());
    b.in_flight = true;
    ctx->bulk_in_flight = true;  // the entry buffer is released by the same ticket as this call's staging buffer
    ctx->bulk_word = b.h_done;
    ctx->bulk_ticket = b.ticket;
    if (ctx->meta_dirty) {
        int wrc = wait_stage_buffer(b);
        if (wrc) return wrc;
    }
    ctx->meta_dirty = false;
    ctx->counts_dirty = false;
    ctx->cur = (ctx->cur + 1) % nvrx_ctx::NBUF;
    return wait_stage_buffer(ctx->buf[ctx->cur]);
}

int nvrx_ring_push_device(nvrx_ctx *ctx, int row, const float *d_values, int n, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    if (n < 0 || (n > 0 && !d_values)) return fail(NVRX_ERR_INVALID, "bad d_values/n");
    if (n == 0) return NVRX_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    mark_side_work(ctx);  // (a later report is only re-homed onto another stream once this is known to be done)
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    rc = flush_locked(ctx, st);  // earlier staged samples must land first
    if (rc) return rc;
    rc = guard_ring_writer(ctx, st);  // the copies below write ring slots
    if (rc) return rc;
    const int cap = ctx->ring_cap;
    // only the newest `cap` samples can survive
    uint64_t first = 0;
    if (n > cap) first = (uint64_t)(n - cap);
    uint64_t pos = ctx->total[row] + first;
    int left = n - (int)first;
    const float *src = d_values + first;
    float *base = ctx->d_samples + (size_t)row * (size_t)ctx->row_stride;
    while (left > 0) {
        const int slot = (int)(pos % (uint64_t)cap);
        const int chunk = std::min(left, cap - slot);
        HIP_TRY(hipMemcpyAsync(base + slot, src, (size_t)chunk * sizeof(float), hipMemcpyDeviceToDevice, st));
        src += chunk;
        pos += (uint64_t)chunk;
        left -= chunk;
    }
    ctx->total[row] += (uint64_t)n;
    touch_row(ctx, row);
    ctx->counts_dirty = true;
    return NVRX_OK;
}

// A [n_rows][ld] device matrix appended to n_rows consecutive rows at once.  Rows whose write position is the same (the
// usual case: rings filled together) take ONE copy launch per ring segment (k_append_rows; two when the append wraps)
// instead of a copy per row; otherwise the rows go one by one.
int nvrx_ring_push_device_rows(nvrx_ctx *ctx, int first_row, int n_rows, const float *d_values, int n, int ld, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (n_rows < 0 || first_row < 0 || first_row + n_rows > ctx->rows)
        return fail(NVRX_ERR_INVALID, "rows [%d,%d) outside [0,%d)", first_row, first_row + n_rows, ctx->rows);
    if (n < 0 || ld < n || (n > 0 && n_rows > 0 && !d_values)) return fail(NVRX_ERR_INVALID, "bad d_values/n/ld");
    if (n == 0 || n_rows == 0) return NVRX_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    mark_side_work(ctx);
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    rc = flush_locked(ctx, st);  // earlier staged samples must land first
    if (rc) return rc;
    rc = guard_ring_writer(ctx, st);
    if (rc) return rc;
    const uint64_t cap = (uint64_t)ctx->ring_cap;
    const uint64_t first = (uint64_t)n > cap ? (uint64_t)n - cap : 0;  // only the newest `cap` samples can survive
    bool together = true;
    for (int r = 1; r < n_rows && together; r++)
        together = ctx->total[(size_t)(first_row + r)] % cap == ctx->total[(size_t)first_row] % cap;
    for (int r = 0; r < (together ? 1 : n_rows); r++) {
        const int row = first_row + r;
        uint64_t pos = ctx->total[(size_t)row] + first;
        int left = n - (int)first;
        const float *src = d_values + (size_t)r * (size_t)ld + first;
        float *base = ctx->d_samples + (size_t)row * (size_t)ctx->row_stride;
        while (left > 0) {
            const int slot = (int)(pos % cap);
            const int chunk = std::min(left, (int)cap - slot);
            // (a kernel of our own, not hipMemcpy2DAsync: nothing but the bytes named here is touched, and it is ordered on
            //  `st` like every other ring writer)
            const int bx = std::min((chunk + 255) / 256, 64);
            const int ny = together ? n_rows : 1;
            hipLaunchKernelGGL(k_append_rows, dim3((unsigned)bx, (unsigned)std::min(ny, 65535)), dim3(256), 0, st, base + slot,
                               (size_t)ctx->row_stride, src, (size_t)ld, chunk, ny);
            HIP_TRY(hipGetLastError());
            src += chunk;
            pos += (uint64_t)chunk;
            left -= chunk;
        }
    }
    for (int r = 0; r < n_rows; r++) ctx->total[(size_t)(first_row + r)] += (uint64_t)n;
    touch_row(ctx, first_row + n_rows - 1);
    ctx->counts_dirty = true;
    return NVRX_OK;
}

int nvrx_ring_set_count(nvrx_ctx *ctx, int row, int n) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    if (n < 0 || n > ctx->ring_cap) return fail(NVRX_ERR_INVALID, "count %d outside [0,%d]", n, ctx->ring_cap);
    std::lock_guard<std::mutex> lk(ctx->mu);
    // samples of this row still sitting in the staging buffer belong to the content being replaced:
    // drop them, otherwise they would land on slots the new content is about to reuse
    StagedSample *e = ctx->buf[ctx->cur].h_entries;
    int kept = 0;
    for (int i = 0; i < ctx->n_staged; i++)
        if ((int)(e[i].row_slot >> 16) != row) e[kept++] = e[i];
    ctx->n_staged = kept;
    ctx->stage_cnt[(size_t)row] = 0;
    ctx->total[row] = (uint64_t)n;
    touch_row(ctx, row);
    ctx->counts_dirty = true;
    return NVRX_OK;
}

int nvrx_ring_set_count_all(nvrx_ctx *ctx, int n) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (n < 0 || n > ctx->ring_cap) return fail(NVRX_ERR_INVALID, "count %d outside [0,%d]", n, ctx->ring_cap);
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->n_staged = 0;
    ctx->flush_gen++;
    std::fill(ctx->total.begin(), ctx->total.end(), (uint64_t)n);
    ctx->rows_hi = ctx->rows;
    ctx->counts_dirty = true;
    return NVRX_OK;
}

int nvrx_ring_count(const nvrx_ctx *ctx, int row) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    std::lock_guard<std::mutex> lk(const_cast<nvrx_ctx *>(ctx)->mu);  // (the kernel tracer's thread appends concurrently)
    return (int)std::min<uint64_t>(ctx->total[(size_t)row], (uint64_t)ctx->ring_cap);
}

int nvrx_ring_counts(const nvrx_ctx *ctx, int32_t *out, int n) {
    if (!ctx || !out) return fail(NVRX_ERR_INVALID, "null argument");
    if (n < 0 || n > ctx->rows) return fail(NVRX_ERR_INVALID, "n %d outside [0,%d]", n, ctx->rows);
    std::lock_guard<std::mutex> lk(const_cast<nvrx_ctx *>(ctx)->mu);  // (the kernel tracer's thread appends concurrently)
    for (int r = 0; r < n; r++) out[r] = (int32_t)std::min<uint64_t>(ctx->total[(size_t)r], (uint64_t)ctx->ring_cap);
    return NVRX_OK;
}

int nvrx_ring_occupancy_changed(nvrx_ctx *ctx, int n) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (n < 0 || n > ctx->rows) return fail(NVRX_ERR_INVALID, "n %d outside [0,%d]", n, ctx->rows);
    std::lock_guard<std::mutex> lk(ctx->mu);  // (the kernel tracer's thread appends concurrently)
    int changed = ctx->occupied_rows != n;
    if (changed) {
        ctx->occupied_seen.assign((size_t)n, 0);
        ctx->occupied_rows = n;
    }
    for (int r = 0; r < n; r++) {
        const uint8_t now = ctx->total[(size_t)r] != 0;
        if (ctx->occupied_seen[(size_t)r] != now) {
            ctx->occupied_seen[(size_t)r] = now;
            changed = 1;
        }
    }
    return changed;
}

int nvrx_ring_reset(nvrx_ctx *ctx) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    // staged samples belong to the window being dropped
    ctx->n_staged = 0;
    ctx->flush_gen++;
    std::fill(ctx->total.begin(), ctx->total.end(), 0);
    ctx->counts_dirty = true;
    return NVRX_OK;
}

int nvrx_history_reset(nvrx_ctx *ctx, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    mark_side_work(ctx);  // (a later report is only re-homed onto another stream once this is known to be done)
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_f32, dim3((ctx->rows + 255) / 256), dim3(256), 0, as_stream(stream), ctx->d_hist_min,
                       (size_t)ctx->rows, INFINITY);
    HIP_TRY(hipGetLastError());
    return NVRX_OK;
}

int nvrx_ring_flush(nvrx_ctx *ctx, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    mark_side_work(ctx);  // (a later report is only re-homed onto another stream once this is known to be done)
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    return flush_locked(ctx, as_stream(stream));
}

int nvrx_ring_read(nvrx_ctx *ctx, int row, float *out, int n, void *stream) {
    if (!ctx || !out) return fail(NVRX_ERR_INVALID, "null argument");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    if (n < 0 || n > ctx->row_stride) return fail(NVRX_ERR_INVALID, "n %d outside [0,%d]", n, ctx->row_stride);
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    rc = flush_locked(ctx, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, ctx->d_samples + (size_t)row * (size_t)ctx->row_stride, (size_t)n * sizeof(float),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NVRX_OK;
}

// ------------------------------------------------------------------------------------------------
// hipEvent region timing
// ------------------------------------------------------------------------------------------------
// A stream that is being captured into a hipGraph runs nothing now: a timestamp launched on it would land in the graph
// with this entry's ring slot baked in, and the host would count a sample no kernel has written.  Such an entry gets no GPU
// time (the reference records none either: no kernel executes while a capture is open, CuptiProfiler.cpp:168-207 sees only
// the replays), the caller keeps the section's wall time on the host path.  (A query the runtime refuses -- the legacy
// stream while another stream captures in global mode -- counts as "not capturing" and its error is not left behind.)
static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

// Argument-free stamp slots (g_stamp_slots) that are held by an OPEN region of some context of this process.
static std::mutex g_stamp_slot_mu;
static uint64_t g_stamp_slot_open = 0;  // bit i: slot i is held
static unsigned g_stamp_slot_next = 0;
static_assert(NVRX_NSTAMP <= 64, "the open-slot mask is one 64-bit word");

static int stamp_slot_take() {
    std::lock_guard<std::mutex> lk(g_stamp_slot_mu);
    for (int tries = 0; tries < NVRX_NSTAMP; tries++) {
        const int cand = (int)(g_stamp_slot_next % (unsigned)NVRX_NSTAMP);
        g_stamp_slot_next++;
        if (!(g_stamp_slot_open >> cand & 1ull)) {
            g_stamp_slot_open |= 1ull << cand;
            return cand;
        }
    }
    return -1;
}

static void stamp_slot_give(int slot) {
    std::lock_guard<std::mutex> lk(g_stamp_slot_mu);
    g_stamp_slot_open &= ~(1ull << slot);
}

static bool take_skipped_region(nvrx_ctx *ctx, int row) {
    for (int i = (int)ctx->skipped_regions.size() - 1; i >= 0; i--)
        if (ctx->skipped_regions[(size_t)i] == row) {
            ctx->skipped_regions.erase(ctx->skipped_regions.begin() + i);
            return true;
        }
    return false;
}

int nvrx_event_begin(nvrx_ctx *ctx, int row, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (stream_is_capturing(as_stream(stream))) {
        ctx->skipped_regions.push_back(row);
        ctx->regions_skipped++;
        return NVRX_REGION_SKIPPED;
    }
    int idx = -1;
    int rc = get_pair(ctx->pool, ctx->free_pairs, &idx);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->pool[(size_t)idx].start, as_stream(stream)));
    ctx->open.push_back({row, idx});
    return NVRX_OK;
}

int nvrx_event_end(nvrx_ctx *ctx, int row, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (take_skipped_region(ctx, row)) return NVRX_REGION_SKIPPED;
    for (int i = (int)ctx->open.size() - 1; i >= 0; i--) {
        if (ctx->open[(size_t)i].row == row) {
            const nvrx_ctx::Open o = ctx->open[(size_t)i];
            ctx->open.erase(ctx->open.begin() + i);
            if (stream_is_capturing(as_stream(stream))) {  // opened before the capture began: the pair cannot be closed now
                ctx->free_pairs.push_back(o.pair);
                ctx->regions_skipped++;
                return NVRX_REGION_SKIPPED;
            }
            HIP_TRY(hipEventRecord(ctx->pool[(size_t)o.pair].end, as_stream(stream)));
            ctx->pending.push_back(o);
            return NVRX_OK;
        }
    }
    return fail(NVRX_ERR_STATE, "nvrx_event_end(row=%d) without a matching nvrx_event_begin", row);
}

int nvrx_event_harvest(nvrx_ctx *ctx, int wait) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    while (!ctx->pending.empty()) {
        const nvrx_ctx::Open o = ctx->pending.front();
        EventPair &p = ctx->pool[(size_t)o.pair];
        if (wait) {
            HIP_TRY(hipEventSynchronize(p.end));
        } else {
            hipError_t q = hipEventQuery(p.end);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return fail(NVRX_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        }
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.start, p.end));
        ctx->pending.pop_front();
        ctx->free_pairs.push_back(o.pair);
        int rc = push_locked(ctx, o.row, ms * 1000.0f);  // microseconds, as CuptiProfiler.cpp:191
        if (rc) return rc;
    }
    return (int)ctx->pending.size();
}

// ------------------------------------------------------------------------------------------------
// device-side region timing
// ------------------------------------------------------------------------------------------------
int nvrx_stamp_begin(nvrx_ctx *ctx, int row, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (row < 0 || row >= ctx->rows) return fail(NVRX_ERR_INVALID, "row %d out of range [0,%d)", row, ctx->rows);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if ((int)ctx->open_stamps.size() >= nvrx_ctx::NSTAMP / 2) return fail(NVRX_ERR_STATE, "too many open GPU-timed regions");
    if (stream_is_capturing(as_stream(stream))) {
        ctx->skipped_regions.push_back(row);
        ctx->regions_skipped++;
        return NVRX_REGION_SKIPPED;
    }
    int slot = -1;
    bool own = false;
    if (ctx->stamps_argfree) {
        // The slots are the device's, shared by every context of the process on it, so they are handed out process-wide --
        // and a slot is not handed out again while the region that holds it is OPEN (between its begin and its end call) in
        // ANY context: 64 later entries of other regions (other contexts, regions nested through the C ABI) would otherwise
        // overwrite a long region's begin timestamp.  (Regions that are closed follow each other in stream order, so the
        // round-robin reuse of their slots is safe on one stream however far the host runs ahead.)
        slot = stamp_slot_take();
        if (slot >= 0) {
            hipLaunchKernelGGL(g_stamp_begin_fn[slot], dim3(1), dim3(1), 0, as_stream(stream));
        } else {
            // every argument-free slot is held by an open region: this one gets a slot of the context's own (one-argument kernel)
            if (!ctx->d_stamps_own) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_stamps_own), nvrx_ctx::NSTAMP * sizeof(unsigned long long)));
                HIP_TRY(hipMemset(ctx->d_stamps_own, 0, nvrx_ctx::NSTAMP * sizeof(unsigned long long)));
            }
            own = true;
        }
    } else {
        own = true;
    }
    if (own) {
        unsigned long long *base = ctx->stamps_argfree ? ctx->d_stamps_own : ctx->d_stamps;
        int &next = ctx->stamps_argfree ? ctx->stamp_next_own : ctx->stamp_next;
        // (the context's own slots: skip those its open regions hold; open_stamps.size() < NSTAMP / 2, so one is free)
        for (int tries = 0; tries < nvrx_ctx::NSTAMP; tries++) {
            const int cand = next;
            next = (next + 1) % nvrx_ctx::NSTAMP;
            bool held = false;
            for (const auto &o : ctx->open_stamps) held = held || (o.own && o.slot == cand);
            if (!held) {
                slot = cand;
                break;
            }
        }
        hipLaunchKernelGGL(k_stamp_begin, dim3(1), dim3(1), 0, as_stream(stream), base + slot);
    }
    HIP_TRY(hipGetLastError());
    ctx->open_stamps.push_back({row, slot, own});
    return NVRX_OK;
}

int nvrx_stamp_end(nvrx_ctx *ctx, int row, int cpu_row, float cpu_value, void *stream) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    if (cpu_row >= ctx->rows) return fail(NVRX_ERR_INVALID, "cpu_row %d out of range [0,%d)", cpu_row, ctx->rows);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (take_skipped_region(ctx, row)) return NVRX_REGION_SKIPPED;
    for (int i = (int)ctx->open_stamps.size() - 1; i >= 0; i--) {
        if (ctx->open_stamps[(size_t)i].row != row) continue;
        const int slot = ctx->open_stamps[(size_t)i].slot;
        const bool own = ctx->open_stamps[(size_t)i].own;
        const unsigned long long *slot_ptr = (own && ctx->stamps_argfree ? ctx->d_stamps_own : ctx->d_stamps) + slot;
        ctx->open_stamps.erase(ctx->open_stamps.begin() + i);
        if (!own) stamp_slot_give(slot);
        if (stream_is_capturing(as_stream(stream))) {  // opened before the capture began: no sample, nothing enqueued
            ctx->regions_skipped++;
            return NVRX_REGION_SKIPPED;
        }
        const uint64_t cap = (uint64_t)ctx->ring_cap;
        float *dst_gpu = ctx->d_samples + (size_t)row * (size_t)ctx->row_stride + (size_t)(ctx->total[(size_t)row] % cap);
        ctx->total[(size_t)row]++;
        touch_row(ctx, row);
        float *dst_cpu = nullptr;
        if (cpu_row >= 0) {
            dst_cpu = ctx->d_samples + (size_t)cpu_row * (size_t)ctx->row_stride + (size_t)(ctx->total[(size_t)cpu_row] % cap);
            ctx->total[(size_t)cpu_row]++;
            touch_row(ctx, cpu_row);
        }
        ctx->counts_dirty = true;
        hipStream_t st = as_stream(stream);
        {
            // an asynchronous report may still be reading the rings: this stream's ring writes follow its statistics kernel
            int grc = guard_ring_writer(ctx, st);
            if (grc) return grc;
        }
        ctx->stamps_used = true;
        hipLaunchKernelGGL(k_stamp_end, dim3(1), dim3(1), 0, st, slot_ptr, ctx->us_per_tick, dst_gpu, dst_cpu, cpu_value);
        HIP_TRY(hipGetLastError());
        if (std::find(ctx->stamp_streams.begin(), ctx->stamp_streams.end(), st) == ctx->stamp_streams.end())
            ctx->stamp_streams.push_back(st);
        return NVRX_OK;
    }
    return fail(NVRX_ERR_STATE, "nvrx_stamp_end(row=%d) without a matching nvrx_stamp_begin", row);
}

// ------------------------------------------------------------------------------------------------
// report (local half)
// ------------------------------------------------------------------------------------------------
static int report_local_impl(nvrx_ctx *ctx, float *d_stats, float *d_send, int K, int S, int names_ok, int rows_active,
                             void *stream, unsigned long long *rowg, uint32_t epoch);

int nvrx_report_local(nvrx_ctx *ctx, float *d_stats, float *d_send, int K, int S, int names_ok, int rows_active,
                      void *stream) {
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        mark_side_work(ctx);  // the caller decides when this stream is waited for: no re-homing until that is known
    }
    return report_local_impl(ctx, d_stats, d_send, K, S, names_ok, rows_active, stream, nullptr, 0);
}

static int report_local_impl(nvrx_ctx *ctx, float *d_stats, float *d_send, int K, int S, int names_ok, int rows_active,
                             void *stream, unsigned long long *rowg, uint32_t epoch) {
    hipStream_t st = as_stream(stream);
    if (!ctx || !d_stats) return fail(NVRX_ERR_INVALID, "null argument");
    if (K < 0 || S < 0) return fail(NVRX_ERR_INVALID, "bad K/S");
    if (rows_active < 0 || rows_active > ctx->rows_per_rank) return fail(NVRX_ERR_INVALID, "rows_active %d outside [0,%d]", rows_active, ctx->rows_per_rank);
    if (rows_active == 0) rows_active = ctx->rows_per_rank;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ctx_set_device(ctx);
    if (rc) return rc;
    int uniform_n = -1;
    rc = flush_locked(ctx, st, &uniform_n, rows_active);
    if (rc) return rc;
    report_clk(2);
    Epilogue ep{};
    ep.gid = ctx->d_gid;
    // history minima only advance on a real report (d_send given), not on a statistics peek
    ep.hist_min = d_send ? ctx->d_hist_min : nullptr;
    ep.send = d_send;
    ep.rows_per_rank = ctx->rows_per_rank;
    ep.rows_active = rows_active;
    ep.K = K;
    ep.KS = K + S;
    ep.L = NVRX_TABLE_LEN(K, S);
    ep.names_ok = names_ok ? 1.0f : 0.0f;
    ep.rowg = rowg;
    ep.epoch = epoch;
    int pair = -1;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    if (ctx->timing) {
        rc = get_pair(ctx->timing_pairs, ctx->timing_free, &pair);
        if (rc) return rc;
        ev_start = ctx->timing_pairs[(size_t)pair].start;
        ev_stop = ctx->timing_pairs[(size_t)pair].end;
    }
    rc = launch_row_stats(ctx->d_samples, ctx->d_counts, ctx->d_kinds, ctx->local_ranks * rows_active, ctx->row_stride,
                          d_stats, ep, st, ev_start, ev_stop, uniform_n);
    if (rc) return rc;
    if (pair >= 0) ctx->timing_used.push_back(pair);
    report_clk(3);
    return NVRX_OK;
}

// ------------------------------------------------------------------------------------------------
// peer-window exchange: state (the entry points follow the report)
// ------------------------------------------------------------------------------------------------
struct nvrx_peer {
    int device = 0, world = 0, rank = 0, stride = 0;
    unsigned long long *window = nullptr;            // this process' window (fine-grained device memory)
    std::vector<unsigned long long *> mapped;        // [world] every window as mapped in this process
    std::vector<bool> opened;                        // mapped[p] came from hipIpcOpenMemHandle
    unsigned long long **d_windows = nullptr;        // device copy of `mapped`
    uint32_t *h_err = nullptr, *d_err = nullptr;     // pinned error word and its device address
    uint32_t epoch = 0;
    double timeout_s = 1800.0;
    int wall_khz = 100000;
    bool ready = false;
};

// Arguments of one exchange on `p` (advances the epoch: every process counts its exchanges the same way).
static int peer_fill_args(nvrx_peer *p, const void *send, void *recv, size_t count, PeerArgs *a) {
    if (!p || !p->ready || !send || !recv) return fail(NVRX_ERR_INVALID, "peer exchange is not ready");
    if (count == 0 || count > (size_t)p->stride) return fail(NVRX_ERR_RANGE, "%zu floats per rank exceed the window's %d", count, p->stride);
    *a = PeerArgs{};
    a->windows = p->d_windows;
    a->send = static_cast<const float *>(send);
    a->recv = static_cast<float *>(recv);
    a->err = p->d_err;
    a->world = p->world;
    a->rank = p->rank;
    a->count = (int)count;
    a->stride = p->stride;
    p->epoch = (p->epoch % 0x7FFFFFFFu) + 1u;  // never 0
    a->epoch = p->epoch;
    a->timeout_ticks = (unsigned long long)(p->timeout_s * 1e3 * (double)p->wall_khz);
    return NVRX_OK;
}

// ------------------------------------------------------------------------------------------------
// report (everything, one call)
// ------------------------------------------------------------------------------------------------
int nvrx_report_clocks(double *out8) {
    if (!out8) return fail(NVRX_ERR_INVALID, "null argument");
    memcpy(out8, g_report_clk, sizeof(g_report_clk));
    return NVRX_OK;
}

int nvrx_report(nvrx_ctx *ctx, nvrx_report_desc *d, void *stream) {
    report_clk(0);
    if (!ctx || !d) return fail(NVRX_ERR_INVALID, "null argument");
    if (!d->d_stats || !d->d_send || !d->d_scores || !d->d_meta) return fail(NVRX_ERR_INVALID, "null buffer in the report descriptor");
    const bool exchanging = d->allgather_fn != nullptr;
    if (exchanging && (!d->d_table || d->d_table == d->d_send || d->send_count <= 0))
        return fail(NVRX_ERR_INVALID, "an exchange needs a separate table buffer and a positive send_count");
    // Re-homing.  A synchronous report that has to follow the work of exactly ONE other stream -- the stream the
    // region stamps of this window were launched on and / or the caller's current stream -- is enqueued ON that stream:
    // the stream order is the dependency, and the report issues nothing but kernel launches.  The event record +
    // stream-wait pairs it replaces are cheap between two reports a millisecond apart, but a report at production
    // cadence (one per ~60 s, S/straggler.py:125) meets them cold: 75 us of the 98 us the call took went into them
    // (tools/cadence_detector_breakdown.py), against 3 us for each of the launches, whose code the training loop keeps
    // warm.  Not for asynchronous reports (they are meant to run BESIDE the next step; re-homed they measured 2.7 % per
    // step instead of 2.0-2.4 % and no gain at cadence), not while work of ours may still be running on the context's
    // own stream (side_work), NVRX_DEBUG_REPORT_REHOME=0 turns it off.
    // Asynchronous reports are re-homed too when they are rare (round 4): one that stays on the detector's own stream needs
    // an event record + stream-wait pair to follow the training stream's stamps, and at production cadence those two
    // cold calls were 45-52 us of the 51-116 us an enqueue cost (profiles/r04o).  Re-homed, the report's kernels (~20 us of
    // GPU time) sit in the training stream once per interval instead of beside it -- which is why reports closer together
    // than NVRX_DEBUG_ASYNC_REHOME_GAP_US (default 5000) stay where they were: a report EVERY step is cheaper beside the step.
    const bool is_async = d->h_seq_word == nullptr;
    bool async_gap_ok = false;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        const double now_us = g_report_clk[0];
        async_gap_ok = ctx->async_rehome_gap_us > 0.0 && ctx->last_report_us > 0.0 && now_us - ctx->last_report_us >= ctx->async_rehome_gap_us;
        ctx->last_report_us = now_us;
        ctx->last_rehome_verdict = 0;  // (not a candidate, unless the block below says otherwise)
        if (d->prev_settled) {
            // the caller has seen this context's previous (asynchronous) report complete: nothing has to be guarded against
            // it any more, and what was enqueued on the context's stream in front of it is done as well
            ctx->guard_done_epoch = ctx->report_epoch;
            if (ctx->async_on_ctx && ctx->async_gen == ctx->side_gen) ctx->side_work = false;
            ctx->async_on_ctx = false;
        }
    }
    bool rehomed = false;
    if (ctx->rehome_mode && ((!is_async && !d->guard_rings) || (is_async && d->guard_rings && async_gap_ok))) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        hipStream_t only = nullptr;
        int distinct = 0;
        auto see = [&](hipStream_t s) {
            if (distinct == 0 || s != only) {
                only = s;
                distinct++;
            }
        };
        for (hipStream_t s : ctx->stamp_streams) see(s);
        if (d->order_after_enabled) see(as_stream(d->order_after_stream));
        ctx->last_rehome_verdict = !(distinct == 1 && only != as_stream(stream)) ? -1 : ctx->side_work ? -2 : ctx->async_on_ctx ? -3 : ctx->timing ? -4 : 1;
        if (ctx->last_rehome_verdict == 1) {
            stream = only;
            rehomed = true;
            ctx->stamp_streams.clear();
            ctx->reports_rehomed++;
        }
    }
    if (!rehomed && d->order_after_enabled && d->order_after_stream != stream) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIP_TRY(hipEventRecord(ctx->order_ev, as_stream(d->order_after_stream)));
        HIP_TRY(hipStreamWaitEvent(as_stream(stream), ctx->order_ev, 0));
    }
    report_clk(1);
    // Resident scorer: the score kernel goes to its own stream NEXT TO the statistics kernel and picks the rows' results
    // up as they are published (8-byte tagged granules), so neither kernel has a queued successor / predecessor.
    // Synchronous reports only, no exchange or the peer-window exchange (an RCCL all-gather needs the stream order).
    if (d->R > 64) {  // (only large jobs keep score scratch per stream, score_launch)
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (std::find(ctx->report_streams.begin(), ctx->report_streams.end(), as_stream(stream)) == ctx->report_streams.end())
            ctx->report_streams.push_back(as_stream(stream));
    }
    const int rows_launch = (d->rows_active > 0 ? d->rows_active : ctx->rows_per_rank) * ctx->local_ranks;
    const bool peer_route = exchanging && d->allgather_fn == reinterpret_cast<void *>(&nvrx_peer_allgather) && peer_prologue_enabled();
    // When it pays (same-box A/B, r02k-m): the resident score kernel must not have to be ordered after other streams' work
    // -- the event waits of a report behind a training step cost more than the kernel boundary they avoid (+74 vs +51 us
    // per step with a report every step) -- so it is used when the report has nothing to wait for: 64 rows 21.2 vs 22.4
    // us per report, 512 rows (two row workgroups on every CU, the score kernel squeezed in beside two of them) 24.2-25.9
    // resident vs 24.0-24.4 queued depending on the box, with the statistics kernel itself at 8.5 instead of 9.9 us.
    bool cross_stream = rehomed || (d->order_after_enabled && d->order_after_stream != stream);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        cross_stream = cross_stream || !ctx->stamp_streams.empty();
    }
    const int rmode = ctx->resident_mode;
    const bool resident = d->resident && d->h_seq_word && !d->guard_rings && (!exchanging || peer_route) && rows_launch > 0 &&
                          score_fits_single_wg(d->R, d->K, d->S, d->d_scores, d->d_flags) &&
                          (rmode == 2 || (rmode == 1 && !cross_stream));
    if (resident) {
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            HIP_TRY(hipSetDevice(ctx->device));
            if (!ctx->score_stream) {
                // the resident scorer's stream follows the rule of the detector's own (backend.py, _stream_priority): high
                // priority in a single-process job -- its one workgroup is dispatched ahead of a busy training stream's next
                // ones --, normal in a multi-rank job; NVRX_STREAM_PRIORITY=high|normal names it outright
                bool high = true;
                const char *pe = getenv("NVRX_STREAM_PRIORITY");
                if (pe && *pe) {
                    high = !(pe[0] == 'n' || pe[0] == 'N' || pe[0] == '0' || pe[0] == 'o' || pe[0] == 'O');
                } else {
                    for (const char *var : {"WORLD_SIZE", "SLURM_NTASKS", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE"}) {
                        const char *v = getenv(var);
                        if (!v || !*v) continue;
                        const long n = strtol(v, nullptr, 10);
                        if (n > 1) high = false;
                        if (n > 1 || strcmp(var, "WORLD_SIZE") == 0) break;  // WORLD_SIZE, where present, has the word (ktrace._job_size)
                    }
                }
                if (high) {
                    int least = 0, greatest = 0;  // (numerically lower = higher priority)
                    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
                    HIP_TRY(hipStreamCreateWithPriority(&ctx->score_stream, hipStreamNonBlocking, greatest));
                } else {
                    HIP_TRY(hipStreamCreateWithFlags(&ctx->score_stream, hipStreamNonBlocking));
                }
            }
            // the score kernel becomes dispatchable when the statistics kernel does, not before
            if (d->order_after_enabled && d->order_after_stream != stream)
                HIP_TRY(hipStreamWaitEvent(ctx->score_stream, ctx->order_ev, 0));
            if (!ctx->stamp_streams.empty()) {
                int orc = order_after_stamps(ctx, as_stream(stream), ctx->score_stream);
                if (orc) return orc;
            }
        }
        ctx->gran_epoch = (ctx->gran_epoch % 0x7FFFFFFFu) + 1u;
        unsigned long long *slice = ctx->d_rowg + (size_t)(ctx->gran_epoch & 1u) * (size_t)ctx->rows * ROW_GRANULES;
        int rc2 = report_local_impl(ctx, d->d_stats, d->d_send, d->K, d->S, d->names_ok, d->rows_active, stream, slice,
                                    ctx->gran_epoch);
        if (rc2) return rc2;
        GatherArgs ga{};
        ga.g = slice;
        ga.gid = ctx->d_gid;
        ga.send = d->d_send;
        ga.stats_out = (d->d_stats_dst && d->stats_rows > 0) ? reinterpret_cast<nvrx_f4 *>(d->d_stats_dst) : nullptr;
        ga.err = ctx->d_gather_err;
        ga.n_blocks = rows_launch;
        ga.rows_active = d->rows_active > 0 ? d->rows_active : ctx->rows_per_rank;
        ga.rows_per_rank = ctx->rows_per_rank;
        ga.local_ranks = ctx->local_ranks;
        ga.names_ok = d->names_ok ? 1.0f : 0.0f;
        ga.epoch = ctx->gran_epoch;
        ga.timeout_ticks = (unsigned long long)((d->timeout_s > 0.0 ? d->timeout_s : 1e9) * 1e3 * (double)ctx->wall_khz);
        ga.poll_naps = ctx->poll_naps;
        PeerArgs pa{};
        if (peer_route) {
            rc2 = peer_fill_args(static_cast<nvrx_peer *>(d->comm), d->d_send, d->d_table, (size_t)d->send_count, &pa);
            if (rc2) return rc2;
        }
        d->seq = (d->seq % 0x7FFFFFFFu) + 1u;
        rc2 = score_launch(exchanging ? d->d_table : d->d_send, d->R, d->K, d->S, d->do_indiv, d->do_rel, d->thresholds,
                           d->d_scores, d->d_flags, d->d_meta, d->d_done_counter, d->seq, nullptr, nullptr, 0,
                           ctx->score_stream, peer_route ? &pa : nullptr, &ga);
        if (rc2) return rc2;
        report_clk(4);
        report_clk(5);
        rc2 = nvrx_poll_u32(d->h_seq_word, d->seq, d->timeout_s > 0.0 ? d->timeout_s : 1e30);
        report_clk(6);
        if (rc2) return rc2;
        if (*static_cast<volatile uint32_t *>(ctx->h_gather_err) == ctx->gran_epoch)
            return fail(NVRX_ERR_TIMEOUT, "the score kernel gave up waiting for the statistics kernel's rows (epoch %u)", ctx->gran_epoch);
        if (as_stream(stream) == ctx->default_stream) {
            // every row's granules were consumed: the statistics kernel, last on the context's in-order stream, is done
            std::lock_guard<std::mutex> lk(ctx->mu);
            ctx->side_work = false;
            ctx->async_on_ctx = false;
        }
        return NVRX_OK;
    }
    int rc = report_local_impl(ctx, d->d_stats, d->d_send, d->K, d->S, d->names_ok, d->rows_active, stream, nullptr, 0);
    if (rc) return rc;
    if (d->guard_rings) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->guard_stream = as_stream(stream);
        if (rehomed || !ctx->stamps_used) {
            // writers on this very stream follow by stream order; anybody else records the event (guard_ring_writer).  Also
            // when no stamp kernel has ever written these rings (per-kernel timing, sections without GPU time): every writer
            // there is -- the scatter of staged samples, from the training thread or the kernel tracer's -- is on the context's
            // own stream, the one this report is on, and the eager record below would be 5 us of a cold enqueue for nobody.
            // With stamps the event stays eager: recorded lazily it would sit behind the SCORE kernel too, which on an
            // in-stream exchange route waits for the slowest peer -- and the training stream's next stamp with it.
            ctx->guard_recorded = false;
        } else {
            HIP_TRY(hipEventRecord(ctx->report_ev, as_stream(stream)));
            ctx->guard_recorded = true;
        }
        ctx->report_epoch++;
    }
    PeerArgs pa{};
    bool prologue = false;
    if (exchanging && d->allgather_fn == reinterpret_cast<void *>(&nvrx_peer_allgather) &&
        score_fits_single_wg(d->R, d->K, d->S, d->d_scores, d->d_flags) && peer_prologue_enabled()) {
        // peer windows + a table the single-workgroup score kernel takes: the exchange runs as that kernel's prologue
        rc = peer_fill_args(static_cast<nvrx_peer *>(d->comm), d->d_send, d->d_table, (size_t)d->send_count, &pa);
        if (rc) return rc;
        prologue = true;
    } else if (exchanging) {
        // ncclAllGather(sendbuff, recvbuff, sendcount, ncclFloat32 = 7, comm, stream), enqueued between the two
        // kernels on the same stream: this rank's rows -> every rank's [R, L] table (reporting.py:281,397)
        using AllGatherFn = int (*)(const void *, void *, size_t, int, void *, void *);
        const int nrc = reinterpret_cast<AllGatherFn>(d->allgather_fn)(d->d_send, d->d_table, (size_t)d->send_count, 7,
                                                                       d->comm, stream);
        if (nrc != 0) {
            return fail(NVRX_ERR_HIP, "all-gather of the exchange rows failed (ncclResult %d)", nrc);
        }
    }
    report_clk(4);
    d->seq = (d->seq % 0x7FFFFFFFu) + 1u;
    rc = score_launch(exchanging ? d->d_table : d->d_send, d->R, d->K, d->S, d->do_indiv, d->do_rel, d->thresholds,
                      d->d_scores, d->d_flags, d->d_meta, d->d_done_counter, d->seq, d->d_stats, d->d_stats_dst,
                      d->stats_rows, stream, prologue ? &pa : nullptr);
    if (rc) return rc;
    report_clk(5);
    if (d->h_seq_word) {
        rc = nvrx_poll_u32(d->h_seq_word, d->seq, d->timeout_s > 0.0 ? d->timeout_s : 1e30);
        report_clk(6);
        if (rc == NVRX_OK && !rehomed && as_stream(stream) == ctx->default_stream) {
            // the completion word was stored by the last kernel of this report on the context's own in-order stream:
            // everything of ours enqueued there before it has finished
            std::lock_guard<std::mutex> lk(ctx->mu);
            ctx->side_work = false;
            ctx->async_on_ctx = false;
        }
        return rc;
    }
    if (!rehomed) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->async_on_ctx = true;  // an asynchronous report is in flight on the context's own stream,
        ctx->async_gen = ctx->side_gen;  // behind all the side work counted so far
    }
    return NVRX_OK;
}

// ------------------------------------------------------------------------------------------------
// peer-window exchange
// ------------------------------------------------------------------------------------------------
// One report window in ONE call: what Detector.generate_report does around nvrx_report in the steady state -- wait for the
// window's kernel records (or harvest the region events), make sure the set of occupied rows is the one the caller's name
// tables were built for, run the report, empty the rings.  At production cadence every one of those steps used to be a
// Python call of its own that ran cold (profiles/r06a_kernels_mode_breakdown.txt: 237 us per report, 49 of them in C).
static thread_local double g_window_clk[2] = {0.0, 0.0};

int nvrx_window_clocks(double *out2) {
    if (!out2) return fail(NVRX_ERR_INVALID, "out2 is null");
    out2[0] = g_window_clk[0], out2[1] = g_window_clk[1];
    return NVRX_OK;
}

int nvrx_window_report(nvrx_ctx *ctx, nvrx_report_desc *desc, void *stream, nvrx_window_desc *w) {
    if (!ctx || !desc || !w) return fail(NVRX_ERR_INVALID, "null argument");
    g_window_clk[0] = g_window_clk[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const bool per_kernel = w->kt_sync != nullptr;
    if (per_kernel && (!w->kt_hold || !w->kt_counter)) return fail(NVRX_ERR_INVALID, "a kernel tracer needs sync, hold and counter");
    const bool enqueue_only = w->asynchronous != 0;
    bool held = false;
    auto leave = [&](int rc) {
        if (held) w->kt_hold(0);
        return rc;
    };
    w->out_names_ok = 1;
    if (per_kernel) {
        if (enqueue_only) {  // durations that arrive from here to the ring reset below stay with the tracer's thread
            w->kt_hold(1);
            held = true;
            // (test hook, NVRX_DEBUG_WINDOW_HOLD_US: stay under the hold for a while, so that the tracer's thread consumes a batch
            //  -- parks it -- before the window is judged; tests/test_gpu_01_ktrace_datapath.py, the changed-rows case)
            static const int hold_us = [] { const char *e = getenv("NVRX_DEBUG_WINDOW_HOLD_US"); return e ? atoi(e) : 0; }();
            if (hold_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(hold_us));
        }
        const int missing = w->kt_sync(enqueue_only ? 0.0 : w->kt_patience_s);
        if (missing < 0) return leave(fail(NVRX_ERR_STATE, "the kernel tracer's sync failed (%d)", missing));
        // still missing after the patience: the caller waits the long way; new kernel keys: it has names to learn first
        if ((missing > 0 && !enqueue_only) || w->kt_counter(10) != w->kt_rows_known || w->kt_counter(6) != w->kt_keys_without_row)
            return leave(NVRX_WINDOW_MISS);
    } else if (w->harvest_regions) {
        const int rc = nvrx_event_harvest(ctx, 1);  // (event pairs of the window are waited for in either mode, as Detector does)
        if (rc < 0) return rc;
    }
    {
        // (a look, nothing stored: when the set changed the caller's general path asks nvrx_ring_occupancy_changed itself)
        // leave() lifts the tracer's hold, and lifting it hands parked durations to the sink, which takes ctx->mu (the tracer's
        // lock order is its own mutex, then the context's): never called with ctx->mu held -- an asynchronous per-kernel report
        // that found the occupied rows changed used to do exactly that and could deadlock against the tracer's thread
        // (tools/soak.py in per-kernel mode, once in ~10 000 reports)
        const int n = w->rows_used;
        bool bad = false, changed = false;
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            bad = n < 0 || n > ctx->rows;
            if (!bad) {
                changed = ctx->occupied_rows != n;
                for (int r = 0; r < n && !changed; r++) changed = ctx->occupied_seen[(size_t)r] != (uint8_t)(ctx->total[(size_t)r] != 0);
            }
        }
        if (bad) return leave(fail(NVRX_ERR_INVALID, "rows_used %d outside [0,%d]", n, ctx->rows));
        if (changed) return leave(NVRX_WINDOW_MISS);
    }
    g_window_clk[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const int rc = nvrx_report(ctx, desc, stream);
    if (rc < 0) return leave(rc);
    if (!enqueue_only && desc->h_seq_word) {
        // the completion word has been seen: meta[0], four words below it, says whether every rank had ids for all its names
        const uint32_t names = *const_cast<const volatile uint32_t *>(desc->h_seq_word - 4);
        if (names != 1u) {
            w->out_names_ok = 0;
            return leave(NVRX_WINDOW_NAMES);  // the rings are left as they are: the caller syncs names and reports again
        }
    }
    const int rrc = nvrx_ring_reset(ctx);
    return leave(rrc < 0 ? rrc : NVRX_OK);
}

int nvrx_peer_create(int device, int world, int rank, int max_floats_per_rank, nvrx_peer **out) {
    if (!out) return fail(NVRX_ERR_INVALID, "out is null");
    *out = nullptr;
    if (world <= 0 || rank < 0 || rank >= world || max_floats_per_rank <= 0) return fail(NVRX_ERR_INVALID, "bad peer geometry");
    HIP_TRY(hipSetDevice(device));
    nvrx_peer *p = new (std::nothrow) nvrx_peer();
    if (!p) return fail(NVRX_ERR_NOMEM, "host allocation failed");
    p->device = device;
    p->world = world;
    p->rank = rank;
    p->stride = (max_floats_per_rank + 1) & ~1;
    p->mapped.assign((size_t)world, nullptr);
    p->opened.assign((size_t)world, false);
    const size_t bytes = 2ull * (size_t)world * (size_t)p->stride * sizeof(unsigned long long);
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&p->window), bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        delete p;
        return fail(NVRX_ERR_HIP, "hipExtMallocWithFlags(fine-grained window, %zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    e = hipMemset(p->window, 0, bytes);  // tag 0 is never a valid epoch
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&p->h_err), 64, hipHostMallocMapped);
    if (e == hipSuccess) {
        p->h_err[0] = 0;
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&p->d_err), p->h_err, 0);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p->d_windows), (size_t)world * sizeof(void *));
    if (e != hipSuccess) {
        int rc = fail(NVRX_ERR_HIP, "peer window set-up failed: %s", hipGetErrorString(e));
        nvrx_peer_destroy(p);
        return rc;
    }
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) p->wall_khz = khz;
    p->mapped[(size_t)rank] = p->window;
    *out = p;
    return NVRX_OK;
}

int nvrx_peer_ipc_handle(nvrx_peer *p, void *handle64) {
    if (!p || !handle64) return fail(NVRX_ERR_INVALID, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipIpcGetMemHandle(&h, p->window));
    memcpy(handle64, &h, sizeof(h));
    return NVRX_OK;
}

int nvrx_peer_device_id(const nvrx_peer *p, char *out, int len) {
    if (!p || !out || len < 16) return fail(NVRX_ERR_INVALID, "nvrx_peer_device_id: a buffer of at least 16 bytes is needed");
    HIP_TRY(hipDeviceGetPCIBusId(out, len, p->device));
    return NVRX_OK;
}

// Preconditions of a window on ANOTHER device, checked before anything is mapped: the peer's GPU (named by its PCI bus id,
// which is the same in every process whatever HIP_VISIBLE_DEVICES says) must be reachable from this one with peer-to-peer
// stores.  0: same device, or peer access is possible; 1: the peer's device is not visible to this process, nothing
// could be checked (hipIpcOpenMemHandle decides); negative: the devices cannot reach each other -- with the reason.
int nvrx_peer_check_access(const nvrx_peer *p, int peer_rank, const char *peer_pci_bus_id) {
    if (!p || !peer_pci_bus_id) return fail(NVRX_ERR_INVALID, "null argument");
    int peer_dev = -1;
    if (hipDeviceGetByPCIBusId(&peer_dev, peer_pci_bus_id) != hipSuccess || peer_dev < 0) {
        (void)hipGetLastError();
        return 1;
    }
    if (peer_dev == p->device) return NVRX_OK;
    int can = 0;
    hipError_t e = hipDeviceCanAccessPeer(&can, p->device, peer_dev);
    if (e != hipSuccess)
        return fail(NVRX_ERR_HIP, "hipDeviceCanAccessPeer(device %d -> device %d [%s], rank %d) failed: %s", p->device, peer_dev,
                    peer_pci_bus_id, peer_rank, hipGetErrorString(e));
    if (!can)
        return fail(NVRX_ERR_STATE,
                    "peer windows need peer-to-peer access between the GPUs of one node: device %d cannot access device %d [%s] "
                    "(rank %d; hipDeviceCanAccessPeer = 0 -- no xGMI / PCIe P2P path, or it is disabled by IOMMU / ACS settings). "
                    "Use NVRX_EXCHANGE=rccl.", p->device, peer_dev, peer_pci_bus_id, peer_rank);
    return NVRX_OK;
}

int nvrx_peer_connect(nvrx_peer *p, int peer_rank, const void *handle64) {
    if (!p || !handle64) return fail(NVRX_ERR_INVALID, "null argument");
    if (peer_rank < 0 || peer_rank >= p->world) return fail(NVRX_ERR_INVALID, "peer rank %d out of range", peer_rank);
    if (peer_rank == p->rank) return NVRX_OK;
    HIP_TRY(hipSetDevice(p->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void *ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess || !ptr) {
        (void)hipGetLastError();
        return fail(NVRX_ERR_HIP,
                    "hipIpcOpenMemHandle of rank %d's window failed on device %d: %s (the window is fine-grained device memory "
                    "exported with hipIpcGetMemHandle; both processes need HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose driver "
                    "only supports dmabuf IPC, and the two GPUs need a peer-to-peer path)", peer_rank, p->device,
                    hipGetErrorString(e));
    }
    p->mapped[(size_t)peer_rank] = static_cast<unsigned long long *>(ptr);
    p->opened[(size_t)peer_rank] = true;
    return NVRX_OK;
}

int nvrx_peer_ready(nvrx_peer *p, double timeout_s) {
    if (!p) return fail(NVRX_ERR_INVALID, "null argument");
    for (int r = 0; r < p->world; r++)
        if (!p->mapped[(size_t)r]) return fail(NVRX_ERR_STATE, "window of rank %d is not connected", r);
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemcpy(p->d_windows, p->mapped.data(), (size_t)p->world * sizeof(void *), hipMemcpyHostToDevice));
    if (timeout_s > 0.0) p->timeout_s = timeout_s;
    p->ready = true;
    return NVRX_OK;
}

// ncclAllGather-compatible signature (plugs into nvrx_report_desc.allgather_fn; `comm` is the nvrx_peer).
// Returns 0 or a positive code (ncclResult-style): nvrx_last_error() holds the message.
int nvrx_peer_allgather(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream) {
    nvrx_peer *p = static_cast<nvrx_peer *>(comm);
    if (dtype != 7) return -fail(NVRX_ERR_INVALID, "peer exchange carries f32 only");
    PeerArgs a{};
    const int frc = peer_fill_args(p, send, recv, count, &a);
    if (frc) return -frc;
    hipLaunchKernelGGL(k_peer_allgather, dim3(1), dim3(PEER_THREADS), 0, as_stream(stream), a);
    if (hipGetLastError() != hipSuccess) return -fail(NVRX_ERR_HIP, "k_peer_allgather launch failed");
    return 0;
}

// Epoch of the last exchange that gave up waiting for a peer (0 = none), as raised by the kernel.
int nvrx_peer_error(const nvrx_peer *p, uint32_t *epoch_out) {
    if (!p || !epoch_out) return fail(NVRX_ERR_INVALID, "null argument");
    *epoch_out = *static_cast<volatile uint32_t *>(p->h_err);
    return NVRX_OK;
}

void *nvrx_peer_allgather_address(void) { return reinterpret_cast<void *>(&nvrx_peer_allgather); }

int nvrx_peer_destroy(nvrx_peer *p) {
    if (!p) return NVRX_OK;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < p->world; r++)
        if (p->opened[(size_t)r] && p->mapped[(size_t)r]) (void)hipIpcCloseMemHandle(p->mapped[(size_t)r]);
    if (p->window) (void)hipFree(p->window);
    if (p->d_windows) (void)hipFree(p->d_windows);
    if (p->h_err) (void)hipHostFree(p->h_err);
    delete p;
    return NVRX_OK;
}

int nvrx_timing_enable(nvrx_ctx *ctx, int on) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->timing = on != 0;
    return NVRX_OK;
}

int nvrx_timing_read(nvrx_ctx *ctx, double *total_us, int *launches, int reset) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (int idx : ctx->timing_used) {
        EventPair &p = ctx->timing_pairs[(size_t)idx];
        HIP_TRY(hipEventSynchronize(p.end));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.start, p.end));
        ctx->timing_total_us += (double)ms * 1000.0;
        ctx->timing_launches++;
        ctx->timing_free.push_back(idx);
    }
    ctx->timing_used.clear();
    if (total_us) *total_us = ctx->timing_total_us;
    if (launches) *launches = ctx->timing_launches;
    if (reset) {
        ctx->timing_total_us = 0.0;
        ctx->timing_launches = 0;
    }
    return NVRX_OK;
}

// ------------------------------------------------------------------------------------------------
// host buffers and completion
// ------------------------------------------------------------------------------------------------
int nvrx_host_alloc(void **out, void **out_device, size_t bytes) {
    if (!out || bytes == 0) return fail(NVRX_ERR_INVALID, "bad arguments");
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocMapped));
    memset(*out, 0, bytes);
    if (out_device) HIP_TRY(hipHostGetDevicePointer(out_device, *out, 0));
    return NVRX_OK;
}

int nvrx_poll_u32(const uint32_t *h_word, uint32_t expected, double timeout_s) {
    if (!h_word) return fail(NVRX_ERR_INVALID, "null argument");
    const volatile uint32_t *p = h_word;
    if (*p == expected) return NVRX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const uint32_t cur = *p;
        if (cur == expected) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return NVRX_OK;
        }
        // sequence words only move forward: one that is already PAST the awaited value will never show it (the block
        // was reused by a later report before this one was collected) -- an error, not a wait
        if (cur != 0 && (uint32_t)(cur - expected) < 0x10000000u)
            return fail(NVRX_ERR_STATE, "completion word is at %u, past the awaited %u: the result block was reused", cur, expected);
        __builtin_ia32_pause();
        if ((spins & 0x3FFu) == 0x3FFu) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > timeout_s) return fail(NVRX_ERR_TIMEOUT, "completion word %u not seen after %.3f s (have %u)", expected, dt, *p);
        }
    }
}

int nvrx_device_alloc(void **out, size_t bytes) {
    if (!out || bytes == 0) return fail(NVRX_ERR_INVALID, "nvrx_device_alloc: bad arguments");
    *out = nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return fail(NVRX_ERR_NOMEM, "hipMalloc of %zu bytes failed", bytes);
    }
    hipError_t e = hipMemset(p, 0, bytes);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(NVRX_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return NVRX_OK;
}

int nvrx_device_free(void *p) {
    if (!p) return NVRX_OK;
    HIP_TRY(hipFree(p));
    return NVRX_OK;
}

int nvrx_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return NVRX_OK;
}

int nvrx_d2h_sync(void *h_dst, const void *d_src, size_t bytes, void *stream) {
    if (!h_dst || !d_src) return fail(NVRX_ERR_INVALID, "null argument");
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NVRX_OK;
}

int nvrx_copy_to_host(nvrx_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream) {
    if (!ctx || !h_dst || !d_src) return fail(NVRX_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t st = as_stream(stream);
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(ctx->copy_done, st));
    ctx->copy_pending = true;
    return NVRX_OK;
}

int nvrx_wait(nvrx_ctx *ctx) {
    if (!ctx) return fail(NVRX_ERR_INVALID, "ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->copy_pending) {
        HIP_TRY(hipEventSynchronize(ctx->copy_done));
        ctx->copy_pending = false;
    }
    return NVRX_OK;
}

}  // extern "C"
