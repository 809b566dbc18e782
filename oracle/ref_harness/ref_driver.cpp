/*
 * ref_driver.cpp -- C-ABI driver around the REFERENCE's own native profiler sources.
 *
 * TEST INFRASTRUCTURE ONLY (see cupti.h in this directory).  This translation unit #includes the
 * reference's CuptiProfiler.cpp / BufferPool.cpp / CircularBuffer.h from /root/reference (include
 * path given by oracle/Makefile; nothing is copied) and supplies a fake CUPTI activity feed, so the
 * reference's real computeStats() (CuptiProfiler.cpp:44-74), CircularBuffer (CircularBuffer.h:22-70)
 * and bufferCompleted() record handling (CuptiProfiler.cpp:168-207) can be executed here and used
 * to (1) validate oracle/straggler_oracle.c and (2) generate tests/golden/native_*.json.
 * Output goes to oracle/_ref/libnvrx_ref.so (git-ignored, but shipped to the GPU box).
 */
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

// bufferRequested/bufferCompleted trampolines and computeStats are private/static in the
// reference; widen access for this harness only.
// Pre-include everything the reference headers pull in, so the access widening below only
// touches the reference's own class definitions.
#include <algorithm>
#include <cmath>
#include <iostream>
#include <memory>
#include <mutex>
#include <numeric>
#include <sstream>
#include <unordered_map>
#include <pybind11/pybind11.h>
#include "cupti.h"
#include "cuda_runtime_api.h"
#include "BufferPool.h"
#include "CircularBuffer.h"
#define private public
#include "CuptiProfiler.h"
#undef private
#include "BufferPool.cpp"
#include "CuptiProfiler.cpp"

// ----------------------------------------------------------------------------- fake CUPTI
static CUpti_BuffersCallbackRequestFunc g_req = nullptr;
static CUpti_BuffersCallbackCompleteFunc g_done = nullptr;
static bool g_enabled = false;
static std::vector<CUpti_ActivityKernel4> g_pending;  // records "in flight" inside fake CUPTI
static std::vector<std::string> g_names;              // owns the name strings

extern "C" {
CUptiResult cuptiGetResultString(CUptiResult, const char **str) {
    *str = "fake cupti error";
    return CUPTI_SUCCESS;
}
CUptiResult cuptiActivityRegisterCallbacks(CUpti_BuffersCallbackRequestFunc req,
                                           CUpti_BuffersCallbackCompleteFunc done) {
    g_req = req;
    g_done = done;
    return CUPTI_SUCCESS;
}
CUptiResult cuptiFinalize(void) {
    g_req = nullptr;
    g_done = nullptr;
    return CUPTI_SUCCESS;
}
CUptiResult cuptiActivityEnable(CUpti_ActivityKind) {
    g_enabled = true;
    return CUPTI_SUCCESS;
}
CUptiResult cuptiActivityDisable(CUpti_ActivityKind) {
    g_enabled = false;
    return CUPTI_SUCCESS;
}
CUptiResult cuptiActivityGetNextRecord(uint8_t *buffer, size_t validBytes, CUpti_Activity **record) {
    auto *first = reinterpret_cast<CUpti_ActivityKernel4 *>(buffer);
    size_t n = validBytes / sizeof(CUpti_ActivityKernel4);
    CUpti_ActivityKernel4 *next =
        (*record == nullptr) ? first : reinterpret_cast<CUpti_ActivityKernel4 *>(*record) + 1;
    if (next >= first + n) return CUPTI_ERROR_MAX_LIMIT_REACHED;
    *record = reinterpret_cast<CUpti_Activity *>(next);
    return CUPTI_SUCCESS;
}
CUptiResult cuptiActivityFlushAll(uint32_t) {
    // deliver everything pending through the registered callbacks, one pool buffer at a time
    size_t pos = 0;
    while (g_req && g_done && pos < g_pending.size()) {
        uint8_t *buf = nullptr;
        size_t size = 0, maxrec = 0;
        g_req(&buf, &size, &maxrec);
        if (!buf) break;  // pool exhausted: records dropped (BufferPool.cpp:46-48)
        size_t cap = size / sizeof(CUpti_ActivityKernel4);
        size_t cnt = std::min(cap, g_pending.size() - pos);
        std::memcpy(buf, g_pending.data() + pos, cnt * sizeof(CUpti_ActivityKernel4));
        pos += cnt;
        g_done(nullptr, 0, buf, size, cnt * sizeof(CUpti_ActivityKernel4));
    }
    g_pending.clear();
    return CUPTI_SUCCESS;
}
}  // extern "C"

// ----------------------------------------------------------------------------- C ABI for tests
static CuptiProfiler *g_prof = nullptr;
static std::map<std::string, KernelStats> g_last_stats;
static std::vector<std::string> g_last_keys;

extern "C" {

// the reference's computeStats, verbatim behaviour
int ref_compute_stats(const float *x, int n, float *out5) {
    std::vector<float> v(x, x + n);
    KernelStats s = computeStats(v);
    out5[0] = s.min;
    out5[1] = s.max;
    out5[2] = s.median;
    out5[3] = s.avg;
    out5[4] = s.stddev;
    return s.num_calls;
}

// the reference's CircularBuffer<float>: push all values, linearize into out (cap >= capacity)
int ref_ring_run(const float *vals, int n, int capacity, float *out) {
    CircularBuffer<float> cb((size_t)capacity);
    for (int i = 0; i < n; i++) cb.push_back(vals[i]);
    std::vector<float> lin = cb.linearize();
    for (size_t i = 0; i < lin.size(); i++) out[i] = lin[i];
    return (int)lin.size();
}

// the reference's CuptiProfiler lifecycle driven by fake kernel-activity records
int ref_profiler_create(long bufferSize, long numBuffers, long statsMaxLen) {
    try {
        g_prof = new CuptiProfiler((size_t)bufferSize, (size_t)numBuffers, (size_t)statsMaxLen);
    } catch (const std::exception &) {
        return -1;  // "Only one CuptiProfiler instance is allowed." (CuptiProfiler.cpp:86-88)
    }
    return 0;
}
void ref_profiler_destroy() {
    delete g_prof;
    g_prof = nullptr;
    g_pending.clear();
    g_names.clear();
}
void ref_profiler_initialize() { g_prof->initializeProfiling(); }
void ref_profiler_shutdown() { g_prof->shutdownProfiling(); }
void ref_profiler_start() { g_prof->startProfiling(); }
void ref_profiler_stop() { g_prof->stopProfiling(); }
void ref_profiler_reset() { g_prof->reset(); }

// a "kernel launch": recorded only while activity collection is enabled, like real CUPTI
void ref_profiler_launch(const char *name, int bx, int by, int bz, int gx, int gy, int gz,
                         uint64_t start_ns, uint64_t end_ns) {
    if (!g_enabled) return;
    g_names.emplace_back(name);
    CUpti_ActivityKernel4 k;
    std::memset(&k, 0, sizeof(k));
    k.kind = CUPTI_ACTIVITY_KIND_CONCURRENT_KERNEL;
    k.start = start_ns;
    k.end = end_ns;
    k.blockX = bx; k.blockY = by; k.blockZ = bz;
    k.gridX = gx; k.gridY = gy; k.gridZ = gz;
    k.name = nullptr;  // patched below (g_names may reallocate)
    g_pending.push_back(k);
    for (size_t i = 0; i < g_pending.size(); i++)
        g_pending[i].name = g_names[g_names.size() - g_pending.size() + i].c_str();
}

int ref_profiler_get_stats() {
    g_last_stats = g_prof->getStats();
    g_last_keys.clear();
    for (auto &kv : g_last_stats) g_last_keys.push_back(kv.first);
    return (int)g_last_keys.size();
}
const char *ref_profiler_key(int i) { return g_last_keys[(size_t)i].c_str(); }
int ref_profiler_stats(int i, float *out5) {
    const KernelStats &s = g_last_stats[g_last_keys[(size_t)i]];
    out5[0] = s.min;
    out5[1] = s.max;
    out5[2] = s.median;
    out5[3] = s.avg;
    out5[4] = s.stddev;
    return s.num_calls;
}

}  // extern "C"
