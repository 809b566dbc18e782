// Is a report's pair of launches (k_row_stats on the detector's stream, the resident k_score1 on its own) cheaper as ONE
// hipGraphLaunch of a two-node graph?  Two independent trivial kernels, each storing a sequence number into its own
// pinned host word; measured per variant: host time of the launch call(s), and launch -> both words visible.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/graph_launch.cpp -o tools/kb/graph_launch && tools/kb/graph_launch
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

__global__ void k_word(volatile unsigned *word, const unsigned *seq_src, unsigned seq_arg) {
    const unsigned seq = seq_src ? *seq_src : seq_arg;  // graph variant: the sequence number comes from memory
    __hip_atomic_store(const_cast<unsigned *>(word), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double med(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned *h = nullptr, *d = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void **>(&h), 256, hipHostMallocMapped));
    CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&d), h, 0));
    volatile unsigned *wa = h, *wb = h + 16;
    unsigned *seq_host = h + 32;  // read by the graph's kernels
    h[0] = h[16] = h[32] = 0;
    const int N = 2000, WARM = 200;

    auto wait_both = [&](unsigned seq) {
        while (*wa != seq || *wb != seq) __builtin_ia32_pause();
    };

    // (1) two plain launches on two streams
    std::vector<double> call, total;
    for (int i = 1; i <= N + WARM; i++) {
        const unsigned seq = (unsigned)i;
        const double t0 = now_us();
        hipLaunchKernelGGL(k_word, dim3(1), dim3(64), 0, sa, d, nullptr, seq);
        hipLaunchKernelGGL(k_word, dim3(1), dim3(64), 0, sb, d + 16, nullptr, seq);
        const double t1 = now_us();
        wait_both(seq);
        const double t2 = now_us();
        if (i > WARM) {
            call.push_back(t1 - t0);
            total.push_back(t2 - t0);
        }
    }
    printf("two launches on two streams      : calls %.2f us, launch -> both words visible %.2f us\n", med(call), med(total));

    // (2) one graph, two independent kernel nodes; the sequence number travels through a pinned word
    hipGraph_t g;
    CK(hipGraphCreate(&g, 0));
    hipGraphNode_t na, nb;
    void *pa = d, *pb = d + 16, *ps = d + 32;
    unsigned zero = 0;
    void *args_a[] = {&pa, &ps, &zero}, *args_b[] = {&pb, &ps, &zero};
    hipKernelNodeParams kp{};
    kp.func = reinterpret_cast<void *>(k_word);
    kp.gridDim = dim3(1);
    kp.blockDim = dim3(64);
    kp.kernelParams = args_a;
    CK(hipGraphAddKernelNode(&na, g, nullptr, 0, &kp));
    kp.kernelParams = args_b;
    CK(hipGraphAddKernelNode(&nb, g, nullptr, 0, &kp));
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    call.clear();
    total.clear();
    for (int i = 1; i <= N + WARM; i++) {
        const unsigned seq = 100000u + (unsigned)i;
        const double t0 = now_us();
        *seq_host = seq;
        CK(hipGraphLaunch(ge, sa));
        const double t1 = now_us();
        wait_both(seq);
        const double t2 = now_us();
        if (i > WARM) {
            call.push_back(t1 - t0);
            total.push_back(t2 - t0);
        }
    }
    printf("one hipGraphLaunch, two nodes     : call  %.2f us, launch -> both words visible %.2f us\n", med(call), med(total));

    // (3) the same graph with both nodes' parameters rewritten before every launch (sequence number as an argument)
    call.clear();
    total.clear();
    void *none = nullptr;
    for (int i = 1; i <= N + WARM; i++) {
        unsigned seq = 200000u + (unsigned)i;
        void *ua[] = {&pa, &none, &seq}, *ub[] = {&pb, &none, &seq};
        const double t0 = now_us();
        kp.kernelParams = ua;
        CK(hipGraphExecKernelNodeSetParams(ge, na, &kp));
        kp.kernelParams = ub;
        CK(hipGraphExecKernelNodeSetParams(ge, nb, &kp));
        CK(hipGraphLaunch(ge, sa));
        const double t1 = now_us();
        wait_both(seq);
        const double t2 = now_us();
        if (i > WARM) {
            call.push_back(t1 - t0);
            total.push_back(t2 - t0);
        }
    }
    printf("graph + 2 x ExecKernelNodeSetParams: calls %.2f us, launch -> both words visible %.2f us\n", med(call), med(total));

    // (4) one plain launch, for scale
    call.clear();
    total.clear();
    for (int i = 1; i <= N + WARM; i++) {
        const unsigned seq = 300000u + (unsigned)i;
        const double t0 = now_us();
        hipLaunchKernelGGL(k_word, dim3(1), dim3(64), 0, sa, d, nullptr, seq);
        const double t1 = now_us();
        while (*wa != seq) __builtin_ia32_pause();
        const double t2 = now_us();
        if (i > WARM) {
            call.push_back(t1 - t0);
            total.push_back(t2 - t0);
        }
    }
    printf("one launch                        : call  %.2f us, launch -> word visible %.2f us\n", med(call), med(total));
    CK(hipDeviceSynchronize());
    return 0;
}
