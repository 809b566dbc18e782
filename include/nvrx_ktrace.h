/*
 * nvrx_ktrace.h -- C ABI of libnvrx_ktrace.so: per-kernel GPU durations with real kernel names on MI355X,
 * the rocprofiler-sdk counterpart of the reference's CUPTI activity tracing
 * (/root/reference/src/nvidia_resiliency_ext/attribution/straggler/cupti_src/CuptiProfiler.cpp:96-207).
 *
 * The reference enables CUPTI_ACTIVITY_KIND_CONCURRENT_KERNEL while a profiled section is open and, on
 * CUPTI's thread, turns every kernel record into key = "<name>_blk_x_y_z_grid_x_y_z", value =
 * (end - start) / 1000.0f microseconds, appended to that key's overwrite-oldest CircularBuffer
 * (CuptiProfiler.cpp:168-207, CircularBuffer.h:53-61).  Here a rocprofiler-sdk context with the buffered
 * KERNEL_DISPATCH tracing service does the same job: start/stop map to rocprofiler_start_context /
 * rocprofiler_stop_context (cuptiActivityEnable / Disable, CuptiProfiler.cpp:116-133), finished dispatches arrive
 * from the SDK (completion callbacks by default, buffered records with NVRX_DEBUG_KTRACE_DELIVERY=buffer), and a thread
 * of the tracer -- never the one that trains -- appends every duration to its key's ring: the rings are the device
 * rings of libnvrx_straggler_hip.so, reached through a SINK of two plain function pointers
 * (nvrx_ktrace_set_sink), so the two libraries do not link against each other.  Overflow keeps the NEWEST
 * ring_cap durations per key, memory is bounded by the rings, and the thread that trains does no per-record
 * work: at report time it calls nvrx_ktrace_sync (wait until every dispatch enqueued so far has been appended)
 * and nothing else.  The statistics (computeStats, CuptiProfiler.cpp:44-74) stay on the GPU (k_row_stats,
 * NVRX_KIND_KERNEL).
 *
 * It is a separate library because it exports rocprofiler_configure: the SDK only accepts tools before the
 * HIP/HSA runtime initialises, so either nvrx_ktrace_setup() runs before the first HIP call of the process
 * (importing nvrx_straggler in a multi-rank job, or with NVRX_GPU_TIMING=kernels, does that), or the library
 * is named in ROCP_TOOL_LIBRARIES.  Plain C, no exceptions across the ABI; 0 / negative errno-style returns as
 * in nvrx_straggler.h.
 */
#ifndef NVRX_KTRACE_H
#define NVRX_KTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVRX_KTRACE_OK 0
#define NVRX_KTRACE_ERR_STATE (-1)    /* too late (runtime already initialised) / not set up */
#define NVRX_KTRACE_ERR_SDK (-5)      /* a rocprofiler-sdk call failed; see nvrx_ktrace_last_error() */
#define NVRX_KTRACE_ERR_INVALID (-22) /* bad argument */
#define NVRX_KTRACE_ERR_UNSAFE (-16)  /* the tool-search guard was refused: other threads of the process are running (EBUSY) */

/* One kernel execution without a sink: key id (see nvrx_ktrace_key_name) and duration in microseconds, f32,
 * computed exactly as CuptiProfiler.cpp:191. */
typedef struct nvrx_ktrace_record {
    uint32_t key;
    float us;
} nvrx_ktrace_record;

/* Where the durations go.  push(ctx, rows, values, n) appends values[i] to ring row rows[i] for all i, in order,
 * overwrite-oldest (nvrx_sink_push of nvrx_straggler.h: nvrx_ring_push_staged with this exact type); row_alloc(ctx, kind) hands
 * out the ring row of a key seen for the first time, negative when none is left (nvrx_sink_row_alloc: nvrx_row_alloc).  Both are called on the
 * SDK's callback thread, never on the thread that launches kernels. */
typedef struct nvrx_ktrace_sink {
    void *ctx;
    int (*push)(void *ctx, const int32_t *rows, const float *values, int n);
    int (*row_alloc)(void *ctx, int kind);
    int32_t kind; /* what row_alloc is asked for: NVRX_KIND_KERNEL */
} nvrx_ktrace_sink;

/* Register the tool with rocprofiler-sdk (rocprofiler_force_configure).  Must run before the HIP runtime
 * initialises; NVRX_KTRACE_ERR_STATE if the SDK is already locked.  Idempotent.  max_pending bounds the records
 * held for nvrx_ktrace_drain while NO sink is installed (<= 0: 1 << 20); beyond it the OLDEST are dropped and
 * counted -- the newest survive, as in the reference's per-key rings (CircularBuffer.h:53-61).
 * The SDK's tool search is kept off the large libraries of the process for the duration of the call (see
 * nvrx_ktrace.cpp, "tool discovery guard"): only while every other thread of the process is asleep, otherwise
 * NVRX_KTRACE_ERR_UNSAFE and nothing has happened (NVRX_KTRACE_SCAN_GUARD=0: no guard, full search;
 * =force: guard without the check). */
int nvrx_ktrace_setup(int max_pending);
/* The bound of the pending queue alone (<= 0: 1 << 20); records beyond it are dropped right away, oldest first. */
int nvrx_ktrace_set_max_pending(int max_pending);
/* rocprofiler_force_configure leaves ROCPROFILER_REGISTER_FORCE_LOAD=1 (and five GLOG_* settings) in the process
 * environment; they are needed until this process' HIP runtime has started and would make every child process load and
 * configure rocprofiler-sdk on `import torch`.  Puts the variables back as they were before nvrx_ktrace_setup; returns how
 * many changed.  Call once nvrx_ktrace_ready() is 1 (nvrx_ktrace_start does it by itself). */
int nvrx_ktrace_release_env(void);
/* Libraries whose names the last nvrx_ktrace_setup kept out of rocprofiler-sdk's tool search. */
int nvrx_ktrace_hidden_libraries(void);
/* 1 once the SDK has called the tool's initialiser and its context is valid (after the first HIP call). */
int nvrx_ktrace_ready(void);
/* Install (or with NULL remove) the sink.  Returns after any batch in progress on the SDK's thread is through:
 * once nvrx_ktrace_set_sink(NULL) has returned the old ctx is not touched again.  Keys keep their ids; rows are
 * handed out afresh under a new sink. */
int nvrx_ktrace_set_sink(const nvrx_ktrace_sink *sink);
/* While on, the tracer's thread keeps the durations that arrive to itself instead of handing them to the sink; turning it
 * off hands them over, in order.  An ASYNCHRONOUS report brackets "statistics launch + ring reset" with it, so that a
 * duration arriving in between lands in the NEXT window instead of being wiped with the old one.  Never blocks the
 * tracer's thread. */
int nvrx_ktrace_hold(int on);
/* The reference accepts only CUPTI_ACTIVITY_KIND_CONCURRENT_KERNEL records (CuptiProfiler.cpp:118,179): a cudaMemset or
 * cudaMemcpy never becomes a key.  ROCm carries hipMemset / device-to-device hipMemcpy (and libraries' workspace fills) out
 * with ROCclr's built-in blit KERNELS (__amd_rocclr_fillBufferAligned, __amd_rocclr_copyBuffer, ...), which arrive as
 * kernel dispatches; they are left out by default (counter 11 counts them) so that the keys are the ones CUPTI would have
 * produced.  on != 0 records them like any other kernel (the decision applies to records that arrive from now on). */
int nvrx_ktrace_include_blits(int on);
/* Diagnostics: while on, a copy of every duration handed to the sink is also queued for nvrx_ktrace_drain (bounded by
 * max_pending, oldest dropped): how a test reads the very durations the rings were given. */
int nvrx_ktrace_tap(int on);
/* Enable / disable kernel-dispatch tracing (cuptiActivityEnable / Disable, CuptiProfiler.cpp:116-133).  Kernels
 * enqueued while tracing was on are recorded even if they finish after nvrx_ktrace_stop. */
int nvrx_ktrace_start(void);
int nvrx_ktrace_stop(void);
/* Wait until every kernel enqueued (while tracing was on) BEFORE this call has finished on the GPU and its
 * duration has been appended to the sink / the pending queue: the role of torch.cuda.synchronize() +
 * cuptiActivityFlushAll in the reference (straggler.py:234, CuptiProfiler.cpp:138) without waiting for anything
 * else on the device.  Returns 0 when complete, the number of dispatches still missing (> 0) after timeout_s
 * seconds (<= 0: do not wait, just look), or a negative error.  Dispatches are counted by an ENQUEUE callback
 * on the launching thread (one atomic increment); NVRX_DEBUG_KTRACE_COUNT=0 turns the counting off and this call into
 * "flush until two flushes bring nothing new" (the caller then has to synchronise the device first). */
int nvrx_ktrace_sync(double timeout_s);
/* Give up on the dispatches nvrx_ktrace_sync is still missing (call after the device has been synchronised and
 * nvrx_ktrace_flush has run: whatever has not arrived by then never will).  Should such a record arrive after all, the
 * forgiveness is taken back (counter 7 goes down again), so later syncs do not under-wait. */
int nvrx_ktrace_forgive(void);
/* Make every completed dispatch visible (cuptiActivityFlushAll, CuptiProfiler.cpp:138): flushes the SDK's buffer
 * until two consecutive flushes bring nothing new. */
int nvrx_ktrace_flush(void);
/* Without a sink: pop up to cap records (oldest first); returns how many were written, or a negative error. */
int nvrx_ktrace_drain(nvrx_ktrace_record *out, int cap);
/* Records waiting to be drained / dropped so far because the pending queue was full (oldest first). */
int nvrx_ktrace_pending(void);
uint64_t nvrx_ktrace_dropped(void);
/* Counters, monotonic over the process: 0 dispatches enqueued while tracing, 1 dispatch records arrived, 2 durations
 * handed to the sink, 3 durations lost because the sink had no row left for their key, 4 sink errors, 5 records
 * of this library's own engine kernels left out, 7 forgiven dispatches, 8 SDK buffer flushes issued by the pump
 * thread (callback delivery: drains of the inbox), 9 = 1 if dispatches are counted, 11 records of the runtime's memset /
 * memcpy (blit) kernels left out, 12 = 1 if finished dispatches are delivered by completion callbacks (0: buffered
 * records, NVRX_DEBUG_KTRACE_DELIVERY=buffer); under the
 * CURRENT sink: 6 keys that found no row left, 10 keys that were given a row (a host polls this one to learn when new
 * names have turned up). */
uint64_t nvrx_ktrace_counter(int what);
/* Keys seen so far, and the name of one: "<kernel name>_blk_x_y_z_grid_x_y_z" (CuptiProfiler.cpp:186-189;
 * grid counts workgroups like CUDA's gridDim, not work-items).  The pointer stays valid for the process. */
int nvrx_ktrace_num_keys(void);
const char *nvrx_ktrace_key_name(uint32_t key);
/* Ring row of a key under the CURRENT sink: >= 0, -1 none left when it was asked for, -2 not asked for yet. */
int nvrx_ktrace_key_row(uint32_t key);
/* Forget pending records (keys keep their ids): CuptiProfiler::reset, CuptiProfiler.cpp:148-152. */
int nvrx_ktrace_reset(void);
const char *nvrx_ktrace_last_error(void);

/* ---- feed without a GPU ---------------------------------------------------------------------------------------
 * The records below take exactly the path of the SDK's records from the callback thread on (key cache, sink, rings,
 * counters), on the calling thread: how the CPU tests and the benchmark's feeder thread drive the data path, and how
 * a host with its own source of kernel timings could use the library.  No counterpart in the reference. */
typedef struct nvrx_ktrace_dispatch {
    uint64_t kernel_id;
    uint32_t workgroup[3];
    uint32_t grid[3]; /* work-items, as HSA counts */
    uint64_t start_ns, end_ns;
} nvrx_ktrace_dispatch;
/* Name a kernel id as a code-object callback would ("<mangled name>"; own != 0: one of the engine's kernels). */
int nvrx_ktrace_feed_kernel_name(uint64_t kernel_id, const char *name, int own);
/* counted & 1: the dispatches also count as enqueued (as if the ENQUEUE callback had seen them); recs == NULL with
 * counted & 1 only counts n dispatches as enqueued -- their records follow in a later call with counted == 0 (a kernel
 * that is still running when somebody calls nvrx_ktrace_sync).  counted & 2: the records are left in the tracer's INBOX,
 * exactly as the SDK's completion callback leaves them, instead of being consumed on the calling thread -- the pump
 * thread or the next nvrx_ktrace_sync / nvrx_ktrace_flush brings them in. */
int nvrx_ktrace_feed(const nvrx_ktrace_dispatch *recs, int n, int counted);

#ifdef __cplusplus
}
#endif
#endif /* NVRX_KTRACE_H */
