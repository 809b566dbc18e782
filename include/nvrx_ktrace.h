/*
 * nvrx_ktrace.h -- C ABI of libnvrx_ktrace.so: per-kernel GPU durations with real kernel names on MI355X,
 * the rocprofiler-sdk counterpart of the reference's CUPTI activity tracing
 * (/root/reference/src/nvidia_resiliency_ext/attribution/straggler/cupti_src/CuptiProfiler.cpp:96-207).
 *
 * The reference enables CUPTI_ACTIVITY_KIND_CONCURRENT_KERNEL while a profiled section is open and, on
 * CUPTI's thread, turns every kernel record into key = "<name>_blk_x_y_z_grid_x_y_z", value =
 * (end - start) / 1000.0f microseconds (CuptiProfiler.cpp:168-207).  Here a rocprofiler-sdk context with
 * the buffered KERNEL_DISPATCH tracing service does the same job: start/stop map to
 * rocprofiler_start_context / rocprofiler_stop_context (cuptiActivityEnable / Disable,
 * CuptiProfiler.cpp:116-133), records arrive on the SDK's callback thread, and the host drains
 * (key id, microseconds) pairs which the Python layer appends to the device rings of
 * libnvrx_straggler_hip.so -- the statistics themselves (computeStats, CuptiProfiler.cpp:44-74) stay on
 * the GPU (k_row_stats, NVRX_KIND_KERNEL).
 *
 * It is a separate library because it exports rocprofiler_configure: the SDK only accepts tools before the
 * HIP/HSA runtime initialises, so either nvrx_ktrace_setup() runs before the first HIP call of the process
 * (importing nvrx_straggler with NVRX_GPU_TIMING=kernels does that), or the library is named in
 * ROCP_TOOL_LIBRARIES.  Plain C, no exceptions across the ABI; 0 / negative errno-style returns as in
 * nvrx_straggler.h.
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

/* One drained kernel execution: key id (see nvrx_ktrace_key_name) and duration in microseconds, f32,
 * computed exactly as CuptiProfiler.cpp:191. */
typedef struct nvrx_ktrace_record {
    uint32_t key;
    float us;
} nvrx_ktrace_record;

/* Register the tool with rocprofiler-sdk (rocprofiler_force_configure).  Must run before the HIP runtime
 * initialises; NVRX_KTRACE_ERR_STATE if the SDK is already locked.  Idempotent.  max_pending bounds the
 * records held between two drains (<= 0: 1 << 20); beyond it records are dropped and counted, the way the
 * reference drops records when its buffer pool is exhausted (BufferPool.cpp:46-48). */
int nvrx_ktrace_setup(int max_pending);
/* Libraries whose names the last nvrx_ktrace_setup kept out of rocprofiler-sdk's tool search (the SDK reads every
 * library of the link map front to back while it looks for tools; see nvrx_ktrace.cpp "tool discovery guard"). */
int nvrx_ktrace_hidden_libraries(void);
/* 1 once the SDK has called the tool's initialiser and its context is valid (after the first HIP call). */
int nvrx_ktrace_ready(void);
/* Enable / disable kernel-dispatch tracing (cuptiActivityEnable / Disable, CuptiProfiler.cpp:116-133). */
int nvrx_ktrace_start(void);
int nvrx_ktrace_stop(void);
/* Make every completed dispatch visible to nvrx_ktrace_drain (cuptiActivityFlushAll, CuptiProfiler.cpp:138). */
int nvrx_ktrace_flush(void);
/* Pop up to cap records (oldest first); returns how many were written, or a negative error. */
int nvrx_ktrace_drain(nvrx_ktrace_record *out, int cap);
/* Records waiting to be drained / dropped so far because the pending queue was full. */
int nvrx_ktrace_pending(void);
uint64_t nvrx_ktrace_dropped(void);
/* Keys seen so far, and the name of one: "<kernel name>_blk_x_y_z_grid_x_y_z" (CuptiProfiler.cpp:186-189;
 * grid counts workgroups like CUDA's gridDim, not work-items).  The pointer stays valid for the process. */
int nvrx_ktrace_num_keys(void);
const char *nvrx_ktrace_key_name(uint32_t key);
/* Forget pending records (keys keep their ids): CuptiProfiler::reset, CuptiProfiler.cpp:148-152. */
int nvrx_ktrace_reset(void);
const char *nvrx_ktrace_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NVRX_KTRACE_H */
