/*
 * nvrx_straggler.h -- C ABI of libnvrx_straggler_hip.so, the MI355X-native (gfx950 HIP) engine behind
 * the straggler-detection scoring hot path of nvidia-resiliency-ext.
 *
 * The reference has NO C/FFI operator interface for this path: its boundary is (1) the public Python
 * API (straggler.Detector / ReportGenerator / Report) and (2) the pybind11 native module
 * `nvrx_cupti_module` (cupti_src/cupti_module_py.cpp:33-55).  This library sits below both; the Python
 * package in nvidia-resiliency-ext_amd/ keeps (1) and (2) intact and calls the entry points below via
 * ctypes.  Each entry point cites the reference code it replaces; paths are relative to
 * /root/reference/src/nvidia_resiliency_ext/attribution/straggler/ .
 *
 * Conventions: plain C, no exceptions across the ABI, no Python/torch types.  Every function returns
 * 0 (NVRX_OK) or a negative errno-style code; nvrx_last_error() returns the calling thread's last
 * message.  Device buffers named d_* are caller-owned HIP device pointers (e.g. tensor.data_ptr());
 * `stream` is a hipStream_t passed as void* (0 = the null stream).  A context is thread-compatible:
 * calls on one context must be serialised by the caller (the Python layer holds a lock, as the
 * reference's CuptiManager does, cupti.py:44).
 */
#ifndef NVRX_STRAGGLER_H
#define NVRX_STRAGGLER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVRX_ABI_VERSION 2

#define NVRX_OK 0
#define NVRX_ERR_INVALID (-22) /* bad argument (EINVAL) */
#define NVRX_ERR_NOMEM (-12)   /* host or device allocation failed (ENOMEM) */
#define NVRX_ERR_HIP (-5)      /* a HIP runtime call failed (EIO); see nvrx_last_error() */
#define NVRX_ERR_STATE (-1)    /* call not valid in the current state (EPERM) */
#define NVRX_ERR_RANGE (-34)   /* size outside what the kernels support (ERANGE) */
#define NVRX_ERR_TIMEOUT (-62) /* wait timed out (ETIME) */

/* Row statistics layout: NVRX_STATS_STRIDE floats per row.
 * Mirrors the Statistic enum (statistics.py:19-35) + the kernel weight NUM*AVG (reporting.py:246). */
#define NVRX_STATS_STRIDE 8
#define NVRX_STAT_MIN 0
#define NVRX_STAT_MAX 1
#define NVRX_STAT_MED 2
#define NVRX_STAT_AVG 3
#define NVRX_STAT_STD 4
#define NVRX_STAT_NUM 5
#define NVRX_STAT_WEIGHT 6

/* Row kinds select which of the reference's two statistics conventions applies. */
#define NVRX_KIND_SECTION 0 /* straggler.py:185-195: LOWER median, UNBIASED std (NaN if n==1) */
#define NVRX_KIND_KERNEL 1  /* CuptiProfiler.cpp:44-74: mean-of-middles median, POPULATION std */

/* Largest row (samples per ring) the statistics kernel keeps resident in registers. */
#define NVRX_MAX_RING_CAP 65536
#define NVRX_MAX_ROWS 65536

/* Exchange-table row length for K kernel ids and S section ids (see nvrx_score). */
#define NVRX_TABLE_LEN(K, S) (2 * ((K) + (S)) + (K) + 1)
/* Score row length: {gpu_indiv, gpu_rel, indiv[S], rel[S]}  (reporting.py:353-360). */
#define NVRX_SCORE_LEN(S) (2 + 2 * (S))
/* Number of uint32 words of score-kernel metadata. */
#define NVRX_META_WORDS 8

typedef struct nvrx_ctx nvrx_ctx;

/* ------------------------------------------------------------------------------------------------
 * Library
 * ---------------------------------------------------------------------------------------------- */
int nvrx_abi_version(void);
/* Message of the last failing call made by the calling thread ("" if none). */
const char *nvrx_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Stateless operators (caller-owned device buffers).  These are the kernels; the context API below
 * only adds ring storage, staging and event timing around them.
 * ---------------------------------------------------------------------------------------------- */

/* Per-row statistics of `rows` timing rows, one workgroup per row.
 *   d_samples [rows][row_stride] f32, row_stride % 4 == 0 and base 16-byte aligned;
 *   d_counts  [rows] valid samples per row (clamped to row_stride; 0 => NaN stats, NUM 0);
 *   d_kinds   [rows] NVRX_KIND_* per row, or NULL for all-section;
 *   d_stats   [rows][NVRX_STATS_STRIDE] f32 out.
 * Replaces Detector._get_section_summaries (straggler.py:172-197: torch.tensor(deque) + min/max/
 * median/mean/std per section) and computeStats (CuptiProfiler.cpp:44-74: std::sort + stats per
 * kernel).  MIN/MAX/MED/NUM are exact; AVG/STD are accumulated in f64 and rounded to f32. */
int nvrx_row_stats(const float *d_samples, const uint32_t *d_counts, const uint8_t *d_kinds, int rows,
                   int row_stride, float *d_stats, void *stream);

/* Cross-rank scoring of the exchanged table, for all R ranks at once.
 *   d_table [R][L] f32, L = NVRX_TABLE_LEN(K,S); per rank r:
 *      med  [0, K+S)            MED per id (kernel ids first, then section ids); -1 = no stats
 *      hmin [K+S, 2(K+S))       running minimum of MED on rank r (individual-score reference)
 *      w    [2(K+S), 2(K+S)+K)  kernel weights NUM*AVG
 *      flag [L-1]               1.0 if rank r has ids for all its names, else 0.0
 *   thresholds[4] = {gpu_rel, section_rel, gpu_indiv, section_indiv} (Report.identify_stragglers
 *      argument order, reporting.py:84-90); NULL => 0.75 each;
 *   d_scores [R][NVRX_SCORE_LEN(S)] f32 out, NaN where the reference reports NaN / nothing.  When d_scores and
 *      d_flags are both 16-byte aligned they are written in 16-byte units (R <= 64: one workgroup scores the whole
 *      table; larger jobs: one workgroup per tile of 16 ranks): pad each array to a multiple of 16 bytes;
 *   d_flags  [R][NVRX_SCORE_LEN(S)] u8 out, 1 where score < threshold (strict; NaN never flagged);
 *   d_meta   [NVRX_META_WORDS] u32 out: {all ranks' name flags set, R, K, S, seq, seq of the statistics rows,
 *      wait for the rows [10 ns ticks], (last row -> scores staged) << 16 | (last row -> completion word) [10 ns ticks, 16 bits each]}
 *      (the last two: single-workgroup kernel only);
 *   d_done_counter  device word (zero before the first launch) or NULL.  When given, d_scores /
 *      d_flags / d_meta may point into pinned host memory (nvrx_host_alloc): after every block's
 *      results are visible system-wide the kernel stores `seq` into d_meta[4] with release
 *      semantics, so a host thread can wait with nvrx_poll_u32 instead of a stream sync + D2H.
 *   d_stats_src / d_stats_dst / stats_rows  optional: the kernel also forwards `stats_rows`
 *      statistics rows (16-byte aligned) from device memory to d_stats_dst (pinned host memory), so
 *      they arrive with the scores under the same completion word.
 * Replaces _all_reduce_times (reporting.py:255-296), _compute_sections_perf_scores (:196-217),
 * _compute_gpu_perf_score (:219-253), _get_tensor_from_scores/_get_scores_from_tensor (:338-380) and
 * the thresholding of Report.identify_stragglers (:84-151). */
int nvrx_score(const float *d_table, int R, int K, int S, int do_indiv, int do_rel,
               const double *thresholds, float *d_scores, uint8_t *d_flags, uint32_t *d_meta,
               uint32_t *d_done_counter, uint32_t seq, const float *d_stats_src, float *d_stats_dst,
               int stats_rows, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Context: device ring buffers + pinned staging + hipEvent timing for `local_ranks` logical ranks
 * of `rows_per_rank` rows each (one logical rank per GPU in production; several per GPU only when a
 * whole job is folded onto fewer GPUs for benchmarking).
 * Replaces CustomSection.cpu_elapsed_times deques (straggler.py:66-83), the native
 * unordered_map<string, CircularBuffer<float>> (CuptiProfiler.h:66, CircularBuffer.h:22-70) and the
 * CUPTI activity machinery (CuptiProfiler.cpp:96-207, BufferPool.cpp).
 * ---------------------------------------------------------------------------------------------- */
int nvrx_ctx_create(int device, int local_ranks, int rows_per_rank, int ring_cap, nvrx_ctx **out);
int nvrx_ctx_destroy(nvrx_ctx *ctx);
/* Stream used for flushes triggered implicitly by a full staging buffer (default: null stream). */
int nvrx_ctx_set_stream(nvrx_ctx *ctx, void *stream);
/* Geometry queries: 0 local_ranks, 1 rows_per_rank, 2 ring_cap, 3 row_stride, 4 device; diagnostics: 5 reports re-homed,
 * 6 verdict of the last re-homing decision, 7 GPU-timed regions skipped because their stream was being captured; 8 rows
 * of a logical rank handed out by nvrx_row_alloc so far. */
int nvrx_ctx_info(const nvrx_ctx *ctx, int what);

/* Hand out the next unused row of a logical rank (the same index in every logical rank of the context), configured as
 * `kind`, not exchanged: the entry of a section name / kernel key seen for the first time -- unordered_map::emplace of
 * CuptiProfiler.cpp:199-203, CustomSection creation of straggler.py:301-305.  Returns the row (>= 0) or NVRX_ERR_RANGE
 * when all rows_per_rank rows are taken.  Thread-safe: the per-kernel tracer calls it on its own thread. */
int nvrx_row_alloc(nvrx_ctx *ctx, int kind);
/* Row metadata: kind (NVRX_KIND_*) and position `gid` in the exchange table (-1 = not exchanged).
 * `row` indexes [0, local_ranks*rows_per_rank). Takes effect at the next flush. */
int nvrx_row_configure(nvrx_ctx *ctx, int row, int kind, int gid);

/* Append one sample (overwrite-oldest beyond ring_cap): deque.append (straggler.py:343) /
 * CircularBuffer::push_back (CircularBuffer.h:53-61).  Staged in pinned host memory; reaches the
 * device ring at the next flush.  O(1), no HIP call unless the staging buffer is full. */
int nvrx_ring_push(nvrx_ctx *ctx, int row, float value);
int nvrx_ring_push_many(nvrx_ctx *ctx, int row, const float *values, int n);
/* Append n (row, value) pairs in arrival order with ONE scatter launch on the context's stream, whatever the number of
 * rows they touch (rows[i] < 0 skips pair i).  Same ring semantics as n calls of nvrx_ring_push.  This is how the
 * per-kernel tracer's drained records reach the rings: the reference appends every CUPTI record to its key's
 * CircularBuffer on the host (CuptiProfiler.cpp:186-207, CircularBuffer.h:53-61). */
int nvrx_ring_push_pairs(nvrx_ctx *ctx, const int32_t *rows, const float *values, int n);
/* The same n pairs appended one by one and STAGED like nvrx_ring_push: nothing is launched unless a staging buffer fills
 * up; thread-safe.  This is the `push` of the per-kernel tracer's sink (include/nvrx_ktrace.h): the tracer's thread
 * appends every kernel record to its key's ring as the records arrive, the way the reference does on CUPTI's thread
 * (CuptiProfiler.cpp:168-207) -- the newest ring_cap durations per key survive (CircularBuffer.h:53-61). */
int nvrx_ring_push_staged(nvrx_ctx *ctx, const int32_t *rows, const float *values, int n);
/* nvrx_ring_push_staged / nvrx_row_alloc with the exact C types of nvrx_ktrace_sink's two function pointers (`void *ctx`):
 * these are the addresses to put into a sink. */
int nvrx_sink_push(void *ctx, const int32_t *rows, const float *values, int n);
int nvrx_sink_row_alloc(void *ctx, int kind);
/* Append n samples that already live in device memory (device-to-device, wraps as needed). */
int nvrx_ring_push_device(nvrx_ctx *ctx, int row, const float *d_values, int n, void *stream);
/* The same for n_rows consecutive rows at once: row first_row + r gets the n samples at d_values + r * ld (ld >= n).  Rows
 * that stand at the same ring position are written with one strided device copy per ring segment.  This is how a whole
 * [sections][samples] matrix (the reference's per-section deques, straggler.py:80-83) is handed over in one call. */
int nvrx_ring_push_device_rows(nvrx_ctx *ctx, int first_row, int n_rows, const float *d_values, int n, int ld, void *stream);
/* Declare that `row` currently holds n valid samples in its device ring (no data movement). */
int nvrx_ring_set_count(nvrx_ctx *ctx, int row, int n);
/* Same for every row of the context at once (benchmark re-arm of resident data). */
int nvrx_ring_set_count_all(nvrx_ctx *ctx, int n);
/* Valid samples in `row` (including staged ones), min(total pushed, ring_cap). */
int nvrx_ring_count(const nvrx_ctx *ctx, int row);
/* Valid-sample counts of rows [0, n) in one call. */
int nvrx_ring_counts(const nvrx_ctx *ctx, int32_t *out, int n);
/* 1 if the SET of rows among [0, n) that hold samples differs from what the previous call saw (or n does, or this is the
 * first call), else 0: a report's name tables -- which sections / kernels have a summary this window, straggler.py:185-195
 * leaves out what holds no samples -- only need rebuilding when this says so.  One call, nothing copied. */
int nvrx_ring_occupancy_changed(nvrx_ctx *ctx, int n);
/* Drop all samples of every row: deque.clear (straggler.py:223-225) / reset (CuptiProfiler.cpp:148-152).
 * History minima and row configuration are kept, as in the reference (reporting.py:186-191). */
int nvrx_ring_reset(nvrx_ctx *ctx);
/* Forget the history minima too (new Detector.initialize). */
int nvrx_history_reset(nvrx_ctx *ctx, void *stream);
/* Push staged samples + row metadata to the device (one kernel reading pinned memory). */
int nvrx_ring_flush(nvrx_ctx *ctx, void *stream);
/* Debug/test: copy a row's ring storage (row_stride floats) to host after a flush. */
int nvrx_ring_read(nvrx_ctx *ctx, int row, float *out, int n, void *stream);

/* GPU timing of a code region with a hipEvent pair on `stream`: the MI355X replacement for the
 * CUPTI activity records (CuptiProfiler.cpp:116-133,168-207).  Elapsed time is appended to `row`
 * in MICROSECONDS f32 (CuptiProfiler.cpp:191) when harvested.  begin/end may nest. */
int nvrx_event_begin(nvrx_ctx *ctx, int row, void *stream);
int nvrx_event_end(nvrx_ctx *ctx, int row, void *stream);
/* Move completed event pairs into their rows.  wait!=0 blocks until every ended pair completes
 * (the role of torch.cuda.synchronize() in straggler.py:234, without a device-wide sync).
 * Returns the number of pairs still pending (>=0) or a negative error. */
int nvrx_event_harvest(nvrx_ctx *ctx, int wait);

/* GPU time of a code region measured on the device (replaces the CUPTI activity records of
 * cupti_src/CuptiProfiler.cpp:168-207 without a hipEvent read-back): nvrx_stamp_begin enqueues a one-thread
 * kernel on `stream` that stores the constant-rate wall clock; nvrx_stamp_end enqueues one that appends the
 * elapsed MICROSECONDS (CuptiProfiler.cpp:191) to `row`'s ring and, when cpu_row >= 0, also appends the
 * host-measured `cpu_value` to `cpu_row`'s ring (the section's wall time, straggler.py:343), so a profiled
 * section entry costs no pinned-memory staging.  Reports are ordered after these kernels on the device;
 * the host never waits for them.  Regions on one row nest LIFO.
 * A region whose stream is being captured into a hipGraph (hipStreamIsCapturing) records nothing -- a captured timestamp
 * would replay into one fixed ring slot, and the reference sees no kernel during a capture either: begin AND the matching
 * end return NVRX_REGION_SKIPPED (> 0), nothing is enqueued, cpu_value is NOT taken (the caller pushes it itself).  The
 * same holds for nvrx_event_begin / nvrx_event_end.  nvrx_ctx_info(ctx, 7) counts such regions. */
#define NVRX_REGION_SKIPPED 1
int nvrx_stamp_begin(nvrx_ctx *ctx, int row, void *stream);
int nvrx_stamp_end(nvrx_ctx *ctx, int row, int cpu_row, float cpu_value, void *stream);

/* Local half of a report: flush -> row statistics for every row -> write this GPU's
 * `local_ranks` exchange rows.
 *   d_stats [local_ranks*rows_per_rank][NVRX_STATS_STRIDE] out
 *   d_send  [local_ranks][L] out, L = NVRX_TABLE_LEN(K,S) (rows without a gid are not exchanged);
 *           NULL = statistics only: nothing is exchanged and the history minima are left alone
 *   names_ok: value of the flag word for these ranks (has ids for all names);
 *   rows_active: only rows [0, rows_active) of every logical rank are processed (0 = all).
 * Replaces straggler.py:236-237 + the packing loops of reporting.py:273-279. Also folds in
 * _update_local_min_times (reporting.py:298-314): hmin[row] = min(hmin[row], MED). */
int nvrx_report_local(nvrx_ctx *ctx, float *d_stats, float *d_send, int K, int S, int names_ok,
                      int rows_active, void *stream);
/* A whole report in ONE call: flush -> row statistics -> [all-gather of the exchange rows] -> scoring of the
 * table -> (optionally) wait for the completion word.  The descriptor is filled once per report shape and reused;
 * the library advances `seq` itself, so a steady-state report is a single FFI crossing.
 * `stream` is where the report runs unless it is RE-HOMED: a synchronous report (h_seq_word given, guard_rings 0) that
 * must follow the work of exactly one other stream -- the stream the window's region stamps (nvrx_stamp_end) were
 * launched on and / or order_after_stream -- is enqueued on THAT stream instead, so that the stream order is the
 * dependency and no event is recorded or waited for (NVRX_DEBUG_REPORT_REHOME=0 keeps it on `stream` behind event waits).
 * Replaces Detector.generate_report's body (straggler.py:236-239) + ReportGenerator.generate_report
 * (reporting.py:421-554) for the case where the summaries never leave the device. */
typedef struct nvrx_report_desc {
    int32_t R, K, S;          /* table shape: ranks, kernel ids, section ids */
    int32_t names_ok;         /* this process has ids for all of its names */
    int32_t rows_active;      /* rows per logical rank to process (0 = all) */
    int32_t do_indiv, do_rel; /* score families to compute */
    int32_t stats_rows;       /* statistics rows to forward to d_stats_dst */
    double thresholds[4];     /* gpu_rel, section_rel, gpu_indiv, section_indiv */
    float *d_stats;           /* [local_ranks*rows_per_rank][NVRX_STATS_STRIDE] device */
    float *d_send;            /* [local_ranks][L] device: this process' exchange rows */
    float *d_table;           /* [R][L] device: all ranks' rows (ignored when allgather_fn is NULL) */
    float *d_scores;          /* result block, device-visible addresses (nvrx_host_alloc's *out_device):    */
    uint8_t *d_flags;         /*   16-byte aligned, each array padded to a multiple of 16 bytes              */
    uint32_t *d_meta;         /*   NVRX_META_WORDS words; [4] is the completion word                         */
    float *d_stats_dst;       /*   or NULL                                                                   */
    uint32_t *d_done_counter; /* device word, zero before the first report */
    void *allgather_fn;       /* NULL = single process, no exchange; else an ncclAllGather-compatible function
                                 int (*)(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream) */
    void *comm;               /* its communicator */
    int32_t send_count;       /* floats this process contributes (local_ranks * L) */
    uint32_t seq;             /* last published sequence number (in/out) */
    const uint32_t *h_seq_word; /* host address of meta[4]; NULL = do not wait */
    double timeout_s;         /* <= 0: wait for ever, like a blocking collective */
    void *order_after_stream; /* hipStream_t whose already-enqueued work the report must follow (device-side wait,
                                 the host does not block), or NULL; takes the place of torch.cuda.synchronize()
                                 in straggler.py:234 for the collectives the caller has enqueued there */
    int32_t order_after_enabled; /* 0: order_after_stream is ignored */
    int32_t resident;         /* nonzero: the library may run the score kernel RESIDENT on a stream of its own next to the
                                 statistics kernel (rows handed over as 8-byte tagged granules instead of through the
                                 stream order); used for synchronous reports without an exchange or with the peer-window
                                 exchange that have no other stream's work to wait for (NVRX_DEBUG_RESIDENT_SCORER=0|1|2: never /
                                 that rule / always).  The statistics rows then land under their own completion word d_meta[5]. */
    int32_t prev_settled;     /* nonzero: the caller has seen this context's previous asynchronous report complete (it polled
                                 that report's completion word): ring writers need not wait for it any more, and an
                                 asynchronous report that comes rarely enough may be enqueued on the one stream it follows */
    int32_t guard_rings;      /* asynchronous reports (h_seq_word == NULL, the caller polls later): nonzero makes later
                                 device-side ring writers on other streams (nvrx_stamp_end) wait, on the device, for this
                                 report's statistics kernel */
} nvrx_report_desc;
int nvrx_report(nvrx_ctx *ctx, nvrx_report_desc *desc, void *stream);
/* One report WINDOW in one call: what straggler.py:228-244 does around the report in the steady state -- wait for the
 * window's GPU measurements (torch.cuda.synchronize() + the profiler's get_stats there; here the kernel tracer's sync, or a
 * harvest of the region events), check that the set of rows holding samples is the one the caller's name tables were built
 * for, run nvrx_report, empty the rings (straggler.py:241-242).  The kernel tracer lives in another library
 * (include/nvrx_ktrace.h): its three entry points are handed over as plain function pointers.
 * Returns NVRX_OK (the report ran -- for asynchronous ones: was enqueued -- and the rings are empty), NVRX_WINDOW_MISS
 * (NOTHING ran: dispatches still missing after the patience, kernel keys the host has not learnt yet, or the occupied rows
 * changed -- the caller takes its general path), NVRX_WINDOW_NAMES (synchronous only: the report ran and its table says some
 * rank has names without ids; the rings were NOT emptied -- the caller syncs names and reports again), or a negative error. */
#define NVRX_WINDOW_MISS 1
#define NVRX_WINDOW_NAMES 2
typedef struct nvrx_window_desc {
    int (*kt_sync)(double timeout_s);  /* nvrx_ktrace_sync, or NULL: GPU time is measured per region */
    int (*kt_hold)(int on);            /* nvrx_ktrace_hold: brackets "statistics launch + ring reset" of an asynchronous report */
    uint64_t (*kt_counter)(int what);  /* nvrx_ktrace_counter */
    double kt_patience_s;              /* how long a synchronous report waits for the window's kernel records */
    uint64_t kt_rows_known;            /* the host's copies of tracer counters 10 and 6: a difference = kernel keys to learn */
    uint64_t kt_keys_without_row;
    int32_t rows_used;                 /* rows the caller's name tables cover (as for nvrx_ring_occupancy_changed) */
    int32_t asynchronous;              /* nonzero: desc->h_seq_word is NULL, the report is only enqueued, nothing is waited for */
    int32_t harvest_regions;           /* region timing: nvrx_event_harvest before the report */
    int32_t out_names_ok;              /* out: 0 when NVRX_WINDOW_NAMES was returned */
} nvrx_window_desc;
int nvrx_window_report(nvrx_ctx *ctx, nvrx_report_desc *desc, void *stream, nvrx_window_desc *window);
/* Diagnostics: host clocks of this thread's last nvrx_window_report, microseconds on the monotonic clock -- [0] entry, [1]
 * the wait for the window's measurements and the occupancy look are done, nvrx_report begins (its own clocks follow). */
int nvrx_window_clocks(double *out2);
/* Diagnostics: host clocks of this thread's last nvrx_report, microseconds on the monotonic clock -- [0] entry, [1] stream
 * ordering done, [2] staged samples flushed, [3] statistics kernel launched, [4] exchange enqueued, [5] score kernel
 * launched, [6] completion word seen (synchronous reports), [7] spare.  No counterpart in the reference. */
int nvrx_report_clocks(double *out8);
/* sizeof(nvrx_report_desc) as this library was compiled: lets an FFI binding check its own struct layout. */
int nvrx_report_desc_size(void);
/* ------------------------------------------------------------------------------------------------
 * Peer-window exchange: the report's one collective as direct xGMI peer stores between the processes of ONE node.
 * Replaces all_reduce(MIN) + gather of reporting.py:281,397 (as the RCCL all-gather does) without an RCCL kernel:
 * every process owns a window of 8-byte {epoch, f32} granules in fine-grained device memory, mapped by all
 * processes through HIP IPC; nvrx_peer_allgather enqueues ONE single-workgroup kernel that stores this process'
 * floats into every window and polls its own window until every rank's floats of this epoch have arrived.
 * Collective set-up (cold): create -> exchange the 64-byte IPC handles out of band -> connect each peer -> ready.
 * Every process must call nvrx_peer_allgather the same number of times (the epoch is counted, not passed).
 * ---------------------------------------------------------------------------------------------- */
typedef struct nvrx_peer nvrx_peer;
int nvrx_peer_create(int device, int world, int rank, int max_floats_per_rank, nvrx_peer **out);
int nvrx_peer_ipc_handle(nvrx_peer *peer, void *handle64 /* 64 bytes out */);
int nvrx_peer_connect(nvrx_peer *peer, int peer_rank, const void *handle64);
/* PCI bus id of the window's device ("0000:05:00.0"): the same string in every process, whatever HIP_VISIBLE_DEVICES. */
int nvrx_peer_device_id(const nvrx_peer *peer, char *out, int len);
/* Before nvrx_peer_connect: can this window's device store into the device named by peer_pci_bus_id?  0 yes (or the same
 * device); 1 the peer's device is not visible to this process (nothing to check, the IPC mapping decides); negative with
 * nvrx_last_error() saying why not (hipDeviceCanAccessPeer = 0).  The reference has no counterpart: its exchange is
 * torch.distributed (reporting.py:281,397), whose transport NCCL picks. */
int nvrx_peer_check_access(const nvrx_peer *peer, int peer_rank, const char *peer_pci_bus_id);
/* timeout_s: how long the kernel polls for a late peer before it gives up (<= 0 keeps the default 1800 s). */
int nvrx_peer_ready(nvrx_peer *peer, double timeout_s);
/* ncclAllGather-compatible: (send, recv, floats per rank, dtype = 7 (f32), comm = the nvrx_peer, stream); returns 0
 * or a positive code.  Usable as nvrx_report_desc.allgather_fn (address: nvrx_peer_allgather_address()). */
int nvrx_peer_allgather(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream);
void *nvrx_peer_allgather_address(void);
/* Epoch of the last exchange in which the kernel gave up waiting for a peer (0 = never). */
int nvrx_peer_error(const nvrx_peer *peer, uint32_t *epoch_out);
int nvrx_peer_destroy(nvrx_peer *peer);

/* Re-initialise an exchange buffer with the "no stats" sentinels (call when ids change). */
int nvrx_send_init(float *d_send, int rows, int K, int S, void *stream);

/* Kernel-time instrumentation for the benchmark: when enabled, nvrx_report_local launches its
 * statistics kernel through hipExtLaunchKernel with a start/stop hipEvent pair, which receive the
 * kernel's own begin/end timestamps on the launch stream; totals are read back here (blocks until
 * the recorded events complete). */
int nvrx_timing_enable(nvrx_ctx *ctx, int on);
int nvrx_timing_read(nvrx_ctx *ctx, double *total_us, int *launches, int reset);

/* Small asynchronous D2H into pinned memory + completion tracking for the report results. */
/* Pinned, device-mapped host memory; *out_device (optional) receives the address kernels use. */
int nvrx_host_alloc(void **out, void **out_device, size_t bytes);
/* Zero-initialised device memory on the current device, for hosts without an allocator of their own (the Python package
 * hands over tensor.data_ptr(); tests/c_abi/abi_host.c, plain C, uses these).  No counterpart in the reference. */
int nvrx_device_alloc(void **out, size_t bytes);
int nvrx_device_free(void *p);
/* Spin until *h_word == expected (acquire); NVRX_ERR_TIMEOUT after timeout_s seconds. */
int nvrx_poll_u32(const uint32_t *h_word, uint32_t expected, double timeout_s);
int nvrx_host_free(void *p);
int nvrx_copy_to_host(nvrx_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream);
/* Stateless: asynchronous D2H on `stream`, then wait for the stream (report results -> pinned host). */
int nvrx_d2h_sync(void *h_dst, const void *d_src, size_t bytes, void *stream);
/* Block (spin) until the last nvrx_copy_to_host on this context has landed. */
int nvrx_wait(nvrx_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* NVRX_STRAGGLER_H */
