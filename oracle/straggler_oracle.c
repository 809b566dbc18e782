/*
 * straggler_oracle.c -- CPU restatement of the reference's straggler-scoring arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package under
 * nvidia-resiliency-ext_amd/) may import, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity pinning: the functions below are checked against
 *   (a) golden vectors produced by importing the reference's own Python code in the build
 *       container (tests/golden/make_golden.py -> tests/golden/*.json), and
 *   (b) the reference's own C++ computeStats()/CircularBuffer compiled from where they lie
 *       (oracle/Makefile target `ref` -> oracle/_ref/libnvrx_ref.so), when /root/reference exists.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src/nvidia_resiliency_ext/attribution/straggler/).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* --------------------------------------------------------------------------------------------
 * Ring buffer: fixed capacity, overwrite-oldest, linearize() returns oldest -> newest.
 * Restates cupti_src/CircularBuffer.h:22-70 (push_back :53-61, linearize :63-69) and the
 * Python deque(maxlen=8192) used for sections (straggler.py:80-83).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float *buf;
    size_t head, tail, size, cap;
} oracle_ring;

oracle_ring *oracle_ring_new(size_t cap) {
    oracle_ring *r = (oracle_ring *)calloc(1, sizeof(*r));
    r->buf = (float *)calloc(cap ? cap : 1, sizeof(float));
    r->cap = cap;
    return r;
}

void oracle_ring_free(oracle_ring *r) {
    if (r) {
        free(r->buf);
        free(r);
    }
}

void oracle_ring_push(oracle_ring *r, float v) {
    /* CircularBuffer.h:53-61: write at tail, advance tail, if already full advance head too. */
    r->buf[r->tail] = v;
    r->tail = (r->tail + 1) % r->cap;
    if (r->size == r->cap) {
        r->head = (r->head + 1) % r->cap;
    } else {
        r->size++;
    }
}

size_t oracle_ring_size(const oracle_ring *r) { return r->size; }

size_t oracle_ring_linearize(const oracle_ring *r, float *out) {
    /* CircularBuffer.h:63-69 */
    for (size_t i = 0; i < r->size; i++) out[i] = r->buf[(r->head + i) % r->cap];
    return r->size;
}

void oracle_ring_clear(oracle_ring *r) { r->head = r->tail = r->size = 0; }

/* --------------------------------------------------------------------------------------------
 * Section statistics.  Restates Detector._get_section_summaries, straggler.py:185-195:
 *   t = torch.tensor(deque, dtype=float64); MIN=torch.min, MAX=torch.max,
 *   MED=torch.median (LOWER median: sorted[(n-1)/2]), AVG=torch.mean,
 *   STD=torch.std (unbiased, n-1) if n>1 else NaN, NUM=n.
 * out = {MIN, MAX, MED, AVG, STD, NUM}.  n==0 rows are skipped by the caller (straggler.py:183).
 * ------------------------------------------------------------------------------------------ */
static int cmp_f64(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

void oracle_section_stats(const double *x, int n, double *out) {
    if (n <= 0) {
        for (int i = 0; i < 5; i++) out[i] = NAN;
        out[5] = 0.0;
        return;
    }
    double *s = (double *)malloc((size_t)n * sizeof(double));
    memcpy(s, x, (size_t)n * sizeof(double));
    qsort(s, (size_t)n, sizeof(double), cmp_f64);
    out[0] = s[0];
    out[1] = s[n - 1];
    out[2] = s[(n - 1) / 2]; /* torch.median returns the lower of the two middles for even n */
    long double sum = 0.0L;
    for (int i = 0; i < n; i++) sum += x[i];
    double mean = (double)(sum / n);
    out[3] = mean;
    if (n > 1) {
        long double sq = 0.0L;
        for (int i = 0; i < n; i++) sq += ((long double)x[i] - mean) * ((long double)x[i] - mean);
        out[4] = (double)sqrtl(sq / (n - 1)); /* Bessel-corrected, torch.std default */
    } else {
        out[4] = NAN; /* straggler.py:192-193 */
    }
    out[5] = (double)n;
    free(s);
}

/* --------------------------------------------------------------------------------------------
 * Kernel (GPU-timed) statistics.  Restates computeStats, cupti_src/CuptiProfiler.cpp:44-74,
 * INCLUDING its float32 arithmetic:
 *   sort; min=front; max=back; median = (s[n/2-1]+s[n/2])/2 for even n else s[n/2] (:56-61);
 *   avg = std::accumulate(sorted, 0.0f)/n  -- sequential f32 accumulation over SORTED data (:63);
 *   stddev = sqrt( accumulate((x-avg)^2, 0.0f) / n ) -- population, f32 (:64-69); num_calls=n.
 * Empty input: all NaN, num_calls 0 (CuptiProfiler.h:39-44).
 * out = {min, max, median, avg, stddev}; returns num_calls.
 * ------------------------------------------------------------------------------------------ */
static int cmp_f32(const void *a, const void *b) {
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

int oracle_kernel_stats(const float *x, int n, float *out) {
    if (n <= 0) {
        for (int i = 0; i < 5; i++) out[i] = NAN;
        return 0;
    }
    float *s = (float *)malloc((size_t)n * sizeof(float));
    memcpy(s, x, (size_t)n * sizeof(float));
    qsort(s, (size_t)n, sizeof(float), cmp_f32);
    out[0] = s[0];
    out[1] = s[n - 1];
    if (n % 2 == 0) {
        volatile float t = s[n / 2 - 1] + s[n / 2];
        out[2] = t / 2;
    } else {
        out[2] = s[n / 2];
    }
    volatile float acc = 0.0f;
    for (int i = 0; i < n; i++) acc = acc + s[i];
    float avg = acc / (float)n; /* size_t n converts to float in `float / size_t` */
    out[3] = avg;
    volatile float sq = 0.0f;
    for (int i = 0; i < n; i++) {
        volatile float d = s[i] - avg;
        volatile float dd = d * d;
        sq = sq + dd;
    }
    out[4] = sqrtf(sq / (float)n);
    free(s);
    return n;
}

/* --------------------------------------------------------------------------------------------
 * Batched helpers used by the parity tests and the cpu_baseline leg of bench.py.
 * samples: [rows][stride] f32 (our ring layout), counts[rows] valid entries per row.
 * kinds[row]: 0 = section semantics (f64 pipeline), 1 = kernel semantics (f32 pipeline).
 * out: [rows][6] f64 = {MIN, MAX, MED, AVG, STD, NUM}.
 * ------------------------------------------------------------------------------------------ */
void oracle_rows_stats(const float *samples, const uint32_t *counts, const uint8_t *kinds, int rows,
                       int stride, double *out) {
    double *tmp = (double *)malloc((size_t)(stride > 0 ? stride : 1) * sizeof(double));
    for (int r = 0; r < rows; r++) {
        int n = (int)counts[r];
        const float *x = samples + (size_t)r * (size_t)stride;
        double *o = out + (size_t)r * 6;
        if (kinds && kinds[r]) {
            float f[5];
            int num = oracle_kernel_stats(x, n, f);
            for (int i = 0; i < 5; i++) o[i] = (double)f[i];
            o[5] = (double)num;
        } else {
            for (int i = 0; i < n; i++) tmp[i] = (double)x[i];
            oracle_section_stats(tmp, n, o);
        }
    }
    free(tmp);
}

/* --------------------------------------------------------------------------------------------
 * Cross-rank scoring on the exchanged table.  Restates, for ALL ranks at once, what each rank's
 * ReportGenerator computes and rank 0 gathers (reporting.py):
 *   _all_reduce_times   :255-296  f32 tensor filled -1, all_reduce(MIN), <0 -> NaN
 *   _compute_sections_perf_scores :196-217   score = reference / MED
 *   _compute_gpu_perf_score       :219-253   sum(score*w)/sum(w), w = NUM*AVG, skip NaN refs,
 *                                            NaN when no kernels or no common kernels
 *   _update_local_min_times       :298-314   (history minima are an INPUT here: hmin)
 *   _get_tensor_from_scores       :338-360   results rounded to f32, NaN default
 *
 * table: [R][L] f32, L = 2*(K+S) + K + 1, per rank r:
 *   med  [0, K+S)        kernel ids first then section ids; -1 = "rank has no stats" sentinel
 *   hmin [K+S, 2(K+S))   this rank's running minimum of MED (individual-score reference)
 *   w    [2(K+S), 2(K+S)+K)  kernel weights NUM*AVG
 *   flag [L-1]           names-complete flag (not used by the scoring)
 * scores: [R][2+2S] f32 = {gpu_indiv, gpu_rel, indiv[S], rel[S]} (reporting.py:353-360 order).
 * Reference precision: scores are computed in f64 from f64 MED and f32-rounded references and
 * then stored to f32; here MED itself is already f32 (our exchange format), so agreement with
 * the reference is to f32 rounding (<= ~2e-7 relative), far inside the 1e-4 contract.
 * ------------------------------------------------------------------------------------------ */
void oracle_score_table(const float *table, int R, int K, int S, int do_indiv, int do_rel,
                        float *scores) {
    const int KS = K + S;
    const int L = 2 * KS + K + 1;
    const int W = 2 + 2 * S;
    double *minmed = (double *)malloc((size_t)(KS > 0 ? KS : 1) * sizeof(double));
    for (int j = 0; j < KS; j++) {
        float m = INFINITY;
        for (int r = 0; r < R; r++) {
            float v = table[(size_t)r * L + j];
            if (v < m) m = v; /* all_reduce MIN over f32, -1 sentinel wins */
        }
        minmed[j] = (m >= 0.0f) ? (double)m : NAN; /* reporting.py:289,295 */
    }
    for (int r = 0; r < R; r++) {
        const float *row = table + (size_t)r * L;
        float *out = scores + (size_t)r * W;
        for (int i = 0; i < W; i++) out[i] = NAN;
        /* GPU scores */
        double wi = 0.0, si = 0.0, wr = 0.0, sr = 0.0;
        int nk = 0, ncommon = 0;
        for (int k = 0; k < K; k++) {
            double med = (double)row[k];
            if (!(med >= 0.0)) continue; /* rank has no such kernel */
            double w = (double)row[2 * KS + k];
            nk++;
            si += ((double)row[KS + k] / med) * w;
            wi += w;
            if (!isnan(minmed[k])) {
                ncommon++;
                sr += (minmed[k] / med) * w;
                wr += w;
            }
        }
        if (do_indiv && nk > 0) out[0] = (float)(si / wi);
        if (do_rel && ncommon > 0) out[1] = (float)(sr / wr);
        for (int s = 0; s < S; s++) {
            double med = (double)row[K + s];
            if (!(med >= 0.0)) continue; /* section absent on this rank -> NaN default */
            if (do_indiv) out[2 + s] = (float)((double)row[KS + K + s] / med);
            if (do_rel) out[2 + S + s] = (float)(minmed[K + s] / med);
        }
    }
    free(minmed);
}
