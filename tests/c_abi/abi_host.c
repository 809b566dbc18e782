/*
 * abi_host.c -- a host of the C ABI that is neither Python nor PyTorch: plain C99 built by gcc against
 * include/nvrx_straggler.h, linked with libnvrx_straggler_hip.so (and, as the checker, the C oracle).  It does what a
 * reference-side binding would do for one report of a small job: rings -> row statistics -> exchange rows
 * -> scores and flags, with every buffer obtained through the ABI itself (nvrx_host_alloc: pinned, device-mapped), so the
 * program needs no HIP header.  TEST INFRASTRUCTURE (tests/test_gpu_c_abi.py runs it; `abi_host --link-only` is what the
 * CPU suite runs: it proves the binary resolves every symbol it uses without touching a device).
 *
 * Job: 3 logical ranks folded on one GPU (local_ranks = 3), per rank one kernel row (gid 0) and three section rows
 * (gids 1..3); rank 2's section 1 and kernel are 1.5x slower.  Checked:
 *   - ring contents after overwrite-oldest appends == oracle_ring (CircularBuffer.h:53-69, straggler.py:80-83);
 *   - section rows: MIN/MAX/MED/NUM exact, AVG/STD 1e-6 vs oracle_section_stats (straggler.py:185-195);
 *   - kernel rows:  MIN/MAX/MED/NUM exact, AVG/STD 2e-4 vs oracle_kernel_stats (CuptiProfiler.cpp:44-74);
 *   - relative section scores == (float)(min over ranks of MED / MED) bit for bit (reporting.py:196-217,255-296), the
 *     relative GPU score within 2e-6 of the weighted mean of reporting.py:219-253, individual scores 1.0 on a first
 *     report (reporting.py:298-314), flags == score < threshold;
 *   - the same report as ONE call (nvrx_report, descriptor reused three times, buffers from nvrx_device_alloc): scores,
 *     flags and forwarded statistics rows byte-identical to the two-call path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nvrx_straggler.h"

/* oracle/straggler_oracle.c (no header: the oracle is not a product interface) */
typedef struct oracle_ring oracle_ring;
oracle_ring *oracle_ring_new(size_t cap);
void oracle_ring_free(oracle_ring *r);
void oracle_ring_push(oracle_ring *r, float v);
size_t oracle_ring_linearize(const oracle_ring *r, float *out);
void oracle_section_stats(const double *x, int n, double *out);
int oracle_kernel_stats(const float *x, int n, float *out);

#define RANKS 3
#define ROWS_PER_RANK 4 /* row 0: the kernel key, rows 1..3: sections */
#define NK 1
#define NS 3
#define CAP 100

static int failures = 0;
#define CHECK(cond, ...)                                       \
    do {                                                       \
        if (!(cond)) {                                         \
            failures++;                                        \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                      \
            fprintf(stderr, "\n");                             \
        }                                                      \
    } while (0)
#define TRY(call)                                                                        \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ < 0) {                                                                   \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nvrx_last_error());            \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

static int cmp_f32(const void *a, const void *b) {
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

static double rel_err(double got, double want) {
    if (isnan(got) && isnan(want)) return 0.0;
    return fabs(got - want) / fmax(fabs(want), 1e-30);
}

/* a small deterministic generator: the samples only have to be the same on both sides */
static uint32_t lcg_state = 12345u;
static float next_sample(float centre) {
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return centre * (0.9f + 0.2f * (float)(lcg_state >> 8) / 16777216.0f);
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "--link-only") == 0) {
        printf("abi version %d, report descriptor %d bytes (header: %zu)\n", nvrx_abi_version(), nvrx_report_desc_size(),
               sizeof(nvrx_report_desc));
        return (nvrx_abi_version() == NVRX_ABI_VERSION && nvrx_report_desc_size() == (int)sizeof(nvrx_report_desc)) ? 0 : 1;
    }

    nvrx_ctx *ctx = NULL;
    TRY(nvrx_ctx_create(0, RANKS, ROWS_PER_RANK, CAP, &ctx));
    const int rows = RANKS * ROWS_PER_RANK;
    const int stride = nvrx_ctx_info(ctx, 3);

    /* ---- fill the rings: different lengths per row, one of them past the capacity (overwrite-oldest) ---- */
    const int lengths[ROWS_PER_RANK] = {37, 64, 250, 1};
    oracle_ring *rings[RANKS * ROWS_PER_RANK];
    for (int r = 0; r < RANKS; r++)
        for (int i = 0; i < ROWS_PER_RANK; i++) {
            const int row = r * ROWS_PER_RANK + i;
            TRY(nvrx_row_configure(ctx, row, i == 0 ? NVRX_KIND_KERNEL : NVRX_KIND_SECTION, i));
            rings[row] = oracle_ring_new(CAP);
            const float slow = (r == 2 && i <= 1) ? 1.5f : 1.0f;
            float tmp[256];
            for (int j = 0; j < lengths[i]; j++) {
                tmp[j] = next_sample(10.0f * (float)(i + 1) * slow);
                oracle_ring_push(rings[row], tmp[j]);
            }
            /* half of them one by one, the rest in one call: both entry points, one ring */
            for (int j = 0; j < lengths[i] / 2; j++) TRY(nvrx_ring_push(ctx, row, tmp[j]));
            TRY(nvrx_ring_push_many(ctx, row, tmp + lengths[i] / 2, lengths[i] - lengths[i] / 2));
        }

    /* ---- buffers through the ABI: pinned host memory the kernels address directly ---- */
    const int L = NVRX_TABLE_LEN(NK, NS), W = NVRX_SCORE_LEN(NS);
    float *h_stats, *d_stats, *h_send, *d_send, *h_scores, *d_scores;
    uint8_t *h_flags, *d_flags;
    uint32_t *h_meta, *d_meta;
    TRY(nvrx_host_alloc((void **)&h_stats, (void **)&d_stats, (size_t)rows * NVRX_STATS_STRIDE * sizeof(float)));
    TRY(nvrx_host_alloc((void **)&h_send, (void **)&d_send, (size_t)RANKS * L * sizeof(float)));
    TRY(nvrx_host_alloc((void **)&h_scores, (void **)&d_scores, (size_t)RANKS * W * sizeof(float) + 16));
    TRY(nvrx_host_alloc((void **)&h_flags, (void **)&d_flags, (size_t)RANKS * W + 16));
    TRY(nvrx_host_alloc((void **)&h_meta, (void **)&d_meta, NVRX_META_WORDS * sizeof(uint32_t)));

    /* ---- one report: statistics + exchange rows, then the scores of the table (3 local ranks = the whole job) ---- */
    TRY(nvrx_send_init(d_send, RANKS, NK, NS, NULL));
    TRY(nvrx_report_local(ctx, d_stats, d_send, NK, NS, 1, 0, NULL));
    const double thresholds[4] = {0.75, 0.75, 0.75, 0.75};
    TRY(nvrx_score(d_send, RANKS, NK, NS, 1, 1, thresholds, d_scores, d_flags, d_meta, NULL, 1u, NULL, NULL, 0, NULL));
    float *stats = (float *)malloc((size_t)rows * NVRX_STATS_STRIDE * sizeof(float));
    TRY(nvrx_d2h_sync(stats, d_stats, (size_t)rows * NVRX_STATS_STRIDE * sizeof(float), NULL)); /* waits for the stream */

    /* ---- rings and statistics against the oracle ---- */
    double med[RANKS][NK + NS], weight[RANKS];
    float *ring_dev = (float *)malloc((size_t)stride * sizeof(float));
    for (int row = 0; row < rows; row++) {
        float lin[CAP];
        const int n = (int)oracle_ring_linearize(rings[row], lin);
        CHECK(nvrx_ring_count(ctx, row) == n, "row %d count %d != %d", row, nvrx_ring_count(ctx, row), n);
        TRY(nvrx_ring_read(ctx, row, ring_dev, stride, NULL));
        float a[CAP], b[CAP];
        memcpy(a, lin, (size_t)n * sizeof(float));
        memcpy(b, ring_dev, (size_t)n * sizeof(float));
        qsort(a, (size_t)n, sizeof(float), cmp_f32);
        qsort(b, (size_t)n, sizeof(float), cmp_f32);
        CHECK(memcmp(a, b, (size_t)n * sizeof(float)) == 0, "row %d: the device ring does not hold the newest %d samples", row, n);

        const float *st = stats + (size_t)row * NVRX_STATS_STRIDE;
        const int r = row / ROWS_PER_RANK, i = row % ROWS_PER_RANK;
        if (i == 0) {
            float e[5];
            const int calls = oracle_kernel_stats(lin, n, e);
            CHECK(st[NVRX_STAT_MIN] == e[0] && st[NVRX_STAT_MAX] == e[1] && st[NVRX_STAT_MED] == e[2] && (int)st[NVRX_STAT_NUM] == calls,
                  "kernel row %d: min/max/med/num %g %g %g %g vs %g %g %g %d", row, st[0], st[1], st[2], st[5], e[0], e[1], e[2], calls);
            CHECK(rel_err(st[NVRX_STAT_AVG], e[3]) < 2e-4 && rel_err(st[NVRX_STAT_STD], e[4]) < 2e-4, "kernel row %d: avg/std %g %g vs %g %g",
                  row, st[3], st[4], e[3], e[4]);
            weight[r] = (double)st[NVRX_STAT_NUM] * (double)st[NVRX_STAT_AVG];
        } else {
            double x[CAP], e[6];
            for (int j = 0; j < n; j++) x[j] = (double)lin[j];
            oracle_section_stats(x, n, e);
            CHECK(st[NVRX_STAT_MIN] == (float)e[0] && st[NVRX_STAT_MAX] == (float)e[1] && st[NVRX_STAT_MED] == (float)e[2] &&
                      (int)st[NVRX_STAT_NUM] == (int)e[5],
                  "section row %d: min/max/med/num %g %g %g %g vs %g %g %g %g", row, st[0], st[1], st[2], st[5], e[0], e[1], e[2], e[5]);
            CHECK(rel_err(st[NVRX_STAT_AVG], e[3]) < 1e-6, "section row %d: avg %g vs %g", row, st[3], e[3]);
            CHECK(n == 1 ? isnan(st[NVRX_STAT_STD]) : rel_err(st[NVRX_STAT_STD], e[4]) < 1e-6, "section row %d: std %g vs %g", row, st[4], e[4]);
        }
        med[r][i] = (double)st[NVRX_STAT_MED];
    }

    /* ---- scores: the table is what nvrx_report_local packed; expectations restated from the reference's formulas ---- */
    for (int r = 0; r < RANKS; r++) {
        const float *sc = h_scores + (size_t)r * W; /* {gpu_indiv, gpu_rel, indiv[NS], rel[NS]} */
        const uint8_t *fl = h_flags + (size_t)r * W;
        for (int s = 0; s < NS; s++) {
            double ref = med[0][NK + s];
            for (int q = 1; q < RANKS; q++) ref = fmin(ref, med[q][NK + s]);
            const float want = (float)(ref / med[r][NK + s]); /* reporting.py:208-214 on the MIN all-reduced medians (:281) */
            CHECK(sc[2 + NS + s] == want, "rank %d section %d: relative score %.9g != %.9g", r, s, sc[2 + NS + s], want);
            CHECK(sc[2 + s] == 1.0f, "rank %d section %d: individual score %.9g on a first report", r, s, sc[2 + s]);
            CHECK(fl[2 + NS + s] == (want < 0.75f), "rank %d section %d: flag %d for score %g", r, s, fl[2 + NS + s], want);
        }
        double ref = med[0][0];
        for (int q = 1; q < RANKS; q++) ref = fmin(ref, med[q][0]);
        const double want_gpu = (weight[r] * (ref / med[r][0])) / weight[r]; /* reporting.py:237-253 with one kernel */
        CHECK(rel_err(sc[1], want_gpu) < 2e-6, "rank %d: relative GPU score %.9g != %.9g", r, sc[1], want_gpu);
        CHECK(rel_err(sc[0], 1.0) < 2e-6, "rank %d: individual GPU score %.9g on a first report", r, sc[0]);
        CHECK(fl[1] == (sc[1] < 0.75f), "rank %d: GPU flag %d for score %g", r, fl[1], sc[1]);
    }
    CHECK(h_flags[2 * W + 1] == 1 && h_flags[0 * W + 1] == 0 && h_flags[1 * W + 1] == 0, "only rank 2's GPU is slow (1 / 1.5 < 0.75)");
    CHECK(h_flags[2 * W + 2 + NS + 0] == 1 && h_flags[2 * W + 2 + NS + 1] == 0, "only rank 2's first section is slow");
    CHECK(h_meta[0] == 1 && h_meta[1] == RANKS && h_meta[2] == NK && h_meta[3] == NS, "meta %u %u %u %u", h_meta[0], h_meta[1], h_meta[2], h_meta[3]);

    /* ---- the same report as ONE call (nvrx_report): statistics and exchange rows stay in device memory, scores, flags and
     *      the forwarded statistics rows land in the pinned block under the completion word the call waits for ---- */
    {
        float *dev_stats = NULL, *dev_send = NULL, *h_stats2, *d_stats2, *h_scores2, *d_scores2;
        uint8_t *h_flags2, *d_flags2;
        uint32_t *dev_done = NULL, *h_meta2, *d_meta2;
        TRY(nvrx_device_alloc((void **)&dev_stats, (size_t)rows * NVRX_STATS_STRIDE * sizeof(float)));
        TRY(nvrx_device_alloc((void **)&dev_send, (size_t)RANKS * L * sizeof(float)));
        TRY(nvrx_device_alloc((void **)&dev_done, 64));
        TRY(nvrx_host_alloc((void **)&h_stats2, (void **)&d_stats2, (size_t)rows * NVRX_STATS_STRIDE * sizeof(float)));
        TRY(nvrx_host_alloc((void **)&h_scores2, (void **)&d_scores2, (size_t)RANKS * W * sizeof(float) + 16));
        TRY(nvrx_host_alloc((void **)&h_flags2, (void **)&d_flags2, (size_t)RANKS * W + 16));
        TRY(nvrx_host_alloc((void **)&h_meta2, (void **)&d_meta2, NVRX_META_WORDS * sizeof(uint32_t)));
        memset(h_meta2, 0, NVRX_META_WORDS * sizeof(uint32_t));
        TRY(nvrx_send_init(dev_send, RANKS, NK, NS, NULL));
        float scratch;
        TRY(nvrx_d2h_sync(&scratch, dev_send, sizeof(scratch), NULL)); /* (a stream wait: the exchange rows are initialised before a resident scorer reads them) */
        nvrx_report_desc d;
        memset(&d, 0, sizeof(d));
        d.R = RANKS, d.K = NK, d.S = NS, d.names_ok = 1, d.rows_active = 0, d.do_indiv = 1, d.do_rel = 1, d.stats_rows = rows;
        for (int i = 0; i < 4; i++) d.thresholds[i] = 0.75;
        d.d_stats = dev_stats, d.d_send = dev_send, d.d_table = NULL;
        d.d_scores = d_scores2, d.d_flags = d_flags2, d.d_meta = d_meta2, d.d_stats_dst = d_stats2, d.d_done_counter = dev_done;
        d.allgather_fn = NULL, d.comm = NULL, d.send_count = RANKS * L, d.seq = 0;
        d.h_seq_word = &h_meta2[4], d.timeout_s = 30.0, d.resident = 1;
        for (int rep = 0; rep < 3; rep++) { /* the descriptor is reused; the library advances seq */
            TRY(nvrx_report(ctx, &d, NULL));
            CHECK(d.seq == (uint32_t)(rep + 1) && h_meta2[4] == d.seq, "report %d: seq %u, completion word %u", rep, d.seq, h_meta2[4]);
            CHECK(memcmp(h_scores2, h_scores, (size_t)RANKS * W * sizeof(float)) == 0, "report %d: scores differ from the two-call path", rep);
            CHECK(memcmp(h_flags2, h_flags, (size_t)RANKS * W) == 0, "report %d: flags differ from the two-call path", rep);
            if (nvrx_poll_u32(&h_meta2[5], d.seq, 5.0) == NVRX_OK) /* resident scorer: the statistics rows have a word of their own */
                CHECK(memcmp(h_stats2, stats, (size_t)rows * NVRX_STATS_STRIDE * sizeof(float)) == 0, "report %d: forwarded statistics rows differ", rep);
            else
                CHECK(0, "report %d: the statistics rows' completion word never arrived (%u)", rep, h_meta2[5]);
        }
        TRY(nvrx_device_free(dev_stats));
        TRY(nvrx_device_free(dev_send));
        TRY(nvrx_device_free(dev_done));
        TRY(nvrx_host_free(h_stats2));
        TRY(nvrx_host_free(h_scores2));
        TRY(nvrx_host_free(h_flags2));
        TRY(nvrx_host_free(h_meta2));
    }

    /* ---- error behaviour of the ABI: negative codes + a message, never an abort ---- */
    CHECK(nvrx_ring_push(ctx, rows, 1.0f) == NVRX_ERR_INVALID && strlen(nvrx_last_error()) > 0, "out-of-range row must be refused");
    CHECK(nvrx_ring_push(NULL, 0, 1.0f) == NVRX_ERR_INVALID, "null context must be refused");

    for (int row = 0; row < rows; row++) oracle_ring_free(rings[row]);
    free(stats);
    free(ring_dev);
    TRY(nvrx_host_free(h_stats));
    TRY(nvrx_host_free(h_send));
    TRY(nvrx_host_free(h_scores));
    TRY(nvrx_host_free(h_flags));
    TRY(nvrx_host_free(h_meta));
    TRY(nvrx_ctx_destroy(ctx));
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("ABI HOST OK: %d rows, %d ranks scored from plain C\n", rows, RANKS);
    return 0;
}
