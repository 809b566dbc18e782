#define _POSIX_C_SOURCE 200809L
/*
 * ktrace_host.c -- a plain-C host (C11 + pthreads, only include/nvrx_ktrace.h) of the kernel tracer's ABI: test
 * infrastructure.  It plays every thread the library meets in production, without a GPU and without Python:
 *
 *   two FEEDER threads   hand batches of dispatch records to nvrx_ktrace_feed (what the rocprofiler-sdk callback
 *                        thread does with the records of a 256 KB buffer) -- they "enqueue" first, deliver later;
 *   the SINK             two callbacks with the signatures of nvrx_ring_push_staged / nvrx_row_alloc, writing into
 *                        plain C rings under a lock of their own (as the engine's context does);
 *   the TRAINING thread  (main) opens and lifts holds, asks for counters and key names, and waits with
 *                        nvrx_ktrace_sync the way a report does.
 *
 * At the end every key's ring must hold exactly the NEWEST `cap` durations that were fed for it, in order
 * (CircularBuffer.h:53-61), nothing may be lost or counted twice, and the counters must add up.  The same source is
 * built a second time with -fsanitize=thread against a ThreadSanitizer build of the library (make tsan): the mutex
 * discipline of nvrx_ktrace.cpp is then checked by the tool, not by reading.
 *
 * usage: ktrace_host [--link-only]
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nvrx_ktrace.h"

#define KEYS 24
#define CAP 16
#define PER_FEEDER 30000
#define BATCH 173
#define MAX_ROWS 32

#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            fprintf(stderr, "KTRACE HOST FAILED: " __VA_ARGS__); \
            fprintf(stderr, "\n");                        \
            exit(1);                                      \
        }                                                 \
    } while (0)

/* ---- the sink: overwrite-oldest rings ---------------------------------------------------------------------- */
typedef struct {
    pthread_mutex_t mu;
    float ring[MAX_ROWS][CAP];
    uint64_t total[MAX_ROWS];
    int rows_used;
    uint64_t pushes;
} rings_t;

static int sink_push(void *ctx, const int32_t *rows, const float *values, int n) {
    rings_t *r = (rings_t *)ctx;
    pthread_mutex_lock(&r->mu);
    for (int i = 0; i < n; i++) {
        if (rows[i] < 0) continue;
        r->ring[rows[i]][r->total[rows[i]] % CAP] = values[i];
        r->total[rows[i]]++;
    }
    r->pushes++;
    pthread_mutex_unlock(&r->mu);
    return 0;
}

static int sink_row_alloc(void *ctx, int kind) {
    rings_t *r = (rings_t *)ctx;
    (void)kind;
    pthread_mutex_lock(&r->mu);
    const int row = r->rows_used < MAX_ROWS ? r->rows_used++ : -34;
    pthread_mutex_unlock(&r->mu);
    return row;
}

/* ---- feeders ------------------------------------------------------------------------------------------------ */
typedef struct {
    int id;
    uint64_t kernel_base;
} feeder_arg;

/* duration of the j-th launch of key k by feeder f, in ns: distinct, and recomputable by the checker */
static uint64_t dur_ns(int f, int k, uint64_t j) { return 1000u * (uint64_t)(1 + k) + 7u * (uint64_t)f + 100000u * (j % 50000u) + 1000u; }

static void *feeder_main(void *p) {
    feeder_arg *a = (feeder_arg *)p;
    nvrx_ktrace_dispatch batch[BATCH];
    uint64_t launched[KEYS];
    memset(launched, 0, sizeof(launched));
    uint32_t lcg = 12345u + 77u * (uint32_t)a->id;
    int sent = 0;
    while (sent < PER_FEEDER) {
        const int n = PER_FEEDER - sent < BATCH ? PER_FEEDER - sent : BATCH;
        for (int i = 0; i < n; i++) {
            lcg = lcg * 1664525u + 1013904223u;
            /* each feeder owns half of the keys: a key's order of arrival is its feeder's program order */
            const int k = (int)((lcg >> 16) % (KEYS / 2)) * 2 + a->id;
            nvrx_ktrace_dispatch *d = &batch[i];
            memset(d, 0, sizeof(*d));
            d->kernel_id = a->kernel_base + (uint64_t)k;
            d->workgroup[0] = 64u, d->workgroup[1] = 1u, d->workgroup[2] = 1u;
            d->grid[0] = 64u * (uint32_t)(1 + k), d->grid[1] = 1u, d->grid[2] = 1u;
            d->start_ns = 5000u;
            d->end_ns = 5000u + dur_ns(a->id, k, launched[k]++);
        }
        /* "the kernels are enqueued" ... */
        CHECK(nvrx_ktrace_feed(NULL, n, 1) == 0, "enqueue-only feed: %s", nvrx_ktrace_last_error());
        /* ... "and have finished": their records arrive -- feeder 0 plays a thread of the tracer that consumes its batch itself,
         * feeder 1 plays the SDK's completion callback, which only leaves them in the tracer's inbox (counted & 2): the
         * training thread's waits below (and the final one) are what brings those in */
        CHECK(nvrx_ktrace_feed(batch, n, a->id == 1 ? 2 : 0) == 0, "feed: %s", nvrx_ktrace_last_error());
        sent += n;
    }
    return NULL;
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "--link-only") == 0) {
        printf("ktrace abi: ready %d, keys %d, sizeof(record) %zu, sizeof(dispatch) %zu, sizeof(sink) %zu\n", nvrx_ktrace_ready(),
               nvrx_ktrace_num_keys(), sizeof(nvrx_ktrace_record), sizeof(nvrx_ktrace_dispatch), sizeof(nvrx_ktrace_sink));
        return sizeof(nvrx_ktrace_record) == 8 && sizeof(nvrx_ktrace_dispatch) == 48 ? 0 : 1;
    }
    static rings_t rings;
    pthread_mutex_init(&rings.mu, NULL);
    const uint64_t base = 0x7000000000ull;
    char name[64];
    for (int k = 0; k < KEYS; k++) {
        snprintf(name, sizeof(name), "_Z9host_kern%02dPf", k);
        CHECK(nvrx_ktrace_feed_kernel_name(base + (uint64_t)k, name, 0) == 0, "name");
    }
    CHECK(nvrx_ktrace_feed_kernel_name(base + 999u, "k_scatter", 1) == 0, "own kernel name");
    nvrx_ktrace_sink sink = {&rings, sink_push, sink_row_alloc, 1};
    CHECK(nvrx_ktrace_set_sink(&sink) == 0, "set_sink: %s", nvrx_ktrace_last_error());
    const uint64_t enq0 = nvrx_ktrace_counter(0), arr0 = nvrx_ktrace_counter(1), del0 = nvrx_ktrace_counter(2), own0 = nvrx_ktrace_counter(5);

    /* one of the engine's own kernels: counted, arrives, is not recorded */
    nvrx_ktrace_dispatch own;
    memset(&own, 0, sizeof(own));
    own.kernel_id = base + 999u, own.workgroup[0] = 256, own.workgroup[1] = own.workgroup[2] = 1, own.grid[0] = 256, own.grid[1] = own.grid[2] = 1;
    own.start_ns = 1, own.end_ns = 2001;
    CHECK(nvrx_ktrace_feed(&own, 1, 1) == 0, "own feed");

    pthread_t th[2];
    feeder_arg args[2] = {{0, base}, {1, base}};
    for (int i = 0; i < 2; i++) CHECK(pthread_create(&th[i], NULL, feeder_main, &args[i]) == 0, "pthread_create");
    /* the training thread meanwhile: looks, holds and releases, reads names -- as reports do */
    int looks = 0, holds = 0;
    for (int round = 0; round < 400; round++) {
        const int missing = nvrx_ktrace_sync(round % 3 == 0 ? 0.0002 : 0.0); /* a short wait drains the inbox, a look does not */
        CHECK(missing >= 0, "sync: %s", nvrx_ktrace_last_error());
        looks++;
        if (round % 7 == 0) {
            CHECK(nvrx_ktrace_hold(1) == 0, "hold");
            const int n = nvrx_ktrace_num_keys();
            for (int k = 0; k < n; k += 5) (void)nvrx_ktrace_key_name((uint32_t)k), (void)nvrx_ktrace_key_row((uint32_t)k);
            CHECK(nvrx_ktrace_hold(0) == 0, "unhold");
            holds++;
        }
        (void)nvrx_ktrace_counter(10);
        const struct timespec nap = {0, 100000}; /* 0.1 ms: the loop lasts as long as the feeders do */
        nanosleep(&nap, NULL);
    }
    for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
    CHECK(nvrx_ktrace_sync(5.0) == 0, "every dispatch that was enqueued has arrived");

    const uint64_t fed = 2u * (uint64_t)PER_FEEDER;
    CHECK(nvrx_ktrace_counter(0) - enq0 == fed + 1, "enqueued %llu", (unsigned long long)(nvrx_ktrace_counter(0) - enq0));
    CHECK(nvrx_ktrace_counter(1) - arr0 == fed + 1, "arrived %llu", (unsigned long long)(nvrx_ktrace_counter(1) - arr0));
    CHECK(nvrx_ktrace_counter(2) - del0 == fed, "delivered %llu", (unsigned long long)(nvrx_ktrace_counter(2) - del0));
    CHECK(nvrx_ktrace_counter(5) - own0 == 1, "own kernels left out");
    CHECK(nvrx_ktrace_counter(3) == 0 && nvrx_ktrace_counter(4) == 0, "nothing lost, no sink error");
    CHECK(nvrx_ktrace_counter(10) == KEYS && rings.rows_used == KEYS, "rows given out: %d", rings.rows_used);

    /* every key: name in the reference's format, count, and the NEWEST CAP durations in order */
    uint64_t per_key[KEYS];
    memset(per_key, 0, sizeof(per_key));
    {   /* recount what each feeder launched per key (same generator) */
        for (int f = 0; f < 2; f++) {
            uint32_t lcg = 12345u + 77u * (uint32_t)f;
            for (int i = 0; i < PER_FEEDER; i++) {
                lcg = lcg * 1664525u + 1013904223u;
                per_key[(int)((lcg >> 16) % (KEYS / 2)) * 2 + f]++;
            }
        }
    }
    int checked = 0;
    const int nkeys = nvrx_ktrace_num_keys();
    for (int id = 0; id < nkeys; id++) {
        const char *key = nvrx_ktrace_key_name((uint32_t)id);
        int k = -1;
        if (!key || sscanf(key, "_Z9host_kern%02dPf_blk_64_1_1_grid_", &k) != 1) continue;   /* keys of other tests of this process */
        char expect[96];
        snprintf(expect, sizeof(expect), "_Z9host_kern%02dPf_blk_64_1_1_grid_%d_1_1", k, 1 + k);
        CHECK(strcmp(key, expect) == 0, "key %s != %s", key, expect);
        const int row = nvrx_ktrace_key_row((uint32_t)id);
        CHECK(row >= 0 && row < MAX_ROWS, "row of %s: %d", key, row);
        CHECK(rings.total[row] == per_key[k], "%s: %llu samples, expected %llu", key, (unsigned long long)rings.total[row],
              (unsigned long long)per_key[k]);
        const int f = k % 2;
        const uint64_t n = per_key[k], first = n > CAP ? n - CAP : 0;
        for (uint64_t j = first; j < n; j++) {
            const float want = (float)dur_ns(f, k, j) / 1000.0f; /* CuptiProfiler.cpp:191 */
            const float got = rings.ring[row][j % CAP];
            CHECK(got == want, "%s launch %llu: %f != %f", key, (unsigned long long)j, got, want);
        }
        checked++;
    }
    CHECK(checked == KEYS, "checked %d keys", checked);
    CHECK(nvrx_ktrace_set_sink(NULL) == 0, "clear sink");
    CHECK(nvrx_ktrace_key_row(0) == -2, "rows are forgotten with the sink");
    printf("KTRACE HOST OK: %llu records through 2 feeder threads, %d keys, %llu sink calls, %d looks, %d holds\n",
           (unsigned long long)fed, checked, (unsigned long long)rings.pushes, looks, holds);
    return 0;
}
