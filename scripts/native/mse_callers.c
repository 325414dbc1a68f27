/* Closed-loop caller threads for the coalescer measurements (bench.py `concurrent_callers`, tests): the reference's call
 * shape -- T host threads, ONE query per call (src/main.rs:896-934; src/query_disk_index.rs:711-736) -- as native threads,
 * the way a Rust host would drive the C ABI.  Bench / test infrastructure: not part of libmse_hip.so; it reaches the
 * library only through the function pointer it is handed (mse_dispatcher_topk_f16 or a compatible entry point).
 *
 * Thread t issues queries t, t + T, t + 2T, ... (n_queries in all, each exactly once), one per call, and records each
 * call's latency; every answer lands in its query's row of scores / ids so the caller can check all of them. */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

/* mse_dispatcher_topk_f16 (f16 queries, i64 scores, u32 ids) and mse_index_search (f32 queries, f32 distances, i64 labels) share
 * this shape; element sizes are arguments of the run */
typedef int (*topk_fn)(void* handle, const void* queries, size_t nq, size_t k, void* out_a, void* out_b);

typedef struct {
    topk_fn fn;
    void* handle;
    const char* queries;
    size_t n_queries, query_bytes, k, first, stride, a_bytes, b_bytes;
    char* out_a;
    char* out_b;
    double* latency_ms;
    pthread_barrier_t* gate;
    int failures;
} caller_t;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* caller_main(void* p) {
    caller_t* c = (caller_t*)p;
    pthread_barrier_wait(c->gate);
    for (size_t j = c->first; j < c->n_queries; j += c->stride) {
        const double t0 = now_s();
        const int rc = c->fn(c->handle, c->queries + j * c->query_bytes, 1, c->k, c->out_a + j * c->k * c->a_bytes, c->out_b + j * c->k * c->b_bytes);
        c->latency_ms[j] = (now_s() - t0) * 1e3;
        if (rc) c->failures++;
    }
    return NULL;
}

/* returns the wall-clock seconds from the moment all threads were released until the last one finished (< 0: setup failed);
 * *n_failed = calls that returned non-zero */
double mse_callers_run(void* fn, void* handle, const void* queries, size_t n_queries, size_t query_bytes, size_t k, int threads,
                       void* out_a, size_t a_bytes, void* out_b, size_t b_bytes, double* latency_ms, int* n_failed) {
    if (threads <= 0 || !fn) return -1.0;
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    caller_t* cs = (caller_t*)calloc((size_t)threads, sizeof(caller_t));
    pthread_barrier_t gate;
    if (!th || !cs || pthread_barrier_init(&gate, NULL, (unsigned)threads + 1)) { free(th); free(cs); return -1.0; }
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    int started = 0;
    for (int t = 0; t < threads; t++) {
        cs[t] = (caller_t){(topk_fn)fn, handle, (const char*)queries, n_queries, query_bytes, k, (size_t)t, (size_t)threads, a_bytes, b_bytes,
                           (char*)out_a, (char*)out_b, latency_ms, &gate, 0};
        if (pthread_create(&th[t], &attr, caller_main, &cs[t])) break;
        started++;
    }
    double dt = -1.0;
    if (started == threads) {
        pthread_barrier_wait(&gate);
        const double t0 = now_s();
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        dt = now_s() - t0;
        int f = 0;
        for (int t = 0; t < threads; t++) f += cs[t].failures;
        if (n_failed) *n_failed = f;
    } else {
        /* could not start them all: the ones waiting at the gate are released by a gate re-made for their count */
        for (int t = 0; t < started; t++) pthread_cancel(th[t]);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    pthread_attr_destroy(&attr);
    pthread_barrier_destroy(&gate);
    free(th);
    free(cs);
    return dt;
}

/* ---- the graph index's request path in the reference's call shape (round 5) -------------------------------------------------
 * query_disk_index answers one request = one query on its own task (src/query_disk_index.rs:436-540,711-736; perf_test.py:6-29:
 * 1000 one-query requests at concurrency 100).  T native threads, closed loop, ONE query per mse_disk_query_topk /
 * mse_disk_query_topk_f32 call through host pointers; thread t uses searchers[t % n_searchers] (a coalesced call only reads its
 * searcher's base, include/mse.h).  scales: [n_queries][n_desc] per-request descriptor scales or NULL. */
typedef int (*query16_fn)(void* s, void* pq, const void* c, const void* g, const uint32_t* starts, const void* q, const float* luts,
                          const float* scales, size_t nq, int disable_pq, size_t beam, size_t list, size_t k, uint32_t* ids, int64_t* scores,
                          uint32_t* nv, uint32_t* cm, uint32_t* pc);
typedef int (*query32_fn)(void* s, void* pq, const void* c, const void* g, const uint32_t* starts, const void* q, const float* scales, size_t nq,
                          int disable_pq, size_t beam, size_t list, size_t k, uint32_t* ids, int64_t* scores, uint32_t* nv, uint32_t* cm,
                          uint32_t* pc);

typedef struct {
    void* fn;
    int is_f32;
    void* searcher;
    void* pq;
    const void* codes;
    const void* graph;
    const char* queries;
    const float* scales;
    size_t n_queries, query_bytes, n_desc, k, beam, list, first, stride;
    int disable_pq;
    uint32_t* ids;
    int64_t* scores;
    double* latency_ms;
    pthread_barrier_t* gate;
    int failures;
} qcaller_t;

static void* qcaller_main(void* p) {
    qcaller_t* c = (qcaller_t*)p;
    pthread_barrier_wait(c->gate);
    for (size_t j = c->first; j < c->n_queries; j += c->stride) {
        const float* sc = c->scales ? c->scales + j * c->n_desc : NULL;
        const double t0 = now_s();
        int rc;
        if (c->is_f32)
            rc = ((query32_fn)c->fn)(c->searcher, c->pq, c->codes, c->graph, NULL, c->queries + j * c->query_bytes, sc, 1, c->disable_pq, c->beam,
                                     c->list, c->k, c->ids + j * c->k, c->scores + j * c->k, NULL, NULL, NULL);
        else
            rc = ((query16_fn)c->fn)(c->searcher, c->pq, c->codes, c->graph, NULL, c->queries + j * c->query_bytes, NULL, sc, 1, c->disable_pq,
                                     c->beam, c->list, c->k, c->ids + j * c->k, c->scores + j * c->k, NULL, NULL, NULL);
        c->latency_ms[j] = (now_s() - t0) * 1e3;
        if (rc) c->failures++;
    }
    return NULL;
}

double mse_callers_run_query(void* fn, int is_f32, void** searchers, int n_searchers, void* pq, const void* codes, const void* graph,
                             const void* queries, size_t n_queries, size_t query_bytes, const float* scales, size_t n_desc, int disable_pq,
                             size_t beam, size_t list, size_t k, int threads, uint32_t* ids, int64_t* scores, double* latency_ms, int* n_failed) {
    if (threads <= 0 || !fn || n_searchers <= 0) return -1.0;
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    qcaller_t* cs = (qcaller_t*)calloc((size_t)threads, sizeof(qcaller_t));
    pthread_barrier_t gate;
    if (!th || !cs || pthread_barrier_init(&gate, NULL, (unsigned)threads + 1)) { free(th); free(cs); return -1.0; }
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    int started = 0;
    for (int t = 0; t < threads; t++) {
        cs[t] = (qcaller_t){fn, is_f32, searchers[t % n_searchers], pq, codes, graph, (const char*)queries, scales, n_queries, query_bytes, n_desc, k,
                            beam, list, (size_t)t, (size_t)threads, disable_pq, ids, scores, latency_ms, &gate, 0};
        if (pthread_create(&th[t], &attr, qcaller_main, &cs[t])) break;
        started++;
    }
    double dt = -1.0;
    if (started == threads) {
        pthread_barrier_wait(&gate);
        const double t0 = now_s();
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        dt = now_s() - t0;
        int f = 0;
        for (int t = 0; t < threads; t++) f += cs[t].failures;
        if (n_failed) *n_failed = f;
    } else {
        for (int t = 0; t < started; t++) pthread_cancel(th[t]);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    pthread_attr_destroy(&attr);
    pthread_barrier_destroy(&gate);
    free(th);
    free(cs);
    return dt;
}

/* ---- the same requests WITHOUT a thread per request (round 5): ONE thread keeps `window` one-query requests in flight through
 * mse_disk_query_submit_f32 / mse_graph_completions -- what an async host (the reference's monoio tasks,
 * src/query_disk_index.rs:640-655) would do.  Request j's latency runs from its submit to the moment its ticket is collected. */
typedef int (*submit32_fn)(void* s, void* pq, const void* c, const void* g, const void* q, const float* scales, size_t nq, int disable_pq,
                           size_t beam, size_t list, size_t k, uint32_t* ids, int64_t* scores, uint32_t* nv, uint32_t* cm, uint32_t* pc, void* user,
                           void* completion_queue, void** ticket);
typedef long (*completions_fn)(const void* g, void** out, size_t max, long timeout_us);
typedef int (*tstatus_fn)(const void* t);
typedef void* (*tuser_fn)(const void* t);
typedef void (*tfree_fn)(void* t);

typedef struct {
    void *submit, *completions, *status, *user, *release, *searcher, *pq;
    const void *codes, *graph, *queries;
    size_t n_queries, query_bytes, beam, list, k, window;
    int disable_pq;
    uint32_t* ids;
    int64_t* scores;
    double *latency_ms, *t_sub;
    _Atomic size_t *next, *done;
    _Atomic long* in_flight;
    _Atomic int* failures;
    pthread_barrier_t* gate;
} acaller_t;

/* one submitting / collecting thread: the windows are shared (a ticket comes back to whichever thread asks next), so the count in
 * flight and the next query to submit are common to all of them */
static void* acaller_main(void* p) {
    acaller_t* c = (acaller_t*)p;
    enum { CHUNK = 256 };
    void* got[CHUNK];
    if (c->gate) pthread_barrier_wait(c->gate);
    while (atomic_load(c->done) < c->n_queries) {
        while (atomic_load(c->in_flight) < (long)c->window) {
            const size_t j = atomic_fetch_add(c->next, 1);
            if (j >= c->n_queries) break;
            void* ticket = NULL;
            atomic_fetch_add(c->in_flight, 1);
            c->t_sub[j] = now_s();
            if (((submit32_fn)c->submit)(c->searcher, c->pq, c->codes, c->graph, (const char*)c->queries + j * c->query_bytes, NULL, 1, c->disable_pq,
                                         c->beam, c->list, c->k, c->ids + j * c->k, c->scores + j * c->k, NULL, NULL, NULL, (void*)(uintptr_t)(j + 1),
                                         NULL, &ticket)) {
                atomic_fetch_add(c->failures, 1);   /* never queued: nothing will come back for it */
                atomic_fetch_sub(c->in_flight, 1);
                atomic_fetch_add(c->done, 1);
            }
        }
        if (atomic_load(c->done) >= c->n_queries) break;
        const long n = ((completions_fn)c->completions)(c->graph, got, CHUNK, 2000);
        if (n < 0) { atomic_fetch_add(c->failures, 1); break; }
        const double t1 = now_s();
        for (long i = 0; i < n; i++) {
            const size_t j = (size_t)(uintptr_t)((tuser_fn)c->user)(got[i]) - 1;
            if (((tstatus_fn)c->status)(got[i])) atomic_fetch_add(c->failures, 1);
            if (j < c->n_queries) c->latency_ms[j] = (t1 - c->t_sub[j]) * 1e3;
            ((tfree_fn)c->release)(got[i]);
        }
        atomic_fetch_sub(c->in_flight, n);
        atomic_fetch_add(c->done, (size_t)n);
    }
    return NULL;
}

double mse_callers_run_async(void* submit, void* completions, void* status, void* user, void* release, void* searcher, void* pq, const void* codes,
                             const void* graph, const void* queries, size_t n_queries, size_t query_bytes, int disable_pq, size_t beam, size_t list,
                             size_t k, size_t window, int threads, uint32_t* ids, int64_t* scores, double* latency_ms, int* n_failed) {
    if (!submit || !completions || !status || !user || !release || window == 0 || threads <= 0 || threads > 64) return -1.0;
    double* t_sub = (double*)calloc(n_queries ? n_queries : 1, sizeof(double));
    if (!t_sub) return -1.0;
    _Atomic size_t next = 0, done = 0;
    _Atomic long in_flight = 0;
    _Atomic int failures = 0;
    pthread_barrier_t gate;
    if (pthread_barrier_init(&gate, NULL, (unsigned)threads)) { free(t_sub); return -1.0; }
    acaller_t c = {submit, completions, status, user, release, searcher, pq, codes, graph, queries, n_queries, query_bytes, beam, list, k, window,
                   disable_pq, ids, scores, latency_ms, t_sub, &next, &done, &in_flight, &failures, threads > 1 ? &gate : NULL};
    pthread_t th[64];
    int started = 0;
    const double t0 = now_s();
    for (int t = 1; t < threads; t++) {
        if (pthread_create(&th[t], NULL, acaller_main, &c)) break;
        started++;
    }
    double dt = -1.0;
    if (started == threads - 1) {
        acaller_main(&c);                        /* the calling thread is one of them */
        for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
        dt = now_s() - t0;
    } else {
        for (int t = 1; t <= started; t++) pthread_cancel(th[t]);
        for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
    }
    pthread_barrier_destroy(&gate);
    free(t_sub);
    if (n_failed) *n_failed = atomic_load(&failures);
    return dt;
}
