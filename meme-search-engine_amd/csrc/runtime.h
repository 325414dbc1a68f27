// Host-side objects behind the opaque C handles.
#pragma once
#include "common.h"
#include "kernels.h"
#include "dispatch.h"
#include <mutex>
#include <vector>

namespace mse {

// growable device buffer (grows by hipMalloc + free; contents are NOT preserved)
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
    ~DevBuf() { release(); }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

int device_cu_count();

// one level of the selection tournament (topk.hip)
struct LevelRef {
    KeyKind kind;
    const void* ptr;
    size_t q_stride, e_stride, n;  // element (q, i) lives at ptr[q*q_stride + i*e_stride]
    bool group_major;              // float [n][nq_pad] as written by the MFMA scan
    int nq_pad;
};

}  // namespace mse

struct mse_searcher;
namespace mse {
// Tournament descent: ids of the k best level-0 entries per query in *sel_out ([nq][k], best first);
// their raw keys in keys_out when it is not null.  Uses the searcher's scratch and stream.
int descend(mse_searcher* s, const LevelRef& l0, int nq, int k, uint32_t** sel_out, void* keys_out);
// workspace-only searcher (no base vectors): scratch + stream for the PQ scan
mse_searcher* scratch_searcher_new();
}  // namespace mse

struct mse_base {
    const uint16_t* dev = nullptr;
    size_t n = 0, d = 0;
    bool owned = false;
    int n_cu = 256;
    int device = 0;            // HIP ordinal the rows live on (the thread's current device when the base was made)
    // the coalescer that MSE_MODE_AUTO host-pointer searches of every thread meet in (dispatch.hip), made on first use
    mutable std::mutex disp_mu;
    mutable struct mse_dispatcher* disp = nullptr;
    mutable bool disp_failed = false;   // it could not be made once: AUTO calls are answered directly from then on
    // max row norm (float bits) for the MFMA certificate, computed on first use
    mutable std::mutex norm_mu;
    mutable uint32_t* norm_bits_dev = nullptr;
    mutable bool norm_ready = false;
};

struct mse_searcher {
    const mse_base* base = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    mse::DevBuf q_stage;      // padded queries for one pass
    mse::DevBuf scores;       // [pass_q][n] i64 / f32 level 0 (exact mode)
    mse::DevBuf levels[6];    // tournament levels above level 0
    mse::DevBuf sel_a, sel_b; // selected ids ping-pong
    mse::DevBuf sel_keys;     // keys of the final selection
    mse::DevBuf out_scores, out_ids;  // device outputs for the host-pointer API
    mse::DevBuf gmax;         // MFMA group maxima [n_groups][nq_pad]
    mse::DevBuf cand_ids, cand_scores, gkeys, eps, margin;
    mse::DevBuf misc, qpacked;
    mse::DevBuf pq4;          // four-query PQ scan: the packed 12-bit table (136 KiB) + the four queries' certificate parameters
    mse::DevBuf wq, wg, widx, wout;   // per-query widening of the MFMA pass: the compact set's queries, group maxima, indices, results
    mse::DevBuf thr;          // [2][nq] u64: k-th best score key of each tournament level, the floor of the level below
    const unsigned long long* last_kth = nullptr;   // after descend(): k-th best level-0 key per query (sortable), or null
    mse::DevBuf pool[16];     // scratch of the batched graph searches (kept between calls: no hipMalloc on the query path)
    uint32_t last_widened = 0, last_max_groups = 0;
    // optional HIP-event timing of the dominant (scan) kernel, for bench.py's roofline line
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double scan_ms_total = 0.0;
    uint64_t scan_launches = 0;
    // optional measurement of the graph search kernel (mse_searcher_beam_timing, for bench.py's gather roofline): HIP events around
    // beam_search_kernel and device totals of what it gathered -- [0] rows scored exactly (2304-byte row gathers: fetched nodes and
    // exactly scored neighbours), [1] fetched nodes (adjacency lists read), [2] neighbours scored by ADC (64-byte code gathers)
    bool beam_timing = false;
    hipEvent_t bev0 = nullptr, bev1 = nullptr;
    double beam_ms_total = 0.0;
    uint64_t beam_launches = 0, beam_queries = 0;
    mse::DevBuf beam_tot;
    // event pairs of the PQ scan launches of one batch call (several per call on this searcher's stream), read when the call ends
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    // pinned host staging of the fused request path (ONE download per call: scores, ids and counters) and the event of
    // mse_searcher_wait_stream
    void* pin = nullptr;
    size_t pin_cap = 0;
    hipEvent_t ev_wait = nullptr;
};

struct mse_pq {
    size_t n_centroids = 0, d = 0, dpc = 0, n_chunks = 0;
    float* centroids = nullptr;  // device [n_centroids][d]
    float* transform = nullptr;  // device [d][d]
    float* transform_t = nullptr;  // device [d][d], transposed copy for the one-vector (query) path
    std::mutex mu;
    mse::DevBuf a, b, c;              // call scratch (guarded by mu)
    mse_searcher* scratch = nullptr;  // stream + scratch for scan calls that bring no searcher (guarded by mu; made on first use)
    mse_searcher* lane2 = nullptr;    // second and third stream of the batched scan; bound to the base of the call that made them
    mse_searcher* lane3 = nullptr;
    mse::DevBuf t2, lut2, qf2, qf3;   // their transformed query, table and f16 queries
    bool avoid8 = false;              // eight-per-pass (8-bit tables) gave too many uncertified queries on this data: stay with four per pass
    uint32_t last_uncertified = 0;    // four-query scan: queries of the last batch whose certificate failed (re-run through the exact scan)
    int device = 0;                   // HIP ordinal the quantiser was loaded on
    std::mutex co_mu;                 // guards the creation of `co`
    mse::Coalescer* co = nullptr;     // meeting point of one-query mse_pq_scan_topk calls from many threads (api_pq.hip), made on first use
    bool timing = false;              // HIP-event timing of the four-query scan kernel (mse_pq_scan_timing), for bench.py's roofline
    double scan_ms_total = 0.0;
    uint64_t scan_launches = 0;
    double span_ms_total = 0.0;       // first scan start .. last scan end of the batch calls with >= 4 scans (mse_pq_scan_sustained)
    uint64_t span_scans = 0;
    void* pin = nullptr;              // pinned host staging of the scan entry points (one upload + one download per call, both
    size_t pin_cap = 0;               // truly asynchronous: a pageable source makes the runtime stage and block per copy)
};

struct mse_codes {
    uint8_t* codes = nullptr;    // device [n][code_size]
    uint8_t* desc = nullptr;     // device [n][n_desc] or null
    size_t n = 0, code_size = 0, n_desc = 0;
};

namespace mse {
// largest row norm of the base (x 1.0001), computed once and kept on the device as float bits (b->norm_bits_dev)
int ensure_base_norm(const mse_base* b, hipStream_t st);
// device memory the batched graph searches may spend on visited sets per launch: half of the free HBM, 256 MiB .. 64 GiB
// (MSE_VISITED_BUDGET_KB overrides, for tests)
size_t visited_budget_bytes();
}  // namespace mse

namespace mse {
// one query's outputs of the fused request path (any of the counter pointers may be null)
struct QueryDst {
    uint32_t* ids; int64_t* scores; uint32_t *n_visited, *cmps, *pq_cmps; size_t k;
};
}  // namespace mse

struct mse_graph {
    uint32_t* adj = nullptr;   // device [n][max_deg]
    uint32_t* deg = nullptr;   // device [n]
    uint8_t* has_url = nullptr;  // device [n] or null (= all)
    size_t n = 0, max_deg = 0;
    // meeting point of the ONE-query calls of mse_disk_search_batch(_f32) from many threads (beam_search.hip), made on first use
    // and of the small calls of mse_disk_query_topk(_f32) (round 5): the whole request path of every waiting caller in one
    // submission, on a searcher and pinned staging that belong to the WORKER (one set per worker thread)
    mutable std::mutex co_mu;
    mutable mse::Coalescer* co = nullptr;
    mutable std::atomic<mse::Coalescer*> co_fast{nullptr};   // == co once made: the request threads' lock-free way to it
    struct WorkerCtx {
        mse_searcher* s = nullptr;   // made on first use over the callers' base
        void* pin = nullptr;         // gathered inputs (queries, scales, starts, tables)
        size_t pin_cap = 0;
        std::vector<mse::QueryDst> dsts;   // where each gathered query's results go
    };
    mutable std::vector<WorkerCtx> co_ctx;
    size_t co_max_queries = 0;       // mse_graph_set_coalescer: 0 = defaults (1024 queries per pass, 200 us, two workers)
    uint32_t co_max_wait_us = 0;
    int co_workers = 0;
    // entry table of the fused request path (beam_search.hip), one of two kinds:
    //   mse_graph_set_entries          copies of the entry records' vectors + their node ids; entry = exact top-1 over the copies
    //   mse_graph_set_entry_centroids  the reference's rule: f32 shard centroids as keys (transposed [d][n_entries]) + medioid ids
    // Request-path calls hold entry_lock shared for their duration; replacing the table takes it exclusively.
    uint16_t* entry_rows = nullptr;
    uint32_t* entry_ids = nullptr;
    size_t n_entries = 0;
    mse_base* entry_base = nullptr;
    float* entry_keys_t = nullptr;
    size_t entry_keys_d = 0;
    mutable mse::SharedExclusive entry_lock;
    // runtime de-duplication of the request path (src/query_disk_index.rs:482-527) inside mse_disk_query_topk(_f32): 0 = off
    float dedup_threshold = 0.0f;
    // searchers over the entry rows: a fused call borrows one for its duration (its scratch is in use until the call's stream is
    // drained), so that calls from several threads -- each with its own searcher and stream -- overlap instead of queueing
    mutable std::vector<mse_searcher*> entry_pool;
    mutable std::mutex entry_mu;
};
