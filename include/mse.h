/*
 * mse.h -- C ABI of libmse_hip.so: the MI355X (gfx950) scoring hot path of meme-search-engine.
 *
 * Drop-in boundary.  Every entry point names the reference interface it replaces (paths are
 * relative to the reference checkout).  A Rust maintainer binds these with an `extern "C"`
 * block (INTEGRATION.md shows the stubs); nothing here mentions torch, HIP or C++ types.
 *
 * Conventions
 *   - All functions returning `int` return 0 on success, non-zero on failure; the message is
 *     available from mse_last_error() (thread-local).  Nothing aborts or throws across the ABI.
 *   - f16 data is passed as uint16_t IEEE binary16 bit patterns (Rust `half::f16` is
 *     repr(transparent) over u16).
 *   - Scores are i64 fixed point, `(f32 * 2^32) as i64`, exactly as diskann::vector produces
 *     (diskann/src/vector.rs:46-47,408-416).
 *   - Pointers named *_dev are device (HBM) pointers; all others are host pointers.  The caller
 *     owns every buffer it passes in; the library keeps no caller pointer after a call returns
 *     except for the *_wrap_device constructors, which borrow.
 *   - Handles are thread-compatible: an mse_base / mse_codes / mse_pq / mse_index may be shared
 *     read-only by any number of threads; each searching thread owns its own mse_searcher
 *     (mirrors the reference: one `Scratch` + `Rc<Index>` per thread over shared `Arc` maps,
 *     src/query_disk_index.rs:714-731).
 *   - Tie order: equal scores are ordered by ascending id (the reference leaves it unspecified:
 *     sort_unstable_by_key, src/query_disk_index.rs:271).
 */
#ifndef MSE_H
#define MSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define MSE_ID_NONE 0xFFFFFFFFu

/* ---- runtime ------------------------------------------------------------------------- */
const char* mse_last_error(void);
int mse_device_count(void);
int mse_set_device(int ordinal);                 /* selects the HIP device for this thread */
int mse_device_synchronize(void);
int mse_device_mem_info(size_t* free_bytes, size_t* total_bytes);
const char* mse_version(void);

/* ---- fixed-point scale: diskann/src/vector.rs:408-416 ---------------------------------- */
int64_t mse_scale_dot_f32(float x);              /* scale_dot_result */
int64_t mse_scale_dot_f64(double x);             /* scale_dot_result_f64 */

/* ---- base vectors: diskann::vector::VectorList (vector.rs:118-186) kept resident in HBM - */
typedef struct mse_base mse_base;
mse_base* mse_base_from_host(const uint16_t* data, size_t n_rows, size_t d);      /* copies */
mse_base* mse_base_wrap_device(const void* data_dev, size_t n_rows, size_t d);    /* borrows; see mse_base_rows_changed */
mse_base* mse_base_generate(uint32_t seed, uint64_t first_row, size_t n_rows, size_t d); /* synthetic rows made on the device */
void mse_base_free(mse_base* b);
size_t mse_base_len(const mse_base* b);
size_t mse_base_dim(const mse_base* b);
const void* mse_base_device_ptr(const mse_base* b);
int mse_base_read_rows(const mse_base* b, size_t first_row, size_t n_rows, uint16_t* out); /* D2H, for spot checks */
/* A base caches its largest row norm (the bound behind the MFMA scan's exactness certificate and the
 * graph build's).  Borrowed memory must stay unchanged while searches run; after rewriting rows of a
 * wrapped base call this (with no search in flight) so the bound is measured again on next use. */
int mse_base_rows_changed(mse_base* b);

/* fast_dot_noprefetch(x, y) / fast_dot(x, y, _) -- diskann/src/vector.rs:255-306,192-252.
 * Host slices in, one i64 out, computed on the device in the reference's summation order. */
/* NB: one call = two uploads, one launch, one download (tens of microseconds): it pins the arithmetic of
 * the interface for parity tests and documents it; production scoring goes through mse_score_rows_f16 /
 * the searches, which keep rows resident and batch the work. */
int mse_fast_dot_f16(const uint16_t* x, const uint16_t* y, size_t n, int64_t* out);

/* ---- searcher: per-thread scratch + stream (reference: `Scratch`, lib.rs:157-175 and
 * query_disk_index.rs:116-123) --------------------------------------------------------- */
typedef struct mse_searcher mse_searcher;
mse_searcher* mse_searcher_new(const mse_base* b);
void mse_searcher_free(mse_searcher* s);
/* HIP stream the searcher launches on (void* = hipStream_t); default: its own stream. */
int mse_searcher_set_stream(mse_searcher* s, void* hip_stream);
void* mse_searcher_stream(const mse_searcher* s);

#define MSE_MODE_AUTO 0   /* host-pointer searches of at most one pass (mse_queries_per_pass_max): coalesced across threads on a
                             worker the base owns (mse_dispatcher below; the caller's stream, scan timing and last_stats are not
                             involved -- mse_dispatcher_searcher has the worker's).  Larger host-pointer batches, device-pointer
                             searches, and every search if that worker could not be made: on the caller's searcher, exact scan
                             for <= 8 queries, batched MFMA scan above that */
#define MSE_MODE_EXACT 1  /* every row scored in the reference order on the vector ALU */
#define MSE_MODE_MFMA 2   /* f16 MFMA scan for candidates + exact re-score + certificate */

/* Most queries one matrix-core pass over the rows serves at vector width d (320; 256 when d / 64 is odd): batches are cut into
 * passes of this size, and it is the dispatcher's default gather size.  Callers that size their own batches gain nothing
 * from going beyond a multiple of it. */
size_t mse_queries_per_pass_max(size_t d);

/* Brute-force top-k: the scan + ranking of `evaluate` (src/query_disk_index.rs:262-273) for a
 * batch of f16 queries.  scores/ids are [nq][k], best first; unfilled slots (k > n_rows) hold
 * INT64_MIN / MSE_ID_NONE.  Returned ids/scores are identical in every mode. */
int mse_bruteforce_topk_f16(mse_searcher* s, const uint16_t* queries, size_t nq, size_t k, int mode,
                            int64_t* scores, uint32_t* ids);
/* Same with device-resident queries/outputs, asynchronous on the searcher's stream.
 * id_offset is added to every returned id (global id = local id + shard offset). */
int mse_bruteforce_topk_f16_dev(mse_searcher* s, const void* queries_dev, size_t nq, size_t k, int mode,
                                uint64_t id_offset, void* scores_dev, void* ids_dev);
/* All n_rows exact scores of one query (what `matches` holds before the sort, :263-269). */
int mse_bruteforce_scores_f16(mse_searcher* s, const uint16_t* query, int64_t* scores);
/* rank[i] = position of row ids[i] in the (score desc, id asc) order of one query; the
 * rank lookup of evaluate (:271-273,309-316). */
int mse_bruteforce_ranks_f16(mse_searcher* s, const uint16_t* query, const uint32_t* ids, size_t n_ids,
                             uint32_t* ranks);
/* Gather-and-score: out[i] = fast_dot(query, base[ids[i]]) -- the neighbour loop of the in-RAM
 * search (diskann/src/lib.rs:201-207) and the fetched-node re-score (query_disk_index.rs:168-169).
 * ids >= n_rows give INT64_MIN. */
int mse_score_rows_f16(mse_searcher* s, const uint32_t* ids, size_t n_ids, const uint16_t* query, int64_t* out);
/* k-way merge of per-shard results after the all-gather (multi-GPU, one process per GPU):
 * gathered_* are device arrays laid out [n_shards][nq][k] (what ncclAllGather produces from each
 * rank's [nq][k]); out_* are device [nq][k].  Asynchronous on the searcher's stream. */
int mse_merge_topk_dev(mse_searcher* s, const void* gathered_scores_dev, const void* gathered_ids_dev,
                       size_t n_shards, size_t nq, size_t k, void* out_scores_dev, void* out_ids_dev);
/* Same merge for PACKED per-shard blocks: block g = [nq*k] i64 scores then [nq*k] u32 ids, blocks
 * mse_topk_block_bytes(nq, k) apart (what mse_comm_search_dev gathers and mse_shard_group fills). */
size_t mse_topk_block_bytes(size_t nq, size_t k);
int mse_merge_topk_packed_dev(mse_searcher* s, const void* gathered_blocks_dev, size_t n_shards, size_t nq, size_t k,
                              void* out_scores_dev, void* out_ids_dev);
/* Test hook: raw output of the matrix-core scan, out[g][q] = max over rows 32g..32g+31 of the MFMA score of query q
 * (nq <= 256, host arrays), so that tests can measure its distance from the exact-order scores. */
int mse_debug_mfma_group_max(mse_searcher* s, const uint16_t* queries, size_t nq, float* out);
/* HIP-event timing of the scan kernel (the HBM-bound kernel) on the searcher's stream: returns the
 * totals accumulated so far, then sets the mode: enable 0 = off, 1 = on, 2 = on and reset totals. */
int mse_searcher_scan_timing(mse_searcher* s, int enable, double* total_ms, uint64_t* launches);
/* statistics of the last MFMA-mode call: number of queries whose certificate needed a wider
 * candidate set, and the widest group count used. */
int mse_searcher_last_stats(const mse_searcher* s, uint32_t* n_widened, uint32_t* max_groups);

/* ---- cross-thread query coalescer ----------------------------------------------------------
 * The reference serves ONE query per request from many threads at once: `index.search(&query, k)` under a shared read
 * guard per HTTP request (src/main.rs:896-934,1043-1049), and a thread per core with its own Scratch, one search per
 * request (src/query_disk_index.rs:711-736).  A pass over the rows costs one MI355X the same for 1 query as for 128, so
 * those callers must share passes: mse_dispatcher_topk_f16 may be called from any number of threads; callers block, ONE
 * worker thread (on the device that holds the rows) gathers what is waiting -- until as many queries wait as the last
 * pass answered, or max_queries_per_pass, or the oldest is max_wait_us old -- runs a single pass (matrix-core scan +
 * exact re-score + certificate: the same answers as every mode of mse_bruteforce_topk_f16) and hands each caller its rows.
 * A lone caller never waits for company.  Callers may ask for different k (each gets the first k of the largest k's order);
 * argument errors are returned to their caller without entering the queue, and if a shared pass fails every request of
 * it is repeated alone, so a caller only ever sees its own failure.
 * max_queries_per_pass 0 = 256 (one matrix-core pass); max_wait_us 0 = a tenth of a pass over the rows, 200 us .. 5 ms.
 * mse_bruteforce_topk_f16(..., MSE_MODE_AUTO, ...) goes through a dispatcher the base makes on first use, so the reference's
 * thread-per-core loop coalesces without knowing; mse_index_search does the same inside every mse_index. */
typedef struct mse_dispatcher mse_dispatcher;
mse_dispatcher* mse_dispatcher_new(const mse_base* b, size_t max_queries_per_pass, uint32_t max_wait_us);
void mse_dispatcher_free(mse_dispatcher* d);               /* no call may be in flight */
int mse_dispatcher_topk_f16(mse_dispatcher* d, const uint16_t* queries, size_t nq, size_t k, int64_t* scores, uint32_t* ids);
/* out: [0] queries answered, [1] requests, [2] passes, [3] most queries in one pass, [4] passes started by the wait budget,
 * [5] requests repeated alone after a failed shared pass */
int mse_dispatcher_stats(mse_dispatcher* d, uint64_t out[6]);
mse_searcher* mse_dispatcher_searcher(mse_dispatcher* d);   /* the worker's searcher, for scan timing / certificate stats; owned by d */
/* test hook: the next n_passes passes that carry more than one request fail before they start, so that the
 * repeat-each-request-alone path can be exercised (answers must be unaffected) */
int mse_debug_dispatcher_fail_shared(mse_dispatcher* d, uint32_t n_passes);
/* test hook that needs no device (the "not gpu" suite): `threads` host threads x `rounds` one-query requests through the coalescer's
 * queue with a stand-in pass (payload p -> 2 p + 1; payloads divisible by 97 fail, alone).  stats_out as mse_dispatcher_stats;
 * *mismatches = requests that got a wrong answer, a wrong status or no error text. */
int mse_debug_coalescer_selftest(int threads, int rounds, uint32_t max_queries, uint32_t max_wait_us, uint64_t stats_out[6],
                                 uint64_t* mismatches);
/* the same through a coalescer with `workers` worker threads (the graph's request path runs three, csrc/dispatch.h) */
int mse_debug_coalescer_selftest_workers(int threads, int rounds, uint32_t max_queries, uint32_t max_wait_us, int workers,
                                         uint64_t stats_out[6], uint64_t* mismatches);
/* the asynchronous side of the same queue (submit_async / completions), no device needed: async_threads threads keep `window` records
 * each in flight while sync_threads blocking callers share the handle; own_queues != 0: every asynchronous thread has a completion queue
 * of its own and must get back exactly its own records; stats_out[5] = records collected; *mismatches = records handed back twice or
 * never (or to the wrong queue), wrong answers / statuses / error texts */
int mse_debug_coalescer_selftest_async(int async_threads, int window, int n_requests, int sync_threads, uint32_t max_queries, int workers,
                                       int own_queues, uint64_t stats_out[6], uint64_t* mismatches);

/* ---- row-sharded index over the GPUs of one node.  The reference has no multi-GPU
 * code; its query server is a thread per core, each with its own Scratch over shared read-only maps
 * (src/query_disk_index.rs:711-736).  Same shape here with a thread per shard: rows partitioned
 * contiguously (shard g of G holds rows [g*n/G ..), remainder on the first shards), every shard scores
 * the same query batch and returns global ids, the per-shard [nq][k] records meet in ONE buffer on the
 * root device (shard 0's; written over a peer mapping, i.e. xGMI, when the devices allow it) and are
 * merged by (score desc, id asc).  Results equal those of one searcher over all rows.
 * devices[g] = HIP ordinal of shard g (NULL: g mod device count); ordinals may repeat (logical shards). */
typedef struct mse_shard_group mse_shard_group;
mse_shard_group* mse_shard_group_new(const int* devices, size_t n_shards, size_t d);
void mse_shard_group_free(mse_shard_group* g);
size_t mse_shard_group_n_shards(const mse_shard_group* g);
size_t mse_shard_group_len(const mse_shard_group* g);                       /* rows over all shards */
int mse_shard_group_device(const mse_shard_group* g, size_t shard);
int mse_shard_group_peer_mapped(const mse_shard_group* g, size_t shard);    /* 1: writes the gather buffer directly */
mse_searcher* mse_shard_group_searcher(mse_shard_group* g, size_t shard);   /* for scan timing / certificate stats; owned by the group */
/* fill: synthetic rows first_row .. first_row+total_rows made on each shard's device | one host array split
 * over the shards (copied) | one shard borrowed from device memory on that shard's device */
int mse_shard_group_generate(mse_shard_group* g, uint32_t seed, uint64_t first_row, size_t total_rows);
int mse_shard_group_load_host(mse_shard_group* g, const uint16_t* rows, size_t total_rows);
int mse_shard_group_set_shard_device(mse_shard_group* g, size_t shard, const void* rows_dev, size_t n_rows, uint64_t first_row);
/* brute-force top-k over all shards: same contract as mse_bruteforce_topk_f16 (ids are global). */
int mse_shard_group_search(mse_shard_group* g, const uint16_t* queries, size_t nq, size_t k, int mode, int64_t* scores,
                           uint32_t* ids);
/* queries / outputs on the ROOT device (queries complete before the call); returns when the result is complete. */
/* How the per-shard [nq][k] records meet (no reference counterpart: the reference has one address space, src/query_disk_index.rs:711-736).
 * MSE_EXCHANGE_PEER (default): kernels of a shard store its block into the root device's gather buffer through a peer mapping
 * (or one hipMemcpyPeerAsync per shard when the devices cannot map each other); works with several shards per device.
 * MSE_EXCHANGE_RCCL: ONE ncclAllGather of the packed 12-byte records per search among the shards' devices (librccl.so, dlopen'ed),
 * every shard on its own device.  set_exchange returns -1 and leaves the previous exchange in place when RCCL cannot be brought
 * up (shards sharing a device, no librccl, ncclCommInitAll failing): callers degrade, they do not abort. */
#define MSE_EXCHANGE_PEER 0
#define MSE_EXCHANGE_RCCL 1
typedef struct mse_pq mse_pq;         /* (declared in full further down) */
typedef struct mse_codes mse_codes;
typedef struct mse_graph mse_graph;
/* ---- the approximate-search paths over the same shards (rows AND their PQ codes / descriptors / graph) ----
 * A shard's codec, codes (+ descriptor bytes) and graph are ordinary handles made by the caller on the shard's device
 * (mse_set_device(mse_shard_group_device(g, shard)) first) over the shard's rows (mse_shard_group_searcher(g, shard)->base), speaking
 * LOCAL ids; the group adds the shard's first row.  Handles stay the caller's and must outlive their attachment (NULLs detach).
 *   mse_shard_group_pq_scan_topk   mse_pq_scan_topk_batch (ADC top-r, exact fp16 re-score, top-k) over all shards with the answer of the
 *                                  unsharded call bit for bit: (A) every shard's ADC top-r -> exchange -> the index's top-r;
 *                                  (B) every shard re-scores ITS members of it exactly -> exchange -> top-k.  scales: [n_desc] or NULL.
 *   mse_shard_group_query_topk     one graph per shard over its rows (the reference's shards: src/generate_index_shard.rs): every shard
 *                                  answers the batch from its graph (mse_disk_query_topk: its entry table, greedy_search, the k best
 *                                  visited records), ONE exchange, merge by (score desc, id asc) = the merge of the per-shard searches.
 *                                  queries f16 [nq][d] host rows; luts [nq][64*256] (ADC) or NULL with disable_pq; scales [nq][n_desc] or NULL.
 * Both use the group's exchange (peer stores or ONE ncclAllGather of the packed blocks per exchange). */
const mse_base* mse_shard_group_base(const mse_shard_group* g, size_t shard);   /* the shard's rows (owned by the group) */
uint64_t mse_shard_group_first_row(const mse_shard_group* g, size_t shard);
int mse_shard_group_attach_pq(mse_shard_group* g, size_t shard, mse_pq* pq, const mse_codes* codes);
int mse_shard_group_attach_graph(mse_shard_group* g, size_t shard, const mse_graph* graph);
int mse_shard_group_pq_scan_topk(mse_shard_group* g, const float* queries_f32, const float* scales, size_t nq, size_t r, size_t k,
                                 int64_t* scores, uint32_t* ids);
int mse_shard_group_query_topk(mse_shard_group* g, const uint16_t* queries, const float* luts, const float* scales, size_t nq, int disable_pq,
                               size_t beamwidth, size_t search_list, size_t k, int64_t* scores, uint32_t* ids);
int mse_shard_group_set_exchange(mse_shard_group* g, int kind);
int mse_shard_group_exchange(const mse_shard_group* g);
int mse_shard_group_rccl_ranks(const mse_shard_group* g);                   /* ranks as ncclCommCount reports them; 0 = RCCL not up */
/* breakdown of the last search in ms: [0] slowest shard's local search (scan + tournament + re-score + certificate),
 * [1] slowest shard's exchange leg (all-gather / staged copy; ~0 for peer stores), [2] merge on the root, [3] wall clock of the call */
int mse_shard_group_last_timing(mse_shard_group* g, double out_ms[4]);
int mse_shard_group_search_dev(mse_shard_group* g, const void* queries_dev, size_t nq, size_t k, int mode, void* scores_dev,
                               void* ids_dev);

/* One process per GPU instead (the launch shape of torchrun): ONE ncclAllGather of the packed per-shard
 * records over RCCL/xGMI on the searcher's stream, then the same merge on every rank.  RCCL is loaded on
 * first use (librccl.so); the 128-byte id made by rank 0 reaches the other ranks through the host's own
 * rendezvous (any byte transport).  Every rank must call mse_comm_search_dev with the same nq and k. */
typedef struct { char internal[128]; } mse_comm_id;     /* == ncclUniqueId */
typedef struct mse_comm mse_comm;
int mse_comm_unique_id(mse_comm_id* out);
mse_comm* mse_comm_init(const mse_comm_id* id, int rank, int world);  /* collective; on the thread's current device */
void mse_comm_free(mse_comm* c);
int mse_comm_rank(const mse_comm* c);
int mse_comm_size(const mse_comm* c);                   /* rank count as RCCL reports it */
int mse_comm_search_dev(mse_comm* c, mse_searcher* s, const void* queries_dev, size_t nq, size_t k, int mode,
                        uint64_t id_offset, void* scores_dev, void* ids_dev);
/* breakdown of this rank's last mse_comm_search_dev in ms: [0] local search, [1] all-gather (incl. waiting for the slowest rank),
 * [2] merge, [3] their sum; waits for that search to finish */
int mse_comm_last_timing(mse_comm* c, double out_ms[4]);
/* The exchange alone -- this rank's packed block of (score, GLOBAL id) records ([nq*k_in] i64, then [nq*k_in] u32; on its device,
 * complete on the searcher's stream) -> ONE ncclAllGather -> the k best per query of all ranks' records, identical on every rank -- and
 * the two approximate-search paths composed over it, one process per GPU (the protocols of mse_shard_group_pq_scan_topk /
 * _query_topk; first_row = global id of this rank's local row 0; host inputs are the same on every rank; outputs [nq][k] on this
 * rank's device, complete on return). */
int mse_comm_exchange_dev(mse_comm* c, mse_searcher* s, const void* block_dev, size_t nq, size_t k_in, size_t k, void* scores_dev, void* ids_dev);
int mse_comm_pq_scan_topk(mse_comm* c, mse_pq* pq, const mse_codes* codes, mse_searcher* s, const float* queries_f32, const float* scales,
                          size_t nq, size_t r, size_t k, uint64_t first_row, void* scores_dev, void* ids_dev);
int mse_comm_query_topk(mse_comm* c, mse_searcher* s, mse_pq* pq, const mse_codes* codes, const mse_graph* g, const uint16_t* queries,
                        const float* luts, const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k,
                        uint64_t first_row, void* scores_dev, void* ids_dev);

/* ---- flat in-memory index: FAISS IndexScalarQuantizer(QT_fp16, INNER_PRODUCT) as used by
 * src/main.rs:822 (new), :858,:892 (add), :900 (search), :1015,:1053 (ntotal) -------------- */
typedef struct mse_index mse_index;
mse_index* mse_index_new(int d);
void mse_index_free(mse_index* idx);
int mse_index_add(mse_index* idx, const float* x, size_t n);            /* fp32 -> fp16 RNE, appended */
size_t mse_index_ntotal(const mse_index* idx);
/* distances [nq][k] descending, labels [nq][k], -1 where fewer than k vectors exist (:908). */
/* Any number of threads may search at once (the shared `index.read()` of src/main.rs:1046); `add` excludes them (`index.write()`,
 * :1016) and is not starved by them.  Concurrent searches meet in the index's coalescer (above) and share passes: <= 8 waiting
 * queries over a cache-sized index take the exact pass, anything more ONE matrix-core pass + f32 re-score + certificate. */
int mse_index_search(mse_index* idx, const float* queries, size_t nq, size_t k, float* distances, int64_t* labels);
int mse_index_stats(mse_index* idx, uint64_t out[6]);     /* as mse_dispatcher_stats */

/* ---- product quantiser: diskann::vector::ProductQuantizer (vector.rs:308-406) ------------ */
typedef struct mse_pq mse_pq;
mse_pq* mse_pq_load(const float* centroids, size_t n_centroids, const float* transform, size_t n_dims,
                    size_t n_dims_per_code);
void mse_pq_free(mse_pq* pq);
int mse_pq_apply_transform(mse_pq* pq, const float* x, size_t n, float* out);          /* :320-329 */
int mse_pq_quantize_batch(mse_pq* pq, const float* x, size_t n, uint8_t* codes);       /* :331-364 */
int mse_pq_preprocess_query(mse_pq* pq, const float* query, float* lut);               /* :367-384, lut[n_chunks*n_centroids] */
int mse_pq_adc(mse_pq* pq, const float* lut, const uint8_t* codes, size_t n, int64_t* out); /* :387-405 */

/* PQ codes (+ optional descriptor bytes) resident in HBM: the mmap'd index.pq-codes.bin /
 * index.descriptor-codes.bin of src/query_disk_index.rs:686-709. */
typedef struct mse_codes mse_codes;
mse_codes* mse_codes_from_host(const uint8_t* codes, size_t n, size_t code_size, const uint8_t* descriptors,
                               size_t n_descriptors);
/* The same from rows already resident in HBM: codes = quantize_batch (vector.rs:331-364) of the f32 widenings of the base's
 * f16 rows -- the encode step of src/dump_processor.rs:468-481 -- computed on the device, 65536 rows at a time; only the
 * descriptor bytes (optional) cross PCIe.  Codes equal mse_pq_quantize_batch's on the same rows. */
mse_codes* mse_codes_quantize_base(mse_pq* pq, const mse_base* b, const uint8_t* descriptors, size_t n_descriptors);
void mse_codes_free(mse_codes* c);
size_t mse_codes_len(const mse_codes* c);
/* out[i] = adc(lut, codes[ids[i]]) + descriptor_product(scales, ids[i])   (query_disk_index.rs:189-203,135-142);
 * scales may be NULL (no descriptor bias). */
int mse_pq_adc_gather(mse_pq* pq, const mse_codes* c, const float* lut, const float* scales, const uint32_t* ids,
                      size_t n_ids, int64_t* out);
/* Full ADC scan of all codes, top-r by approximate score, exact re-score against `base` with the
 * f16 query, final top-k (BASELINE config 5).  base may be NULL: then scores are the ADC scores. */
int mse_pq_scan_topk(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* query_f32,
                     const float* scales, size_t r, size_t k, int64_t* scores, uint32_t* ids);
/* The same for nq queries ([nq][n_dims] f32) back to back on one stream with one upload and one download;
 * scores / ids are [nq][k].  The scan keeps one maximum per 64 vectors instead of a score per vector; the r best
 * vectors are then found inside the r best groups (exact, ties by lower id). */
int mse_pq_scan_topk_batch(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* queries_f32, size_t nq,
                           const float* scales, size_t r, size_t k, int64_t* scores, uint32_t* ids);
/* A shard's form of the batch call: results as a packed block on the device ([nq*k] i64 scores, [nq*k] u32 ids + id_offset; empty slots
 * INT64_MIN / MSE_ID_NONE), complete on return. */
int mse_pq_scan_topk_block(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* queries_f32, size_t nq, const float* scales,
                           size_t r, size_t k, uint64_t id_offset, void* block_dev);
/* Batches of >= 4 queries go through the codes four (12-bit tables) or eight (8-bit tables, batches of >= 8) queries per pass: an
 * integer nomination scan on the matrix cores whose answer is certified against the reference-order re-score of the nominated
 * vectors (csrc/pq.hip); a query whose certificate does not hold is repeated through the exact scan, so results are identical
 * either way.  This counts such repeats in the last batch call.  (A quantiser whose data defeats the 8-bit certificate -- more than
 * an eighth of a batch's eight-per-pass queries repeated -- goes back to four per pass for good.) */
uint32_t mse_pq_last_uncertified(mse_pq* pq);
/* test hook: the group maxima (best ADC score + descriptor bias of every 64 vectors, INT64_MIN past the end) the flat scan
 * nominates with; lut1 == NULL: the one-query kernel, else the two-queries-per-pass kernel.  out0 / out1: [ceil(n/64)] on the host. */
int mse_debug_pq_group_max(mse_pq* pq, const mse_codes* c, const float* lut0, const float* lut1, const float* scales, int64_t* out0,
                           int64_t* out1);
/* HIP-event timing of the four-queries-per-pass scan kernel inside mse_pq_scan_topk_batch (the dominant kernel, for the
 * roofline report): returns the totals accumulated so far, then sets the mode: 0 off, 1 on, 2 on and reset. */
int mse_pq_scan_timing(mse_pq* pq, int enable, double* total_ms, uint64_t* launches);
/* the sustained figure beside it: over the batch calls with at least four scans made while timing was on, *span_ms = time from the
 * first scan's start to the last scan's end, *scans = scans in those spans (back-to-back passes on two streams); reset with timing */
int mse_pq_scan_sustained(mse_pq* pq, double* span_ms, uint64_t* scans);
/* test hook: the integer nomination scan alone, per_pass = 4 (12-bit tables) or 8 (8-bit tables) queries per pass over the codes.
 * luts [per_pass][64*256] (n_valid of them used), scales NULL or [4]; out [per_pass][ceil(n/64)] u32 group maxima of the integer
 * sums, params_out [per_pass][4] = delta, c, eps, ok of each query's table. */
int mse_debug_pq4_group_max(mse_pq* pq, const mse_codes* c, const float* luts, const float* scales, int n_valid, int per_pass,
                            uint32_t* out, double* params_out);
/* descriptor_product (src/query_disk_index.rs:135-142) for one id, host-side helper. */
int64_t mse_descriptor_product(const float* scales, size_t n_descriptors, const uint8_t* descriptors, uint32_t id);

/* ---- NeighbourBuffer: diskann/src/lib.rs:74-155 (host) ----------------------------------- */
typedef struct mse_nb mse_nb;
mse_nb* mse_nb_new(size_t cap);
void mse_nb_free(mse_nb* b);
void mse_nb_clear(mse_nb* b);
size_t mse_nb_len(const mse_nb* b);
size_t mse_nb_cap(const mse_nb* b);
void mse_nb_insert(mse_nb* b, uint32_t id, int64_t score);
int mse_nb_next_unvisited(mse_nb* b, uint32_t* id);       /* 1 and *id, or 0 when none */
const uint32_t* mse_nb_ids(const mse_nb* b);
const int64_t* mse_nb_scores(const mse_nb* b);

/* In-RAM Vamana greedy search: diskann::greedy_search (lib.rs:183-211).  Traversal on the host,
 * neighbour scoring on the device (gather-and-score).  adj is [n][max_deg], deg[n].  Results
 * are left in `buf`, best first.  *n_distances receives GreedySearchCounters.distances. */
int mse_greedy_search(mse_searcher* s, const uint32_t* adj, const uint32_t* deg, size_t max_deg, uint32_t start,
                      const uint16_t* query, int base_vectors_only, uint32_t query_breakpoint, mse_nb* buf,
                      size_t* n_distances);

/* Disk-index beam search: query_disk_index::greedy_search (src/query_disk_index.rs:144-212) with the records of
 * index.bin (node.vector = the searcher's base rows, node.vertices = adj/deg, node.url.len() > 0 = has_url, NULL
 * meaning "all"), index.pq-codes.bin and index.descriptor-codes.bin (`c`) resident in HBM.  One batched device
 * submission per beam iteration; traversal on the host in the reference's order, including its quirks (entry
 * point inserted with score 0, :153; the pre-buffer is cleared per beam iteration, not per node, :157).
 * lut = mse_pq_preprocess_query output; scales = DescriptorScales (:463-471) or NULL.  Results: `buf` (best first),
 * the visited list in fetch order (ids + exact scores incl. bias; at most visited_cap written, *n_visited is the
 * full count), *cmps and *pq_cmps = the returned `(cmps, pq_cmps)` (:211). */
int mse_disk_greedy_search(mse_searcher* s, mse_pq* pq, const mse_codes* c, const uint32_t* adj, const uint32_t* deg,
                           size_t max_deg, const uint8_t* has_url, uint32_t start, const uint16_t* query, const float* lut,
                           const float* scales, int disable_pq, size_t beamwidth, mse_nb* buf, uint32_t* visited_ids,
                           int64_t* visited_scores, size_t visited_cap, size_t* n_visited, size_t* cmps, size_t* pq_cmps);
/* The same search, GPU-resident and batched over queries: one workgroup per query keeps the
 * NeighbourBuffer, the pre-buffer and the query's distance table in LDS and the visited sets as bit maps in HBM; no
 * host round trip during a search.  The adjacency lives on the device (mse_graph).  Per query q the outputs equal
 * those of mse_disk_greedy_search: buf_ids/buf_scores [nq][search_list] (first buf_len[q] valid, best first),
 * visited_* [nq][visited_cap] in fetch order, n_visited/cmps/pq_cmps [nq].  starts [nq]; queries [nq][d] f16; luts
 * [nq][64*256]; scales [nq][n_descriptors] or NULL.  Limits: 64 x 256 codec, search_list <= 1024, beamwidth <= 8,
 * max_deg <= 128 (merged indexes carry up to SHARD_SPILL x R neighbours per node, src/dump_processor.rs:282-291). */
/* Called with nq = 1 from many threads at once (the reference's request path: one greedy_search per HTTP request on its own task,
 * src/query_disk_index.rs:436-540,711-736), mse_disk_search_batch / _f32 meet in the graph's coalescer: calls that can share a launch
 * (same vectors, codec, codes, graph, search parameters and kind of inputs) run as ONE batched search, a workgroup per query, and
 * every caller gets exactly what its call returns when made alone. */
typedef struct mse_graph mse_graph;
mse_graph* mse_graph_from_host(const uint32_t* adj, const uint32_t* deg, size_t n, size_t max_deg, const uint8_t* has_url);
void mse_graph_free(mse_graph* g);
int mse_disk_search_batch(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts,
                          const uint16_t* queries, const float* luts, const float* scales, size_t nq, int disable_pq,
                          size_t beamwidth, size_t search_list, uint32_t* buf_ids, int64_t* buf_scores, uint32_t* buf_len,
                          uint32_t* visited_ids, int64_t* visited_scores, size_t visited_cap, uint32_t* n_visited,
                          uint32_t* cmps, uint32_t* pq_cmps);
/* The request path in ONE call (src/query_disk_index.rs:436-540 for a batch of queries): entry node, greedy_search, the visited
 * records ordered by exact score and cut to the first k -- nothing but the queries goes up and the k results come down.
 *   mse_graph_set_entries  the entry table: the reference starts a search at the medioid of the shard whose centroid is closest to
 *                          the query (:254-256,447-450); here node_ids name the entry records (shard medioids, or a sample of the
 *                          rows of a one-piece index) and a search starts at the one whose vector has the largest dot product
 *                          with the f16 query (exact top-1 on the device).  Copies of those vectors are kept with the graph.
 *   mse_disk_query_topk    starts == NULL: start nodes by the entry table; otherwise as given.  `queries` ([nq][d] f16) may be a host
 *                          OR a device pointer (embeddings that never left the GPU); everything else is host memory.  luts / scales / disable_pq /
 *                          beamwidth / search_list as mse_disk_search_batch.  ids / scores [nq][k]: the k best visited records by
 *                          (exact score + bias) descending -- equal scores by id ascending (the reference's sort is unstable there) --
 *                          padded with MSE_ID_NONE / INT64_MIN; identical to sorting mse_disk_search_batch's visited list.
 *                          n_visited / cmps / pq_cmps: [nq] or NULL.  Every visited record takes part (no visited_cap to choose).
 *   mse_graph_set_entry_centroids  the reference's entry rule itself (:254-256,447-450): centroids [n_entries][d] f32 are the shard
 *                          centroids of the index header, node_ids the shards' medioids; a search starts at the medioid of the shard
 *                          maximising scale_dot_result_f64(dot(centroid, query)) -- f32 operands (an f16 query widened exactly), the
 *                          sum carried in f64 in index order (as mse_select_shard), the LAST maximum on ties (position_max_by_key).
 *                          Replaces a table set by mse_graph_set_entries and vice versa.  Either setter waits for request-path calls
 *                          in flight and keeps new ones out while it runs.
 *   mse_disk_query_topk_f32  the handler as the reference runs it: f32 queries in (HOST memory); the entry step sees the f32 query, the
 *                          f16 copy (RNE, :477) scores the fetched nodes, preprocess_query (:475) makes the distance tables on the device.
 * THE REFERENCE'S CALL SHAPE (one request = one query on its own task, :436-540,711-736; perf_test.py: 1000 one-query requests at
 * concurrency 100): calls of mse_disk_query_topk(_f32) with nq <= 16 whose queries are host memory meet in the graph's coalescer.
 * Calls that can share a submission (same vectors, codec, codes, graph, disable_pq, beamwidth, search_list, kinds of inputs; k may
 * differ) run as ONE entry step + ONE search launch + ONE select on a searcher owned by the worker thread, and every caller gets
 * exactly what its call returns when made alone.  Such a call only reads `s` for the vectors it names: request threads may share one
 * searcher handle for these calls (4096 request threads do not need 4096 streams).  mse_graph_set_coalescer (before the first such
 * call, or with none in flight): queries per shared submission (0 = 1024), longest wait of the oldest request in microseconds
 * (0 = 200; a lone caller never waits), worker threads (0 = 3: the copies and host side of one submission overlap the kernels of the others).
 * mse_graph_coalescer_stats: {queries, requests, submissions, most queries in one submission, submissions started by the wait
 * budget, microseconds the workers spent executing submissions}.
 * DEVICE-RESIDENT QUERIES: the copy of `queries` runs on the searcher's stream.  If another stream produced them (a tower's), call
 * mse_searcher_wait_stream(s, that_stream) first -- or synchronise that stream -- else the search may read them half written. */
/* WITHOUT A THREAD PER REQUEST (round 5).  The device wants thousands of queries per submission; a sleeping OS thread per request is
 * the wrong vehicle for that (4096 request threads on a 16-core CPU allowance: 25 us of CPU per request just for being woken).  An
 * async host -- the reference serves every connection as a monoio task on a runtime per core (src/query_disk_index.rs:640-655,
 * 716-732) -- keeps its requests in flight as tickets:
 *   mse_disk_query_submit_f32  as mse_disk_query_topk_f32 with nq = 1..16 and entry by the graph's table, but returns as soon as the
 *                          request is queued.  The query (and scales) are copied: the caller's buffers are free at once.  ids / scores
 *                          (/ n_visited / cmps / pq_cmps) are written when the request is executed and must stay valid until its
 *                          ticket has come back.  `user` travels with the ticket (a oneshot sender, a request id).
 *   mse_graph_completions  hands back up to `max` tickets of executed requests of this graph, each exactly once, in completion order;
 *                          sleeps up to timeout_us for the first (0: poll, < 0: no limit).  Returns how many (0: none in time), -1 on
 *                          error.  Any number of threads may submit and collect; a ticket comes back to whichever thread asks next.
 *   mse_completion_queue_* a completion queue of the caller's own.  A host with several event loops -- the reference runs a runtime per
 *                          core -- makes one per loop and passes it as `cq` at submit: those tickets come back through
 *                          mse_completion_queue_wait(q, …) (and its eventfd, mse_completion_queue_fd) and nowhere else, i.e. to the
 *                          loop that submitted them; the shared submissions are the same.  cq = NULL: the graph's own queue
 *                          (mse_graph_completions / mse_graph_completion_fd).  Free a queue only when none of its tickets is out.
 *   mse_graph_completion_fd  an eventfd owned by the graph (valid until its coalescer settings change or it is freed; -1 on error)
 *                          whose counter is bumped once per submission that completed tickets: register it with epoll / io_uring, read
 *                          the 8-byte counter when it fires, then call mse_graph_completions(…, 0) until it returns 0.
 *   mse_ticket_status / _error / _user / _free   0 or the request's error (with its message); the user pointer; release (tickets
 *                          are recycled per thread: a poller that submits and releases on one thread allocates nothing in steady state).
 * Results are those of the synchronous call, bit for bit (the same shared submissions execute both kinds).  Do not free the graph
 * or change its coalescer settings while tickets are out; its entry table may be replaced (a queued request starts from the table
 * that is set when it executes). */
typedef struct mse_ticket mse_ticket;
typedef struct mse_completion_queue mse_completion_queue;
int mse_disk_query_submit_f32(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const float* queries_f32, const float* scales,
                              size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k, uint32_t* ids, int64_t* scores,
                              uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps, void* user, mse_completion_queue* cq,
                              mse_ticket** ticket_out);
/* the same without the copies: queries_f32 (and scales) must stay valid and unchanged until the ticket has come back */
int mse_disk_query_submit_f32_nocopy(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const float* queries_f32,
                                     const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k, uint32_t* ids,
                                     int64_t* scores, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps, void* user, mse_completion_queue* cq,
                                     mse_ticket** ticket_out);
mse_completion_queue* mse_completion_queue_new(void);
void mse_completion_queue_free(mse_completion_queue* q);
int mse_completion_queue_fd(mse_completion_queue* q);
long mse_completion_queue_wait(mse_completion_queue* q, mse_ticket** out, size_t max, long timeout_us);
long mse_graph_completions(const mse_graph* g, mse_ticket** out, size_t max, long timeout_us);
int mse_graph_completion_fd(const mse_graph* g);
int mse_ticket_status(const mse_ticket* t);
const char* mse_ticket_error(const mse_ticket* t);
void* mse_ticket_user(const mse_ticket* t);
void mse_ticket_free(mse_ticket* t);
int mse_graph_set_entries(mse_graph* g, const mse_base* b, const uint32_t* node_ids, size_t n_entries);
int mse_graph_set_entry_centroids(mse_graph* g, const float* centroids, size_t d, const uint32_t* node_ids, size_t n_entries);
int mse_disk_query_topk(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts, const uint16_t* queries,
                        const float* luts, const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k,
                        uint32_t* ids, int64_t* scores, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps);
int mse_disk_query_topk_f32(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts, const float* queries_f32,
                            const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k, uint32_t* ids,
                            int64_t* scores, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps);
/* A shard's form of the call (multi-GPU, below): the [nq][k] results stay on the device as a packed block -- [nq*k] i64 scores, then
 * [nq*k] u32 ids + id_offset (mse_topk_block_bytes(nq, k) bytes at block_dev) -- ready for the exchange; never coalesced. */
int mse_disk_query_topk_block(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts, const uint16_t* queries,
                              const float* luts, const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k,
                              uint64_t id_offset, void* block_dev, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps);
/* Measurement hook (bench.py's gather roofline for the graph search; no counterpart in the reference): HIP events around every
 * beam_search_kernel launch of this searcher + device totals of what the searches gathered.  enable: 0 off, 1 on, 2 on and reset.
 * out (optional, 8 words, read BEFORE `enable` takes effect): kernel microseconds, launches, queries, rows gathered for an exact score
 * (one 2 x d-byte row gather each: the exactly scored neighbours + the entry point, or -- ADC scoring -- the fetched nodes), nodes
 * fetched (one adjacency list each), neighbours
 * scored by ADC (one 64-byte code gather each), beam iterations, iterations whose inserts ran sequentially (equal scores in play). */
int mse_searcher_beam_timing(mse_searcher* s, int enable, uint64_t out[8]);
/* The handler's runtime de-duplication (src/query_disk_index.rs:482-527, DUPLICATES_THRESHOLD 0.95 :99) INSIDE mse_disk_query_topk(_f32)
 * and its block form: before the visited records are ordered, a record whose vector has a dot product above `threshold` with an ALREADY
 * KEPT record (f32 products of the f16 rows, summed k-ascending as mse_dedup_visited; visit order) is dropped -- for every query of the
 * batch on the device.  0 = off (the default: the k best of ALL visited records).  n_visited still counts every visited record.
 * At most 4096 visited records per query (search lists up to ~2000).  Set while no request-path call is in flight. */
int mse_graph_set_dedup(mse_graph* g, float threshold);
int mse_graph_set_coalescer(mse_graph* g, size_t max_queries_per_pass, uint32_t max_wait_us, int workers);
int mse_graph_coalescer_stats(const mse_graph* g, uint64_t out[6]);
/* everything `producer_stream` (a hipStream_t) holds at the time of the call completes before anything this searcher's stream is
 * given afterwards starts: an event recorded there, waited for here; no host synchronisation */
int mse_searcher_wait_stream(mse_searcher* s, void* producer_stream);
/* ---- Vamana graph build on the device (diskann/src/lib.rs:183-389, driven by
 * src/generate_index_shard.rs:85-133) ----
 * The graph being built is an mse_graph with max_deg = r (lists of at most r ids, stride r) that stays in HBM
 * between the passes; vectors are the searcher's base rows (base vectors first, then the query vectors of the
 * OOD-DiskANN variant, ids >= query_breakpoint).  The reference leaves two things to chance, and both are
 * arguments here: the insertion order (rng.shuffle, lib.rs:291-292,333-334) and the random initial graph. */
typedef struct mse_build_config {   /* IndexBuildConfig, lib.rs:42-52; defaults generate_index_shard.rs:22-33,85-94 */
    uint64_t r, l, maxc;            /* degree bound (<= 64), search list (<= 1024), candidate cap (<= 1024) */
    int64_t alpha, query_alpha;     /* relaxation factors times 2^16 */
    uint32_t saturate_graph, query_breakpoint;
    uint64_t max_add_per_stitch_iter;
} mse_build_config;
mse_graph* mse_graph_new(size_t n, size_t max_deg);                        /* IndexGraph::empty (lib.rs:22-31) */
int mse_graph_to_host(const mse_graph* g, uint32_t* adj, uint32_t* deg);   /* [n][max_deg], [n] */
size_t mse_graph_len(const mse_graph* g);
size_t mse_graph_max_degree(const mse_graph* g);
/* random_fill_graph (lib.rs:376-389): every list topped up to r distinct uniformly drawn ids (a node may draw
 * itself).  The reference draws from clock-seeded fastrand forks; here draw k of node i is
 * mulhi(philox4x32-10(ctr = (k,0,i,0), key = (seed, 0xF111))[0], n), so a seed names one graph. */
int mse_graph_random_fill(mse_graph* g, uint32_t seed, size_t r);
/* build_graph (lib.rs:287-324) over order[0..n_order): for each point greedy_search from the medioid (:183-211),
 * merge_existing_neighbours (:215-221), robust_prune (:227-285), then the back edges (:311-322).  The reference
 * feeds the points to rayon workers under per-list locks, so its result depends on thread timing; here the points
 * are taken `batch` at a time: the searches and prunes of a batch see the graph as it was before the batch, then the
 * batch's lists are replaced, then the back edges are applied in (position in batch, position in list) order.
 * batch = 1 is exactly the single-threaded loop the reference keeps in comments (:294,297).  One workgroup per point
 * (search list, candidates, prune state in LDS, visited set as a bit map in HBM), one wave per touched list for the
 * back edges.  Every score that orders a list or is compared against is the reference's fast_dot, bit for bit; the
 * candidate-candidate products of the two prunes, which only feed `(alpha * s) >> 16 >= score`, may come from the
 * matrix cores, and then decide only when the comparison holds across their error bound (the exact dot settles the
 * rest), so the graph is the same as with exact products throughout. */
int mse_build_graph(mse_searcher* s, mse_graph* g, const uint32_t* order, size_t n_order, size_t batch, uint32_t medioid,
                    const mse_build_config* cfg);
/* robust_stitch (lib.rs:326-374): query nodes are removed from the base nodes' lists; each base node that pointed
 * at a query receives up to max_add_per_stitch_iter of that query's out-neighbours, best first.  queries_order
 * [n - query_breakpoint] = the shuffled query ids, applied one after another. */
int mse_robust_stitch(mse_searcher* s, mse_graph* g, const uint32_t* queries_order, const mse_build_config* cfg);
/* robust_prune alone (lib.rs:227-285) on a caller-supplied candidate list (scratch.visited_list); neigh has room for
 * cfg->r ids, *n_neigh receives the count.  n_cand is unbounded (the best maxc are kept, :233-234). */
int mse_robust_prune(mse_searcher* s, const uint32_t* cand_ids, const int64_t* cand_scores, size_t n_cand, uint32_t p,
                     const mse_build_config* cfg, uint32_t* neigh, size_t* n_neigh);
/* diskann::greedy_search (lib.rs:183-211) GPU-resident and batched over queries (the in-RAM scorer, A21): one
 * workgroup per query, outputs as mse_greedy_search leaves them in `buf`: buf_ids/buf_scores [nq][search_list]
 * (first buf_len[q] valid, best first), n_distances [nq] = GreedySearchCounters.distances. */
int mse_graph_search_batch(mse_searcher* s, const mse_graph* g, const uint32_t* starts, const uint16_t* queries, size_t nq,
                           size_t search_list, int base_vectors_only, uint32_t query_breakpoint, uint32_t* buf_ids,
                           int64_t* buf_scores, uint32_t* buf_len, uint32_t* n_distances);

/* The same with the caller's two preparation steps (src/query_disk_index.rs:475-477) done on the device: queries arrive
 * as f32 [nq][d]; their f16 copies (RNE, half::f16::from_f32) score the fetched nodes and preprocess_query
 * (vector.rs:367-384) makes the distance tables in HBM, so 64 KiB per query stay off PCIe. */
int mse_disk_search_batch_f32(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts,
                              const float* queries_f32, const float* scales, size_t nq, int disable_pq, size_t beamwidth,
                              size_t search_list, uint32_t* buf_ids, int64_t* buf_scores, uint32_t* buf_len,
                              uint32_t* visited_ids, int64_t* visited_scores, size_t visited_cap, uint32_t* n_visited,
                              uint32_t* cmps, uint32_t* pq_cmps);
/* Result de-duplication of the visited list (src/query_disk_index.rs:482-527): S = V V^T over the visited rows
 * (ids into the searcher's base, visit order), greedy keep-first filter with S[i][j] > threshold (0.95, :99) against
 * already kept rows.  keep[i] = 1 for survivors. */
int mse_dedup_visited(mse_searcher* s, const uint32_t* ids, size_t n, float threshold, uint8_t* keep);
/* Shard / entry-point selection (src/query_disk_index.rs:254-256,447-450): argmax over shard centroids of
 * scale_dot_result_f64(dot_f32(centroid, query)), LAST maximum on ties (position_max_by_key). */
int mse_select_shard(const float* centroids, size_t n_shards, size_t d, const float* query, size_t* shard_out);
/* medioid (diskann/src/lib.rs:52-68): row with the largest `dot` (vector.rs:49-52) against the f16-rounded running
 * mean of all rows; last maximum on ties. */
int mse_medioid(const mse_base* b, uint32_t* id_out);

/* ---- Index packing (dump_processor; quantize_batch is mse_pq_quantize_batch above) ----
 * ScoreModel::score_batch (src/score_model.rs:13-32): out[b] = down_proj . silu(up_proj . x_b + bias) * d_emb / d_hidden.
 * up_proj [d_hidden][d_emb], bias [d_hidden], down_proj [out_channels][d_hidden], row-major f32 (the safetensors layout). */
typedef struct mse_score_model mse_score_model;
mse_score_model* mse_score_model_load(const float* up_proj, const float* bias, const float* down_proj, size_t d_emb,
                                      size_t d_hidden, size_t out_channels);
void mse_score_model_free(mse_score_model* m);
size_t mse_score_model_output_channels(const mse_score_model* m);
int mse_score_model_score_batch(mse_score_model* m, const float* input, size_t batch, float* out /* [batch][out_channels] */);
/* Descriptor bytes (src/dump_processor.rs:483-491): out[i][j] = position of scores[i][j] in the ascending cdfs[j]
 * as `binary_search_by(|x| x.partial_cmp(score))` reports it (Ok(p) or Err(p) -> p), one byte; cdf_len <= 255
 * (meme-rater/compute_cdf.py: 255 quantiles, 255 = above the last). */
int mse_descriptor_buckets(const float* cdfs, size_t n_desc, size_t cdf_len, const float* scores, size_t n, uint8_t* out);

/* ---- SigLIP ViT image tower: the in-process seam of clip_server.py, `fast_image_fns[batch](images NCHW
 * fp16 on device) -> [batch, 1152]` (clip_server.py:31,66-82,105-112), plus the normalisation and fp16
 * serialisation of do_inference / run_inference (clip_server.py:115,166).  Graph: aitemplate/model.py:13-123;
 * hyper-parameters aitemplate/run.py:47-55; weight names clip_server.py:40-57 without the "visual." prefix. */
typedef struct mse_siglip mse_siglip;
typedef struct mse_siglip_config {
    int img_size;    /* 384 */
    int patch_size;  /* 14 */
    int in_chans;    /* 3 */
    int emb_dim;     /* 1152 */
    int depth;       /* 27 */
    int num_heads;   /* 16 */
    int mlp_dim;     /* 4304 */
    float eps;       /* LayerNorm epsilon, 1e-6 */
    int gelu_tanh;   /* 0 = erf GELU (timm / AITemplate "gelu"), 1 = tanh approximation (HF / big_vision) */
    int max_batch;   /* `max_batch_size` of clip_server_config.json */
} mse_siglip_config;
mse_siglip* mse_siglip_create(const mse_siglip_config* cfg);
void mse_siglip_destroy(mse_siglip* m);
int mse_siglip_n_weights(const mse_siglip* m);
const char* mse_siglip_weight_name(const mse_siglip* m, int idx);   /* names the engine expects, sorted */
/* fp32 host tensor in its state-dict shape (e.g. qkv.weight [3456,1152], patch_embed.proj.weight [1152,3,14,14]) */
int mse_siglip_set_weight(mse_siglip* m, const char* name, const float* data, const size_t* shape, int ndim);
int mse_siglip_finalize(mse_siglip* m);                              /* fails if a weight is missing */
/* images: [batch,3,H,W], dtype 0 = f32 / 1 = f16, already normalised (x/127.5 - 1); batch > max_batch is an
 * error (the reference asserts, clip_server.py:139).  Outputs (either may be NULL) are host [batch, emb_dim].
 * Batch invariance: rows of calls of >= 5 images are bit-equal whatever the batch.  Calls of 1-4 images run small-batch
 * kernels with another summation order: an image embedded alone (query time) and the same image inside a larger batch (index
 * time) agree to bf16 rounding (cosine within 1e-4; both within 1e-3 of the fp32 model), not bit for bit.  A pipeline that needs
 * index/query bit-equality sets MSE_SIGLIP_NOSMALL=1 in the environment before mse_siglip_create (read once, there). */
int mse_siglip_encode_image(mse_siglip* m, const void* images, int dtype, int on_device, int batch, int normalize,
                            float* out_f32, uint16_t* out_f16);
/* Same from decoded RGB bytes [batch][H][W][3] (host): the ToTensor / Normalize(0.5, 0.5) / .half() / stack steps of the
 * preprocessing thread (clip_server.py:131-146) run on the device; x / 127.5 - 1 in fp32, fp16 round-to-nearest-even. */
int mse_siglip_encode_rgb8(mse_siglip* m, const uint8_t* rgb_hwc, int batch, int normalize, float* out_f32, uint16_t* out_f16);
/* Same from the request bytes themselves when they are what the reference's clients send (src/common.rs:31-54: 24-bit
 * uncompressed BMP of exactly image_size): the host reads the 54-byte header, the device does BGR -> RGB, the bottom-up row
 * flip and the normalisation.  Any other file (or size) is an error: decode it on the host and use mse_siglip_encode_rgb8.
 * mse_bmp24_info is the header check alone (0 = plain 24-bit BMP; width / height / pixel_offset / bottom_up may be NULL). */
int mse_bmp24_info(const uint8_t* data, size_t size, uint32_t* width, uint32_t* height, uint32_t* pixel_offset, int* bottom_up);
int mse_siglip_encode_bmp(mse_siglip* m, const uint8_t* const* bmps, const size_t* sizes, int batch, int normalize, float* out_f32,
                          uint16_t* out_f16);
const void* mse_siglip_output_device(const mse_siglip* m, int which);  /* device result of the last call: 0 f32, 1 f16 */
void* mse_siglip_stream(const mse_siglip* m);
int mse_siglip_debug_residual(mse_siglip* m, float* out);             /* test hook: residual stream after the last block */
int mse_debug_gemm_ms(int M, int N, int K, int ablation, int iters, float* ms_out); /* developer hook: GEMM timing/ablation */
/* developer / test hook for the small-batch GEMM kernels of the towers (one image, a few texts): `rows` real rows, bias (epi 0) or
 * bias + GELU (epi 1) epilogue, run by `variant` (0 = large-batch kernels, 1 = chosen by size, 2 = K-split skinny, 3 = 64 x 64 tiles,
 * 4 = 128 x 128 tiles).  ms_out = average launch time with weights streaming from HBM; n_diff (optional, TWO words) = output
 * elements that differ from the large-batch kernels' result, and those more than two bf16 steps apart (the kernels differ in
 * summation order only: the second word must be 0). */
int mse_debug_gemm_small(int rows, int N, int K, int epi, int variant, int iters, float* ms_out, uint64_t* n_diff);

/* ---- SigLIP text tower: `model.encode_text(tokens)` + normalisation + fp16 serialisation
 * (clip_server.py:98-99,128-131,166).  open_clip's TextTransformer is a third-party dependency not vendored in
 * the reference; geometry from misc/clip_accursed.py:31-55 and clip_server.py:107,182: width 1152, 27 layers,
 * 16 heads, mlp 4304, context 64, vocabulary 32000, no causal mask, last position pooled, Linear projection with
 * bias.  Weight names are open_clip's (`text.token_embedding.weight`, `text.transformer.resblocks.N. ...`).
 * Tokenisation (sentencepiece, pad id 1, clip_server.py:129) stays on the host. */
typedef struct mse_siglip_text mse_siglip_text;
typedef struct mse_siglip_text_config {
    int width;           /* 1152 */
    int layers;          /* 27 */
    int heads;           /* 16 */
    int mlp_dim;         /* 4304 */
    int context_length;  /* 64 */
    int vocab_size;      /* 32000 */
    float eps;           /* 1e-6 */
    int gelu_tanh;       /* 0 erf, 1 tanh approximation */
    int max_batch;
} mse_siglip_text_config;
mse_siglip_text* mse_siglip_text_create(const mse_siglip_text_config* cfg);
void mse_siglip_text_destroy(mse_siglip_text* m);
int mse_siglip_text_n_weights(const mse_siglip_text* m);
const char* mse_siglip_text_weight_name(const mse_siglip_text* m, int idx);
int mse_siglip_text_set_weight(mse_siglip_text* m, const char* name, const float* data, const size_t* shape, int ndim);
int mse_siglip_text_finalize(mse_siglip_text* m);
/* tokens: host int64 [batch, context_length]; outputs (either may be NULL) are host [batch, width]. */
int mse_siglip_text_encode(mse_siglip_text* m, const int64_t* tokens, int batch, int normalize, float* out_f32,
                           uint16_t* out_f16);
/* The same forward with the features left ON THE DEVICE, nothing copied back and nothing waited for: the request handler embeds the
 * query text (src/query_disk_index.rs:345-381) and searches with it (:436-540) -- order the searcher behind the engine's stream
 * (mse_searcher_wait_stream(s, mse_siglip_text_stream(m))) and pass mse_siglip_text_output_device(m, 1) as the device-resident f16
 * queries of mse_disk_query_topk.  `tokens` (host) must stay untouched until the engine's stream has read them; the output buffers
 * (which: 0 = f32 [batch][width], 1 = f16) belong to the engine and hold this call's rows until its next call. */
int mse_siglip_text_encode_dev(mse_siglip_text* m, const int64_t* tokens, int batch, int normalize);
const void* mse_siglip_text_output_device(const mse_siglip_text* m, int which);
void* mse_siglip_text_stream(const mse_siglip_text* m);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MSE_H */
