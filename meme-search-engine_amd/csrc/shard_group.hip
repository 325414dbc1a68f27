// Row-sharded brute-force search over the GPUs of one node; C ABI in include/mse.h.
//
// The reference has no multi-GPU code; what it has is a thread per core, each with its own scratch over shared read-only
// maps (src/query_disk_index.rs:711-736).  Scoring is row-independent, so the same shape carries over: base rows are partitioned
// contiguously, every shard scores the same query batch and returns GLOBAL ids (local id + the shard's first row), the
// per-shard [nq][k] records meet in one buffer, and a k-way merge by (score desc, id asc) gives the result of the whole index.
//
// Two ways for the records to meet, both behind this file, neither touching torch:
//   mse_shard_group   ONE process drives every shard: a persistent host thread per shard (hipSetDevice once, own mse_base,
//                     mse_searcher and stream).  Shards may share a device ("logical shards"; how the 8-way layout is tested on a
//                     one-GPU box).  The gather buffer [G] x {[nq][k] i64, [nq][k] u32} lives on the root device; a shard on
//                     another device writes its block straight into it through a peer mapping (hipDeviceEnablePeerAccess:
//                     the finalize kernel's few KB of stores cross xGMI), or, without peer access, by one hipMemcpyPeerAsync.
//   mse_comm          one process per GPU (torchrun's shape): ONE ncclAllGather of the packed 12-byte-per-record block on the
//                     searcher's stream through librccl.so (dlopen'ed here; the unique id travels through the host's own
//                     rendezvous), then the same merge on every rank.
// The merge is the radix select of topk.hip over the G*k candidates of a query (unique composite key, so deterministic).
#include "../../include/mse.h"
#include "runtime.h"
#include <dlfcn.h>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace mse;

namespace {

// packed per-shard block: [nq*k] i64 scores, then [nq*k] u32 ids, padded to 16 bytes
inline size_t block_bytes(size_t nq, size_t k) { return (nq * k * 12 + 15) & ~(size_t)15; }

// merge G packed blocks (device memory of the searcher's device) -> out [nq][k]; asynchronous on the searcher's stream
// k_in records per shard and query in the blocks, the k best of all of them out ([nq][k]; k_in = 0: k_in = k)
int merge_packed(mse_searcher* s, const char* gathered, size_t n_shards, size_t nq, size_t k, void* out_scores_dev,
                 void* out_ids_dev, size_t k_in = 0) {
    if (k_in == 0) k_in = k;
    if (k > (size_t)TOPK_KMAX || k_in > (size_t)TOPK_KMAX) return fail("k too large");
    if (s->misc.ensure(nq * k * 4) || s->sel_keys.ensure(nq * k * 8)) return -1;
    const size_t B = block_bytes(nq, k_in);
    SelectArgs a{};
    a.kind = KEY_I64;
    a.list_keys = gathered;
    a.list_ids = reinterpret_cast<const uint32_t*>(gathered + nq * k_in * 8);
    a.list_stride = k_in;               // query q starts k_in records into each shard's block
    a.list_chunk = k_in;
    a.list_chunk_stride = B / 8;        // next shard, in i64 elements
    a.list_id_chunk_stride = B / 4;     // next shard, in u32 elements
    a.n_list = n_shards * k_in;
    a.k = (int)k; a.out_ids = s->misc.as<uint32_t>(); a.out_keys = s->sel_keys.p; a.out_stride = k; a.nq = (int)nq;
    if (launch_select(a, s->stream)) return -1;
    return launch_finalize(s->misc.as<uint32_t>(), s->sel_keys.as<int64_t>(), k, (int)k, (int)nq, 0,
                           reinterpret_cast<int64_t*>(out_scores_dev), reinterpret_cast<uint32_t*>(out_ids_dev), k, nullptr, 0,
                           0, 0, nullptr, nullptr, s->stream);
}

// rows of the global top-r that live on this shard: local id = global id - first_row, ID_NONE for everybody else's
__global__ void members_local_kernel(const uint32_t* __restrict__ gids, size_t n, uint64_t first_row, uint64_t n_local, uint32_t* __restrict__ local) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = gids[i];
    local[i] = (g != ID_NONE && g >= first_row && g - first_row < n_local) ? (uint32_t)(g - first_row) : ID_NONE;
}

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, mse_comm_id, int) = nullptr;   // ncclUniqueId is a 128-byte struct passed by value
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;     // one process, several devices
    int (*CommAbort)(void*) = nullptr;                         // optional: frees the ranks a failed collective left waiting
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process already holds (the torch wheel bundles one) is reused; otherwise ROCm's
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!r.lib) for (const char* n : names) if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!r.lib) { r.why = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p && r.why.empty()) r.why = std::string("librccl.so lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
    });
    return r;
}

int rccl_ready() {
    Rccl& r = rccl();
    if (!r.why.empty()) return fail(r.why);
    return 0;
}
int rccl_fail(const char* what, int code) {
    Rccl& r = rccl();
    return fail(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(code) : "RCCL error") + " (" + std::to_string(code) + ")");
}

struct Shard {
    int device = 0;
    size_t first_row = 0;       // global id of local row 0
    mse_base* base = nullptr;
    mse_searcher* searcher = nullptr;
    DevBuf q_local, block;      // on this shard's device (used when it cannot reach the root's memory directly, and by RCCL)
    DevBuf gathered;            // RCCL exchange: every shard's block, on this shard's device
    void* comm = nullptr;       // RCCL exchange: this shard's communicator (rank = shard index)
    // approximate-search state of the shard's rows (round 5; handles are the caller's, made on this shard's device):
    mse_pq* pq = nullptr;               // the codec (the same centroids / rotation on every shard)
    const mse_codes* codes = nullptr;   // PQ codes (+ descriptor bytes) of THIS shard's rows
    const mse_graph* graph = nullptr;   // a Vamana graph over THIS shard's rows, with its own entry table
    DevBuf ann_a, ann_b;                // phase-B scratch of the sharded PQ scan: local ids, exact scores
    bool peer = false;          // may read/write root-device memory from kernels
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // step breakdown: start, local search done, exchange done
    float local_ms = 0.f, exch_ms = 0.f;
    std::string err;
    int rc = 0;
};

}  // namespace

struct mse_shard_group {
    size_t d = 0;
    int root_device = 0;
    std::vector<Shard> shards;
    mse_searcher* root = nullptr;       // scratch searcher on the root device: merge + its stream
    DevBuf gathered, q_root, out_s, out_i;
    // persistent workers: run(fn) executes fn(shard index) on every shard's thread and waits
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::function<int(size_t)> job;
    uint64_t job_seq = 0;
    size_t pending = 0;
    bool stop = false;
    std::mutex call_mu;                 // one search at a time per group (a second group = a second set of scratch)
    int exchange = 0;                   // MSE_EXCHANGE_PEER (0): peer stores / staged copies; MSE_EXCHANGE_RCCL (1): ncclAllGather
    int rccl_ranks = 0;                 // as ncclCommCount reports them
    hipEvent_t ev_merge[2] = {nullptr, nullptr};
    double last_ms[4] = {0, 0, 0, 0};   // last search: max local search, max exchange, merge, wall

    void worker(size_t g) {
        (void)hipSetDevice(shards[g].device);
        uint64_t seen = 0;
        for (;;) {
            std::function<int(size_t)> fn;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || job_seq != seen; });
                if (stop) return;
                seen = job_seq;
                fn = job;
            }
            set_error("");
            const int rc = fn(g);
            shards[g].rc = rc;
            shards[g].err = rc ? std::string(mse_last_error()) : std::string();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    int run(std::function<int(size_t)> fn) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(fn);
            pending = shards.size();
            job_seq++;
        }
        cv_job.notify_all();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        for (size_t g = 0; g < shards.size(); g++)
            if (shards[g].rc) return fail("shard " + std::to_string(g) + ": " + shards[g].err);
        return 0;
    }
};

// contiguous split, remainder spread over the first shards (the same rule as the host mirror's shard_range)
static void shard_range(size_t n, size_t g, size_t G, size_t* lo, size_t* hi) {
    const size_t base = n / G, rem = n % G;
    *lo = g * base + std::min(g, rem);
    *hi = *lo + base + (g < rem ? 1 : 0);
}

extern "C" {

size_t mse_topk_block_bytes(size_t nq, size_t k) { return block_bytes(nq, k); }
int mse_merge_topk_packed_dev(mse_searcher* s, const void* gathered_blocks_dev, size_t n_shards, size_t nq, size_t k,
                              void* out_scores_dev, void* out_ids_dev) {
    if (!s) return fail("null searcher");
    if (nq == 0 || k == 0 || n_shards == 0) return 0;
    return merge_packed(s, reinterpret_cast<const char*>(gathered_blocks_dev), n_shards, nq, k, out_scores_dev, out_ids_dev);
}

mse_shard_group* mse_shard_group_new(const int* devices, size_t n_shards, size_t d) {
    if (n_shards == 0 || n_shards > 1024) { fail("shard group: 1..1024 shards"); return nullptr; }
    if (d == 0 || d % 64 != 0 || d > (size_t)D_MAX) { fail("vector width must be a positive multiple of 64"); return nullptr; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { fail("shard group: no HIP device"); return nullptr; }
    for (size_t g = 0; g < n_shards; g++)
        if (devices && (devices[g] < 0 || devices[g] >= n_dev)) { fail("shard group: device ordinal out of range"); return nullptr; }
    mse_shard_group* G = new (std::nothrow) mse_shard_group();
    if (!G) { fail("out of host memory"); return nullptr; }
    G->d = d;
    G->shards.resize(n_shards);
    for (size_t g = 0; g < n_shards; g++) G->shards[g].device = devices ? devices[g] : (int)(g % (size_t)n_dev);
    G->root_device = G->shards[0].device;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(G->root_device);
    G->root = scratch_searcher_new();
    (void)hipSetDevice(prev);
    if (!G->root) { delete G; return nullptr; }
    try {
        for (size_t g = 0; g < n_shards; g++) G->threads.emplace_back([G, g] { G->worker(g); });
    } catch (...) {   // nothing may be thrown across the ABI: a thread that could not start is an error like any other
        fail("shard group: could not start the worker threads");
        G->shards.resize(G->threads.size());   // run() waits for exactly the workers that exist
        mse_shard_group_free(G);
        return nullptr;
    }
    // peer mappings towards the root, once per distinct device
    // test hook: MSE_SHARD_NO_PEER=1 sends every shard down the staged path (queries copied in, block copied back) that devices
    // without a peer mapping take
    const bool no_peer = getenv("MSE_SHARD_NO_PEER") != nullptr;
    const int rc = G->run([G, no_peer](size_t g) -> int {
        Shard& sh = G->shards[g];
        if (no_peer) { sh.peer = false; return 0; }
        if (sh.device == G->root_device) { sh.peer = true; return 0; }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, sh.device, G->root_device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(G->root_device, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) sh.peer = true;
            (void)hipGetLastError();
        }
        return 0;
    });
    if (rc) { mse_shard_group_free(G); return nullptr; }
    return G;
}

void mse_shard_group_free(mse_shard_group* G) {
    if (!G) return;
    if (!G->threads.empty()) {
        (void)G->run([G](size_t g) -> int {   // handles are released on the thread (and device) that made them
            Shard& sh = G->shards[g];
            if (sh.searcher) mse_searcher_free(sh.searcher);
            if (sh.base) mse_base_free(sh.base);
            sh.searcher = nullptr; sh.base = nullptr;
            sh.q_local.release(); sh.block.release(); sh.gathered.release();
            if (sh.comm) { (void)rccl().CommDestroy(sh.comm); sh.comm = nullptr; }
            for (hipEvent_t& e : sh.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            return 0;
        });
        {
            std::lock_guard<std::mutex> lk(G->mu);
            G->stop = true;
        }
        G->cv_job.notify_all();
        for (auto& t : G->threads) t.join();
    }
    for (hipEvent_t& e : G->ev_merge) if (e) (void)hipEventDestroy(e);
    if (G->root) mse_searcher_free(G->root);
    delete G;
}

size_t mse_shard_group_n_shards(const mse_shard_group* G) { return G ? G->shards.size() : 0; }
size_t mse_shard_group_len(const mse_shard_group* G) {
    size_t n = 0;
    if (G) for (const Shard& s : G->shards) n += s.base ? s.base->n : 0;
    return n;
}
int mse_shard_group_device(const mse_shard_group* G, size_t shard) {
    return (G && shard < G->shards.size()) ? G->shards[shard].device : -1;
}
const mse_base* mse_shard_group_base(const mse_shard_group* G, size_t shard) {
    return (G && shard < G->shards.size()) ? G->shards[shard].base : nullptr;
}
uint64_t mse_shard_group_first_row(const mse_shard_group* G, size_t shard) {
    return (G && shard < G->shards.size()) ? (uint64_t)G->shards[shard].first_row : 0;
}
mse_searcher* mse_shard_group_searcher(mse_shard_group* G, size_t shard) {
    return (G && shard < G->shards.size()) ? G->shards[shard].searcher : nullptr;
}
int mse_shard_group_peer_mapped(const mse_shard_group* G, size_t shard) {
    return (G && shard < G->shards.size()) ? (G->shards[shard].peer ? 1 : 0) : 0;
}

static int install(Shard& sh, mse_base* b, size_t first_row) {
    if (!b) return -1;
    mse_searcher* s = mse_searcher_new(b);
    if (!s) { mse_base_free(b); return -1; }
    if (sh.searcher) mse_searcher_free(sh.searcher);
    if (sh.base) mse_base_free(sh.base);
    sh.base = b; sh.searcher = s; sh.first_row = first_row;
    return 0;
}

int mse_shard_group_generate(mse_shard_group* G, uint32_t seed, uint64_t first_row, size_t total_rows) {
    if (!G) return fail("null shard group");
    if (first_row + total_rows > 0xFFFFFFFEull) return fail("row ids are u32: too many rows");
    std::lock_guard<std::mutex> call(G->call_mu);
    return G->run([=](size_t g) -> int {
        size_t lo, hi;
        shard_range(total_rows, g, G->shards.size(), &lo, &hi);
        return install(G->shards[g], mse_base_generate(seed, first_row + lo, hi - lo, G->d), first_row + lo);
    });
}

int mse_shard_group_load_host(mse_shard_group* G, const uint16_t* rows, size_t total_rows) {
    if (!G) return fail("null shard group");
    if (total_rows > 0xFFFFFFFEull) return fail("row ids are u32: too many rows");
    std::lock_guard<std::mutex> call(G->call_mu);
    return G->run([=](size_t g) -> int {
        size_t lo, hi;
        shard_range(total_rows, g, G->shards.size(), &lo, &hi);
        return install(G->shards[g], mse_base_from_host(rows + lo * G->d, hi - lo, G->d), lo);
    });
}

int mse_shard_group_set_shard_device(mse_shard_group* G, size_t shard, const void* rows_dev, size_t n_rows, uint64_t first_row) {
    if (!G) return fail("null shard group");
    if (shard >= G->shards.size()) return fail("shard index out of range");
    if (first_row + n_rows > 0xFFFFFFFEull) return fail("row ids are u32: too many rows");
    std::lock_guard<std::mutex> call(G->call_mu);
    return G->run([=](size_t g) -> int {
        if (g != shard) return 0;
        return install(G->shards[g], mse_base_wrap_device(rows_dev, n_rows, G->d), (size_t)first_row);
    });
}

}  // extern "C"

// the search proper; the caller holds G->call_mu (one search at a time per group: gathered / q_root / out_* are group scratch)
// What a shard does with the payload (queries, ...) to fill its packed block [nq * k_in i64 scores | nq * k_in u32 GLOBAL ids] on its own
// device, asynchronously on its searcher's stream or complete on return.  q: the payload where this shard can read it (its own copy, or
// the root's memory through the peer mapping).
using LocalFn = std::function<int(Shard& sh, const void* q, char* blk)>;

// One sharded search: every shard's local step, ONE exchange of the packed blocks, the k best of all shards' k_in records per query.
static int search_dev_locked(mse_shard_group* G, const void* queries_dev, size_t qbytes, size_t nq, size_t k_in, size_t k, const LocalFn& local,
                             void* scores_dev, void* ids_dev) {
    for (const Shard& s : G->shards) if (!s.searcher) return fail("shard group: a shard holds no rows yet");
    const auto wall0 = std::chrono::steady_clock::now();
    const size_t n_shards = G->shards.size();
    const size_t B = block_bytes(nq, k_in);
    const bool use_rccl = G->exchange == MSE_EXCHANGE_RCCL;
    int prev = 0;
    (void)hipGetDevice(&prev);
    MSE_HIP_TRY(hipSetDevice(G->root_device));
    int rc = use_rccl ? 0 : G->gathered.ensure(B * n_shards);
    const char* merged_from = nullptr;
    if (!rc) {
        char* const gathered = G->gathered.as<char>();
        rc = G->run([=](size_t g) -> int {
            Shard& sh = G->shards[g];
            hipStream_t st = sh.searcher->stream;
            for (hipEvent_t& e : sh.ev) if (!e) MSE_HIP_TRY(hipEventCreate(&e));
            MSE_HIP_TRY(hipEventRecord(sh.ev[0], st));
            const void* q = queries_dev;
            const bool on_root = sh.device == G->root_device;
            if (use_rccl) {
                // every shard on its own device: queries come over by one peer copy (a device cannot be assumed to map the root's
                // memory here), the block stays local; the all-gather is issued in a second round, once EVERY shard has its
                // block (a rank that failed before its collective would leave the others waiting in theirs for ever)
                if (sh.block.ensure(B) || sh.gathered.ensure(B * n_shards)) return -1;
                if (!on_root && qbytes) {
                    if (sh.q_local.ensure(qbytes)) return -1;
                    MSE_HIP_TRY(hipMemcpyPeerAsync(sh.q_local.p, sh.device, queries_dev, G->root_device, qbytes, st));
                    q = sh.q_local.p;
                }
                char* blk = sh.block.as<char>();
                if (local(sh, q, blk)) return -1;
                MSE_HIP_TRY(hipEventRecord(sh.ev[1], st));
                return 0;
            } else {
                char* blk = gathered + g * B;
                if (!sh.peer) {   // no mapping of the root's memory: stage the queries here, copy the block back
                    if (sh.q_local.ensure(std::max<size_t>(qbytes, 16)) || sh.block.ensure(B)) return -1;
                    if (qbytes) MSE_HIP_TRY(hipMemcpyPeerAsync(sh.q_local.p, sh.device, queries_dev, G->root_device, qbytes, st));
                    q = sh.q_local.p;
                    blk = sh.block.as<char>();
                }
                if (local(sh, q, blk)) return -1;
                MSE_HIP_TRY(hipEventRecord(sh.ev[1], st));
                if (!sh.peer) MSE_HIP_TRY(hipMemcpyPeerAsync(gathered + g * B, G->root_device, blk, sh.device, B, st));
            }
            MSE_HIP_TRY(hipEventRecord(sh.ev[2], st));
            MSE_HIP_TRY(hipStreamSynchronize(st));
            (void)hipEventElapsedTime(&sh.local_ms, sh.ev[0], sh.ev[1]);
            (void)hipEventElapsedTime(&sh.exch_ms, sh.ev[1], sh.ev[2]);
            return 0;
        });
        if (!rc && use_rccl) {  // ONE ncclAllGather of the packed records per search: rank g = shard g, on its own thread and stream
            // A rank that fails here (the collective's launch, its event, its stream) leaves the OTHER ranks inside a collective that
            // can never complete: nobody waits blindly -- every rank polls its completion event and, once any rank has failed,
            // aborts its own communicator (ncclCommAbort frees the stream).  Afterwards the group drops back to the peer-store
            // exchange with no communicators, and this search is repeated over it below.
            std::atomic<int> failed{0};
            std::atomic<int>* const failed_p = &failed;
            rc = G->run([=](size_t g) -> int {
                Shard& sh = G->shards[g];
                hipStream_t st = sh.searcher->stream;
                const int nrc = rccl().AllGather(sh.block.p, sh.gathered.p, B, /*ncclInt8*/ 0, sh.comm, st);
                if (nrc) { failed_p->store(1); return rccl_fail("ncclAllGather", nrc); }
                if (hipEventRecord(sh.ev[2], st) != hipSuccess) { failed_p->store(1); return fail("shard group: event after the all-gather"); }
                for (;;) {
                    const hipError_t qe = hipEventQuery(sh.ev[2]);
                    if (qe == hipSuccess) break;
                    if (qe != hipErrorNotReady) { failed_p->store(1); return fail(std::string("shard group: all-gather stream: ") + hipGetErrorString(qe)); }
                    if (failed_p->load()) {
                        if (rccl().CommAbort && sh.comm) { (void)rccl().CommAbort(sh.comm); sh.comm = nullptr; }
                        return fail("shard group: another rank's all-gather failed; this rank's communicator was aborted");
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(20));
                }
                (void)hipEventElapsedTime(&sh.local_ms, sh.ev[0], sh.ev[1]);
                (void)hipEventElapsedTime(&sh.exch_ms, sh.ev[1], sh.ev[2]);
                return 0;
            });
            if (rc) {
                const std::string why = mse_last_error();
                for (Shard& sh : G->shards)
                    if (sh.comm) { if (rccl().CommAbort) (void)rccl().CommAbort(sh.comm); else (void)rccl().CommDestroy(sh.comm); sh.comm = nullptr; }
                G->exchange = MSE_EXCHANGE_PEER;
                G->rccl_ranks = 0;
                (void)hipSetDevice(prev);
                const int rc2 = search_dev_locked(G, queries_dev, qbytes, nq, k_in, k, local, scores_dev, ids_dev);   // over the peer-store exchange
                if (rc2) return rc2;
                set_error("RCCL exchange failed and was shut down (" + why + "); the search was answered over the peer-store exchange");
                return 0;
            }
        }
        merged_from = use_rccl ? G->shards[0].gathered.as<char>() : gathered;   // shard 0 lives on the root device
        hipStream_t rs = G->root->stream;
        if (!rc) for (hipEvent_t& e : G->ev_merge) if (!e && hipEventCreate(&e) != hipSuccess) rc = fail("shard group: event");
        if (!rc && hipEventRecord(G->ev_merge[0], rs) != hipSuccess) rc = fail("shard group: event");
        if (!rc) rc = merge_packed(G->root, merged_from, n_shards, nq, k, scores_dev, ids_dev, k_in);
        if (!rc && hipEventRecord(G->ev_merge[1], rs) != hipSuccess) rc = fail("shard group: event");
        if (!rc && hipStreamSynchronize(rs) != hipSuccess) rc = fail("shard group: merge failed");
        if (!rc) {
            float mm = 0.f;
            (void)hipEventElapsedTime(&mm, G->ev_merge[0], G->ev_merge[1]);
            double lmax = 0, emax = 0;
            for (const Shard& sh : G->shards) { lmax = std::max(lmax, (double)sh.local_ms); emax = std::max(emax, (double)sh.exch_ms); }
            G->last_ms[0] = lmax; G->last_ms[1] = emax; G->last_ms[2] = mm;
            G->last_ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
        }
    }
    (void)hipSetDevice(prev);
    return rc;
}

// RCCL for the in-process shape: one communicator per shard device, made by ONE ncclCommInitAll from the calling thread;
// afterwards each shard's own host thread issues its rank's ncclAllGather on its searcher's stream (the multi-thread,
// one-device-per-thread usage RCCL documents; no group call is needed because no thread drives two devices).
static int bring_up_rccl(mse_shard_group* G) {
    const size_t n = G->shards.size();
    for (size_t a = 0; a < n; a++)
        for (size_t b = a + 1; b < n; b++)
            if (G->shards[a].device == G->shards[b].device)
                return fail("RCCL exchange needs every shard on its own device (shards " + std::to_string(a) + " and " + std::to_string(b) +
                            " share device " + std::to_string(G->shards[a].device) + "): logical shards keep the peer-store exchange");
    if (rccl_ready()) return -1;
    if (!rccl().CommInitAll) return fail("librccl.so lacks ncclCommInitAll");
    std::vector<void*> comms(n, nullptr);
    std::vector<int> devs(n);
    for (size_t g = 0; g < n; g++) devs[g] = G->shards[g].device;
    int prev = 0;
    (void)hipGetDevice(&prev);
    const int rc = rccl().CommInitAll(comms.data(), (int)n, devs.data());
    (void)hipSetDevice(prev);
    if (rc) return rccl_fail("ncclCommInitAll", rc);
    int count = 0;
    if (rccl().CommCount(comms[0], &count) || count != (int)n) {
        for (void* c : comms) if (c) (void)rccl().CommDestroy(c);
        return fail("RCCL reports " + std::to_string(count) + " ranks for " + std::to_string(n) + " shards");
    }
    for (size_t g = 0; g < n; g++) G->shards[g].comm = comms[g];
    G->rccl_ranks = count;
    return 0;
}

// the brute-force local step: exact top-k of the shard's rows with global ids (the shard's first row added)
static LocalFn bruteforce_local(size_t nq, size_t k, int mode) {
    return [=](Shard& sh, const void* q, char* blk) -> int {
        return mse_bruteforce_topk_f16_dev(sh.searcher, q, nq, k, mode, sh.first_row, blk, blk + nq * k * 8);
    };
}

extern "C" {

// queries_dev: [nq][d] f16 on the ROOT device (device of shard 0), complete before the call; outputs [nq][k] on the root device.
// Returns when the merged result is complete.
int mse_shard_group_search_dev(mse_shard_group* G, const void* queries_dev, size_t nq, size_t k, int mode,
                               void* scores_dev, void* ids_dev) {
    if (!G) return fail("null shard group");
    if (nq == 0 || k == 0) return 0;
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    std::lock_guard<std::mutex> call(G->call_mu);
    return search_dev_locked(G, queries_dev, nq * G->d * 2, nq, k, k, bruteforce_local(nq, k, mode), scores_dev, ids_dev);
}

int mse_shard_group_search(mse_shard_group* G, const uint16_t* queries, size_t nq, size_t k, int mode, int64_t* scores,
                           uint32_t* ids) {
    if (!G) return fail("null shard group");
    if (nq == 0 || k == 0) return 0;
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    int prev = 0;
    (void)hipGetDevice(&prev);
    MSE_HIP_TRY(hipSetDevice(G->root_device));
    int rc = 0;
    {
        // ONE lock across upload, search and download: two host threads on the same group must not see each other's queries
        // in q_root or each other's answers in out_s / out_i
        std::lock_guard<std::mutex> call(G->call_mu);
        rc = G->q_root.ensure(nq * G->d * 2) || G->out_s.ensure(nq * k * 8) || G->out_i.ensure(nq * k * 4);
        if (!rc && hipMemcpy(G->q_root.p, queries, nq * G->d * 2, hipMemcpyHostToDevice) != hipSuccess) rc = fail("query upload failed");
        if (!rc) rc = search_dev_locked(G, G->q_root.p, nq * G->d * 2, nq, k, k, bruteforce_local(nq, k, mode), G->out_s.p, G->out_i.p);
        if (!rc && (hipMemcpy(scores, G->out_s.p, nq * k * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(ids, G->out_i.p, nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = fail("result download failed");
    }
    (void)hipSetDevice(prev);
    return rc;
}

// ---- the approximate-search paths over the same shards (round 5; SURVEY 8(e): "rows (and their PQ codes / descriptors)") ---------
// A shard's codes, descriptors and graph cover exactly its rows and speak LOCAL ids; what leaves a shard is the same packed block of
// (score, global id) records as in the brute-force search, and the same ONE exchange + merge brings the blocks together.

int mse_shard_group_attach_pq(mse_shard_group* G, size_t shard, mse_pq* pq, const mse_codes* codes) {
    if (!G) return fail("null shard group");
    if (shard >= G->shards.size()) return fail("shard index out of range");
    std::lock_guard<std::mutex> call(G->call_mu);
    Shard& sh = G->shards[shard];
    if (pq && codes) {
        if (!sh.base) return fail("shard group: the shard holds no rows yet");
        if (codes->n != sh.base->n) return fail("shard group: codes and rows of the shard differ in length");
        if (pq->d != G->d || codes->code_size != pq->n_chunks) return fail("shard group: codec does not fit the vectors / codes");
        if (pq->device != sh.device) return fail("shard group: the codec lives on another device than the shard");
    } else if (pq || codes) {
        return fail("shard group: codec and codes come together");
    }
    sh.pq = pq; sh.codes = codes;
    return 0;
}

int mse_shard_group_attach_graph(mse_shard_group* G, size_t shard, const mse_graph* graph) {
    if (!G) return fail("null shard group");
    if (shard >= G->shards.size()) return fail("shard index out of range");
    std::lock_guard<std::mutex> call(G->call_mu);
    Shard& sh = G->shards[shard];
    if (graph && (!sh.base || graph->n != sh.base->n)) return fail("shard group: graph and rows of the shard differ in length");
    sh.graph = graph;
    return 0;
}

// phase B of the sharded PQ scan on one shard: exact score (+ descriptor bias) of ITS members of the index's top-r, everybody else's
// slots empty; q16: f16 queries [nq][d], gids: the merged top-r [nq][r] global ids, scales_dev: [n_desc] or null -- all readable from this
// device; the block is written on `st`
static int pq_rescore_members(const mse_base* b, const mse_codes* codes, uint64_t first_row, const void* q16, const uint32_t* gids,
                              const float* scales_dev, size_t nq, size_t r, DevBuf& a, DevBuf& bb, char* blk, hipStream_t st) {
    const size_t n = nq * r;
    if (a.ensure(n * 4) || bb.ensure(n * 8)) return -1;
    uint32_t* local = a.as<uint32_t>();
    int64_t* ex = bb.as<int64_t>();
    hipLaunchKernelGGL(members_local_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gids, n, first_row, (uint64_t)b->n, local);
    MSE_HIP_TRY(hipGetLastError());
    if (launch_score_rows(b->dev, b->n, (int)b->d, q16, false, local, n, r, ex, nullptr, st)) return -1;
    if (scales_dev && codes->n_desc && launch_add_descriptor(local, n, codes->desc, (int)codes->n_desc, codes->n, scales_dev, ex, st)) return -1;
    return launch_block_finish(ex, local, n, first_row, reinterpret_cast<int64_t*>(blk), reinterpret_cast<uint32_t*>(blk + n * 8), st);
}

// The OPQ/PQ flat scan + exact re-rank (mse_pq_scan_topk_batch: top-r by ADC, exact fp16 re-score, top-k) over sharded codes, with the
// answer of the unsharded call bit for bit.  A per-shard re-rank would not give that (a shard's r-th best by ADC is not the index's), so
// the two selections are done on the whole index, each through one exchange:
//   A  every shard: ADC top-r of ITS codes (global ids)            -> exchange -> the index's top-r by (ADC score desc, id asc)
//   B  every shard: exact score (+ descriptor bias) of ITS members of that top-r, (INT64_MIN, none) for the others'
//                                                                   -> exchange -> top-k by (exact score desc, id asc)
// Records exchanged: 2 x shards x nq x r x 12 bytes (r = 200, 32 queries, 8 shards: 1.2 MB in all).
int mse_shard_group_pq_scan_topk(mse_shard_group* G, const float* queries_f32, const float* scales, size_t nq, size_t r, size_t k,
                                 int64_t* scores, uint32_t* ids) {
    if (!G || !queries_f32 || !scores || !ids) return fail("shard group pq scan: null argument");
    if (nq == 0 || k == 0) return 0;
    if (r < k) r = k;
    if (r > (size_t)TOPK_KMAX - 64) return fail("r too large (max 1984)");
    for (const Shard& sh : G->shards) if (!sh.pq || !sh.codes) return fail("shard group: a shard has no codec / codes attached (mse_shard_group_attach_pq)");
    const size_t d = G->d, n_desc = G->shards[0].codes->n_desc;
    const bool bias = scales && n_desc;
    int prev = 0;
    (void)hipGetDevice(&prev);
    MSE_HIP_TRY(hipSetDevice(G->root_device));
    int rc = 0;
    {
        std::lock_guard<std::mutex> call(G->call_mu);
        // root staging: [f16 queries nq x d][merged top-r ids nq x r][scales n_desc f32] -- phase B's payload, one buffer so that a shard
        // that cannot map the root's memory gets it in one peer copy
        const size_t off_ids = (nq * d * 2 + 255) & ~(size_t)255, off_sc = (off_ids + nq * r * 4 + 255) & ~(size_t)255;
        const size_t pay_bytes = off_sc + (bias ? n_desc * 4 : 0) + 16, off_q32 = (pay_bytes + 255) & ~(size_t)255;
        rc = G->q_root.ensure(off_q32 + nq * d * 4) || G->out_s.ensure(nq * r * 8) || G->out_i.ensure(nq * r * 4);
        hipStream_t rs = G->root->stream;
        char* const pay = G->q_root.as<char>();
        float* const q32_dev = reinterpret_cast<float*>(pay + off_q32);   // behind the payload: the f32 queries, only to make their f16 copies
        if (!rc && hipMemcpyAsync(q32_dev, queries_f32, nq * d * 4, hipMemcpyHostToDevice, rs) != hipSuccess) rc = fail("query upload failed");
        if (!rc) rc = launch_f32_to_f16(q32_dev, nq * d, reinterpret_cast<uint16_t*>(pay), rs);   // half::f16::from_f32 (RNE), query_disk_index.rs:477
        if (!rc && bias && hipMemcpyAsync(pay + off_sc, scales, n_desc * 4, hipMemcpyHostToDevice, rs) != hipSuccess) rc = fail("scales upload failed");
        if (!rc && hipStreamSynchronize(rs) != hipSuccess) rc = fail("query upload failed");
        // phase A: the index's top-r by ADC (+ bias)
        if (!rc)
            rc = search_dev_locked(G, nullptr, 0, nq, r, r, [=](Shard& sh, const void*, char* blk) -> int {
                return mse_pq_scan_topk_block(sh.pq, sh.codes, nullptr, queries_f32, nq, bias ? scales : nullptr, r, r, sh.first_row, blk);
            }, G->out_s.p, G->out_i.p);
        double tA[4] = {G->last_ms[0], G->last_ms[1], G->last_ms[2], G->last_ms[3]};
        if (!rc && hipMemcpyAsync(pay + off_ids, G->out_i.p, nq * r * 4, hipMemcpyDeviceToDevice, rs) != hipSuccess) rc = fail("shard group: copy failed");
        if (!rc && hipStreamSynchronize(rs) != hipSuccess) rc = fail("shard group: copy failed");
        // phase B: exact re-score of the members, top-k
        if (!rc)
            rc = search_dev_locked(G, pay, pay_bytes, nq, r, k, [=](Shard& sh, const void* q, char* blk) -> int {
                const char* p = static_cast<const char*>(q);
                return pq_rescore_members(sh.base, sh.codes, sh.first_row, p, reinterpret_cast<const uint32_t*>(p + off_ids),
                                          bias ? reinterpret_cast<const float*>(p + off_sc) : nullptr, nq, r, sh.ann_a, sh.ann_b, blk, sh.searcher->stream);
            }, G->out_s.p, G->out_i.p);
        if (!rc) for (int i = 0; i < 4; i++) G->last_ms[i] += tA[i];   // both phases
        if (!rc && (hipMemcpy(scores, G->out_s.p, nq * k * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(ids, G->out_i.p, nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = fail("result download failed");
    }
    (void)hipSetDevice(prev);
    return rc;
}

// The graph index sharded: ONE Vamana graph per shard over its own rows (the reference's shards are exactly that:
// src/generate_index_shard.rs builds a shard's graph over the shard's vectors).  Every shard answers the whole query batch from ITS
// graph -- entry by its own entry table, query_disk_index::greedy_search, the k best visited records (mse_disk_query_topk_block) --
// and the blocks meet in the one exchange: the merged result is the merge of the per-shard searches.  queries: f16 [nq][d] host rows;
// neighbours scored exactly (disable_pq = 1) or by ADC through the shard's codec and codes with tables from `luts` ([nq][64*256] f32).
int mse_shard_group_query_topk(mse_shard_group* G, const uint16_t* queries, const float* luts, const float* scales, size_t nq, int disable_pq,
                               size_t beamwidth, size_t search_list, size_t k, int64_t* scores, uint32_t* ids) {
    if (!G || !queries || !scores || !ids) return fail("shard group query: null argument");
    if (nq == 0 || k == 0) return 0;
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    for (const Shard& sh : G->shards) {
        if (!sh.graph) return fail("shard group: a shard has no graph attached (mse_shard_group_attach_graph)");
        if (!disable_pq && (!sh.pq || !sh.codes)) return fail("shard group: ADC-scored search needs a codec and codes on every shard");
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    MSE_HIP_TRY(hipSetDevice(G->root_device));
    int rc = 0;
    {
        std::lock_guard<std::mutex> call(G->call_mu);
        rc = G->q_root.ensure(nq * G->d * 2) || G->out_s.ensure(nq * k * 8) || G->out_i.ensure(nq * k * 4);
        if (!rc && hipMemcpy(G->q_root.p, queries, nq * G->d * 2, hipMemcpyHostToDevice) != hipSuccess) rc = fail("query upload failed");
        if (!rc)
            rc = search_dev_locked(G, G->q_root.p, nq * G->d * 2, nq, k, k, [=](Shard& sh, const void* q, char* blk) -> int {
                return mse_disk_query_topk_block(sh.searcher, sh.pq, sh.codes, sh.graph, nullptr, static_cast<const uint16_t*>(q), luts, scales, nq, disable_pq,
                                                 beamwidth, search_list, k, sh.first_row, blk, nullptr, nullptr, nullptr);
            }, G->out_s.p, G->out_i.p);
        if (!rc && (hipMemcpy(scores, G->out_s.p, nq * k * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(ids, G->out_i.p, nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = fail("result download failed");
    }
    (void)hipSetDevice(prev);
    return rc;
}

int mse_shard_group_set_exchange(mse_shard_group* G, int kind) {
    if (!G) return fail("null shard group");
    if (kind != MSE_EXCHANGE_PEER && kind != MSE_EXCHANGE_RCCL) return fail("unknown exchange kind");
    std::lock_guard<std::mutex> call(G->call_mu);
    if (kind == MSE_EXCHANGE_RCCL && !G->shards[0].comm && bring_up_rccl(G)) return -1;   // the previous exchange stays
    G->exchange = kind;
    return 0;
}
int mse_shard_group_exchange(const mse_shard_group* G) { return G ? G->exchange : -1; }
int mse_shard_group_rccl_ranks(const mse_shard_group* G) { return G ? G->rccl_ranks : 0; }
int mse_shard_group_last_timing(mse_shard_group* G, double out_ms[4]) {
    if (!G || !out_ms) return fail("null shard group / output");
    std::lock_guard<std::mutex> call(G->call_mu);
    for (int i = 0; i < 4; i++) out_ms[i] = G->last_ms[i];
    return 0;
}

}  // extern "C"

struct mse_comm {
    void* comm = nullptr;
    int rank = 0, world = 1;
    DevBuf local, gathered, ann_a, ann_b;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // start, local search done, all-gather done, merge done
    hipStream_t ev_stream = nullptr;
    bool timed = false;
};

extern "C" {

int mse_comm_unique_id(mse_comm_id* out) {
    if (!out) return fail("null id");
    if (rccl_ready()) return -1;
    const int rc = rccl().GetUniqueId(out);
    return rc ? rccl_fail("ncclGetUniqueId", rc) : 0;
}

mse_comm* mse_comm_init(const mse_comm_id* id, int rank, int world) {
    if (!id || world <= 0 || rank < 0 || rank >= world) { fail("mse_comm_init: bad rank / world"); return nullptr; }
    if (rccl_ready()) return nullptr;
    mse_comm* c = new (std::nothrow) mse_comm();
    if (!c) { fail("out of host memory"); return nullptr; }
    c->rank = rank; c->world = world;
    const int rc = rccl().CommInitRank(&c->comm, world, *id, rank);
    if (rc) { rccl_fail("ncclCommInitRank", rc); delete c; return nullptr; }
    return c;
}

void mse_comm_free(mse_comm* c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    for (hipEvent_t& e : c->ev) if (e) (void)hipEventDestroy(e);
    delete c;
}

int mse_comm_rank(const mse_comm* c) { return c ? c->rank : -1; }
int mse_comm_size(const mse_comm* c) {
    if (!c) return 0;
    int n = 0;
    if (rccl().CommCount(c->comm, &n)) return 0;   // what RCCL itself reports, not what the caller claimed
    return n;
}

// local search over this rank's shard + ONE all-gather of the packed records + merge; outputs [nq][k] on this rank's device,
// identical on every rank.  Asynchronous on the searcher's stream after the local search (which synchronises internally).
int mse_comm_search_dev(mse_comm* c, mse_searcher* s, const void* queries_dev, size_t nq, size_t k, int mode,
                        uint64_t id_offset, void* scores_dev, void* ids_dev) {
    if (!c || !s) return fail("null communicator / searcher");
    if (nq == 0 || k == 0) return 0;
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    const size_t B = block_bytes(nq, k);
    if (c->local.ensure(B) || c->gathered.ensure(B * (size_t)c->world)) return -1;
    char* blk = c->local.as<char>();
    for (hipEvent_t& e : c->ev) if (!e) MSE_HIP_TRY(hipEventCreate(&e));
    c->timed = false;
    MSE_HIP_TRY(hipEventRecord(c->ev[0], s->stream));
    if (mse_bruteforce_topk_f16_dev(s, queries_dev, nq, k, mode, id_offset, blk, blk + nq * k * 8)) return -1;
    MSE_HIP_TRY(hipEventRecord(c->ev[1], s->stream));
    const int rc = rccl().AllGather(blk, c->gathered.p, B, /*ncclInt8*/ 0, c->comm, s->stream);
    if (rc) return rccl_fail("ncclAllGather", rc);
    MSE_HIP_TRY(hipEventRecord(c->ev[2], s->stream));
    if (merge_packed(s, c->gathered.as<char>(), (size_t)c->world, nq, k, scores_dev, ids_dev)) return -1;
    MSE_HIP_TRY(hipEventRecord(c->ev[3], s->stream));
    c->timed = true;
    return 0;
}

// The exchange alone: this rank's packed block ([nq*k_in] i64 scores | [nq*k_in] u32 GLOBAL ids, on its device, complete on the
// searcher's stream) -> ONE all-gather -> the k best of all ranks' records per query, identical on every rank.
int mse_comm_exchange_dev(mse_comm* c, mse_searcher* s, const void* block_dev, size_t nq, size_t k_in, size_t k, void* scores_dev, void* ids_dev) {
    if (!c || !s || !block_dev) return fail("null communicator / searcher / block");
    if (nq == 0 || k == 0) return 0;
    if (k_in == 0) k_in = k;
    if (k > (size_t)TOPK_KMAX - 64 || k_in > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    const size_t B = block_bytes(nq, k_in);
    if (c->gathered.ensure(B * (size_t)c->world)) return -1;
    const int rc = rccl().AllGather(block_dev, c->gathered.p, B, /*ncclInt8*/ 0, c->comm, s->stream);
    if (rc) return rccl_fail("ncclAllGather", rc);
    return merge_packed(s, c->gathered.as<char>(), (size_t)c->world, nq, k, scores_dev, ids_dev, k_in);
}

// One rank of the sharded PQ scan + exact re-rank (the protocol of mse_shard_group_pq_scan_topk): this rank's codes / rows start at
// global row first_row; queries_f32 [nq][d] and scales [n_desc] are host memory, the same on every rank.  Outputs [nq][k] on this
// rank's device, identical on every rank, complete on return.
int mse_comm_pq_scan_topk(mse_comm* c, mse_pq* pq, const mse_codes* codes, mse_searcher* s, const float* queries_f32, const float* scales,
                          size_t nq, size_t r, size_t k, uint64_t first_row, void* scores_dev, void* ids_dev) {
    if (!c || !pq || !codes || !s || !s->base || !queries_f32) return fail("comm pq scan: null argument");
    if (nq == 0 || k == 0) return 0;
    if (r < k) r = k;
    if (r > (size_t)TOPK_KMAX - 64) return fail("r too large (max 1984)");
    const size_t d = s->base->d, n_desc = codes->n_desc, n = nq * r;
    const bool bias = scales && n_desc;
    hipStream_t st = s->stream;
    const size_t B = block_bytes(nq, r);
    // scratch: block | merged scores | merged ids | f32 queries | f16 queries | scales
    const size_t o_ms = B, o_mi = o_ms + n * 8, o_q32 = (o_mi + n * 4 + 255) & ~(size_t)255, o_q16 = o_q32 + nq * d * 4, o_sc = (o_q16 + nq * d * 2 + 255) & ~(size_t)255;
    if (c->local.ensure(o_sc + n_desc * 4 + 64)) return -1;
    char* w = c->local.as<char>();
    if (mse_pq_scan_topk_block(pq, codes, nullptr, queries_f32, nq, bias ? scales : nullptr, r, r, first_row, w)) return -1;   // phase A, complete on return
    if (mse_comm_exchange_dev(c, s, w, nq, r, r, w + o_ms, w + o_mi)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(w + o_q32, queries_f32, nq * d * 4, hipMemcpyHostToDevice, st));
    if (launch_f32_to_f16(reinterpret_cast<const float*>(w + o_q32), nq * d, reinterpret_cast<uint16_t*>(w + o_q16), st)) return -1;
    if (bias) MSE_HIP_TRY(hipMemcpyAsync(w + o_sc, scales, n_desc * 4, hipMemcpyHostToDevice, st));
    if (pq_rescore_members(s->base, codes, first_row, w + o_q16, reinterpret_cast<const uint32_t*>(w + o_mi), bias ? reinterpret_cast<const float*>(w + o_sc) : nullptr,
                           nq, r, c->ann_a, c->ann_b, w, st)) return -1;
    if (mse_comm_exchange_dev(c, s, w, nq, r, k, scores_dev, ids_dev)) return -1;
    MSE_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// One rank of the sharded graph index (mse_shard_group_query_topk's protocol): this rank's graph over its own rows answers the batch,
// ONE all-gather, merge.  queries: f16 [nq][d], host or device; outputs [nq][k] on this rank's device, complete on return.
int mse_comm_query_topk(mse_comm* c, mse_searcher* s, mse_pq* pq, const mse_codes* codes, const mse_graph* g, const uint16_t* queries,
                        const float* luts, const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k,
                        uint64_t first_row, void* scores_dev, void* ids_dev) {
    if (!c || !s || !g || !queries) return fail("comm query: null argument");
    if (nq == 0 || k == 0) return 0;
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    if (c->local.ensure(block_bytes(nq, k))) return -1;
    if (mse_disk_query_topk_block(s, pq, codes, g, nullptr, queries, luts, scales, nq, disable_pq, beamwidth, search_list, k, first_row, c->local.p,
                                  nullptr, nullptr, nullptr)) return -1;
    if (mse_comm_exchange_dev(c, s, c->local.p, nq, k, k, scores_dev, ids_dev)) return -1;
    MSE_HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

// breakdown of the last mse_comm_search_dev on this rank, ms: [0] local search, [1] all-gather (includes waiting for the slowest
// rank), [2] merge, [3] the three together.  Waits for that search to finish.
int mse_comm_last_timing(mse_comm* c, double out_ms[4]) {
    if (!c || !out_ms) return fail("null communicator / output");
    if (!c->timed) return fail("no search has been issued on this communicator");
    MSE_HIP_TRY(hipEventSynchronize(c->ev[3]));
    float a = 0, b = 0, m = 0;
    MSE_HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    MSE_HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    MSE_HIP_TRY(hipEventElapsedTime(&m, c->ev[2], c->ev[3]));
    out_ms[0] = a; out_ms[1] = b; out_ms[2] = m; out_ms[3] = (double)a + b + m;
    return 0;
}

}  // extern "C"
