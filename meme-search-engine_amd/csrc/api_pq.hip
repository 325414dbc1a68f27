// C ABI, part 2: evaluator ranks, flat fp16 index (FAISS SQfp16-IP semantics), product quantiser.
#include "../../include/mse.h"
#include "runtime.h"
#include "dispatch.h"
#include <algorithm>
#include <atomic>
#include <memory>
#include <cfloat>
#include <cstring>
#include <new>
#include <vector>

using namespace mse;

// Flat index.  Concurrency is the reference's: searches share, `add` excludes (the RwLock of src/main.rs:1016 write / :1046
// read).  Searches of all threads meet in the index's coalescer (dispatch.h) and are executed by its ONE worker thread, which
// is the only user of `scratch` while readers hold the lock; `add` runs on the caller's thread with every reader out.
struct mse_index {
    int d = 0;
    int device = 0;
    size_t n = 0, cap = 0;
    uint16_t* codes = nullptr;   // [cap][d] fp16, device
    mse_base view;               // non-owning view of codes[0..n) for the matrix-core scan; its norm bound grows with every add
    mse_searcher* scratch = nullptr;
    mse::SharedExclusive rw;
    std::unique_ptr<mse::Coalescer> co;
    void* pin = nullptr;         // pinned staging of the worker: queries up, [distances | labels] down
    size_t pin_cap = 0;
    mse::DevBuf q32, out;
    std::atomic<uint64_t> retried_alone{0};
};

extern "C" {

// ---- evaluator ranks (src/query_disk_index.rs:271-273,309-316) ---------------------------------
int mse_bruteforce_ranks_f16(mse_searcher* s, const uint16_t* query, const uint32_t* ids, size_t n_ids,
                             uint32_t* ranks) {
    if (!s || !s->base) return fail("null searcher");
    const mse_base* b = s->base;
    if (n_ids == 0) return 0;
    const size_t d = b->d;
    hipStream_t st = s->stream;
    if (s->q_stage.ensure(8 * d * 2) || s->scores.ensure(std::max<size_t>(b->n, 1) * 8)) return -1;
    MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, 8 * d * 2, st));
    MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, query, d * 2, hipMemcpyHostToDevice, st));
    if (launch_scan_exact(b->dev, b->n, (int)d, s->q_stage.p, 1, false, s->scores.as<int64_t>(), b->n, nullptr, s->n_cu,
                          st)) return -1;
    const size_t chunk = (size_t)rank_max_targets();
    if (s->cand_ids.ensure(chunk * 4) || s->cand_scores.ensure(chunk * 8)) return -1;
    std::vector<unsigned long long> counts(chunk);
    for (size_t o = 0; o < n_ids; o += chunk) {
        const size_t m = std::min(chunk, n_ids - o);
        for (size_t i = 0; i < m; i++)
            if (ids[o + i] >= b->n) return fail("rank: id out of range");
        MSE_HIP_TRY(hipMemcpyAsync(s->cand_ids.p, ids + o, m * 4, hipMemcpyHostToDevice, st));
        MSE_HIP_TRY(hipMemsetAsync(s->cand_scores.p, 0, m * 8, st));
        if (launch_rank(s->scores.as<int64_t>(), b->n, s->cand_ids.as<uint32_t>(), (int)m,
                        s->cand_scores.as<unsigned long long>(), s->n_cu, st)) return -1;
        MSE_HIP_TRY(hipMemcpyAsync(counts.data(), s->cand_scores.p, m * 8, hipMemcpyDeviceToHost, st));
        MSE_HIP_TRY(hipStreamSynchronize(st));
        for (size_t i = 0; i < m; i++) ranks[o + i] = (uint32_t)counts[i];
    }
    return 0;
}

// ---- flat index -----------------------------------------------------------------------------------
namespace {

// <= 8 f32 queries against every row in the stated FAISS order (scan_exact.hip, QF32), tournament, ids + f32 keys into
// out_ids / out_keys ([nq][out_stride], ID_NONE-padded)
int index_pass_exact(mse_index* idx, const float* q32_dev, int nqp, int k, uint32_t* out_ids, float* out_keys, size_t out_stride) {
    mse_searcher* s = idx->scratch;
    hipStream_t st = s->stream;
    const size_t d = idx->d, n = idx->n;
    if (s->q_stage.ensure(8 * d * 4) || s->scores.ensure((size_t)nqp * n * 4)) return -1;
    MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, 8 * d * 4, st));
    MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, q32_dev, (size_t)nqp * d * 4, hipMemcpyDeviceToDevice, st));
    if (launch_scan_exact(idx->codes, n, (int)d, s->q_stage.p, nqp, true, nullptr, n, s->scores.as<float>(), s->n_cu, st)) return -1;
    if (s->sel_keys.ensure((size_t)nqp * k * 4)) return -1;
    uint32_t* sel = nullptr;
    LevelRef l0{KEY_F32, s->scores.p, n, 1, n, false, 0};
    if (descend(s, l0, nqp, k, &sel, s->sel_keys.p)) return -1;
    MSE_HIP_TRY(hipMemcpy2DAsync(out_ids, out_stride * 4, sel, (size_t)k * 4, (size_t)k * 4, nqp, hipMemcpyDeviceToDevice, st));
    MSE_HIP_TRY(hipMemcpy2DAsync(out_keys, out_stride * 4, s->sel_keys.p, (size_t)k * 4, (size_t)k * 4, nqp, hipMemcpyDeviceToDevice, st));
    return 0;
}

// Up to 256 f32 queries in ONE pass over the rows.  The matrix cores score f16 roundings of the queries and only nominate:
// group maxima -> tournament -> the rows of the best groups re-scored with the f32 query in the stated order -> top k ->
// certificate.  The bound on what the nomination can have missed is the f16 scan's (api.hip mfma_pass) plus the query
// rounding, |x . (q - f16(q))| <= |x| |q - f16(q)| (measured per query, not assumed).  A query whose certificate fails widens
// its candidate set and finally repeats through the exact pass: answers equal index_pass_exact's.
int index_pass_mfma(mse_index* idx, const float* q32_dev, int nqp, int k, uint32_t* out_ids, float* out_keys, size_t out_stride) {
    mse_searcher* s = idx->scratch;
    hipStream_t st = s->stream;
    const int d = idx->d;
    const size_t n = idx->n;
    const int nq_pad = mfma_pad(nqp, d);
    if (s->q_stage.ensure((size_t)nq_pad * d * 2)) return -1;
    MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, (size_t)nq_pad * d * 2, st));
    if (launch_f32_to_f16(q32_dev, (size_t)nqp * d, s->q_stage.as<uint16_t>(), st)) return -1;
    const size_t n_groups = (n + GROUP_ROWS - 1) / GROUP_ROWS;
    if (s->gmax.ensure(n_groups * (size_t)nq_pad * 4) || s->qpacked.ensure(mfma_packed_bytes(d))) return -1;
    if (launch_scan_mfma(idx->codes, n, d, s->q_stage.as<uint16_t>(), nq_pad, s->qpacked.p, s->gmax.as<float>(), s->n_cu, st,
                         s->timing ? s->ev0 : nullptr, s->timing ? s->ev1 : nullptr)) return -1;
    bool timing_pending = s->timing;
    if (s->eps.ensure((size_t)nqp * 8) || s->margin.ensure((size_t)nqp * 8)) return -1;   // second halves: the widening's compact set
    if (launch_query_eps_f32(q32_dev, s->q_stage.as<uint16_t>(), nqp, d, idx->view.norm_bits_dev, 2.8e-4f, s->eps.as<float>(), st)) return -1;
    std::vector<float> margin_h(nqp);
    const int kg0 = (int)std::min<size_t>(std::max(k + 8, 16), TOPK_KMAX);
    // one round: tournament -> rows of the kg best groups re-scored with the f32 queries -> top k -> certificate margins (host)
    auto round = [&](const float* gm, int gm_pad, const float* q32, int nq, int kg_eff, const float* eps_dev, float* margin_dev, uint32_t* dst_i,
                     float* dst_k, size_t dst_stride) -> int {
        if (s->gkeys.ensure((size_t)nq * kg_eff * 4)) return -1;
        uint32_t* gsel = nullptr;
        LevelRef l0{KEY_F32, gm, 1, (size_t)gm_pad, n_groups, true, gm_pad};
        if (descend(s, l0, nq, kg_eff, &gsel, s->gkeys.p)) return -1;
        const size_t n_cand = (size_t)kg_eff * GROUP_ROWS;
        if (s->cand_ids.ensure((size_t)nq * n_cand * 4) || s->cand_scores.ensure((size_t)nq * n_cand * 4)) return -1;
        if (launch_expand_groups(gsel, kg_eff, kg_eff, GROUP_ROWS, n, s->cand_ids.as<uint32_t>(), n_cand, nq, st)) return -1;
        if (launch_score_rows(idx->codes, n, d, q32, true, s->cand_ids.as<uint32_t>(), (size_t)nq * n_cand, n_cand, nullptr,
                              s->cand_scores.as<float>(), st)) return -1;
        SelectArgs a{};
        a.kind = KEY_F32; a.list_ids = s->cand_ids.as<uint32_t>(); a.list_keys = s->cand_scores.p;
        a.list_stride = n_cand; a.n_list = n_cand; a.k = k; a.out_ids = dst_i; a.out_keys = dst_k; a.out_stride = dst_stride; a.nq = nq;
        if (launch_select(a, st)) return -1;
        if (launch_margin_f32(dst_i, dst_k, dst_stride, k, nq, s->gkeys.as<float>(), kg_eff, kg_eff, n_groups, eps_dev, margin_dev, st)) return -1;
        MSE_HIP_TRY(hipMemcpyAsync(margin_h.data(), margin_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        MSE_HIP_TRY(hipStreamSynchronize(st));
        s->last_max_groups = std::max<uint32_t>(s->last_max_groups, (uint32_t)kg_eff);
        return 0;
    };
    if (round(s->gmax.as<float>(), nq_pad, q32_dev, nqp, kg0, s->eps.as<float>(), s->margin.as<float>(), out_ids, out_keys, out_stride)) return -1;
    if (timing_pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) { s->scan_ms_total += ms; s->scan_launches++; }
        timing_pending = false;
    }
    std::vector<uint32_t> bad;
    for (int i = 0; i < nqp; i++)
        if (!(margin_h[i] > 0.0f)) bad.push_back((uint32_t)i);
    if (bad.empty() || (size_t)kg0 >= n_groups) return 0;
    s->last_widened = std::max<uint32_t>(s->last_widened, (uint32_t)bad.size());
    // only the queries whose certificate failed go on, as a compact set (api.hip mfma_pass has the same structure)
    const int nb = (int)bad.size(), nbp = (nb + 31) / 32 * 32;
    if (s->widx.ensure((size_t)nb * 5) || s->wq.ensure((size_t)nb * d * 6) || s->wg.ensure(n_groups * (size_t)nbp * 4) ||
        s->wout.ensure((size_t)nb * k * 8)) return -1;
    uint32_t* idx_dev = s->widx.as<uint32_t>();
    uint8_t* take_dev = reinterpret_cast<uint8_t*>(idx_dev + nb);
    float* wq32 = s->wq.as<float>();
    uint16_t* wq16 = reinterpret_cast<uint16_t*>(s->wq.as<char>() + (size_t)nb * d * 4);
    MSE_HIP_TRY(hipMemcpyAsync(idx_dev, bad.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
    if (launch_gather_rows16(q32_dev, (size_t)d * 4, idx_dev, nb, wq32, st)) return -1;
    if (launch_gather_rows16(s->q_stage.p, (size_t)d * 2, idx_dev, nb, wq16, st)) return -1;
    if (launch_gather_columns(s->gmax.as<float>(), nq_pad, n_groups, idx_dev, nb, s->wg.as<float>(), nbp, st)) return -1;
    float* eps2 = s->eps.as<float>() + nqp;
    float* margin2 = s->margin.as<float>() + nqp;
    if (launch_query_eps_f32(wq32, wq16, nb, d, idx->view.norm_bits_dev, 2.8e-4f, eps2, st)) return -1;
    uint32_t* w_i = s->wout.as<uint32_t>();
    float* w_k = reinterpret_cast<float*>(s->wout.as<char>() + (size_t)nb * k * 4);
    std::vector<uint8_t> open_q(nb, 1);
    int kg = kg0 * 4;
    for (;;) {
        const int kg_eff = (int)std::min<size_t>(kg, TOPK_KMAX);
        if (round(s->wg.as<float>(), nbp, wq32, nb, kg_eff, eps2, margin2, w_i, w_k, (size_t)k)) return -1;
        std::vector<uint8_t> take(nb, 0);
        int still = 0;
        for (int j = 0; j < nb; j++) {
            if (!open_q[j]) continue;
            if (margin_h[j] > 0.0f || (size_t)kg_eff >= n_groups) { take[j] = 1; open_q[j] = 0; } else still++;
        }
        MSE_HIP_TRY(hipMemcpyAsync(take_dev, take.data(), (size_t)nb, hipMemcpyHostToDevice, st));
        if (launch_scatter_rows4(idx_dev, take_dev, nb, k, w_i, out_ids, out_stride, st) ||
            launch_scatter_rows4(idx_dev, take_dev, nb, k, w_k, out_keys, out_stride, st)) return -1;
        MSE_HIP_TRY(hipStreamSynchronize(st));
        if (still == 0) return 0;
        if (kg_eff >= TOPK_KMAX) break;
        kg = kg_eff * 4;
    }
    for (int j = 0; j < nb; j++) {   // cannot widen further: the exact pass, straight into the query's own output rows
        if (!open_q[j]) continue;
        if (index_pass_exact(idx, wq32 + (size_t)j * d, 1, k, out_ids + (size_t)bad[j] * out_stride, out_keys + (size_t)bad[j] * out_stride,
                             out_stride)) return -1;
    }
    return 0;
}

// one engine call for a group of requests (worker thread; every caller of the group holds the shared lock)
int index_run_group(mse_index* idx, DispatchReq* const* reqs, size_t n_req) {
    mse_searcher* s = idx->scratch;
    hipStream_t st = s->stream;
    const size_t d = idx->d, n = idx->n;
    size_t total = 0, kmax = 0;
    for (size_t i = 0; i < n_req; i++) { total += reqs[i]->nq; kmax = std::max(kmax, reqs[i]->k); }
    if (total == 0 || kmax == 0 || n == 0) return 0;   // outputs were pre-filled with "nothing found"
    const size_t in_bytes = total * d * 4, out_bytes = total * kmax * 8;
    if (idx->pin_cap < std::max(in_bytes, out_bytes)) {
        if (idx->pin) (void)hipHostFree(idx->pin);
        idx->pin = nullptr; idx->pin_cap = 0;
        const size_t want = std::max<size_t>(2 * std::max(in_bytes, out_bytes), (size_t)1 << 20);
        MSE_HIP_TRY(hipHostMalloc(&idx->pin, want, hipHostMallocDefault));
        idx->pin_cap = want;
    }
    if (idx->q32.ensure(in_bytes) || idx->out.ensure(out_bytes)) return -1;
    char* p = static_cast<char*>(idx->pin);
    for (size_t i = 0, o = 0; i < n_req; i++) { memcpy(p + o, reqs[i]->queries, reqs[i]->nq * d * 4); o += reqs[i]->nq * d * 4; }
    MSE_HIP_TRY(hipMemcpyAsync(idx->q32.p, idx->pin, in_bytes, hipMemcpyHostToDevice, st));
    uint32_t* ids_dev = idx->out.as<uint32_t>();
    float* keys_dev = reinterpret_cast<float*>(idx->out.as<char>() + total * kmax * 4);
    s->last_widened = 0; s->last_max_groups = 0;
    // same rule as the f16 dispatcher (dispatch.hip): the matrix-core pass for more than 8 queries, and for any count once the
    // rows have outgrown the caches
    const bool mfma = total > 8 || n >= ((size_t)1 << 22);
    const size_t tile = mfma ? (size_t)mfma_query_tile((int)d) : 8;
    for (size_t q0 = 0; q0 < total; q0 += tile) {
        const int m = (int)std::min(tile, total - q0);
        const int rc = mfma ? index_pass_mfma(idx, idx->q32.as<float>() + q0 * d, m, (int)kmax, ids_dev + q0 * kmax, keys_dev + q0 * kmax, kmax)
                            : index_pass_exact(idx, idx->q32.as<float>() + q0 * d, m, (int)kmax, ids_dev + q0 * kmax, keys_dev + q0 * kmax, kmax);
        if (rc) return -1;
    }
    MSE_HIP_TRY(hipMemcpyAsync(idx->pin, idx->out.p, out_bytes, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    const uint32_t* ids_h = reinterpret_cast<const uint32_t*>(p);
    const float* keys_h = reinterpret_cast<const float*>(p + total * kmax * 4);
    for (size_t i = 0, row = 0; i < n_req; i++) {
        DispatchReq* r = reqs[i];
        float* dist = static_cast<float*>(r->out_a);
        int64_t* lab = static_cast<int64_t*>(r->out_b);
        for (size_t q = 0; q < r->nq; q++, row++)
            for (size_t j = 0; j < r->k; j++) {
                const uint32_t id = ids_h[row * kmax + j];
                if (id == ID_NONE) continue;
                dist[q * r->k + j] = keys_h[row * kmax + j];
                lab[q * r->k + j] = (int64_t)id;
            }
    }
    return 0;
}

void index_run_batch(mse_index* idx, std::vector<DispatchReq*>& batch) {
    if (index_run_group(idx, batch.data(), batch.size()) == 0) {
        for (DispatchReq* r : batch) r->rc = 0;
        return;
    }
    if (batch.size() == 1) { batch[0]->rc = -1; batch[0]->err = mse_last_error(); return; }
    for (DispatchReq* r : batch) {   // a caller only ever sees its own failure
        idx->retried_alone++;
        r->rc = index_run_group(idx, &r, 1);
        if (r->rc) r->err = mse_last_error();
    }
}

}  // namespace

mse_index* mse_index_new(int d) {
    if (d <= 0 || d % 64 != 0 || d > D_MAX) {
        fail("index width must be a positive multiple of 64");
        return nullptr;
    }
    mse_index* idx = new (std::nothrow) mse_index();
    if (!idx) { fail("out of host memory"); return nullptr; }
    idx->d = d;
    if (hipGetDevice(&idx->device) != hipSuccess) idx->device = 0;
    idx->scratch = scratch_searcher_new();
    if (!idx->scratch) { delete idx; return nullptr; }
    idx->view.d = d; idx->view.owned = false; idx->view.n_cu = idx->scratch->n_cu; idx->view.device = idx->device;
    if (hipMalloc((void**)&idx->view.norm_bits_dev, 12) != hipSuccess || hipMemset(idx->view.norm_bits_dev, 0, 12) != hipSuccess) {
        mse_index_free(idx);
        fail("device allocation failed for the index");
        return nullptr;
    }
    idx->view.norm_ready = true;   // kept current by add()
    idx->scratch->base = &idx->view;
    const int device = idx->device;
    // at most one matrix-core pass worth of queries per gather; the wait budget follows the row count (add)
    idx->co.reset(new Coalescer((size_t)mfma_query_tile((int)idx->view.d), 200, [idx](std::vector<DispatchReq*>& b) { index_run_batch(idx, b); },
                                [device] { (void)hipSetDevice(device); }));
    return idx;
}
void mse_index_free(mse_index* idx) {
    if (!idx) return;
    idx->co.reset();   // joins the worker
    if (idx->scratch) mse_searcher_free(idx->scratch);
    if (idx->codes) (void)hipFree(idx->codes);
    if (idx->view.norm_bits_dev) (void)hipFree(idx->view.norm_bits_dev);
    if (idx->pin) (void)hipHostFree(idx->pin);
    delete idx;
}
size_t mse_index_ntotal(const mse_index* idx) { return idx ? idx->n : 0; }

int mse_index_add(mse_index* idx, const float* x, size_t n) {
    if (!idx) return fail("null index");
    if (n == 0) return 0;
    std::lock_guard<SharedExclusive> g(idx->rw);   // index.write() of src/main.rs:1016: waits for the searches in flight, holds new ones off
    const size_t d = idx->d;
    hipStream_t st = idx->scratch->stream;
    if (idx->n + n > 0xFFFFFFFEull) return fail("index full");
    if (idx->n + n > idx->cap) {
        size_t ncap = std::max<size_t>(idx->cap * 2, idx->n + n);
        ncap = std::max<size_t>(ncap, 1024);
        uint16_t* np = nullptr;
        MSE_HIP_TRY(hipMalloc((void**)&np, ncap * d * 2));
        hipError_t ce = idx->n ? hipMemcpyAsync(np, idx->codes, idx->n * d * 2, hipMemcpyDeviceToDevice, st) : hipSuccess;
        if (ce == hipSuccess) ce = hipStreamSynchronize(st);
        if (ce != hipSuccess) {   // the old block stays in place; the new one must not leak
            (void)hipFree(np);
            return fail(std::string("mse_index_add: growing the index failed: ") + hipGetErrorString(ce));
        }
        if (idx->codes) (void)hipFree(idx->codes);
        idx->codes = np;
        idx->cap = ncap;
    }
    // fp32 rows are staged on the device and narrowed there (RNE), as FAISS QT_fp16 encodes them
    DevBuf& stage = idx->scratch->misc;
    if (stage.ensure(n * d * 4)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(stage.p, x, n * d * 4, hipMemcpyHostToDevice, st));
    if (launch_f32_to_f16(stage.as<float>(), n * d, idx->codes + idx->n * d, st)) return -1;
    // the certificate's row-norm bound only ever grows: fold the new rows in (atomic maxima on the device)
    if (launch_row_norm_max(idx->codes + idx->n * d, n, (int)d, idx->view.norm_bits_dev, st)) return -1;
    MSE_HIP_TRY(hipStreamSynchronize(st));
    idx->n += n;
    idx->view.dev = idx->codes;
    idx->view.n = idx->n;
    idx->co->set_max_wait_us(default_wait_us(idx->n, d * 2));
    return 0;
}

int mse_index_search(mse_index* idx, const float* queries, size_t nq, size_t k, float* distances, int64_t* labels) {
    if (!idx) return fail("null index");
    if (nq == 0 || k == 0) return 0;
    if (!queries || !distances || !labels) return fail("null argument");
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    for (size_t i = 0; i < nq * k; i++) { distances[i] = -FLT_MAX; labels[i] = -1; }
    // index.read() of src/main.rs:1046: any number of searches at once; they meet in the coalescer and share passes
    idx->rw.lock_shared();
    DispatchReq r;
    r.queries = queries; r.nq = nq; r.k = k; r.out_a = distances; r.out_b = labels;
    const int rc = idx->co->submit(r);
    idx->rw.unlock_shared();
    return rc;
}

int mse_index_stats(mse_index* idx, uint64_t out[6]) {
    if (!idx || !out) return fail("null argument");
    const DispatchStats st = idx->co->stats();
    out[0] = st.queries; out[1] = st.requests; out[2] = st.passes; out[3] = st.max_pass_queries; out[4] = st.deadline_fires;
    out[5] = idx->retried_alone.load();
    return 0;
}

// ---- product quantiser ----------------------------------------------------------------------------
mse_pq* mse_pq_load(const float* centroids, size_t n_centroids, const float* transform, size_t n_dims,
                    size_t n_dims_per_code) {
    if (n_centroids == 0 || n_centroids > 256) { fail("at most 256 centroids (vector.rs:337)"); return nullptr; }
    if (n_dims == 0 || n_dims_per_code == 0 || n_dims % n_dims_per_code != 0) {
        fail("n_dims must be a positive multiple of n_dims_per_code");
        return nullptr;
    }
    if ((n_dims / n_dims_per_code) * n_centroids * 4 > 160 * 1024) { fail("PQ table exceeds LDS"); return nullptr; }
    mse_pq* pq = new (std::nothrow) mse_pq();
    if (!pq) { fail("out of host memory"); return nullptr; }
    pq->n_centroids = n_centroids; pq->d = n_dims; pq->dpc = n_dims_per_code; pq->n_chunks = n_dims / n_dims_per_code;
    if (hipGetDevice(&pq->device) != hipSuccess) pq->device = 0;
    if (hipMalloc((void**)&pq->centroids, n_centroids * n_dims * 4) != hipSuccess ||
        hipMalloc((void**)&pq->transform, n_dims * n_dims * 4) != hipSuccess ||
        hipMemcpy(pq->centroids, centroids, n_centroids * n_dims * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(pq->transform, transform, n_dims * n_dims * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc((void**)&pq->transform_t, n_dims * n_dims * 4) != hipSuccess) {
        mse_pq_free(pq);
        fail("device allocation/copy failed for the quantiser");
        return nullptr;
    }
    {
        std::vector<float> tt(n_dims * n_dims);
        for (size_t i = 0; i < n_dims; i++)
            for (size_t k = 0; k < n_dims; k++) tt[k * n_dims + i] = transform[i * n_dims + k];
        if (hipMemcpy(pq->transform_t, tt.data(), n_dims * n_dims * 4, hipMemcpyHostToDevice) != hipSuccess) {
            mse_pq_free(pq);
            fail("device copy failed for the quantiser");
            return nullptr;
        }
    }
    return pq;
}
void mse_pq_free(mse_pq* pq) {
    if (!pq) return;
    delete pq->co;   // joins its worker
    pq->co = nullptr;
    if (pq->centroids) (void)hipFree(pq->centroids);
    if (pq->transform) (void)hipFree(pq->transform);
    if (pq->transform_t) (void)hipFree(pq->transform_t);
    if (pq->pin) (void)hipHostFree(pq->pin);
    if (pq->scratch) mse_searcher_free(pq->scratch);
    if (pq->lane2) mse_searcher_free(pq->lane2);
    if (pq->lane3) mse_searcher_free(pq->lane3);
    delete pq;
}

int mse_pq_apply_transform(mse_pq* pq, const float* x, size_t n, float* out) {
    if (!pq) return fail("null quantiser");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> g(pq->mu);
    const size_t bytes = n * pq->d * 4;
    if (pq->a.ensure(bytes) || pq->b.ensure(bytes)) return -1;
    MSE_HIP_TRY(hipMemcpy(pq->a.p, x, bytes, hipMemcpyHostToDevice));
    if (launch_pq_transform(pq->transform, (int)pq->d, pq->a.as<float>(), n, pq->b.as<float>(), nullptr)) return -1;
    MSE_HIP_TRY(hipMemcpy(out, pq->b.p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int mse_pq_quantize_batch(mse_pq* pq, const float* x, size_t n, uint8_t* codes) {
    if (!pq) return fail("null quantiser");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> g(pq->mu);
    const size_t bytes = n * pq->d * 4;
    if (pq->a.ensure(bytes) || pq->b.ensure(bytes) || pq->c.ensure(n * pq->n_chunks)) return -1;
    MSE_HIP_TRY(hipMemcpy(pq->a.p, x, bytes, hipMemcpyHostToDevice));
    if (launch_pq_transform(pq->transform, (int)pq->d, pq->a.as<float>(), n, pq->b.as<float>(), nullptr)) return -1;
    if (launch_pq_quantize(pq->centroids, (int)pq->n_centroids, (int)pq->d, (int)pq->dpc, pq->b.as<float>(), n,
                           pq->c.as<uint8_t>(), nullptr)) return -1;
    MSE_HIP_TRY(hipMemcpy(codes, pq->c.p, n * pq->n_chunks, hipMemcpyDeviceToHost));
    return 0;
}

// device LUT build into pq->c (needs pq->mu held); query_dev = fp32 [d] on device
static int build_lut_locked(mse_pq* pq, const float* query_dev, hipStream_t st) {
    if (pq->b.ensure(pq->d * 4) || pq->c.ensure(pq->n_chunks * pq->n_centroids * 4)) return -1;
    if (launch_pq_transform_vec(pq->transform_t, (int)pq->d, query_dev, pq->b.as<float>(), st)) return -1;
    return launch_pq_lut(pq->centroids, (int)pq->n_centroids, (int)pq->d, (int)pq->dpc, pq->b.as<float>(),
                         pq->c.as<float>(), st);
}

int mse_pq_preprocess_query(mse_pq* pq, const float* query, float* lut) {
    if (!pq) return fail("null quantiser");
    std::lock_guard<std::mutex> g(pq->mu);
    if (pq->a.ensure(pq->d * 4)) return -1;
    MSE_HIP_TRY(hipMemcpy(pq->a.p, query, pq->d * 4, hipMemcpyHostToDevice));
    if (build_lut_locked(pq, pq->a.as<float>(), nullptr)) return -1;
    MSE_HIP_TRY(hipMemcpy(lut, pq->c.p, pq->n_chunks * pq->n_centroids * 4, hipMemcpyDeviceToHost));
    return 0;
}

int mse_pq_adc(mse_pq* pq, const float* lut, const uint8_t* codes, size_t n, int64_t* out) {
    if (!pq) return fail("null quantiser");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> g(pq->mu);
    const size_t lut_bytes = pq->n_chunks * pq->n_centroids * 4;
    if (pq->a.ensure(lut_bytes) || pq->b.ensure(n * pq->n_chunks) || pq->c.ensure(n * 8)) return -1;
    MSE_HIP_TRY(hipMemcpy(pq->a.p, lut, lut_bytes, hipMemcpyHostToDevice));
    MSE_HIP_TRY(hipMemcpy(pq->b.p, codes, n * pq->n_chunks, hipMemcpyHostToDevice));
    if (launch_pq_adc(pq->a.as<float>(), (int)pq->n_chunks, (int)pq->n_centroids, pq->b.as<uint8_t>(), n, nullptr, n,
                      nullptr, 0, nullptr, pq->c.as<int64_t>(), device_cu_count(), nullptr)) return -1;
    MSE_HIP_TRY(hipMemcpy(out, pq->c.p, n * 8, hipMemcpyDeviceToHost));
    return 0;
}

mse_codes* mse_codes_from_host(const uint8_t* codes, size_t n, size_t code_size, const uint8_t* descriptors,
                               size_t n_descriptors) {
    if (code_size == 0) { fail("code_size must be positive"); return nullptr; }
    mse_codes* c = new (std::nothrow) mse_codes();
    if (!c) { fail("out of host memory"); return nullptr; }
    c->n = n; c->code_size = code_size; c->n_desc = descriptors ? n_descriptors : 0;
    // + 4 KiB / 256 B of slack: the full-scan kernel fetches whole groups of 64 vectors (pq.hip)
    bool ok = hipMalloc((void**)&c->codes, n * code_size + 4096) == hipSuccess;
    if (ok && n) ok = hipMemcpy(c->codes, codes, n * code_size, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && c->n_desc) {
        ok = hipMalloc((void**)&c->desc, n * c->n_desc + 256) == hipSuccess;
        if (ok && n) ok = hipMemcpy(c->desc, descriptors, n * c->n_desc, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { mse_codes_free(c); fail("device allocation/copy failed for PQ codes"); return nullptr; }
    return c;
}
void mse_codes_free(mse_codes* c) {
    if (!c) return;
    if (c->codes) (void)hipFree(c->codes);
    if (c->desc) (void)hipFree(c->desc);
    delete c;
}
size_t mse_codes_len(const mse_codes* c) { return c ? c->n : 0; }

int mse_pq_adc_gather(mse_pq* pq, const mse_codes* c, const float* lut, const float* scales, const uint32_t* ids,
                      size_t n_ids, int64_t* out) {
    if (!pq || !c) return fail("null quantiser or codes");
    if (c->code_size != pq->n_chunks) return fail("code size does not match the quantiser");
    if (n_ids == 0) return 0;
    std::lock_guard<std::mutex> g(pq->mu);
    const size_t lut_bytes = pq->n_chunks * pq->n_centroids * 4;
    const size_t sc_bytes = c->n_desc * 4;
    if (pq->a.ensure(lut_bytes + 256 + sc_bytes) || pq->b.ensure(n_ids * 4) || pq->c.ensure(n_ids * 8)) return -1;
    float* scales_dev = nullptr;
    MSE_HIP_TRY(hipMemcpy(pq->a.p, lut, lut_bytes, hipMemcpyHostToDevice));
    if (scales && c->n_desc) {
        scales_dev = reinterpret_cast<float*>(pq->a.as<char>() + ((lut_bytes + 255) & ~(size_t)255));
        MSE_HIP_TRY(hipMemcpy(scales_dev, scales, sc_bytes, hipMemcpyHostToDevice));
    }
    MSE_HIP_TRY(hipMemcpy(pq->b.p, ids, n_ids * 4, hipMemcpyHostToDevice));
    if (launch_pq_adc(pq->a.as<float>(), (int)pq->n_chunks, (int)pq->n_centroids, c->codes, c->n, pq->b.as<uint32_t>(),
                      n_ids, scales_dev ? c->desc : nullptr, (int)c->n_desc, scales_dev, pq->c.as<int64_t>(),
                      device_cu_count(), nullptr)) return -1;
    MSE_HIP_TRY(hipMemcpy(out, pq->c.p, n_ids * 8, hipMemcpyDeviceToHost));
    return 0;
}

static int prep_table(mse_pq* pq, mse_searcher* s, const float* query_dev, float* t_dev, float* lut_dev) {
    if (launch_pq_transform_vec(pq->transform_t, (int)pq->d, query_dev, t_dev, s->stream)) return -1;
    return launch_pq_lut(pq->centroids, (int)pq->n_centroids, (int)pq->d, (int)pq->dpc, t_dev, lut_dev, s->stream);
}

// everything after the scan of one query.  gmax: the scan's group maxima [n_groups] (i64), or null when the codec shape has no
// group-maximum scan (then every vector is scored here).
static int scan_tail_async(mse_pq* pq, const mse_codes* c, mse_searcher* s, const int64_t* gmax, const float* query_dev,
                           const float* lut_dev, uint16_t* qf16_dev, const float* scales_dev, size_t r, size_t k,
                           int64_t* out_scores_dev, uint32_t* out_ids_dev) {
    hipStream_t st = s->stream;
    const size_t d = pq->d;
    const uint8_t* desc = scales_dev ? c->desc : nullptr;
    uint32_t* top_ids = nullptr;           // the r best by approximate score
    const int64_t* top_scores = nullptr;
    if (s->sel_keys.ensure(r * 8)) return -1;
    if (gmax) {
        const size_t n_groups = (c->n + 63) / 64;
        const size_t rg = std::min(r, n_groups);
        if (s->gkeys.ensure(rg * 8) || s->cand_ids.ensure(rg * 64 * 4) || s->cand_scores.ensure(rg * 64 * 8) || s->out_ids.ensure(r * 4))
            return -1;
        uint32_t* gsel = nullptr;
        LevelRef l0{KEY_I64, const_cast<int64_t*>(gmax), n_groups, 1, n_groups, false, 0};
        if (descend(s, l0, 1, (int)rg, &gsel, s->gkeys.p)) return -1;
        if (launch_expand_groups(gsel, rg, rg, 64, c->n, s->cand_ids.as<uint32_t>(), rg * 64, 1, st)) return -1;
        if (launch_pq_adc(lut_dev, (int)pq->n_chunks, (int)pq->n_centroids, c->codes, c->n, s->cand_ids.as<uint32_t>(), rg * 64,
                          desc, (int)c->n_desc, scales_dev, s->cand_scores.as<int64_t>(), s->n_cu, st)) return -1;
        SelectArgs a{};
        a.kind = KEY_I64; a.list_ids = s->cand_ids.as<uint32_t>(); a.list_keys = s->cand_scores.p; a.list_stride = rg * 64;
        a.n_list = rg * 64; a.k = (int)r; a.out_ids = s->out_ids.as<uint32_t>(); a.out_keys = s->sel_keys.p; a.out_stride = r; a.nq = 1;
        // r group maxima reach the r-th best group's key, so r vectors do: the key is a floor for the select.  This leans on the
        // scan kernels and pq_adc_kernel producing the SAME i64 for a vector (same adds in the same order, bias added after the
        // conversion); tests/test_gpu_pq_index_graph.py::test_group_maxima_equal_the_gathered_scores pins exactly that.
        if (rg == r) a.floor_hi = s->last_kth;
        if (launch_select(a, st)) return -1;
        top_ids = s->out_ids.as<uint32_t>();
    } else {
        // other codec shapes: every vector's score, then the tournament over them
        if (s->scores.ensure(c->n * 8)) return -1;
        if (launch_pq_adc(lut_dev, (int)pq->n_chunks, (int)pq->n_centroids, c->codes, c->n, nullptr, c->n, desc, (int)c->n_desc,
                          scales_dev, s->scores.as<int64_t>(), s->n_cu, st)) return -1;
        LevelRef l0{KEY_I64, s->scores.p, c->n, 1, c->n, false, 0};
        if (descend(s, l0, 1, (int)r, &top_ids, s->sel_keys.p)) return -1;
    }
    top_scores = s->sel_keys.as<int64_t>();
    if (s->base) {
        // exact re-score: f16(query) . base[id] (+ descriptor bias), query_disk_index.rs:168-170,477
        const mse_base* b = s->base;
        if (s->cand_scores.ensure(r * 8) || s->misc.ensure(k * 4) || s->gkeys.ensure(k * 8)) return -1;
        if (launch_f32_to_f16(query_dev, d, qf16_dev, st)) return -1;
        if (launch_score_rows(b->dev, b->n, (int)d, qf16_dev, false, top_ids, r, r, s->cand_scores.as<int64_t>(), nullptr, st))
            return -1;
        if (launch_add_descriptor(top_ids, r, desc, (int)c->n_desc, c->n, scales_dev, s->cand_scores.as<int64_t>(), st)) return -1;
        SelectArgs a{};
        a.kind = KEY_I64; a.list_ids = top_ids; a.list_keys = s->cand_scores.p; a.list_stride = r; a.n_list = r;
        a.k = (int)k; a.out_ids = s->misc.as<uint32_t>(); a.out_keys = s->gkeys.p; a.out_stride = k; a.nq = 1;
        if (launch_select(a, st)) return -1;
        top_ids = s->misc.as<uint32_t>();
        top_scores = s->gkeys.as<int64_t>();
    }
    MSE_HIP_TRY(hipMemcpyAsync(out_ids_dev, top_ids, k * 4, hipMemcpyDeviceToDevice, st));
    MSE_HIP_TRY(hipMemcpyAsync(out_scores_dev, top_scores, k * 8, hipMemcpyDeviceToDevice, st));
    return 0;
}

// One query (n_q = 1) or a PAIR of queries (n_q = 2) of the flat ADC scan, entirely on the searcher's stream (no host synchronisation):
//   table(query) -> scan of all codes keeping one maximum per 64 vectors (a pair shares ONE pass over the codes: pq_scan64x2_kernel)
//   -> per query: tournament over the maxima -> the r best groups' r x 64 vectors re-scored by the gather kernel (same arithmetic)
//   -> exact top-r by ADC score (ties: lower id) -> with base vectors: exact fast_dot re-score of those r (+ descriptor bias), top-k
// t_dev: transformed queries (fp32 [2][d]) scratch, lut_dev: table scratch (2 x 64 KiB), qf16_dev: f16 copy of a query (8 rows)
// prepared: lut_dev already holds this group's tables (the batch entry point builds all tables of a call in two launches)
static int scan_topk_async(mse_pq* pq, const mse_codes* c, mse_searcher* s, const float* queries_dev, int n_q, float* t_dev,
                           float* lut_dev, uint16_t* qf16_dev, const float* scales_dev, size_t r, size_t k,
                           int64_t* out_scores_dev, uint32_t* out_ids_dev, bool prepared = false) {
    const size_t d = pq->d, lut_floats = pq->n_chunks * pq->n_centroids;
    const uint8_t* desc = scales_dev ? c->desc : nullptr;
    const bool gm = pq_scan_gmax_supported((int)pq->n_chunks, (int)pq->n_centroids, desc, (int)c->n_desc, scales_dev);
    for (int j = 0; j < n_q && !prepared; j++)
        if (prep_table(pq, s, queries_dev + j * d, t_dev + j * d, lut_dev + j * lut_floats)) return -1;
    const size_t n_groups = (c->n + 63) / 64;
    int64_t* g0 = nullptr;
    int64_t* g1 = nullptr;
    if (gm) {
        if (s->scores.ensure(n_groups * 8) || (n_q == 2 && s->gmax.ensure(n_groups * 8))) return -1;
        g0 = s->scores.as<int64_t>();
        g1 = s->gmax.as<int64_t>();
        if (n_q == 2) {
            if (launch_pq_scan_gmax2(lut_dev, lut_dev + lut_floats, c->codes, c->n, desc, scales_dev, g0, g1, s->n_cu, s->stream)) return -1;
        } else if (launch_pq_scan_gmax(lut_dev, c->codes, c->n, desc, scales_dev, g0, s->n_cu, s->stream)) return -1;
    }
    for (int j = 0; j < n_q; j++)
        if (scan_tail_async(pq, c, s, gm ? (j ? g1 : g0) : nullptr, queries_dev + j * d, lut_dev + j * lut_floats, qf16_dev, scales_dev,
                            r, k, out_scores_dev + j * k, out_ids_dev + j * k)) return -1;
    return 0;
}

// FOUR (12-bit tables) or EIGHT (8-bit tables) queries in one pass over the codes (pq_scan64x4_kernel: integer nomination under a
// certificate, pq.hip).  Per query:
// the n_nom best groups by their integer maximum (+ one more, whose key bounds everything excluded) -> their vectors re-scored in the
// reference's arithmetic (pq_adc_kernel) -> exact top-r among them -> certificate flag (1 = provably the exact top-r of all
// vectors) -> with base vectors the same fast_dot re-score as the other paths.  flags_dev[j] = 0 asks the caller to repeat query j
// through the exact scan.  lut_dev: NQ tables; t_dev: NQ transformed queries.
// How many groups to nominate: the certificate needs every group whose integer maximum lies within ~2 eps of the r-th exact
// score.  With 12-bit tables that is a few tens of vectors at 1e8 codes (r + max(64, r / 2) groups); the 8-bit step is 16 times
// coarser and the band holds ~0.7 r vectors on random codes (simulated at 1e7 codes, r = 200: 339 groups reach it): 2 r + 112.
static size_t pq_nominated(size_t r, int NQ) { return NQ == 4 ? r + std::max<size_t>(64, r / 2) : 2 * r + 112; }
static int scan_topk4_async(mse_pq* pq, const mse_codes* c, mse_searcher* s, const float* queries_dev, float* t_dev, float* lut_dev,
                            uint16_t* qf16_dev, const float* scales_dev, size_t r, size_t k, int64_t* out_scores_dev,
                            uint32_t* out_ids_dev, int* flags_dev, bool prepared, int NQ) {
    hipStream_t st = s->stream;
    const size_t d = pq->d, lut_floats = pq->n_chunks * pq->n_centroids;
    const uint8_t* desc = scales_dev ? c->desc : nullptr;
    for (int j = 0; j < NQ && !prepared; j++)
        if (prep_table(pq, s, queries_dev + j * d, t_dev + j * d, lut_dev + j * lut_floats)) return -1;
    const size_t n_groups = (c->n + 63) / 64;
    // one more group than nominated is selected, only for its key
    const size_t n_nom = std::min(n_groups, pq_nominated(r, NQ));
    const size_t n_sel = std::min(n_groups, n_nom + 1);
    if (n_sel > (size_t)TOPK_KMAX) return fail("pq scan: r too large for the four-query scan");
    if (s->pq4.ensure(pq4_table_bytes() + 8 * sizeof(Pq4Params)) || s->gmax.ensure((size_t)NQ * n_groups * 4)) return -1;
    void* table = s->pq4.p;
    Pq4Params* params = reinterpret_cast<Pq4Params*>(s->pq4.as<char>() + pq4_table_bytes());
    if (launch_pq4_table(lut_dev, scales_dev, NQ, table, params, st, NQ)) return -1;
    hipEvent_t te0 = nullptr, te1 = nullptr;
    if (pq->timing) {
        while (s->ev_pool.size() < s->ev_used + 2) {
            hipEvent_t e = nullptr;
            MSE_HIP_TRY(hipEventCreate(&e));
            s->ev_pool.push_back(e);
        }
        te0 = s->ev_pool[s->ev_used];
        te1 = s->ev_pool[s->ev_used + 1];
        s->ev_used += 2;
        MSE_HIP_TRY(hipEventRecord(te0, st));
    }
    if (launch_pq_scan_gmax4(table, c->codes, c->n, desc, s->gmax.as<uint32_t>(), s->n_cu, st, NQ)) return -1;
    if (te1) MSE_HIP_TRY(hipEventRecord(te1, st));
    // the NQ tails as ONE chain of launches with a query dimension (their kernels are latency-bound: four chains in a row cost more
    // than the scan); every buffer at its final size before the first kernel that uses it
    const size_t n_cand = n_nom * 64;
    const size_t Q = (size_t)NQ;
    if (s->gkeys.ensure(Q * std::max(k, n_sel) * 8) || s->cand_ids.ensure(Q * n_cand * 4) || s->cand_scores.ensure(Q * std::max(r, n_cand) * 8) ||
        s->out_ids.ensure(Q * r * 4) || s->sel_keys.ensure(Q * r * 8) || s->misc.ensure(Q * k * 4) || s->scores.ensure(Q * k * 8)) return -1;
    uint32_t* gsel = nullptr;
    LevelRef l0{KEY_U32, s->gmax.p, 1, (size_t)NQ, n_groups, false, 0};   // group-major: element (q, g) at gmax[g * NQ + q]
#ifdef MSE_DEV_KERNELS
    // developer library, MSE_PQ_TAIL_SKIP=mask: the scan WITHOUT parts of its tail (answers are garbage, every query is reported
    // certified): what the tail costs the scans that run beside it.  1 tournament, 2 expand + re-score, 4 exact top-r, 8 certificate
    static const int tail_skip = getenv("MSE_PQ_TAIL_SKIP") ? atoi(getenv("MSE_PQ_TAIL_SKIP")) : 0;
    if (tail_skip) {
        if (!(tail_skip & 1) && descend(s, l0, NQ, (int)n_sel, &gsel, s->gkeys.p)) return -1;
        if (!(tail_skip & 2) && gsel) {
            if (launch_expand_groups(gsel, n_sel, n_nom, 64, c->n, s->cand_ids.as<uint32_t>(), n_cand, NQ, st)) return -1;
            if (launch_pq_adc(lut_dev, (int)pq->n_chunks, (int)pq->n_centroids, c->codes, c->n, s->cand_ids.as<uint32_t>(), n_cand, desc,
                              (int)c->n_desc, scales_dev, s->cand_scores.as<int64_t>(), s->n_cu, st, NQ, n_cand)) return -1;
        }
        if (!(tail_skip & 4)) {
            SelectArgs a{};
            a.kind = KEY_I64; a.list_ids = s->cand_ids.as<uint32_t>(); a.list_keys = s->cand_scores.p; a.list_stride = n_cand;
            a.n_list = n_cand; a.k = (int)r; a.out_ids = s->out_ids.as<uint32_t>(); a.out_keys = s->sel_keys.p; a.out_stride = r; a.nq = NQ;
            if (launch_select(a, st)) return -1;
        }
        MSE_HIP_TRY(hipMemsetAsync(flags_dev, 1, (size_t)NQ * sizeof(int), st));
        return 0;
    }
#endif
    if (descend(s, l0, NQ, (int)n_sel, &gsel, s->gkeys.p)) return -1;                      // gsel [NQ][n_sel], gkeys u32 [NQ][n_sel]
    if (launch_expand_groups(gsel, n_sel, n_nom, 64, c->n, s->cand_ids.as<uint32_t>(), n_cand, NQ, st)) return -1;
    if (launch_pq_adc(lut_dev, (int)pq->n_chunks, (int)pq->n_centroids, c->codes, c->n, s->cand_ids.as<uint32_t>(), n_cand, desc,
                      (int)c->n_desc, scales_dev, s->cand_scores.as<int64_t>(), s->n_cu, st, NQ, n_cand)) return -1;
    {
        SelectArgs a{};
        a.kind = KEY_I64; a.list_ids = s->cand_ids.as<uint32_t>(); a.list_keys = s->cand_scores.p; a.list_stride = n_cand;
        a.n_list = n_cand; a.k = (int)r; a.out_ids = s->out_ids.as<uint32_t>(); a.out_keys = s->sel_keys.p; a.out_stride = r; a.nq = NQ;
        if (launch_select(a, st)) return -1;
    }
    if (launch_pq4_certify(params, s->gkeys.as<uint32_t>(), (int)n_nom, (int)n_sel, s->out_ids.as<uint32_t>(), s->sel_keys.as<int64_t>(), r,
                           (int)std::min(r, c->n), NQ, flags_dev, st)) return -1;
    const uint32_t* top_ids = s->out_ids.as<uint32_t>();      // [NQ][r]
    const int64_t* top_scores = s->sel_keys.as<int64_t>();
    size_t top_stride = r;
    if (s->base) {
        // exact re-score: f16(query) . base[id] (+ descriptor bias), query_disk_index.rs:168-170,477 -- four queries per launch
        const mse_base* b = s->base;
        if (launch_f32_to_f16(queries_dev, Q * d, qf16_dev, st)) return -1;
        if (launch_score_rows(b->dev, b->n, (int)d, qf16_dev, false, top_ids, Q * r, r, s->cand_scores.as<int64_t>(), nullptr, st)) return -1;
        if (launch_add_descriptor(top_ids, Q * r, desc, (int)c->n_desc, c->n, scales_dev, s->cand_scores.as<int64_t>(), st)) return -1;
        SelectArgs b2{};
        b2.kind = KEY_I64; b2.list_ids = top_ids; b2.list_keys = s->cand_scores.p; b2.list_stride = r; b2.n_list = r;
        b2.k = (int)k; b2.out_ids = s->misc.as<uint32_t>(); b2.out_keys = s->scores.p; b2.out_stride = k; b2.nq = NQ;
        if (launch_select(b2, st)) return -1;
        top_ids = s->misc.as<uint32_t>();
        top_scores = s->scores.as<int64_t>();
        top_stride = k;
    }
    MSE_HIP_TRY(hipMemcpy2DAsync(out_ids_dev, k * 4, top_ids, top_stride * 4, k * 4, Q, hipMemcpyDeviceToDevice, st));
    MSE_HIP_TRY(hipMemcpy2DAsync(out_scores_dev, k * 8, top_scores, top_stride * 8, k * 8, Q, hipMemcpyDeviceToDevice, st));
    return 0;
}

// (a sharded scan hands its results over as a packed block on the device: launch_block_finish, topk.hip)
static int pq_scan_topk_batch_impl(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* queries_f32, size_t nq,
                                   const float* scales, size_t r, size_t k, int64_t* scores, uint32_t* ids, uint64_t id_offset, void* block_dev);

int mse_pq_scan_topk_batch(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* queries_f32, size_t nq,
                           const float* scales, size_t r, size_t k, int64_t* scores, uint32_t* ids) {
    if (!scores || !ids) return fail("null output");
    return pq_scan_topk_batch_impl(pq, c, s_or_null, queries_f32, nq, scales, r, k, scores, ids, 0, nullptr);
}

int mse_pq_scan_topk_block(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* queries_f32, size_t nq, const float* scales,
                           size_t r, size_t k, uint64_t id_offset, void* block_dev) {
    if (!block_dev) return fail("null output block");
    return pq_scan_topk_batch_impl(pq, c, s_or_null, queries_f32, nq, scales, r, k, nullptr, nullptr, id_offset, block_dev);
}

static int pq_scan_topk_batch_impl(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* queries_f32, size_t nq,
                                   const float* scales, size_t r, size_t k, int64_t* scores, uint32_t* ids, uint64_t id_offset, void* block_dev) {
    if (!pq || !c) return fail("null quantiser or codes");
    if (c->code_size != pq->n_chunks) return fail("code size does not match the quantiser");
    if (k == 0 || nq == 0) return 0;
    if (r < k) r = k;
    if (r > (size_t)TOPK_KMAX - 64) return fail("r too large (max 1984)");
    if (!block_dev)
        for (size_t i = 0; i < nq * k; i++) { scores[i] = INT64_MIN; ids[i] = MSE_ID_NONE; }
    if (c->n == 0) {
        if (block_dev) {   // an empty shard still hands over a block: all slots empty
            if (launch_block_finish(nullptr, nullptr, nq * k, 0, reinterpret_cast<int64_t*>(block_dev), reinterpret_cast<uint32_t*>(static_cast<char*>(block_dev) + nq * k * 8), nullptr)) return -1;
            MSE_HIP_TRY(hipStreamSynchronize(nullptr));
        }
        return 0;
    }
    std::lock_guard<std::mutex> g(pq->mu);
    pq->last_uncertified = 0;
    mse_searcher* s = s_or_null;
    if (!s) {
        if (!pq->scratch && !(pq->scratch = scratch_searcher_new())) return -1;
        s = pq->scratch;
    }
    if (s->base && s->base->n != c->n) return fail("base and codes differ in length");
    if (s->base && s->base->d != pq->d) return fail("base width differs from the quantiser");
    hipStream_t st = s->stream;
    const size_t d = pq->d;
    int rc = -1;
    do {
        // all queries (+ scales) go up in ONE copy from pinned memory; every query then runs on the streams; ONE download at the end
        const size_t sc_off = (nq * d * 4 + 255) & ~(size_t)255;
        const size_t sc_bytes = (scales && c->n_desc) ? c->n_desc * 4 : 0;
        const size_t in_bytes = sc_off + sc_bytes, out_bytes = nq * k * 12 + nq * 4;   // scores, ids, certificate flags
        // batches of 4 .. 2048 queries: all transformed queries and tables up front, in two launches (the same kernels the codec's
        // batch entry points use; same arithmetic as the one-vector forms) instead of two small launches per query in front of every scan
        const size_t lut_floats_b = pq->n_chunks * pq->n_centroids;
        const bool prep_all = nq >= 4 && nq <= 2048;
        const size_t n_tab = prep_all ? nq : 8;
        if (pq->a.ensure(in_bytes + 256) || pq->b.ensure(n_tab * d * 4) || pq->c.ensure(n_tab * lut_floats_b * 4)) break;
        if (s->q_stage.ensure(8 * d * 2) || s->out_scores.ensure(out_bytes)) break;
        if (pq->pin_cap < std::max(in_bytes, out_bytes)) {
            if (pq->pin) (void)hipHostFree(pq->pin);
            pq->pin = nullptr; pq->pin_cap = 0;
            const size_t want = std::max<size_t>(std::max(in_bytes, out_bytes) * 2, 1 << 16);
            if (hipHostMalloc(&pq->pin, want, hipHostMallocDefault) != hipSuccess) { fail("pinned staging allocation failed"); break; }
            pq->pin_cap = want;
        }
        memcpy(pq->pin, queries_f32, nq * d * 4);
        if (sc_bytes) memcpy(static_cast<char*>(pq->pin) + sc_off, scales, sc_bytes);
        if (hipMemcpyAsync(pq->a.p, pq->pin, in_bytes, hipMemcpyHostToDevice, st) != hipSuccess) { fail("H2D failed"); break; }
        float* scales_dev = sc_bytes ? reinterpret_cast<float*>(pq->a.as<char>() + sc_off) : nullptr;
        int64_t* const out_scores_dev = s->out_scores.as<int64_t>();
        uint32_t* const out_ids_dev = reinterpret_cast<uint32_t*>(s->out_scores.as<char>() + nq * k * 8);
        // Groups of queries alternate between TWO streams (each with its own scratch): the chain of small kernels that follows a
        // scan (tournament, re-score of the nominated groups, selects, certificate) runs beside the next group's scan, on the CUs
        // that scan leaves free.  (A third stream -- so that group u + 2 would not queue behind group u's tail -- and 16 / 24 / 32
        // spare CUs were measured in round 4: 0.194-0.203 ms per query at 32 per call in every combination, two streams and 8 spare
        // CUs being the best; the code below keeps room for a third lane.)
        mse_searcher* lanes[3] = {s, nullptr, nullptr};
        DevBuf &t2 = pq->t2, &lut2 = pq->lut2;
        DevBuf* qfs[3] = {&s->q_stage, &pq->qf2, &pq->qf3};
        if (nq >= 4) {
            mse_searcher** extra[2] = {&pq->lane2, &pq->lane3};
#ifdef MSE_DEV_KERNELS
            static const int n_extra = getenv("MSE_PQ_LANES") ? std::min(std::max(atoi(getenv("MSE_PQ_LANES")) - 1, 0), 2) : 1;   // developer library: 1-3 streams
#else
            const int n_extra = 1;
#endif
            bool lanes_ok = true;
            for (int e = 0; e < n_extra && lanes_ok; e++) {
                mse_searcher*& ln = *extra[e];
                if (ln && ln->base != s->base) { mse_searcher_free(ln); ln = nullptr; }
                if (!ln) ln = s->base ? mse_searcher_new(s->base) : scratch_searcher_new();
                lanes[e + 1] = ln;
                lanes_ok = ln != nullptr && qfs[e + 1]->ensure(8 * d * 2) == 0;
            }
            if (!lanes_ok || t2.ensure(8 * d * 4) || lut2.ensure(8 * pq->n_chunks * pq->n_centroids * 4)) break;
            if (hipStreamSynchronize(st) != hipSuccess) { fail("H2D failed"); break; }   // uploads visible to every stream
        }
        const int n_lanes = lanes[2] ? 3 : lanes[1] ? 2 : 1;
        // queries go through in groups that share one pass over the codes: EIGHTS (8-bit tables) and FOURS (12-bit tables) -- integer
        // nomination under a certificate, pq_scan64x4_kernel --, then a PAIR (pq_scan64x2_kernel, exact), then a single one; groups
        // alternate between the streams
        int* const flags_dev = reinterpret_cast<int*>(s->out_scores.as<char>() + nq * k * 12);
        if (hipMemsetAsync(flags_dev, 0xff, nq * 4, st) != hipSuccess) { fail("memset failed"); break; }   // non-zero = certified / exact
        if (prep_all && (launch_pq_transform(pq->transform, (int)d, pq->a.as<float>(), nq, pq->b.as<float>(), st) ||
                         launch_pq_lut_batch(pq->centroids, (int)pq->n_centroids, (int)d, (int)pq->dpc, pq->b.as<float>(), nq, pq->c.as<float>(), st)))
            break;
        if (lanes[1] && hipStreamSynchronize(st) != hipSuccess) { fail("memset failed"); break; }
        const uint8_t* desc_dev = scales_dev ? c->desc : nullptr;
        const bool four_ok = pq_scan_gmax_supported((int)pq->n_chunks, (int)pq->n_centroids, desc_dev, (int)c->n_desc, scales_dev) &&
                             r + std::max<size_t>(64, r / 2) + 1 <= (size_t)TOPK_KMAX;
        // eight per pass while the 8-bit certificate holds for this quantiser's data: a batch in which more than an eighth of the
        // eight-per-pass queries had to be repeated through the exact scan switches the handle back to four per pass for good
        const bool eight_ok = four_ok && !pq->avoid8 && pq_nominated(r, 8) + 1 <= (size_t)TOPK_KMAX;
        std::vector<char> in_eight(nq, 0);
        bool ok = true;
        size_t q = 0;
        for (size_t unit = 0; q < nq && ok; unit++) {
            const int n_q = (eight_ok && nq - q >= 8) ? 8 : (four_ok && nq - q >= 4) ? 4 : nq - q >= 2 ? 2 : 1;
            const int w = (int)(unit % (size_t)n_lanes);
            float* const tw = prep_all ? pq->b.as<float>() + q * d : w ? t2.as<float>() : pq->b.as<float>();     // (several lanes imply prep_all)
            float* const lw = prep_all ? pq->c.as<float>() + q * lut_floats_b : w ? lut2.as<float>() : pq->c.as<float>();
            uint16_t* const qw = qfs[w]->as<uint16_t>();
            if (n_q >= 4) {
                ok = scan_topk4_async(pq, c, lanes[w], pq->a.as<float>() + q * d, tw, lw, qw, scales_dev, r, k, out_scores_dev + q * k,
                                      out_ids_dev + q * k, flags_dev + q, prep_all, n_q) == 0;
                if (n_q == 8) for (int j = 0; j < 8; j++) in_eight[q + j] = 1;
            }
            else
                ok = scan_topk_async(pq, c, lanes[w], pq->a.as<float>() + q * d, n_q, tw, lw, qw, scales_dev, r, k,
                                     out_scores_dev + q * k, out_ids_dev + q * k, prep_all) == 0;
            q += n_q;
        }
        for (int w = 1; w < n_lanes; w++)
            if (hipStreamSynchronize(lanes[w]->stream) != hipSuccess) ok = false;
        if (!ok) { if (std::string(mse_last_error()).empty()) fail("scan failed"); break; }
        if (four_ok && nq >= 4) {
            // a query whose certificate did not hold (rare: the band of +-eps around its r-th score reached past the nominated
            // groups) is repeated through the exact one-query scan, into the same output rows
            std::vector<int> flags(nq);
            if (hipMemcpyAsync(flags.data(), flags_dev, nq * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { fail("D2H failed"); break; }
            pq->last_uncertified = 0;
            size_t n8 = 0, bad8 = 0;
            for (size_t j = 0; j < nq; j++) { n8 += in_eight[j]; bad8 += in_eight[j] && !flags[j]; }
            if (n8 && bad8 * 8 > n8) pq->avoid8 = true;
            for (size_t j = 0; j < nq && ok; j++)
                if (!flags[j]) {
                    pq->last_uncertified++;
                    ok = scan_topk_async(pq, c, s, pq->a.as<float>() + j * d, 1, pq->b.as<float>() + (prep_all ? j * d : 0),
                                         pq->c.as<float>() + (prep_all ? j * lut_floats_b : 0), s->q_stage.as<uint16_t>(), scales_dev, r, k,
                                         out_scores_dev + j * k, out_ids_dev + j * k, prep_all) == 0;
                }
            if (!ok) { if (std::string(mse_last_error()).empty()) fail("scan failed"); break; }
        }
        if (block_dev) {
            if (launch_block_finish(out_scores_dev, out_ids_dev, nq * k, id_offset, reinterpret_cast<int64_t*>(block_dev),
                                    reinterpret_cast<uint32_t*>(static_cast<char*>(block_dev) + nq * k * 8), st) || hipStreamSynchronize(st) != hipSuccess) {
                fail("block hand-over failed"); break;
            }
        } else if (hipMemcpyAsync(pq->pin, s->out_scores.p, nq * k * 12, hipMemcpyDeviceToHost, st) != hipSuccess ||
                   hipStreamSynchronize(st) != hipSuccess) { fail("D2H failed"); break; }
        {   // sustained rate: from the first scan's start to the last scan's end of this call, over its scans (they run back to back,
            // alternating between the streams: what a launch costs when the next one is already waiting behind it)
            size_t scans = 0;
            float span = 0.0f;
            for (mse_searcher* ln : lanes)
                if (ln && ln->ev_used >= 2 && lanes[0]->ev_used >= 2) {
                    float ms = 0.0f;
                    if (hipEventElapsedTime(&ms, lanes[0]->ev_pool[0], ln->ev_pool[ln->ev_used - 1]) == hipSuccess) span = std::max(span, ms);
                    scans += ln->ev_used / 2;
                }
            if (scans >= 4 && span > 0.0f) { pq->span_ms_total += span; pq->span_scans += scans; }
        }
        for (mse_searcher* ln : lanes)      // both streams are idle here: the scan launches' event pairs can be read
            if (ln) {
                for (size_t e = 0; e + 1 < ln->ev_used; e += 2) {
                    float ms = 0.0f;
                    if (hipEventElapsedTime(&ms, ln->ev_pool[e], ln->ev_pool[e + 1]) == hipSuccess) { pq->scan_ms_total += ms; pq->scan_launches++; }
                }
                ln->ev_used = 0;
            }
        if (!block_dev) {
            memcpy(scores, pq->pin, nq * k * 8);
            memcpy(ids, static_cast<char*>(pq->pin) + nq * k * 8, nq * k * 4);
            for (size_t i = 0; i < nq * k; i++)
                if (ids[i] == MSE_ID_NONE) scores[i] = INT64_MIN;
        }
        rc = 0;
    } while (0);
    return rc;
}

// test hook: the group maxima the flat scan nominates with -- lut1 == null: pq_scan64_kernel<true> for one table; otherwise
// pq_scan64x2_kernel for the pair.  out0 / out1: [ceil(n / 64)] i64 on the host.
int mse_debug_pq_group_max(mse_pq* pq, const mse_codes* c, const float* lut0, const float* lut1, const float* scales, int64_t* out0,
                           int64_t* out1) {
    if (!pq || !c || !lut0 || !out0 || (lut1 && !out1)) return fail("null argument");
    if (c->code_size != pq->n_chunks) return fail("code size does not match the quantiser");
    const uint8_t* desc = (scales && c->n_desc) ? c->desc : nullptr;
    if (!pq_scan_gmax_supported((int)pq->n_chunks, (int)pq->n_centroids, desc, (int)c->n_desc, scales))
        return fail("the group-maximum scan serves 64 x 256 codecs (and 4 descriptor bytes) only");
    if (c->n == 0) return 0;
    std::lock_guard<std::mutex> g(pq->mu);
    const size_t lut_bytes = pq->n_chunks * pq->n_centroids * 4, n_groups = (c->n + 63) / 64;
    if (pq->a.ensure(2 * lut_bytes + 256) || pq->c.ensure(2 * n_groups * 8)) return -1;
    float* l0 = pq->a.as<float>();
    float* l1 = reinterpret_cast<float*>(pq->a.as<char>() + lut_bytes);
    float* sc = reinterpret_cast<float*>(pq->a.as<char>() + 2 * lut_bytes);
    MSE_HIP_TRY(hipMemcpy(l0, lut0, lut_bytes, hipMemcpyHostToDevice));
    if (lut1) MSE_HIP_TRY(hipMemcpy(l1, lut1, lut_bytes, hipMemcpyHostToDevice));
    if (desc) MSE_HIP_TRY(hipMemcpy(sc, scales, c->n_desc * 4, hipMemcpyHostToDevice));
    int64_t* g0 = pq->c.as<int64_t>();
    int64_t* g1 = g0 + n_groups;
    const int rc = lut1 ? launch_pq_scan_gmax2(l0, l1, c->codes, c->n, desc, desc ? sc : nullptr, g0, g1, device_cu_count(), nullptr)
                        : launch_pq_scan_gmax(l0, c->codes, c->n, desc, desc ? sc : nullptr, g0, device_cu_count(), nullptr);
    if (rc) return -1;
    MSE_HIP_TRY(hipMemcpy(out0, g0, n_groups * 8, hipMemcpyDeviceToHost));
    if (lut1) MSE_HIP_TRY(hipMemcpy(out1, g1, n_groups * 8, hipMemcpyDeviceToHost));
    return 0;
}

// HIP-event timing of the four-queries-per-pass scan kernel (the dominant kernel of a batched scan): returns the totals so far,
// then sets the mode: 0 off, 1 on, 2 on and reset
int mse_pq_scan_timing(mse_pq* pq, int enable, double* total_ms, uint64_t* launches) {
    if (!pq) return fail("null quantiser");
    std::lock_guard<std::mutex> g(pq->mu);
    if (total_ms) *total_ms = pq->scan_ms_total;
    if (launches) *launches = pq->scan_launches;
    if (enable == 2) { pq->scan_ms_total = 0.0; pq->scan_launches = 0; pq->span_ms_total = 0.0; pq->span_scans = 0; }
    pq->timing = enable != 0;
    return 0;
}

// The SUSTAINED figure beside it: over the batch calls made while timing was on that ran at least four scans, the time from the first
// scan's start to the last scan's end (HIP events) and the number of scans in those spans.  span / scans = what a pass costs when passes
// run back to back (two streams, a group's tail beside the next group's scan); reset by mse_pq_scan_timing(pq, 2, ..).
int mse_pq_scan_sustained(mse_pq* pq, double* span_ms, uint64_t* scans) {
    if (!pq) return fail("null quantiser");
    std::lock_guard<std::mutex> g(pq->mu);
    if (span_ms) *span_ms = pq->span_ms_total;
    if (scans) *scans = pq->span_scans;
    return 0;
}

// PQ codes of rows that are already resident in HBM: quantize_batch (vector.rs:331-364) over f32 widenings of the base's f16 rows
// (what dump_processor feeds it, src/dump_processor.rs:468-481), 65536 rows at a time on the device; only the descriptors cross
// PCIe.  The codes equal mse_pq_quantize_batch's on the same rows.
mse_codes* mse_codes_quantize_base(mse_pq* pq, const mse_base* b, const uint8_t* descriptors, size_t n_descriptors) {
    if (!pq || !b) { fail("null quantiser or base"); return nullptr; }
    if (b->d != pq->d) { fail("base width differs from the quantiser"); return nullptr; }
    mse_codes* c = new (std::nothrow) mse_codes();
    if (!c) { fail("out of host memory"); return nullptr; }
    const size_t n = b->n, cs = pq->n_chunks;
    c->n = n; c->code_size = cs; c->n_desc = descriptors ? n_descriptors : 0;
    bool ok = hipMalloc((void**)&c->codes, n * cs + 4096) == hipSuccess;
    if (ok && c->n_desc) {
        ok = hipMalloc((void**)&c->desc, n * c->n_desc + 256) == hipSuccess;
        if (ok && n) ok = hipMemcpy(c->desc, descriptors, n * c->n_desc, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        std::lock_guard<std::mutex> g(pq->mu);
        const size_t step = 65536;
        ok = pq->a.ensure(step * pq->d * 4) == 0 && pq->b.ensure(step * pq->d * 4) == 0;
        for (size_t r0 = 0; ok && r0 < n; r0 += step) {
            const size_t m = std::min(step, n - r0);
            ok = launch_f16_to_f32(b->dev + r0 * b->d, m * b->d, pq->a.as<float>(), nullptr) == 0 &&
                 launch_pq_transform(pq->transform, (int)pq->d, pq->a.as<float>(), m, pq->b.as<float>(), nullptr) == 0 &&
                 launch_pq_quantize(pq->centroids, (int)pq->n_centroids, (int)pq->d, (int)pq->dpc, pq->b.as<float>(), m, c->codes + r0 * cs, nullptr) == 0;
        }
        if (ok) ok = hipDeviceSynchronize() == hipSuccess;
    }
    if (!ok) { mse_codes_free(c); if (std::string(mse_last_error()).empty()) fail("device allocation / quantisation failed"); return nullptr; }
    return c;
}

// test hook: the integer nomination scan by itself -- the four queries' group maxima (u32, [4][ceil(n / 64)]) and their
// certificate parameters (params_out [4][4] = delta, c, eps, ok), so that a test can rebuild the 12-bit tables on the host and
// check every maximum of the matrix-core scan against plain integer sums
int mse_debug_pq4_group_max(mse_pq* pq, const mse_codes* c, const float* luts4, const float* scales, int n_valid, int per_pass, uint32_t* out,
                            double* params_out) {
    if (!pq || !c || !luts4 || !out || !params_out) return fail("null argument");
    if (per_pass != 4 && per_pass != 8) return fail("per_pass must be 4 or 8");
    const size_t NQ = (size_t)per_pass;
    if (c->code_size != pq->n_chunks || pq->n_chunks != 64 || pq->n_centroids != 256) return fail("the four-query scan serves 64 x 256 codecs only");
    if (n_valid < 1 || n_valid > per_pass) return fail("n_valid must be 1 .. per_pass");
    const uint8_t* desc = (scales && c->n_desc) ? c->desc : nullptr;
    if (desc && c->n_desc != 4) return fail("the four-query scan takes 4 descriptor bytes");
    if (c->n == 0) return 0;
    std::lock_guard<std::mutex> g(pq->mu);
    const size_t lut_bytes = NQ * 64 * 256 * 4, n_groups = (c->n + 63) / 64;
    if (pq->a.ensure(lut_bytes + 256) || pq->b.ensure(pq4_table_bytes() + 8 * sizeof(Pq4Params)) || pq->c.ensure(NQ * n_groups * 4)) return -1;
    float* sc = reinterpret_cast<float*>(pq->a.as<char>() + lut_bytes);
    MSE_HIP_TRY(hipMemcpy(pq->a.p, luts4, lut_bytes, hipMemcpyHostToDevice));
    if (desc) MSE_HIP_TRY(hipMemcpy(sc, scales, 16, hipMemcpyHostToDevice));
    Pq4Params* params = reinterpret_cast<Pq4Params*>(pq->b.as<char>() + pq4_table_bytes());
    if (launch_pq4_table(pq->a.as<float>(), desc ? sc : nullptr, n_valid, pq->b.p, params, nullptr, per_pass)) return -1;
    if (launch_pq_scan_gmax4(pq->b.p, c->codes, c->n, desc, pq->c.as<uint32_t>(), device_cu_count(), nullptr, per_pass)) return -1;
    Pq4Params ph[8];
    {   // the scan writes group-major [n_groups][NQ]; the hook hands out [NQ][n_groups]
        std::vector<uint32_t> gm(NQ * n_groups);
        MSE_HIP_TRY(hipMemcpy(gm.data(), pq->c.p, NQ * n_groups * 4, hipMemcpyDeviceToHost));
        for (size_t g2 = 0; g2 < n_groups; g2++)
            for (size_t j = 0; j < NQ; j++) out[j * n_groups + g2] = gm[g2 * NQ + j];
    }
    MSE_HIP_TRY(hipMemcpy(ph, params, NQ * sizeof(Pq4Params), hipMemcpyDeviceToHost));
    for (int j = 0; j < per_pass; j++) { params_out[4 * j] = ph[j].delta; params_out[4 * j + 1] = ph[j].c; params_out[4 * j + 2] = ph[j].eps; params_out[4 * j + 3] = ph[j].ok; }
    return 0;
}

// queries of the last mse_pq_scan_topk_batch call whose four-query certificate did not hold and that were repeated through the
// exact scan (0 for batches of fewer than four queries)
uint32_t mse_pq_last_uncertified(mse_pq* pq) {
    if (!pq) return 0;
    std::lock_guard<std::mutex> g(pq->mu);
    return pq->last_uncertified;
}

// One query per call, possibly from many threads at once (the reference: one greedy_search / evaluate per request on a thread per
// core, src/query_disk_index.rs:711-736).  Calls meet in the quantiser's coalescer; the worker sorts what it gathered into groups
// that can share a batch call -- same codes, same base rows (or none), same r and k, the same descriptor scales byte for byte --
// and runs each group through mse_pq_scan_topk_batch (four queries per pass over the codes), on the searcher of the group's first
// caller (its owner is blocked in this call, so it is free to borrow).  Results are those of the call made alone.
namespace {
struct PqKey {
    const mse_codes* c; const mse_base* base; size_t r, k; const float* scales; size_t n_sc;
    bool same(const PqKey& o) const {
        if (c != o.c || base != o.base || r != o.r || k != o.k || (scales == nullptr) != (o.scales == nullptr)) return false;
        return !scales || (n_sc == o.n_sc && memcmp(scales, o.scales, n_sc * 4) == 0);
    }
};
PqKey pq_key_of(const DispatchReq* q) {
    const mse_codes* c = static_cast<const mse_codes*>(q->aux0);
    const mse_searcher* s = static_cast<const mse_searcher*>(q->aux1);
    return PqKey{c, s ? s->base : nullptr, q->aux_n, q->k, static_cast<const float*>(q->aux2), c->n_desc};
}
void pq_run_batch(mse_pq* pq, std::vector<DispatchReq*>& batch) {
    std::vector<char> taken(batch.size(), 0);
    std::vector<float> qs;
    std::vector<int64_t> sc;
    std::vector<uint32_t> id;
    for (size_t i = 0; i < batch.size(); i++) {
        if (taken[i]) continue;
        const PqKey key = pq_key_of(batch[i]);
        std::vector<DispatchReq*> grp;
        for (size_t j = i; j < batch.size(); j++)
            if (!taken[j] && key.same(pq_key_of(batch[j]))) { taken[j] = 1; grp.push_back(batch[j]); }
        const size_t d = pq->d, k = key.k, n = grp.size();
        qs.resize(n * d); sc.resize(n * k); id.resize(n * k);
        for (size_t j = 0; j < n; j++) memcpy(qs.data() + j * d, grp[j]->queries, d * 4);
        mse_searcher* s = const_cast<mse_searcher*>(static_cast<const mse_searcher*>(grp[0]->aux1));
        int rc = mse_pq_scan_topk_batch(pq, key.c, s, qs.data(), n, key.scales, key.r, k, sc.data(), id.data());
        if (rc == 0) {
            for (size_t j = 0; j < n; j++) {
                memcpy(grp[j]->out_a, sc.data() + j * k, k * 8);
                memcpy(grp[j]->out_b, id.data() + j * k, k * 4);
                grp[j]->rc = 0;
            }
            continue;
        }
        // the shared call failed: each caller is repeated alone with its own searcher and sees only its own outcome
        const std::string why = mse_last_error();
        for (size_t j = 0; j < n; j++) {
            if (n == 1) { grp[j]->rc = rc; grp[j]->err = why; break; }
            mse_searcher* sj = const_cast<mse_searcher*>(static_cast<const mse_searcher*>(grp[j]->aux1));
            grp[j]->rc = mse_pq_scan_topk_batch(pq, key.c, sj, static_cast<const float*>(grp[j]->queries), 1, key.scales, key.r, k,
                                                static_cast<int64_t*>(grp[j]->out_a), static_cast<uint32_t*>(grp[j]->out_b));
            if (grp[j]->rc) grp[j]->err = mse_last_error();
        }
    }
}
}  // namespace

int mse_pq_scan_topk(mse_pq* pq, const mse_codes* c, mse_searcher* s_or_null, const float* query_f32,
                     const float* scales, size_t r, size_t k, int64_t* scores, uint32_t* ids) {
    if (!pq || !c) return fail("null quantiser or codes");
    if (!query_f32 || !scores || !ids) return fail("null argument");
    // argument errors belong to this caller alone: they never enter the queue
    if (c->code_size != pq->n_chunks) return fail("code size does not match the quantiser");
    if (k == 0) return 0;
    if (std::max(r, k) > (size_t)TOPK_KMAX - 64) return fail("r too large (max 1984)");
    if (s_or_null && s_or_null->base && s_or_null->base->n != c->n) return fail("base and codes differ in length");
    if (s_or_null && s_or_null->base && s_or_null->base->d != pq->d) return fail("base width differs from the quantiser");
    {
        std::lock_guard<std::mutex> g(pq->co_mu);
        if (!pq->co) {
            const int device = pq->device;
            // up to 64 queries per gather (sixteen passes of four); wait budget from the size of the first code array seen
            pq->co = new (std::nothrow) Coalescer(64, default_wait_us(c->n, c->code_size + c->n_desc),
                                                  [pq](std::vector<DispatchReq*>& b) { pq_run_batch(pq, b); },
                                                  [device] { (void)hipSetDevice(device); });
            if (!pq->co) return fail("out of host memory");
        }
    }
    DispatchReq q;
    q.queries = query_f32; q.nq = 1; q.k = k; q.out_a = scores; q.out_b = ids;
    q.aux0 = c; q.aux1 = s_or_null; q.aux2 = (scales && c->n_desc) ? scales : nullptr; q.aux_n = r;
    return pq->co->submit(q);
}

int64_t mse_descriptor_product(const float* scales, size_t n_descriptors, const uint8_t* descriptors, uint32_t id) {
    // src/query_disk_index.rs:135-142 (host helper; the device paths fold the same sum into their kernels)
    int64_t r = 0;
    for (size_t j = 0; j < n_descriptors; j++)
        r += scale_dot_result(scales[j] * (float)descriptors[(size_t)id * n_descriptors + j]);
    return r;
}

}  // extern "C"
