// C ABI (include/mse.h): runtime, base vectors, searcher, brute-force search, flat index.
#include <cstdlib>
#include "../../include/mse.h"
#include "runtime.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <new>

namespace mse {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(const std::string& msg) {
    g_last_error = msg;
    return -1;
}

int ensure_dyn_lds(const void* kernel, int bytes) {
    int dev = 0;
    MSE_HIP_TRY(hipGetDevice(&dev));
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> granted;
    std::lock_guard<std::mutex> lk(mu);
    int& have = granted[{kernel, dev}];
    if (have >= bytes) return 0;
    MSE_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
    return 0;
}

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    release();
    size_t want = (bytes + 255) & ~(size_t)255;
    MSE_HIP_TRY(hipMalloc(&p, want));
    cap = want;
    return 0;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

constexpr size_t DENSE_MAX = 16384;  // a level this small is selected from directly

mse_searcher* scratch_searcher_new() {
    mse_searcher* s = new (std::nothrow) mse_searcher();
    if (!s) { fail("out of host memory"); return nullptr; }
    s->n_cu = device_cu_count();
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        delete s; fail("hipStreamCreate failed"); return nullptr;
    }
    s->own_stream = true;
    return s;
}

// Tournament descent (topk.hip header).  Leaves the ids of the k best level-0 entries per query in
// *sel_out ([nq][k], best first) and, when keys_out != nullptr, their raw keys in keys_out.
int descend(mse_searcher* s, const LevelRef& l0, int nq, int k, uint32_t** sel_out, void* keys_out) {
    hipStream_t st = s->stream;
    std::vector<LevelRef> lv;
    lv.push_back(l0);
    int li = 0;
    while (lv.back().n > DENSE_MAX) {
        const LevelRef cur = lv.back();
        const size_t n_out = (cur.n + TOPK_FANOUT - 1) / TOPK_FANOUT;
        const bool k64 = (cur.kind == KEY_I64 || cur.kind == KEY_U64);
        if (li >= 6) return fail("descend: too many levels");
        if (s->levels[li].ensure((size_t)nq * n_out * (k64 ? 8 : 4))) return -1;
        if (cur.group_major) {
            if (launch_reduce_max_gq(reinterpret_cast<const float*>(cur.ptr), cur.nq_pad, cur.n,
                                     s->levels[li].as<uint32_t>(), n_out, n_out, nq, st)) return -1;
        } else {
            if (launch_reduce_max(cur.kind, cur.ptr, cur.q_stride, cur.n, s->levels[li].p, n_out, n_out, nq, st, cur.e_stride))
                return -1;
        }
        lv.push_back(LevelRef{k64 ? KEY_U64 : KEY_U32, s->levels[li].p, n_out, 1, n_out, false, 0});
        li++;
    }
    if (s->sel_a.ensure((size_t)nq * k * 4) || s->sel_b.ensure((size_t)nq * k * 4) || s->thr.ensure((size_t)nq * 16)) return -1;
    uint32_t* cur_sel = s->sel_a.as<uint32_t>();
    uint32_t* nxt_sel = s->sel_b.as<uint32_t>();
    // each level hands the score key of its k-th best entry down as a floor: every one of the k best parents has a child with
    // exactly its key, so at least k children reach the floor and everything below it can be skipped unread by the radix passes
    unsigned long long* kth_cur = s->thr.as<unsigned long long>();
    unsigned long long* kth_nxt = kth_cur + nq;
    const int top = (int)lv.size() - 1;
    {
        SelectArgs a{};
        const LevelRef& L = lv[top];
        a.kind = L.kind; a.in = L.ptr; a.in_stride = L.q_stride; a.n_in = L.n;
        a.k = k; a.out_ids = cur_sel; a.out_keys = top == 0 ? keys_out : nullptr; a.out_stride = k; a.nq = nq;
        a.kth_hi_out = kth_cur;
        if (launch_select_strided(a, L.e_stride, st)) return -1;
    }
    for (int l = top - 1; l >= 0; l--) {
        SelectArgs a{};
        const LevelRef& L = lv[l];
        a.kind = L.kind; a.in = L.ptr; a.in_stride = L.q_stride; a.n_in = L.n;
        a.parents = cur_sel; a.par_stride = k; a.n_par = k; a.fanout = TOPK_FANOUT;
        a.k = k; a.out_ids = nxt_sel; a.out_keys = l == 0 ? keys_out : nullptr; a.out_stride = k; a.nq = nq;
        a.floor_hi = kth_cur; a.kth_hi_out = kth_nxt;
        if (launch_select_strided(a, L.e_stride, st)) return -1;
        std::swap(cur_sel, nxt_sel);
        std::swap(kth_cur, kth_nxt);
    }
    s->last_kth = kth_cur;
    *sel_out = cur_sel;
    return 0;
}

size_t visited_budget_bytes() {
    const char* e = getenv("MSE_VISITED_BUDGET_KB");
    if (e) return (size_t)atoll(e) * 1024;
    // half of what is free right now, between 256 MiB and 64 GiB (an index that fills the HBM leaves little; a small one leaves
    // room for every query of a batch at once)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)4 << 30;
    size_t b = free_b / 2;
    if (b < ((size_t)256 << 20)) b = (size_t)256 << 20;
    if (b > ((size_t)64 << 30)) b = (size_t)64 << 30;
    return b;
}

int ensure_base_norm(const mse_base* b, hipStream_t st) {
    std::lock_guard<std::mutex> g(b->norm_mu);
    if (b->norm_ready) return 0;
    if (!b->norm_bits_dev) MSE_HIP_TRY(hipMalloc((void**)&b->norm_bits_dev, 12));   // [max norm, max subnormal mass of a row, max |x_i|]
    MSE_HIP_TRY(hipMemsetAsync(b->norm_bits_dev, 0, 12, st));
    if (launch_row_norm_max(b->dev, b->n, (int)b->d, b->norm_bits_dev, st)) return -1;
    MSE_HIP_TRY(hipStreamSynchronize(st));
    b->norm_ready = true;
    return 0;
}

// exact mode, one pass of <= 8 queries already staged (padded) in s->q_stage
static int exact_pass(mse_searcher* s, int nq_pass, int k, uint64_t id_offset, int64_t* out_scores, uint32_t* out_ids,
                      size_t out_stride) {
    const mse_base* b = s->base;
    if (s->scores.ensure((size_t)nq_pass * b->n * 8)) return -1;
    if (launch_scan_exact(b->dev, b->n, (int)b->d, s->q_stage.p, nq_pass, false, s->scores.as<int64_t>(), b->n, nullptr,
                          s->n_cu, s->stream)) return -1;
    if (s->sel_keys.ensure((size_t)nq_pass * k * 8)) return -1;
    uint32_t* sel = nullptr;
    LevelRef l0{KEY_I64, s->scores.p, b->n, 1, b->n, false, 0};
    if (descend(s, l0, nq_pass, k, &sel, s->sel_keys.p)) return -1;
    return launch_finalize(sel, s->sel_keys.as<int64_t>(), k, k, nq_pass, id_offset, out_scores, out_ids, out_stride,
                           nullptr, 0, 0, 0, nullptr, nullptr, s->stream);
}

// How many queries one call of mfma_pass may take: one pass over the rows (mfma_query_tile) for a large base; for a SMALL base -- group
// maxima of all queries within 256 MiB -- up to 8192, scanned pass by pass into one wide array of group maxima and finished by ONE
// tournament / re-score / certificate over all of them.  The fixed cost of a pass (a dozen small launches and a host synchronisation
// for the margins) is what a small base pays for: 4096 queries against a 4096-row entry table (the request path's entry step,
// beam_search.hip) took 13 passes x 0.28 ms.
static size_t mfma_call_tile(const mse_base* b, size_t k) {
    const size_t tile = (size_t)mfma_query_tile((int)b->d);
    const size_t n_groups = (b->n + GROUP_ROWS - 1) / GROUP_ROWS;
    size_t fit = ((size_t)256 << 20) / (std::max<size_t>(n_groups, 1) * 4) / tile * tile;
    // the first round re-scores (k + 8) groups of 32 rows per query: ids + scores of all queries within 1 GiB
    const size_t per_query = std::min<size_t>(std::max<size_t>(k + 8, 16), TOPK_KMAX) * GROUP_ROWS * 12;
    fit = std::min(fit, ((size_t)1 << 30) / per_query / tile * tile);
    if (fit > 8192 / tile * tile) fit = 8192 / tile * tile;
    return std::max(fit, tile);
}

// MFMA mode for up to mfma_call_tile(base) queries (device pointer to [nq][d] f16, contiguous)
static int mfma_pass(mse_searcher* s, const uint16_t* q_dev, int nq_pass, int k, uint64_t id_offset,
                     int64_t* out_scores, uint32_t* out_ids, size_t out_stride) {
    const mse_base* b = s->base;
    hipStream_t st = s->stream;
    const int d = (int)b->d;
    // one pass over the rows serves up to 320 queries (padded to 128 / 192 / 256 / 320); more queries (small base only) = full passes
    // and a last one, their columns side by side in the array of group maxima
    const int tile = mfma_query_tile(d);
    const int n_full = nq_pass / tile, rem = nq_pass - n_full * tile;
    const int nq_pad = n_full * tile + (rem ? mfma_pad(rem, d) : 0);
    if (ensure_base_norm(b, st)) return -1;
    // padded query tile
    if (s->q_stage.ensure((size_t)nq_pad * d * 2)) return -1;
    MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, (size_t)nq_pad * d * 2, st));
    MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, q_dev, (size_t)nq_pass * d * 2, hipMemcpyDeviceToDevice, st));
    const size_t n_groups = (b->n + GROUP_ROWS - 1) / GROUP_ROWS;
    if (s->gmax.ensure(n_groups * (size_t)nq_pad * 4)) return -1;
    // the full passes go out as ONE launch (a small base has few row tiles: its passes fill the chip side by side), then the remainder
    const size_t one_tile_packed = (size_t)(d / 64) * tile * 128;
    if (s->qpacked.ensure(std::max(mfma_packed_bytes(d), (size_t)std::max(n_full, 1) * one_tile_packed))) return -1;
    if (n_full &&
        launch_scan_mfma(b->dev, b->n, d, s->q_stage.as<uint16_t>(), tile, s->qpacked.p, s->gmax.as<float>(), s->n_cu, st,
                         s->timing ? s->ev0 : nullptr, s->timing && !rem ? s->ev1 : nullptr, nq_pad, n_full)) return -1;
    if (rem &&
        launch_scan_mfma(b->dev, b->n, d, s->q_stage.as<uint16_t>() + (size_t)n_full * tile * d, nq_pad - n_full * tile, s->qpacked.p,
                         s->gmax.as<float>() + n_full * tile, s->n_cu, st, s->timing && !n_full ? s->ev0 : nullptr,
                         s->timing ? s->ev1 : nullptr, nq_pad, 1)) return -1;
    bool timing_pending = s->timing;
    if (s->eps.ensure((size_t)nq_pass * 8) || s->margin.ensure((size_t)nq_pass * 8)) return -1;   // second halves: the widening's compact set
    // |mfma score - exact-order score| <= 2 * gamma_1151 * sum|x_i q_i| <= 1.4e-4 * |x||q|; doubled again
    // because the matrix core's internal rounding is not documented.
    if (launch_query_eps(s->q_stage.as<uint16_t>(), nq_pass, d, b->norm_bits_dev, 2.8e-4f, s->eps.as<float>(), st))
        return -1;

    std::vector<float> margin_h(nq_pass);
    const int kg0 = (int)std::min<size_t>(std::max(k + 8, 16), TOPK_KMAX);
    s->last_widened = 0;
    // One round of: tournament over the group maxima -> the kg best groups' rows re-scored exactly -> exact top-k -> certificate.
    // gm: group maxima [n_groups][gm_pad] of the nq queries in `qs` ([nq][d] f16); results go to dst_* with stride dst_stride;
    // margins (> 0 = certified) come back in margin_h[0..nq).
    auto round = [&](const float* gm, int gm_pad, const uint16_t* qs, int nq, int kg_eff, const float* eps_dev, float* margin_dev,
                     int64_t* dst_s, uint32_t* dst_i, size_t dst_stride, uint64_t id_off) -> int {
        if (s->gkeys.ensure((size_t)nq * kg_eff * 4)) return -1;
        uint32_t* gsel = nullptr;
        LevelRef l0{KEY_F32, gm, 1, (size_t)gm_pad, n_groups, true, gm_pad};
        if (descend(s, l0, nq, kg_eff, &gsel, s->gkeys.p)) return -1;
        const size_t n_cand = (size_t)kg_eff * GROUP_ROWS;
        if (s->cand_ids.ensure((size_t)nq * n_cand * 4) || s->cand_scores.ensure((size_t)nq * n_cand * 8)) return -1;
        if (launch_expand_groups(gsel, kg_eff, kg_eff, GROUP_ROWS, b->n, s->cand_ids.as<uint32_t>(), n_cand, nq, st)) return -1;
        if (launch_score_rows(b->dev, b->n, d, qs, false, s->cand_ids.as<uint32_t>(), (size_t)nq * n_cand, n_cand,
                              s->cand_scores.as<int64_t>(), nullptr, st)) return -1;
        // final exact selection among the re-scored candidates
        if (s->sel_keys.ensure((size_t)nq * k * 8) || s->misc.ensure((size_t)nq * k * 4)) return -1;
        SelectArgs a{};
        a.kind = KEY_I64; a.list_ids = s->cand_ids.as<uint32_t>(); a.list_keys = s->cand_scores.p;
        a.list_stride = n_cand; a.n_list = n_cand; a.k = k; a.out_ids = s->misc.as<uint32_t>();
        a.out_keys = s->sel_keys.p; a.out_stride = k; a.nq = nq;
        if (launch_select(a, st)) return -1;
        if (launch_finalize(s->misc.as<uint32_t>(), s->sel_keys.as<int64_t>(), k, k, nq, id_off, dst_s, dst_i, dst_stride,
                            s->gkeys.as<float>(), kg_eff, kg_eff, n_groups, eps_dev, margin_dev, st)) return -1;
        MSE_HIP_TRY(hipMemcpyAsync(margin_h.data(), margin_dev, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        MSE_HIP_TRY(hipStreamSynchronize(st));
        s->last_max_groups = std::max<uint32_t>(s->last_max_groups, (uint32_t)kg_eff);
        return 0;
    };
    if (round(s->gmax.as<float>(), nq_pad, s->q_stage.as<uint16_t>(), nq_pass, kg0, s->eps.as<float>(), s->margin.as<float>(), out_scores,
              out_ids, out_stride, id_offset)) return -1;
    if (timing_pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) { s->scan_ms_total += ms; s->scan_launches++; }
        timing_pending = false;
    }
    std::vector<uint32_t> bad;
    for (int i = 0; i < nq_pass; i++)
        if (!(margin_h[i] > 0.0f)) bad.push_back((uint32_t)i);
    if (bad.empty() || (size_t)kg0 >= n_groups) return 0;
    s->last_widened = (uint32_t)bad.size();
    // The queries whose certificate failed (near-duplicate rows around their k-th score, ties) are carried on as a COMPACT set: their
    // columns of the group maxima, their query rows.  Widening then costs what those few queries cost -- not a 4x, 16x, 64x larger
    // re-score for all 256 (a clustered 1e8-row set: 84 ms per pass of 256 queries instead of 58, before this).
    const int nb = (int)bad.size(), nbp = (nb + 31) / 32 * 32;
    if (s->widx.ensure((size_t)nb * 5) || s->wq.ensure((size_t)(nb + 8) * d * 2) || s->wg.ensure(n_groups * (size_t)nbp * 4) ||
        s->wout.ensure((size_t)std::max(nb, 8) * k * 12)) return -1;
    uint32_t* idx_dev = s->widx.as<uint32_t>();
    uint8_t* take_dev = reinterpret_cast<uint8_t*>(idx_dev + nb);
    MSE_HIP_TRY(hipMemcpyAsync(idx_dev, bad.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st));
    if (launch_gather_rows16(s->q_stage.p, (size_t)d * 2, idx_dev, nb, s->wq.p, st)) return -1;
    if (launch_gather_columns(s->gmax.as<float>(), nq_pad, n_groups, idx_dev, nb, s->wg.as<float>(), nbp, st)) return -1;
    float* eps2 = s->eps.as<float>() + nq_pass;
    float* margin2 = s->margin.as<float>() + nq_pass;
    if (launch_query_eps(s->wq.as<uint16_t>(), nb, d, b->norm_bits_dev, 2.8e-4f, eps2, st)) return -1;
    int64_t* w_s = s->wout.as<int64_t>();
    uint32_t* w_i = reinterpret_cast<uint32_t*>(s->wout.as<char>() + (size_t)nb * k * 8);
    std::vector<uint8_t> open_q(nb, 1);   // still uncertified
    int kg = kg0 * 4;
    for (;;) {
        const int kg_eff = (int)std::min<size_t>(kg, TOPK_KMAX);
        if (round(s->wg.as<float>(), nbp, s->wq.as<uint16_t>(), nb, kg_eff, eps2, margin2, w_s, w_i, (size_t)k, id_offset)) return -1;
        // rows of the queries certified in this round (or examined completely) go to their places
        std::vector<uint8_t> take(nb, 0);
        int still = 0;
        for (int j = 0; j < nb; j++) {
            if (!open_q[j]) continue;
            if (margin_h[j] > 0.0f || (size_t)kg_eff >= n_groups) { take[j] = 1; open_q[j] = 0; } else still++;
        }
        MSE_HIP_TRY(hipMemcpyAsync(take_dev, take.data(), (size_t)nb, hipMemcpyHostToDevice, st));
        if (launch_scatter_topk(idx_dev, take_dev, nb, k, w_s, w_i, out_scores, out_ids, out_stride, st)) return -1;
        MSE_HIP_TRY(hipStreamSynchronize(st));   // `take` is a stack-owned source
        if (still == 0) return 0;
        if (kg_eff >= TOPK_KMAX) break;
        kg = kg_eff * 4;
    }
    // cannot widen further: the exact scan for what is left, 8 queries at a time
    std::vector<uint32_t> rest;
    for (int j = 0; j < nb; j++)
        if (open_q[j]) rest.push_back((uint32_t)j);
    for (size_t r0 = 0; r0 < rest.size(); r0 += 8) {
        const int nqp = (int)std::min<size_t>(8, rest.size() - r0);
        if (s->q_stage.ensure((size_t)8 * d * 2)) return -1;
        MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, (size_t)8 * d * 2, st));
        for (int j = 0; j < nqp; j++)
            MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.as<char>() + (size_t)j * d * 2, s->wq.as<char>() + (size_t)rest[r0 + j] * d * 2, (size_t)d * 2,
                                       hipMemcpyDeviceToDevice, st));
        if (exact_pass(s, nqp, k, id_offset, w_s, w_i, (size_t)k)) return -1;
        std::vector<uint32_t> dst(nqp);
        for (int j = 0; j < nqp; j++) dst[j] = bad[rest[r0 + j]];
        MSE_HIP_TRY(hipMemcpyAsync(idx_dev, dst.data(), (size_t)nqp * 4, hipMemcpyHostToDevice, st));
        if (launch_scatter_topk(idx_dev, nullptr, nqp, k, w_s, w_i, out_scores, out_ids, out_stride, st)) return -1;
        MSE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

}  // namespace mse

using namespace mse;

extern "C" {

const char* mse_last_error(void) { return g_last_error.c_str(); }
const char* mse_version(void) { return "mse-hip 0.1 (gfx950)"; }

int mse_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int mse_set_device(int ordinal) {
    MSE_HIP_TRY(hipSetDevice(ordinal));
    return 0;
}
int mse_device_synchronize(void) {
    MSE_HIP_TRY(hipDeviceSynchronize());
    return 0;
}
int mse_device_mem_info(size_t* free_bytes, size_t* total_bytes) {
    size_t f = 0, t = 0;
    MSE_HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}

int64_t mse_scale_dot_f32(float x) { return scale_dot_result(x); }
int64_t mse_scale_dot_f64(double x) { return scale_dot_result_f64(x); }

// ---- base ------------------------------------------------------------------------------------
static mse_base* base_alloc(size_t n, size_t d, bool owned) {
    if (d == 0 || d % 64 != 0 || d > (size_t)D_MAX) {
        fail("vector width must be a positive multiple of 64 (fast_dot asserts len % 64 == 0)");
        return nullptr;
    }
    if (n > 0xFFFFFFFEull) {
        fail("row ids are u32: too many rows");
        return nullptr;
    }
    mse_base* b = new (std::nothrow) mse_base();
    if (!b) { fail("out of host memory"); return nullptr; }
    b->n = n; b->d = d; b->owned = owned; b->n_cu = device_cu_count();
    if (hipGetDevice(&b->device) != hipSuccess) b->device = 0;
    return b;
}
mse_base* mse_base_from_host(const uint16_t* data, size_t n_rows, size_t d) {
    mse_base* b = base_alloc(n_rows, d, true);
    if (!b) return nullptr;
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n_rows * d * 2, 256);
    if (hipMalloc(&p, bytes) != hipSuccess) { delete b; fail("hipMalloc failed for base vectors"); return nullptr; }
    if (n_rows && hipMemcpy(p, data, n_rows * d * 2, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p); delete b; fail("hipMemcpy failed for base vectors"); return nullptr;
    }
    b->dev = reinterpret_cast<const uint16_t*>(p);
    return b;
}
mse_base* mse_base_wrap_device(const void* data_dev, size_t n_rows, size_t d) {
    mse_base* b = base_alloc(n_rows, d, false);
    if (!b) return nullptr;
    b->dev = reinterpret_cast<const uint16_t*>(data_dev);
    // the device that holds the rows, not the one that happens to be current on the calling thread: worker threads made for this
    // base (coalescer, shard group) select b->device
    hipPointerAttribute_t at{};
    if (data_dev && hipPointerGetAttributes(&at, data_dev) == hipSuccess) {
        if (at.type == hipMemoryTypeDevice) b->device = at.device;
    } else {
        (void)hipGetLastError();
    }
    return b;
}
mse_base* mse_base_generate(uint32_t seed, uint64_t first_row, size_t n_rows, size_t d) {
    mse_base* b = base_alloc(n_rows, d, true);
    if (!b) return nullptr;
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n_rows * d * 2, 256);
    if (hipMalloc(&p, bytes) != hipSuccess) { delete b; fail("hipMalloc failed for base vectors"); return nullptr; }
    b->dev = reinterpret_cast<const uint16_t*>(p);
    if (launch_generate_rows(reinterpret_cast<uint16_t*>(p), seed, first_row, n_rows, (int)d, nullptr) ||
        hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p); delete b; if (g_last_error.empty()) fail("row generation failed"); return nullptr;
    }
    return b;
}
void mse_base_free(mse_base* b) {
    if (!b) return;
    if (b->disp) mse_dispatcher_free(b->disp);   // joins its worker; no search may be in flight (as for the rows themselves)
    if (b->owned && b->dev) (void)hipFree(const_cast<uint16_t*>(b->dev));
    if (b->norm_bits_dev) (void)hipFree(b->norm_bits_dev);
    delete b;
}
int mse_base_rows_changed(mse_base* b) {
    if (!b) return fail("null base");
    std::lock_guard<std::mutex> g(b->norm_mu);
    b->norm_ready = false;
    return 0;
}
size_t mse_base_len(const mse_base* b) { return b ? b->n : 0; }
size_t mse_base_dim(const mse_base* b) { return b ? b->d : 0; }
const void* mse_base_device_ptr(const mse_base* b) { return b ? b->dev : nullptr; }
int mse_base_read_rows(const mse_base* b, size_t first_row, size_t n_rows, uint16_t* out) {
    if (!b) return fail("null base");
    if (first_row + n_rows > b->n) return fail("row range out of bounds");
    if (n_rows == 0) return 0;
    MSE_HIP_TRY(hipMemcpy(out, b->dev + first_row * b->d, n_rows * b->d * 2, hipMemcpyDeviceToHost));
    return 0;
}

int mse_fast_dot_f16(const uint16_t* x, const uint16_t* y, size_t n, int64_t* out) {
    if (n == 0 || n % 64 != 0 || n > (size_t)D_MAX) return fail("fast_dot: length must be a positive multiple of 64");
    DevBuf buf;
    if (buf.ensure(n * 4 + 64)) return -1;
    char* p = buf.as<char>();
    uint32_t zero = 0;
    MSE_HIP_TRY(hipMemcpy(p, x, n * 2, hipMemcpyHostToDevice));
    MSE_HIP_TRY(hipMemcpy(p + n * 2, y, n * 2, hipMemcpyHostToDevice));
    MSE_HIP_TRY(hipMemcpy(p + n * 4, &zero, 4, hipMemcpyHostToDevice));
    if (launch_score_rows(reinterpret_cast<const uint16_t*>(p + n * 2), 1, (int)n, p, false,
                          reinterpret_cast<const uint32_t*>(p + n * 4), 1, 1, reinterpret_cast<int64_t*>(p + n * 4 + 8),
                          nullptr, nullptr)) return -1;
    MSE_HIP_TRY(hipMemcpy(out, p + n * 4 + 8, 8, hipMemcpyDeviceToHost));
    return 0;
}

// ---- searcher --------------------------------------------------------------------------------
mse_searcher* mse_searcher_new(const mse_base* b) {
    if (!b) { fail("null base"); return nullptr; }
    mse_searcher* s = new (std::nothrow) mse_searcher();
    if (!s) { fail("out of host memory"); return nullptr; }
    s->base = b;
    s->n_cu = b->n_cu;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        delete s; fail("hipStreamCreate failed"); return nullptr;
    }
    s->own_stream = true;
    return s;
}
void mse_searcher_free(mse_searcher* s) {
    if (!s) return;
    if (s->own_stream && s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->ev_wait) (void)hipEventDestroy(s->ev_wait);
    if (s->bev0) (void)hipEventDestroy(s->bev0);
    if (s->bev1) (void)hipEventDestroy(s->bev1);
    if (s->pin) (void)hipHostFree(s->pin);
    delete s;
}
int mse_searcher_set_stream(mse_searcher* s, void* hip_stream) {
    if (!s) return fail("null searcher");
    if (s->own_stream && s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    s->stream = reinterpret_cast<hipStream_t>(hip_stream);
    s->own_stream = false;
    return 0;
}
void* mse_searcher_stream(const mse_searcher* s) { return s ? (void*)s->stream : nullptr; }
int mse_searcher_scan_timing(mse_searcher* s, int enable, double* total_ms, uint64_t* launches) {
    if (!s) return fail("null searcher");
    if (total_ms) *total_ms = s->scan_ms_total;
    if (launches) *launches = s->scan_launches;
    if (enable && !s->ev0) {
        MSE_HIP_TRY(hipEventCreate(&s->ev0));
        MSE_HIP_TRY(hipEventCreate(&s->ev1));
    }
    if (enable == 2) { s->scan_ms_total = 0.0; s->scan_launches = 0; }
    s->timing = enable != 0;
    return 0;
}
int mse_searcher_last_stats(const mse_searcher* s, uint32_t* n_widened, uint32_t* max_groups) {
    if (!s) return fail("null searcher");
    if (n_widened) *n_widened = s->last_widened;
    if (max_groups) *max_groups = s->last_max_groups;
    return 0;
}

size_t mse_queries_per_pass_max(size_t d) { return d && d % 64 == 0 ? (size_t)mfma_query_tile((int)d) : 0; }

int mse_bruteforce_topk_f16_dev(mse_searcher* s, const void* queries_dev, size_t nq, size_t k, int mode,
                                uint64_t id_offset, void* scores_dev, void* ids_dev) {
    if (!s) return fail("null searcher");
    if (nq == 0 || k == 0) return 0;
    if (k > (size_t)TOPK_KMAX - 64) return fail("k too large (max 1984)");
    const mse_base* b = s->base;
    const int d = (int)b->d;
    int64_t* out_scores = reinterpret_cast<int64_t*>(scores_dev);
    uint32_t* out_ids = reinterpret_cast<uint32_t*>(ids_dev);
    const uint16_t* q = reinterpret_cast<const uint16_t*>(queries_dev);
    if (mode == MSE_MODE_AUTO) mode = nq <= 8 ? MSE_MODE_EXACT : MSE_MODE_MFMA;
    s->last_widened = 0;
    s->last_max_groups = 0;
    if (b->n == 0) {
        // nothing to score: every slot is empty
        std::vector<int64_t> hs(nq * k, INT64_MIN);
        std::vector<uint32_t> hi(nq * k, MSE_ID_NONE);
        MSE_HIP_TRY(hipMemcpyAsync(out_scores, hs.data(), hs.size() * 8, hipMemcpyHostToDevice, s->stream));
        MSE_HIP_TRY(hipMemcpyAsync(out_ids, hi.data(), hi.size() * 4, hipMemcpyHostToDevice, s->stream));
        MSE_HIP_TRY(hipStreamSynchronize(s->stream));
        return 0;
    }
    if (mode == MSE_MODE_EXACT) {
        for (size_t q0 = 0; q0 < nq; q0 += 8) {
            const int nqp = (int)std::min<size_t>(8, nq - q0);
            if (s->q_stage.ensure((size_t)8 * d * 2)) return -1;
            MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, (size_t)8 * d * 2, s->stream));
            MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, q + q0 * d, (size_t)nqp * d * 2, hipMemcpyDeviceToDevice, s->stream));
            if (exact_pass(s, nqp, (int)k, id_offset, out_scores + q0 * k, out_ids + q0 * k, k)) return -1;
        }
        return 0;
    }
    if (mode == MSE_MODE_MFMA) {
        const size_t tile = mfma_call_tile(b, k);
        for (size_t q0 = 0; q0 < nq; q0 += tile) {
            const int nqp = (int)std::min<size_t>(tile, nq - q0);
            if (mfma_pass(s, q + q0 * d, nqp, (int)k, id_offset, out_scores + q0 * k, out_ids + q0 * k, k)) return -1;
        }
        return 0;
    }
    return fail("unknown mode");
}

// test hook: the raw output of the matrix-core scan, so that its deviation from the exact-order scores can be MEASURED
// (tests/test_gpu_bruteforce.py) instead of assumed: out[g][q] = max over rows 32g .. 32g+31 of the MFMA score of query q
int mse_debug_mfma_group_max(mse_searcher* s, const uint16_t* queries, size_t nq, float* out) {
    if (!s || !s->base) return fail("null searcher");
    const mse_base* b = s->base;
    if (nq == 0 || nq > (size_t)mfma_query_tile((int)b->d) || b->n == 0) return fail("mfma_group_max: 1..320 queries (256 when d / 64 is odd), non-empty base");
    const int d = (int)b->d;
    const int nq_pad = mfma_pad((int)nq, d);
    const size_t n_groups = (b->n + GROUP_ROWS - 1) / GROUP_ROWS;
    if (s->q_stage.ensure((size_t)nq_pad * d * 2) || s->gmax.ensure(n_groups * (size_t)nq_pad * 4) ||
        s->qpacked.ensure(mfma_packed_bytes(d))) return -1;
    MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, (size_t)nq_pad * d * 2, s->stream));
    MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, queries, nq * d * 2, hipMemcpyHostToDevice, s->stream));
    if (launch_scan_mfma(b->dev, b->n, d, s->q_stage.as<uint16_t>(), nq_pad, s->qpacked.p, s->gmax.as<float>(), s->n_cu, s->stream))
        return -1;
    MSE_HIP_TRY(hipMemcpy2DAsync(out, nq * 4, s->gmax.p, (size_t)nq_pad * 4, nq * 4, n_groups, hipMemcpyDeviceToHost, s->stream));
    MSE_HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

int mse_bruteforce_topk_f16(mse_searcher* s, const uint16_t* queries, size_t nq, size_t k, int mode, int64_t* scores,
                            uint32_t* ids) {
    if (!s) return fail("null searcher");
    if (nq == 0 || k == 0) return 0;
    if (mode == MSE_MODE_AUTO && s->base && nq <= (size_t)mfma_query_tile((int)s->base->d)) {
        // The reference's call shape is a thread per core, each with its own Scratch and ONE query per request
        // (src/query_disk_index.rs:711-736): such callers meet in the base's coalescer and share a pass over the rows.
        // Answers are those of every other mode; a lone caller fires its pass at once (dispatch.h).  Only requests that fit one pass
        // go there: a larger batch fills passes on its own and stays on the caller's searcher (its stream, its timing, its
        // last_stats).  If the coalescer cannot be made (no memory for its worker's scratch) the call is answered directly as well.
        const mse_base* b = s->base;
        mse_dispatcher* disp = nullptr;
        {
            std::lock_guard<std::mutex> g(b->disp_mu);
            if (!b->disp && !b->disp_failed) {
                b->disp = mse_dispatcher_new(b, 0, 0);
                if (!b->disp) b->disp_failed = true;
            }
            disp = b->disp;
        }
        if (disp) return mse_dispatcher_topk_f16(disp, queries, nq, k, scores, ids);
    }
    const size_t d = s->base->d;
    DevBuf qd;
    if (qd.ensure(nq * d * 2)) return -1;
    if (s->out_scores.ensure(nq * k * 8) || s->out_ids.ensure(nq * k * 4)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(qd.p, queries, nq * d * 2, hipMemcpyHostToDevice, s->stream));
    if (mse_bruteforce_topk_f16_dev(s, qd.p, nq, k, mode, 0, s->out_scores.p, s->out_ids.p)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(scores, s->out_scores.p, nq * k * 8, hipMemcpyDeviceToHost, s->stream));
    MSE_HIP_TRY(hipMemcpyAsync(ids, s->out_ids.p, nq * k * 4, hipMemcpyDeviceToHost, s->stream));
    MSE_HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

int mse_merge_topk_dev(mse_searcher* s, const void* gathered_scores_dev, const void* gathered_ids_dev,
                       size_t n_shards, size_t nq, size_t k, void* out_scores_dev, void* out_ids_dev) {
    if (!s) return fail("null searcher");
    if (nq == 0 || k == 0 || n_shards == 0) return 0;
    if (k > (size_t)TOPK_KMAX) return fail("k too large");
    if (s->misc.ensure(nq * k * 4) || s->sel_keys.ensure(nq * k * 8)) return -1;
    SelectArgs a{};
    a.kind = KEY_I64;
    a.list_ids = reinterpret_cast<const uint32_t*>(gathered_ids_dev);
    a.list_keys = gathered_scores_dev;
    a.list_stride = k;                 // query q starts k records into each shard block
    a.list_chunk = k;
    a.list_chunk_stride = nq * k;      // next shard
    a.n_list = n_shards * k;
    a.k = (int)k; a.out_ids = s->misc.as<uint32_t>(); a.out_keys = s->sel_keys.p; a.out_stride = k; a.nq = (int)nq;
    if (launch_select(a, s->stream)) return -1;
    return launch_finalize(s->misc.as<uint32_t>(), s->sel_keys.as<int64_t>(), k, (int)k, (int)nq, 0,
                           reinterpret_cast<int64_t*>(out_scores_dev), reinterpret_cast<uint32_t*>(out_ids_dev), k,
                           nullptr, 0, 0, 0, nullptr, nullptr, s->stream);
}

int mse_bruteforce_scores_f16(mse_searcher* s, const uint16_t* query, int64_t* scores) {
    if (!s) return fail("null searcher");
    const mse_base* b = s->base;
    if (b->n == 0) return 0;
    const size_t d = b->d;
    if (s->q_stage.ensure(8 * d * 2) || s->scores.ensure(b->n * 8)) return -1;
    MSE_HIP_TRY(hipMemsetAsync(s->q_stage.p, 0, 8 * d * 2, s->stream));
    MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, query, d * 2, hipMemcpyHostToDevice, s->stream));
    if (launch_scan_exact(b->dev, b->n, (int)d, s->q_stage.p, 1, false, s->scores.as<int64_t>(), b->n, nullptr, s->n_cu,
                          s->stream)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(scores, s->scores.p, b->n * 8, hipMemcpyDeviceToHost, s->stream));
    MSE_HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

int mse_score_rows_f16(mse_searcher* s, const uint32_t* ids, size_t n_ids, const uint16_t* query, int64_t* out) {
    if (!s) return fail("null searcher");
    if (n_ids == 0) return 0;
    const mse_base* b = s->base;
    const size_t d = b->d;
    if (s->q_stage.ensure(8 * d * 2) || s->cand_ids.ensure(n_ids * 4) || s->cand_scores.ensure(n_ids * 8)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, query, d * 2, hipMemcpyHostToDevice, s->stream));
    MSE_HIP_TRY(hipMemcpyAsync(s->cand_ids.p, ids, n_ids * 4, hipMemcpyHostToDevice, s->stream));
    if (launch_score_rows(b->dev, b->n, (int)d, s->q_stage.p, false, s->cand_ids.as<uint32_t>(), n_ids, n_ids,
                          s->cand_scores.as<int64_t>(), nullptr, s->stream)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(out, s->cand_scores.p, n_ids * 8, hipMemcpyDeviceToHost, s->stream));
    MSE_HIP_TRY(hipStreamSynchronize(s->stream));
    return 0;
}

}  // extern "C"
