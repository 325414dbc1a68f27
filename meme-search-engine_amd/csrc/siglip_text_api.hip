// C ABI of the SigLIP text tower: `model.encode_text(tokens)` of clip_server.py:98 followed by the
// normalisation (:99) and fp16 serialisation (:166).  The reference holds no restatement of this tower (it is
// open_clip's TextTransformer, third-party and absent): known from the repository are the output width
// (model.text.text_projection.out_features, clip_server.py:107,182) and the constants of
// misc/clip_accursed.py:31-55 (width 1152, 27 layers, context 64, vocabulary 32000, pad id 1).  Published
// architecture: token + positional embedding, pre-LN blocks WITHOUT causal mask, final LayerNorm, the last
// position pooled, Linear projection with bias.  Weight names follow open_clip (`text.*`).  The blocks run on
// the same kernels as the image tower (siglip_kernels.hip).  Tokenisation stays on the host (Python).
#include "../../include/mse.h"
#include "runtime.h"
#include "siglip.h"
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace mse;
using namespace mse::siglip;

namespace {

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }
// rows of a part (a range of sequences on one stream) from which the LayerNorm-fused batch kernels are used: below, launch_gemm picks
// the small-batch tiles (<= 3072 rows), which have no fused form
constexpr int FUSED_MIN_ROWS = 3072;

struct TBlock {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    uint16_t *wqkv, *wproj, *w1, *w2;
    float *bqkv, *bproj, *b1, *b2;
    // LayerNorm folded into the GEMMs around it (large batches; siglip_kernels.hip "Fused LayerNorm"): fp16 gamma-folded weights,
    // their row sums and the beta-folded biases of the two consumers (QKV, fc1)
    uint16_t *wqkv16 = nullptr, *w116 = nullptr;
    float *cqkv = nullptr, *bqkv2 = nullptr, *c1 = nullptr, *b12 = nullptr;
};
struct TSlot {
    bool bf16;
    void* dst;
    size_t rows, cols, rows_pad, cols_pad;
    bool loaded = false;
};

}  // namespace

struct mse_siglip_text {
    mse_siglip_text_config cfg{};
    int D = 0, H = 0, dh = 0, mlp = 0, mlp_pad = 0, ctx = 0, n_pad = 0, dh_pad = 96, dv_pad = 80;
    int max_batch = 0;
    size_t m_pad = 0;
    static constexpr int MAX_PARTS = 4;
    hipStream_t stream = nullptr;
    hipStream_t part_s[MAX_PARTS] = {};                // streams of the parts of a large batch beyond the first (index 0 unused)
    hipEvent_t ev_fork = nullptr, part_join[MAX_PARTS] = {};
    int n_parts = 2;                                   // parts a batch of >= 32 texts runs as (MSE_SIGLIP_TEXT_PARTS)
    hipStream_t side[MAX_PARTS] = {};                  // per part: where the 128-column remainder launches of its GEMMs run
    hipEvent_t side_ev[MAX_PARTS][2] = {};
    std::mutex call_mu;   // one call at a time: token upload, kernels and scratch of a call share one stream (see mse_siglip)
    std::vector<void*> allocs;
    std::map<std::string, TSlot> slots;
    bool finalized = false;
    float *tok_emb = nullptr, *pos = nullptr, *lnf_g = nullptr, *lnf_b = nullptr, *bproj = nullptr;
    uint16_t* wproj = nullptr;
    std::vector<TBlock> blocks;
    int64_t* tokens_dev = nullptr;
    uint16_t* x = nullptr;   // residual stream [M][D], fp16
    float *pooled = nullptr, *feat = nullptr, *out_f32 = nullptr;
    uint16_t *h = nullptr, *dlt = nullptr, *mlp_h = nullptr, *qb = nullptr, *kb = nullptr, *vtb = nullptr, *out_f16 = nullptr;
    float* kparts = nullptr;   // fp32 partial sums of a K-split fc2 (few rows; launch_gemm GEMM_EPI_PART)
    // fused-LayerNorm path of large batches (round 6, as the image tower since round 2); MSE_SIGLIP_NOFUSE=1 keeps the LayerNorms
    // as kernels of their own for every batch size
    bool fused = false;
    int dp = 0;                  // width rounded up to whole 256-column tiles: rows of the proj / fc2 weights (zero rows behind D)
    float* ln_stats = nullptr;   // [m_pad] (mean, 1/std)
    float* ln_part = nullptr;    // [D / 64][m_pad] (sum, M2)
    void* sink = nullptr;
    float* stage = nullptr; size_t stage_elems = 0;

    template <typename T> T* dalloc(size_t n, bool zero = false) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n * sizeof(T), 256)) != hipSuccess) return nullptr;
        if (zero && hipMemset(p, 0, std::max<size_t>(n * sizeof(T), 256)) != hipSuccess) return nullptr;
        allocs.push_back(p);
        return reinterpret_cast<T*>(p);
    }
    void add_f32(const std::string& name, float** dst, size_t rows, size_t cols, size_t cols_pad = 0) {
        const size_t cp = cols_pad ? cols_pad : cols;
        *dst = dalloc<float>(rows * cp, true);
        slots[name] = TSlot{false, *dst, rows, cols, rows, cp};
    }
    void add_bf16(const std::string& name, uint16_t** dst, size_t rows, size_t cols, size_t rows_pad, size_t cols_pad) {
        *dst = dalloc<uint16_t>(rows_pad * cols_pad, true);
        slots[name] = TSlot{true, *dst, rows, cols, rows_pad, cols_pad};
    }
};

extern "C" {

mse_siglip_text* mse_siglip_text_create(const mse_siglip_text_config* c) {
    if (!c) { fail("null config"); return nullptr; }
    if (c->width % 128 || c->heads <= 0 || c->width / c->heads != 72 || c->context_length % 32 || c->context_length <= 0 ||
        c->vocab_size <= 0 || c->max_batch <= 0) {
        fail("siglip text: unsupported geometry (width % 128 == 0, head_dim == 72, context % 32 == 0 required)");
        return nullptr;
    }
    mse_siglip_text* m = new (std::nothrow) mse_siglip_text();
    if (!m) { fail("out of host memory"); return nullptr; }
    m->cfg = *c;
    m->D = c->width; m->H = c->heads; m->dh = m->D / m->H; m->mlp = c->mlp_dim; m->mlp_pad = (int)round_up(m->mlp, 128);
    m->ctx = c->context_length; m->n_pad = m->ctx; m->max_batch = c->max_batch;
    m->m_pad = round_up((size_t)m->max_batch * m->ctx, 256);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; fail("hipStreamCreate failed"); return nullptr; }
    {
        const char* e = getenv("MSE_SIGLIP_TEXT_PARTS");
        m->n_parts = std::min(std::max(e ? atoi(e) : 2, 1), (int)mse_siglip_text::MAX_PARTS);
    }
    if (hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); m->n_parts = 1; }
    for (int pt = 1; pt < m->n_parts; pt++)
        if (hipStreamCreateWithFlags(&m->part_s[pt], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->part_join[pt], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            m->n_parts = pt;   // fewer streams then: correct, only slower
            break;
        }
    for (int hlf = 0; hlf < m->n_parts; hlf++)
        if (hipStreamCreateWithFlags(&m->side[hlf], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->side_ev[hlf][0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->side_ev[hlf][1], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (m->side[hlf]) { (void)hipStreamDestroy(m->side[hlf]); m->side[hlf] = nullptr; }
        }
    const size_t D = m->D, MP = m->mlp_pad, DP = round_up(m->D, 256);
    m->dp = (int)DP;
    m->add_f32("text.token_embedding.weight", &m->tok_emb, c->vocab_size, D);
    m->add_f32("text.positional_embedding", &m->pos, m->ctx, D);
    m->blocks.resize(c->layers);
    for (int i = 0; i < c->layers; i++) {
        TBlock& b = m->blocks[i];
        const std::string p = "text.transformer.resblocks." + std::to_string(i) + ".";
        m->add_f32(p + "ln_1.weight", &b.ln1_g, 1, D); m->add_f32(p + "ln_1.bias", &b.ln1_b, 1, D);
        m->add_bf16(p + "attn.in_proj_weight", &b.wqkv, 3 * D, D, 3 * D, D); m->add_f32(p + "attn.in_proj_bias", &b.bqkv, 1, 3 * D);
        // proj and fc2 write the residual branch: N = D padded to whole 256-column tiles (zero weight rows) so that the fused path's
        // persistent 256 x 256 kernel covers them without a 128-column remainder launch; the unfused path reads the first D rows
        m->add_bf16(p + "attn.out_proj.weight", &b.wproj, D, D, DP, D); m->add_f32(p + "attn.out_proj.bias", &b.bproj, 1, D, DP);
        m->add_f32(p + "ln_2.weight", &b.ln2_g, 1, D); m->add_f32(p + "ln_2.bias", &b.ln2_b, 1, D);
        m->add_bf16(p + "mlp.c_fc.weight", &b.w1, m->mlp, D, MP, D); m->add_f32(p + "mlp.c_fc.bias", &b.b1, 1, m->mlp, MP);
        m->add_bf16(p + "mlp.c_proj.weight", &b.w2, D, m->mlp, DP, MP); m->add_f32(p + "mlp.c_proj.bias", &b.b2, 1, D, DP);
    }
    {
        const char* e = getenv("MSE_SIGLIP_NOFUSE");
        m->fused = !(e && atoi(e)) && gemm_fused_ok((int)m->m_pad, (int)D, (int)MP, m->H, m->dh, m->ctx, m->n_pad, 8) &&
                   m->m_pad > (size_t)FUSED_MIN_ROWS;
    }
    bool fused_alloc_ok = true;
    if (m->fused) {
        for (int i = 0; i < c->layers; i++) {
            TBlock& b = m->blocks[i];
            b.wqkv16 = m->dalloc<uint16_t>(3 * D * D); b.cqkv = m->dalloc<float>(3 * D); b.bqkv2 = m->dalloc<float>(3 * D);
            b.w116 = m->dalloc<uint16_t>(MP * D); b.c1 = m->dalloc<float>(MP); b.b12 = m->dalloc<float>(MP);
            fused_alloc_ok = fused_alloc_ok && b.wqkv16 && b.cqkv && b.bqkv2 && b.w116 && b.c1 && b.b12;
        }
        m->ln_stats = m->dalloc<float>(2 * m->m_pad, true);
        m->ln_part = m->dalloc<float>(2 * (D / 64) * m->m_pad, true);
        m->sink = m->dalloc<char>(4096, true);
        fused_alloc_ok = fused_alloc_ok && m->ln_stats && m->ln_part && m->sink;
    }
    m->add_f32("text.ln_final.weight", &m->lnf_g, 1, D); m->add_f32("text.ln_final.bias", &m->lnf_b, 1, D);
    m->add_bf16("text.text_projection.weight", &m->wproj, D, D, D, D); m->add_f32("text.text_projection.bias", &m->bproj, 1, D);
    const size_t B = m->max_batch, M = m->m_pad, BH = B * m->H;
    m->tokens_dev = m->dalloc<int64_t>(B * m->ctx);
    m->x = m->dalloc<uint16_t>(M * D, true);
    m->h = m->dalloc<uint16_t>(M * D, true);
    m->dlt = m->dalloc<uint16_t>(M * D, true);
    m->kparts = m->dalloc<float>((size_t)4 * 768 * D, true);   // K-split partial sums of fc2 for up to 12 texts (4 ranges x 768 rows)
    m->mlp_h = m->dalloc<uint16_t>(M * MP, true);
    // + 4 sequences of slack: the GEMM epilogues store the rows of the M padding (up to 255) unconditionally
    m->qb = m->dalloc<uint16_t>((BH + 4 * m->H) * m->n_pad * m->dh_pad, true);
    m->kb = m->dalloc<uint16_t>((BH + 4 * m->H) * m->n_pad * attention_k_stride(), true);
    m->vtb = m->dalloc<uint16_t>((BH + 4 * m->H) * m->dv_pad * m->n_pad, true);
    m->pooled = m->dalloc<float>(B * D); m->feat = m->dalloc<float>(B * D);
    m->out_f32 = m->dalloc<float>(B * D); m->out_f16 = m->dalloc<uint16_t>(B * D);
    bool ok = m->tokens_dev && m->x && m->h && m->dlt && m->kparts && m->mlp_h && m->qb && m->kb && m->vtb && m->pooled && m->feat && m->out_f32 && m->out_f16;
    ok = ok && fused_alloc_ok;
    for (auto& kv : m->slots) ok = ok && kv.second.dst;
    if (!ok) { mse_siglip_text_destroy(m); fail("siglip text: device allocation failed"); return nullptr; }
    (void)hipDeviceSynchronize();   // the zero fills above ran on the null stream; m->stream does not wait for it
    if (launch_vt_ones_row(m->vtb, BH + 4 * m->H, m->dh, m->dv_pad, m->n_pad, m->stream) || hipStreamSynchronize(m->stream) != hipSuccess) {
        mse_siglip_text_destroy(m);
        return nullptr;
    }
    return m;
}

void mse_siglip_text_destroy(mse_siglip_text* m) {
    if (!m) return;
    for (int hlf = 0; hlf < mse_siglip_text::MAX_PARTS; hlf++) {
        if (m->side[hlf]) { (void)hipStreamSynchronize(m->side[hlf]); (void)hipStreamDestroy(m->side[hlf]); }
        for (hipEvent_t e : m->side_ev[hlf]) if (e) (void)hipEventDestroy(e);
        if (m->part_s[hlf]) { (void)hipStreamSynchronize(m->part_s[hlf]); (void)hipStreamDestroy(m->part_s[hlf]); }
        if (m->part_join[hlf]) (void)hipEventDestroy(m->part_join[hlf]);
    }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->stage) (void)hipFree(m->stage);
    delete m;
}

int mse_siglip_text_n_weights(const mse_siglip_text* m) { return m ? (int)m->slots.size() : 0; }
const char* mse_siglip_text_weight_name(const mse_siglip_text* m, int idx) {
    if (!m || idx < 0 || idx >= (int)m->slots.size()) return nullptr;
    auto it = m->slots.begin();
    std::advance(it, idx);
    return it->first.c_str();
}

int mse_siglip_text_set_weight(mse_siglip_text* m, const char* name, const float* data, const size_t* shape, int ndim) {
    if (!m || !name || !data) return fail("siglip_text_set_weight: null argument");
    auto it = m->slots.find(name);
    if (it == m->slots.end()) return fail(std::string("siglip text: unknown weight '") + name + "'");
    TSlot& s = it->second;
    size_t total = 1;
    for (int i = 0; i < ndim; i++) total *= shape[i];
    if (total != s.rows * s.cols) return fail(std::string("siglip text: wrong size for '") + name + "'");
    if (m->stage_elems < total) {
        if (m->stage) (void)hipFree(m->stage);
        m->stage = nullptr;
        MSE_HIP_TRY(hipMalloc((void**)&m->stage, total * 4));
        m->stage_elems = total;
    }
    MSE_HIP_TRY(hipMemcpyAsync(m->stage, data, total * 4, hipMemcpyHostToDevice, m->stream));
    if (!s.bf16) {
        if (s.cols_pad == s.cols) MSE_HIP_TRY(hipMemcpyAsync(s.dst, m->stage, total * 4, hipMemcpyDeviceToDevice, m->stream));
        else MSE_HIP_TRY(hipMemcpy2DAsync(s.dst, s.cols_pad * 4, m->stage, s.cols * 4, s.cols * 4, s.rows, hipMemcpyDeviceToDevice, m->stream));
    } else if (launch_f32_to_bf16_pad(m->stage, (int)s.rows, (int)s.cols, (int)s.cols, reinterpret_cast<uint16_t*>(s.dst),
                                      (int)s.rows_pad, (int)s.cols_pad, m->stream)) {
        return -1;
    }
    MSE_HIP_TRY(hipStreamSynchronize(m->stream));
    s.loaded = true;
    m->finalized = false;
    return 0;
}

int mse_siglip_text_finalize(mse_siglip_text* m) {
    if (!m) return fail("null engine");
    for (auto& kv : m->slots)
        if (!kv.second.loaded) return fail("siglip text: weight '" + kv.first + "' was never set");
    if (m->fused) {
        for (TBlock& b : m->blocks) {
            if (launch_ln_fold(b.wqkv, 3 * m->D, m->D, b.ln1_g, b.ln1_b, b.bqkv, b.wqkv16, b.cqkv, b.bqkv2, m->stream)) return -1;
            if (launch_ln_fold(b.w1, m->mlp_pad, m->D, b.ln2_g, b.ln2_b, b.b1, b.w116, b.c1, b.b12, m->stream)) return -1;
        }
        MSE_HIP_TRY(hipStreamSynchronize(m->stream));
    }
    m->finalized = true;
    return 0;
}

}  // extern "C"

// every kernel of one forward, enqueued on the engine's stream (and its part streams, joined back): tokens up, features into
// m->out_f32 / m->out_f16 on the device.  Nothing is copied back and nothing is waited for.
static int text_forward(mse_siglip_text* m, const int64_t* tokens, int batch, int normalize) {
    hipStream_t st = m->stream;
    const mse_siglip_text_config& c = m->cfg;
    const int D = m->D, T = m->ctx, M = batch * T;
    MSE_HIP_TRY(hipMemcpyAsync(m->tokens_dev, tokens, (size_t)M * 8, hipMemcpyHostToDevice, st));
    if (launch_embed_tokens(m->tokens_dev, m->tok_emb, m->pos, c.vocab_size, T, D, M, m->x, st)) return -1;
    // The blocks over the sequences [b0, b0 + nb) on stream `ss`: rows b0 * T .. of every activation buffer, (sequence, head) matrices
    // b0 * H .. of the attention operands.  b0 * T is a multiple of 256, so the GEMMs' row padding stays inside the range's own rows
    // (or behind the last range).
    // Up to 12 texts (768 rows): fc2 (K = 4352 for 1152 columns) is split four ways along K across workgroups, its partial sums and
    // bias added by the LayerNorm that consumes the branch (siglip_kernels.hip gemm_small_ksplit).  Such a call is one range of rows.
    const int ksp = gemm_small_ksplit(M, D, m->mlp_pad);
    const int kspp = gemm_small_ksplit_short(M, D, D);   // the output projection of ONE text (the slabs share fc2's buffer: consumed in turn)
    LnDelta fc2_delta_call;   // what the LayerNorm after an fc2 adds to x (bias filled in per block; the bf16 branch per range of rows)
    if (ksp > 1) { fc2_delta_call.parts = m->kparts; fc2_delta_call.n_parts = ksp; fc2_delta_call.part_stride = (size_t)gemm_small_ksplit_rows(M) * D; fc2_delta_call.ldp = D; }
    // parts of a large batch (decided here because the fused path is chosen by the size of a part, the same for every part of a call)
    const int parts = batch >= 32 ? std::min(m->n_parts, batch / 16) : 1;
    const int per = parts > 1 ? std::max(4, (batch / parts) / 4 * 4) : batch;
    const bool fused_call = m->fused && c.layers > 0 && per * T > FUSED_MIN_ROWS;
    auto blocks = [&](hipStream_t ss, int b0, int nb, int half) -> int {
        const size_t r0 = (size_t)b0 * T;
        // large halves: the remainder launches of the N = 1152 / 3456 GEMMs beside their full column tiles (32-64 workgroups that ran
        // alone for 43 us after the 56 us of the four full tiles)
        hipStream_t sd = nb * T > 512 ? m->side[half] : nullptr;
        auto with_side = [&](GemmLaunch& g) { g.side = sd; g.ev_fork = m->side_ev[half][0]; g.ev_join = m->side_ev[half][1]; };
        const int Ms = nb * T, Msp = (int)round_up(Ms, 256);
        uint16_t *x = m->x + r0 * D, *h = m->h + r0 * D, *dlt = m->dlt + r0 * D, *mlp_h = m->mlp_h + r0 * m->mlp_pad;
        LnDelta fc2_delta = fc2_delta_call;
        if (ksp <= 1) { fc2_delta.bf16 = dlt; fc2_delta.ldd = D; }
        uint16_t* qb = m->qb + (size_t)b0 * m->H * m->n_pad * m->dh_pad;
        uint16_t* kb = m->kb + (size_t)b0 * m->H * m->n_pad * attention_k_stride();
        uint16_t* vtb = m->vtb + (size_t)b0 * m->H * m->dv_pad * m->n_pad;
        if (fused_call) {
            // Large batch: LN1 / LN2 folded into the GEMMs around them (the image tower's path, siglip_api.hip): proj / fc2 add their
            // tile to the fp16 residual stream in place and emit per-row (sum, M2) of their 64-column groups; QKV / fc1 read the
            // residual rows themselves against gamma-folded weights and correct with (mean, 1/std).  No LayerNorm pass, no bf16
            // round trip of the branch, no 128-column remainder launch behind proj / fc2 (their N is padded to 1280).
            float* ln_stats = m->ln_stats + 2 * r0;
            float* ln_part = m->ln_part + 2 * r0;
            if (launch_row_stats(x, D, D, (size_t)Msp, c.eps, ln_stats, ss)) return -1;
            for (int i = 0; i < c.layers; i++) {
                const TBlock& b = m->blocks[i];
                {
                    GemmLaunch g; g.x = x; g.w = b.wqkv16; g.bias = b.bqkv2; g.csum = b.cqkv; g.ln_stats = ln_stats;
                    g.M = Msp; g.N = 3 * D; g.K = D; g.m_valid = Ms; g.tokens = T;
                    g.q = qb; g.k = kb; g.vt = vtb; g.heads = m->H; g.dh = m->dh; g.dh_pad = m->dh_pad; g.n_pad = m->n_pad;
                    g.dv_pad = m->dv_pad; g.kdh_pad = attention_k_stride();
                    if (launch_gemm_fused(GEMM_EPI_QKV, g, ss)) return -1;
                }
                if (launch_attention(qb, kb, vtb, nb, m->H, T, m->n_pad, m->dh, m->dh_pad, m->dv_pad, h, D, T, ss)) return -1;
                {
                    GemmLaunch g; g.x = h; g.w = b.wproj; g.bias = b.bproj; g.M = Msp; g.N = m->dp; g.K = D; g.m_valid = Ms;
                    g.xres = x; g.ldr = D; g.part = ln_part; g.part_rows = m->m_pad; g.n_valid = D; g.sink = m->sink;
                    if (launch_gemm_fused(GEMM_EPI_RESID_LN, g, ss)) return -1;   // x += attention branch, statistics for LN2
                }
                if (launch_ln_finalize(ln_part, m->m_pad, D / 64, (size_t)Msp, c.eps, ln_stats, ss)) return -1;
                {
                    GemmLaunch g; g.x = x; g.w = b.w116; g.bias = b.b12; g.csum = b.c1; g.ln_stats = ln_stats;
                    g.M = Msp; g.N = m->mlp_pad; g.K = D; g.m_valid = Ms; g.out_bf16 = mlp_h; g.ldo = m->mlp_pad; g.gelu_tanh = c.gelu_tanh;
                    if (launch_gemm_fused(GEMM_EPI_GELU, g, ss)) return -1;
                }
                {
                    GemmLaunch g; g.x = mlp_h; g.w = b.w2; g.bias = b.b2; g.M = Msp; g.N = m->dp; g.K = m->mlp_pad; g.m_valid = Ms;
                    g.xres = x; g.ldr = D; g.part = ln_part; g.part_rows = m->m_pad; g.n_valid = D; g.sink = m->sink;
                    if (launch_gemm_fused(GEMM_EPI_RESID_LN, g, ss)) return -1;   // x += MLP branch, statistics for the next LN1
                }
                if (i + 1 < c.layers && launch_ln_finalize(ln_part, m->m_pad, D / 64, (size_t)Msp, c.eps, ln_stats, ss)) return -1;
            }
            return 0;
        }
        for (int i = 0; i < c.layers; i++) {
            const TBlock& b = m->blocks[i];
            // x += (fc2 output of the previous block), then LayerNorm
            LnDelta d1;
            if (i) { d1 = fc2_delta; d1.bias = m->blocks[i - 1].b2; }
            if (launch_layernorm_d(x, 1, D, d1, b.ln1_g, b.ln1_b, c.eps, D, Ms, h, D, nullptr, ss)) return -1;
            {
                GemmLaunch g; g.skinny = 1; g.x = h; g.w = b.wqkv; g.bias = b.bqkv; g.M = Msp; g.N = 3 * D; g.K = D; g.m_valid = Ms; g.tokens = T;
                g.q = qb; g.k = kb; g.vt = vtb; g.heads = m->H; g.dh = m->dh; g.dh_pad = m->dh_pad; g.n_pad = m->n_pad;
                g.dv_pad = m->dv_pad; g.kdh_pad = attention_k_stride();
                with_side(g);
                if (launch_gemm(GEMM_EPI_QKV, g, ss)) return -1;
            }
            if (launch_attention(qb, kb, vtb, nb, m->H, T, m->n_pad, m->dh, m->dh_pad, m->dv_pad, h, D, T, ss)) return -1;
            {
                GemmLaunch g; g.skinny = 1; g.x = h; g.w = b.wproj; g.bias = b.bproj; g.M = Msp; g.N = D; g.K = D; g.m_valid = Ms;
                g.out_bf16 = dlt; g.ldo = D;   // residual branch: added to x by the next LayerNorm
                with_side(g);
                if (kspp > 1) { g.kpart = m->kparts; g.kpart_stride = (size_t)gemm_small_ksplit_rows(M) * D; g.ksplit = kspp; g.ldr = D; }
                if (launch_gemm(kspp > 1 ? GEMM_EPI_PART : GEMM_EPI_BF16, g, ss)) return -1;
            }
            {   // x += attention branch (bf16, or the projection's K-split slabs + bias: one text), then LayerNorm
                LnDelta d2;
                if (kspp > 1) { d2.parts = m->kparts; d2.n_parts = kspp; d2.part_stride = (size_t)gemm_small_ksplit_rows(M) * D; d2.ldp = D; d2.bias = b.bproj; }
                else { d2.bf16 = dlt; d2.ldd = D; }
                if (launch_layernorm_d(x, 1, D, d2, b.ln2_g, b.ln2_b, c.eps, D, Ms, h, D, nullptr, ss)) return -1;
            }
            {
                GemmLaunch g; g.skinny = 1; g.x = h; g.w = b.w1; g.bias = b.b1; g.M = Msp; g.N = m->mlp_pad; g.K = D; g.m_valid = Ms;
                g.out_bf16 = mlp_h; g.ldo = m->mlp_pad; g.gelu_tanh = c.gelu_tanh;
                if (launch_gemm(GEMM_EPI_GELU, g, ss)) return -1;
            }
            {
                GemmLaunch g; g.skinny = 1; g.x = mlp_h; g.w = b.w2; g.bias = b.b2; g.M = Msp; g.N = D; g.K = m->mlp_pad; g.m_valid = Ms;
                g.out_bf16 = dlt; g.ldo = D;   // residual branch: added to x by the next LayerNorm
                with_side(g);
                if (ksp > 1) { g.kpart = m->kparts; g.kpart_stride = fc2_delta.part_stride; g.ksplit = ksp; g.ldr = D; }
                if (launch_gemm(ksp > 1 ? GEMM_EPI_PART : GEMM_EPI_BF16, g, ss)) return -1;
            }
        }
        return 0;
    };
    // A large batch runs as TWO halves on two streams (round 5, as the image tower does since round 2): the output-projection and fc2
    // GEMMs have 4.5 column tiles, so their last round of 256 x 256 tiles leaves most of the chip idle -- the other half's next kernel
    // takes those CUs.  The first half is a multiple of four sequences (256 rows).
    // (MSE_SIGLIP_TEXT_PARTS, read when the engine is created: 1..4 parts; every part but the last is a multiple of four sequences.)
    if (parts > 1) {
        MSE_HIP_TRY(hipEventRecord(m->ev_fork, st));
        for (int pt = 1; pt < parts; pt++) {
            const int b0 = pt * per, nb = pt + 1 == parts ? batch - b0 : per;
            MSE_HIP_TRY(hipStreamWaitEvent(m->part_s[pt], m->ev_fork, 0));
            if (blocks(m->part_s[pt], b0, nb, pt)) return -1;
            MSE_HIP_TRY(hipEventRecord(m->part_join[pt], m->part_s[pt]));
        }
    }
    if (blocks(st, 0, parts > 1 ? per : batch, 0)) return -1;
    for (int pt = 1; pt < parts; pt++) MSE_HIP_TRY(hipStreamWaitEvent(st, m->part_join[pt], 0));
    // final LayerNorm of the LAST position only (pool_type "last"), then the projection with bias
    {
        LnDelta df;   // rows b * T + (T - 1): row stride T * D of x, of the bf16 branch and of the partial sums alike
        if (fused_call) {
            // (x already holds the last block's MLP branch)
        } else if (c.layers && ksp > 1) {
            df = fc2_delta_call; df.parts += (size_t)(T - 1) * D; df.ldp = T * D; df.bias = m->blocks[c.layers - 1].b2;
        } else if (c.layers) {
            df.bf16 = m->dlt + (size_t)(T - 1) * D; df.ldd = T * D;
        }
        if (launch_layernorm_d(m->x + (size_t)(T - 1) * D, 1, T * D, df, m->lnf_g, m->lnf_b, c.eps, D, batch, nullptr, D, m->pooled, st)) return -1;
    }
    if (launch_small_linear(m->pooled, D, m->wproj, D, m->bproj, D, D, batch, 0, nullptr, 0, m->feat, D, st)) return -1;
    if (launch_l2norm(m->feat, D, D, batch, normalize, m->out_f32, m->out_f16, st)) return -1;
    return 0;
}

extern "C" {

int mse_siglip_text_encode(mse_siglip_text* m, const int64_t* tokens, int batch, int normalize, float* out_f32, uint16_t* out_f16) {
    if (!m || !tokens) return fail("siglip text: null engine or tokens");
    if (!m->finalized) return fail("siglip text: call mse_siglip_text_finalize after loading the weights");
    std::lock_guard<std::mutex> call_lock(m->call_mu);
    if (batch <= 0 || batch > m->max_batch) return fail("siglip text: batch exceeds max_batch");  // clip_server.py:136
    if (text_forward(m, tokens, batch, normalize)) return -1;
    hipStream_t st = m->stream;
    const int D = m->D;
    if (out_f32) MSE_HIP_TRY(hipMemcpyAsync(out_f32, m->out_f32, (size_t)batch * D * 4, hipMemcpyDeviceToHost, st));
    if (out_f16) MSE_HIP_TRY(hipMemcpyAsync(out_f16, m->out_f16, (size_t)batch * D * 2, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

// The same forward with the features LEFT ON THE DEVICE and no wait: the query path hands them to the search without a host round
// trip (src/query_disk_index.rs:345-381 embeds the text, :436-540 searches with it).  `tokens` must stay valid until the engine's
// stream has consumed them (pinned memory, or wait for the stream before reusing the buffer); the result buffers are the engine's
// own and hold this call's rows until the next call on the engine.
int mse_siglip_text_encode_dev(mse_siglip_text* m, const int64_t* tokens, int batch, int normalize) {
    if (!m || !tokens) return fail("siglip text: null engine or tokens");
    if (!m->finalized) return fail("siglip text: call mse_siglip_text_finalize after loading the weights");
    std::lock_guard<std::mutex> call_lock(m->call_mu);
    if (batch <= 0 || batch > m->max_batch) return fail("siglip text: batch exceeds max_batch");
    return text_forward(m, tokens, batch, normalize);
}
const void* mse_siglip_text_output_device(const mse_siglip_text* m, int which) {
    return m ? (which ? (const void*)m->out_f16 : (const void*)m->out_f32) : nullptr;
}
void* mse_siglip_text_stream(const mse_siglip_text* m) { return m ? (void*)m->stream : nullptr; }

}  // extern "C"
