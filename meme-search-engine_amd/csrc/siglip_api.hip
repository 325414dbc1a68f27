// C ABI of the SigLIP image engine: the in-process seam the reference fills with its AITemplate engines,
// `fast_image_fns[batch](images NCHW fp16 on device) -> [batch, 1152]` (clip_server.py:31,66-82,105-112),
// followed by the L2 normalisation and fp16 serialisation of do_inference / run_inference
// (clip_server.py:115,166).  Weights are addressed by the timm state-dict names the reference's loader maps
// (clip_server.py:40-57, without the "visual." prefix).
#include "../../include/mse.h"
#include "runtime.h"
#include "siglip.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace mse;
using namespace mse::siglip;

namespace {

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

struct Block {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    uint16_t *wqkv, *wproj, *w1, *w2;
    float *bqkv, *bproj, *b1, *b2;
    // LayerNorm folded into QKV / fc1 (built by mse_siglip_finalize): fp16 w * gamma, its row sums, bias + w . beta
    uint16_t *wqkv16 = nullptr, *w116 = nullptr;
    float *cqkv = nullptr, *c1 = nullptr, *bqkv2 = nullptr, *b12 = nullptr;
};

struct Slot {
    enum Kind { F32, BF16_PAD } kind;
    void* dst;
    size_t rows, cols;          // logical shape of the source (product of leading dims, last dim)
    size_t rows_pad, cols_pad;  // destination shape for BF16_PAD
    bool loaded = false;
};

}  // namespace

struct mse_siglip {
    mse_siglip_config cfg{};
    int tokens = 0, D = 0, H = 0, dh = 0, mlp = 0, mlp_pad = 0, kpe = 0, kpe_pad = 0;
    int n_pad = 0, dh_pad = 96, dv_pad = 80;
    int max_batch = 0;
    size_t m_pad = 0;
    hipStream_t stream = nullptr;
    // one call at a time per engine: a call enqueues its uploads and kernels on the engine's streams into shared scratch, so two
    // threads inside one engine would read each other's images (replicas are separate engines and run side by side)
    std::recursive_mutex call_mu;
    static constexpr int MAX_SIDE = 3;
    static constexpr int SMALL_BATCH = 4;   // images per call that still take the small-batch GEMM kernels (4 x 736 rows <= 3072)
    hipStream_t side[MAX_SIDE] = {};   // further streams of a forward pass (MSE_SIGLIP_STREAMS = 1 + how many are used; default 2)
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SIDE] = {};
    int n_side = 0;
    std::vector<void*> allocs;
    std::map<std::string, Slot> slots;
    bool finalized = false;
    // parameters
    uint16_t* wpe = nullptr; float* bpe = nullptr; float* pos = nullptr;
    std::vector<Block> blocks;
    float *lnf_g = nullptr, *lnf_b = nullptr;
    float* latent = nullptr; uint16_t *wq = nullptr, *wkv = nullptr, *wpp = nullptr, *wp1 = nullptr, *wp2 = nullptr;
    float *bq = nullptr, *bkv = nullptr, *bpp = nullptr, *lnp_g = nullptr, *lnp_b = nullptr, *bp1 = nullptr, *bp2 = nullptr;
    float* qlat = nullptr;
    // activations
    int dp = 0;              // emb_dim rounded up to whole 256-column GEMM tiles (ld of the residual-branch buffer)
    void* img_dev = nullptr;
    uint16_t *patches = nullptr, *h = nullptr, *dlt = nullptr, *mlp_h = nullptr, *qb = nullptr, *kb = nullptr, *vtb = nullptr, *kvb = nullptr;
    uint16_t* x = nullptr;   // residual stream [M][D], fp16
    float *pool_a = nullptr, *pool_o = nullptr;
    uint16_t *pool_a16 = nullptr, *pool_ln16 = nullptr, *pool_h16 = nullptr;   // bf16 operands of the MAP head's GEMMs, rows padded to 256
    size_t pool_rows = 0;
    float* out_f32 = nullptr; uint16_t* out_f16 = nullptr;
    float* stage = nullptr; size_t stage_elems = 0;
    int last_batch = 0;
    // fused LayerNorm path (siglip_kernels.hip "Fused LayerNorm"); MSE_SIGLIP_NOFUSE=1 keeps LN1 / LN2 as kernels of their own
    bool fused = false;
    bool no_small = false;       // MSE_SIGLIP_NOSMALL=1: calls of 1..4 images run the batch kernels too (bit-equal to rows of larger batches)
    float* ln_stats = nullptr;   // [m_pad] (mean, 1/std)
    float* ln_part = nullptr;    // [D / 64][m_pad] (sum, M2)
    float* kparts = nullptr;     // small-batch path: fp32 partial sums of a K-split fc2 (launch_gemm GEMM_EPI_PART)
    void* sink = nullptr;

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
        slots[name] = Slot{Slot::F32, *dst, rows, cols, rows, cp};
    }
    void add_bf16(const std::string& name, uint16_t** dst, size_t rows, size_t cols, size_t rows_pad, size_t cols_pad) {
        *dst = dalloc<uint16_t>(rows_pad * cols_pad, true);
        slots[name] = Slot{Slot::BF16_PAD, *dst, rows, cols, rows_pad, cols_pad};
    }
};

extern "C" {

mse_siglip* mse_siglip_create(const mse_siglip_config* c) {
    if (!c) { fail("null config"); return nullptr; }
    // 384 / 14 = 27 patches per side: the stride-14 convolution simply ignores the last 6 pixels
    if (c->emb_dim % 128 || c->num_heads <= 0 || c->emb_dim / c->num_heads != 72 || c->patch_size <= 0 ||
        c->img_size < c->patch_size || c->max_batch <= 0) {
        fail("siglip: unsupported geometry (emb_dim % 128 == 0, head_dim == 72 required)");
        return nullptr;
    }
    mse_siglip* m = new (std::nothrow) mse_siglip();
    if (!m) { fail("out of host memory"); return nullptr; }
    m->cfg = *c;
    const int g = c->img_size / c->patch_size;
    m->tokens = g * g; m->D = c->emb_dim; m->H = c->num_heads; m->dh = m->D / m->H; m->mlp = c->mlp_dim;
    m->mlp_pad = (int)round_up(m->mlp, 128);
    m->kpe = c->in_chans * c->patch_size * c->patch_size; m->kpe_pad = (int)round_up(m->kpe, 64);
    m->n_pad = (int)round_up(m->tokens, 32);
    m->max_batch = c->max_batch;
    // token rows of image b are rows b * n_pad + t of every [M][..] buffer: images start on 32-row boundaries, so
    // 8- and 16-token pieces never straddle two images (the transposed V scatter of the QKV epilogue needs that)
    m->m_pad = round_up((size_t)m->max_batch * m->n_pad, 256);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; fail("hipStreamCreate failed"); return nullptr; }
    {
        const char* e = getenv("MSE_SIGLIP_STREAMS");
        m->n_side = std::min(std::max((e ? atoi(e) : 2) - 1, 0), (int)mse_siglip::MAX_SIDE);
        bool ok = m->n_side == 0 || hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < m->n_side; i++)
            ok = hipStreamCreateWithFlags(&m->side[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { mse_siglip_destroy(m); fail("hipStreamCreate failed"); return nullptr; }
    }
    const size_t D = m->D, MP = m->mlp_pad, DP = round_up(D, 256);
    m->dp = (int)DP;
    // parameters (names: clip_server.py:40-57)
    m->add_bf16("trunk.patch_embed.proj.weight", &m->wpe, D, m->kpe, D, m->kpe_pad);
    m->add_f32("trunk.patch_embed.proj.bias", &m->bpe, 1, D);
    m->pos = m->dalloc<float>((size_t)m->n_pad * D, true);   // padded to the row stride (EPI_PATCH indexes it by m % n_pad)
    m->slots["trunk.pos_embed"] = Slot{Slot::F32, m->pos, (size_t)m->tokens, D, (size_t)m->tokens, D};
    m->blocks.resize(c->depth);
    for (int i = 0; i < c->depth; i++) {
        Block& b = m->blocks[i];
        const std::string p = "trunk.blocks." + std::to_string(i) + ".";
        m->add_f32(p + "norm1.weight", &b.ln1_g, 1, D); m->add_f32(p + "norm1.bias", &b.ln1_b, 1, D);
        m->add_bf16(p + "attn.qkv.weight", &b.wqkv, 3 * D, D, 3 * D, D); m->add_f32(p + "attn.qkv.bias", &b.bqkv, 1, 3 * D);
        // proj and fc2 write the residual branch: their N = D output columns are padded to whole 256-column tiles (zero weight
        // rows, ld DP) so that the persistent 256 x 256 kernel covers them without the half-efficiency 128-column remainder launch
        m->add_bf16(p + "attn.proj.weight", &b.wproj, D, D, DP, D); m->add_f32(p + "attn.proj.bias", &b.bproj, 1, D, DP);
        m->add_f32(p + "norm2.weight", &b.ln2_g, 1, D); m->add_f32(p + "norm2.bias", &b.ln2_b, 1, D);
        m->add_bf16(p + "mlp.fc1.weight", &b.w1, m->mlp, D, MP, D); m->add_f32(p + "mlp.fc1.bias", &b.b1, 1, m->mlp, MP);
        m->add_bf16(p + "mlp.fc2.weight", &b.w2, D, m->mlp, DP, MP); m->add_f32(p + "mlp.fc2.bias", &b.b2, 1, D, DP);
    }
    {
        const char* e = getenv("MSE_SIGLIP_NOSMALL");
        m->no_small = e && atoi(e);
    }
    {
        const char* e = getenv("MSE_SIGLIP_NOFUSE");
        m->fused = !(e && atoi(e)) && gemm_fused_ok((int)m->m_pad, (int)D, (int)MP, m->H, m->dh, m->n_pad, m->n_pad, 8);
    }
    bool fused_alloc_ok = true;
    if (m->fused) {
        for (int i = 0; i < c->depth; i++) {
            Block& b = m->blocks[i];
            b.wqkv16 = m->dalloc<uint16_t>(3 * D * D); b.cqkv = m->dalloc<float>(3 * D); b.bqkv2 = m->dalloc<float>(3 * D);
            b.w116 = m->dalloc<uint16_t>(MP * D); b.c1 = m->dalloc<float>(MP); b.b12 = m->dalloc<float>(MP);
            fused_alloc_ok = fused_alloc_ok && b.wqkv16 && b.cqkv && b.bqkv2 && b.w116 && b.c1 && b.b12;
        }
        m->ln_stats = m->dalloc<float>(2 * m->m_pad, true);
        m->ln_part = m->dalloc<float>(2 * (D / 64) * m->m_pad, true);
        m->sink = m->dalloc<char>(4096, true);
        fused_alloc_ok = fused_alloc_ok && m->ln_stats && m->ln_part && m->sink;
    }
    m->add_f32("trunk.norm.weight", &m->lnf_g, 1, D); m->add_f32("trunk.norm.bias", &m->lnf_b, 1, D);
    const std::string ap = "trunk.attn_pool.";
    m->add_f32(ap + "latent", &m->latent, 1, D);
    m->add_bf16(ap + "q.weight", &m->wq, D, D, D, D); m->add_f32(ap + "q.bias", &m->bq, 1, D);
    m->add_bf16(ap + "kv.weight", &m->wkv, 2 * D, D, 2 * D, D); m->add_f32(ap + "kv.bias", &m->bkv, 1, 2 * D);
    m->add_bf16(ap + "proj.weight", &m->wpp, D, D, D, D); m->add_f32(ap + "proj.bias", &m->bpp, 1, D);
    m->add_f32(ap + "norm.weight", &m->lnp_g, 1, D); m->add_f32(ap + "norm.bias", &m->lnp_b, 1, D);
    m->add_bf16(ap + "mlp.fc1.weight", &m->wp1, m->mlp, D, MP, D); m->add_f32(ap + "mlp.fc1.bias", &m->bp1, 1, m->mlp, MP);
    m->add_bf16(ap + "mlp.fc2.weight", &m->wp2, D, m->mlp, D, MP); m->add_f32(ap + "mlp.fc2.bias", &m->bp2, 1, D);
    // activations
    const size_t B = m->max_batch, M = m->m_pad, BH = B * m->H;
    m->img_dev = m->dalloc<float>(B * c->in_chans * c->img_size * c->img_size);
    m->patches = m->dalloc<uint16_t>(M * m->kpe_pad, true);
    m->x = m->dalloc<uint16_t>(M * D, true);   // residual stream, fp16
    m->h = m->dalloc<uint16_t>(M * D, true);
    m->dlt = m->dalloc<uint16_t>(M * DP, true);
    m->mlp_h = m->dalloc<uint16_t>(M * MP, true);
    m->qb = m->dalloc<uint16_t>((BH + m->H) * m->n_pad * m->dh_pad, true);   // one image of slack (rows of the M padding)
    m->kb = m->dalloc<uint16_t>((BH + m->H) * m->n_pad * attention_k_stride(), true);   // 224-byte rows (attention DMA)
    m->vtb = m->dalloc<uint16_t>((BH + m->H) * m->dv_pad * m->n_pad, true);
    m->kvb = m->dalloc<uint16_t>(M * 2 * D, true);
    m->qlat = m->dalloc<float>(D, true);
    m->kparts = m->dalloc<float>((size_t)4 * 768 * D, true);   // K-split partial sums of fc2 for ONE image (4 ranges x 768 rows)
    m->pool_rows = round_up(B, 256);
    m->pool_a = m->dalloc<float>(B * D); m->pool_o = m->dalloc<float>(m->pool_rows * D, true);
    m->pool_a16 = m->dalloc<uint16_t>(m->pool_rows * D, true); m->pool_ln16 = m->dalloc<uint16_t>(m->pool_rows * D, true);
    m->pool_h16 = m->dalloc<uint16_t>(m->pool_rows * MP, true);
    m->out_f32 = m->dalloc<float>(B * D); m->out_f16 = m->dalloc<uint16_t>(B * D);
    bool ok = m->img_dev && m->patches && m->x && m->h && m->dlt && m->mlp_h && m->qb && m->kb && m->vtb && m->kvb && m->qlat && m->kparts &&
              m->pool_a && m->pool_o && m->pool_a16 && m->pool_ln16 && m->pool_h16 && m->out_f32 && m->out_f16;
    ok = ok && fused_alloc_ok;
    for (auto& kv : m->slots) ok = ok && kv.second.dst;
    if (!ok) { mse_siglip_destroy(m); fail("siglip: device allocation failed"); return nullptr; }
    (void)hipDeviceSynchronize();   // the zero fills above ran on the null stream; m->stream does not wait for it
    if (launch_vt_ones_row(m->vtb, BH + m->H, m->dh, m->dv_pad, m->n_pad, m->stream) || hipStreamSynchronize(m->stream) != hipSuccess) {
        mse_siglip_destroy(m);
        return nullptr;
    }
    return m;
}

void mse_siglip_destroy(mse_siglip* m) {
    if (!m) return;
    for (int i = 0; i < mse_siglip::MAX_SIDE; i++) {
        if (m->side[i]) { (void)hipStreamSynchronize(m->side[i]); (void)hipStreamDestroy(m->side[i]); }
        if (m->ev_join[i]) (void)hipEventDestroy(m->ev_join[i]);
    }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->stage) (void)hipFree(m->stage);
    delete m;
}

int mse_siglip_n_weights(const mse_siglip* m) { return m ? (int)m->slots.size() : 0; }

// name of weight #idx (sorted), or NULL; lets a loader enumerate what the engine expects
const char* mse_siglip_weight_name(const mse_siglip* m, int idx) {
    if (!m || idx < 0 || idx >= (int)m->slots.size()) return nullptr;
    auto it = m->slots.begin();
    std::advance(it, idx);
    return it->first.c_str();
}

int mse_siglip_set_weight(mse_siglip* m, const char* name, const float* data, const size_t* shape, int ndim) {
    if (!m || !name || !data) return fail("siglip_set_weight: null argument");
    auto it = m->slots.find(name);
    if (it == m->slots.end()) return fail(std::string("siglip: unknown weight '") + name + "'");
    Slot& s = it->second;
    size_t total = 1;
    for (int i = 0; i < ndim; i++) total *= shape[i];
    if (total != s.rows * s.cols) return fail(std::string("siglip: wrong size for '") + name + "'");
    if (m->stage_elems < total) {
        if (m->stage) (void)hipFree(m->stage);
        m->stage = nullptr;
        MSE_HIP_TRY(hipMalloc((void**)&m->stage, total * 4));
        m->stage_elems = total;
    }
    MSE_HIP_TRY(hipMemcpyAsync(m->stage, data, total * 4, hipMemcpyHostToDevice, m->stream));
    if (s.kind == Slot::F32) {
        if (s.cols_pad == s.cols) {
            MSE_HIP_TRY(hipMemcpyAsync(s.dst, m->stage, total * 4, hipMemcpyDeviceToDevice, m->stream));
        } else {
            MSE_HIP_TRY(hipMemcpy2DAsync(s.dst, s.cols_pad * 4, m->stage, s.cols * 4, s.cols * 4, s.rows, hipMemcpyDeviceToDevice,
                                         m->stream));
        }
    } else {
        if (launch_f32_to_bf16_pad(m->stage, (int)s.rows, (int)s.cols, (int)s.cols, reinterpret_cast<uint16_t*>(s.dst),
                                   (int)s.rows_pad, (int)s.cols_pad, m->stream)) return -1;
    }
    MSE_HIP_TRY(hipStreamSynchronize(m->stream));
    s.loaded = true;
    m->finalized = false;
    return 0;
}

int mse_siglip_finalize(mse_siglip* m) {
    if (!m) return fail("null engine");
    for (auto& kv : m->slots)
        if (!kv.second.loaded) return fail("siglip: weight '" + kv.first + "' was never set");
    // the pooling query does not depend on the input: q = Linear(latent)   (model.py:94-95)
    if (launch_small_linear(m->latent, m->D, m->wq, m->D, m->bq, m->D, m->D, 1, 0, nullptr, 0, m->qlat, m->D, m->stream)) return -1;
    if (m->fused) {
        for (Block& b : m->blocks) {
            if (launch_ln_fold(b.wqkv, 3 * m->D, m->D, b.ln1_g, b.ln1_b, b.bqkv, b.wqkv16, b.cqkv, b.bqkv2, m->stream)) return -1;
            if (launch_ln_fold(b.w1, m->mlp_pad, m->D, b.ln2_g, b.ln2_b, b.b1, b.w116, b.c1, b.b12, m->stream)) return -1;
        }
    }
    MSE_HIP_TRY(hipStreamSynchronize(m->stream));
    m->finalized = true;
    return 0;
}

// Decoded RGB bytes in, features out: the ToTensor / Normalize / .half() / stack / H2D steps of the reference's
// preprocessing thread (clip_server.py:131-146) run on the device, and the PCIe copy is the u8 image (1 B/element)
int mse_siglip_encode_rgb8(mse_siglip* m, const uint8_t* rgb_hwc, int batch, int normalize, float* out_f32, uint16_t* out_f16) {
    if (!m || !rgb_hwc) return fail("null engine or image");
    std::lock_guard<std::recursive_mutex> call_lock(m->call_mu);
    if (batch <= 0 || batch > m->max_batch) return fail("siglip: batch exceeds max_batch");
    const mse_siglip_config& c = m->cfg;
    const size_t elems = (size_t)batch * c.in_chans * c.img_size * c.img_size;
    uint8_t* u8 = reinterpret_cast<uint8_t*>(m->img_dev) + 2 * (size_t)m->max_batch * c.in_chans * c.img_size * c.img_size;
    MSE_HIP_TRY(hipMemcpyAsync(u8, rgb_hwc, elems, hipMemcpyHostToDevice, m->stream));
    if (launch_rgb8_to_nchw_f16(u8, m->img_dev, batch, c.in_chans, c.img_size, c.img_size, m->stream)) return -1;
    return mse_siglip_encode_image(m, m->img_dev, 1, 1, batch, normalize, out_f32, out_f16);
}

// BITMAPFILEHEADER (14 bytes) + BITMAPINFOHEADER or one of its longer successors: what `image::codecs::bmp::BmpEncoder`
// writes for Rgb8 (54-byte header, bottom-up rows) and what PIL writes.  Anything else (palettes, RLE, 32 bits, OS/2 headers)
// is not "plain" and goes through the host decoder instead.
int mse_bmp24_info(const uint8_t* data, size_t size, uint32_t* width, uint32_t* height, uint32_t* pixel_offset, int* bottom_up) {
    auto u16 = [&](size_t o) { return (uint32_t)data[o] | ((uint32_t)data[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (!data || size < 54 || data[0] != 'B' || data[1] != 'M') return fail("not a BMP file");
    const uint32_t off = u32(10), dib = u32(14);
    if (dib < 40 || 14 + (size_t)dib > size) return fail("BMP: unsupported header");
    const int32_t w = (int32_t)u32(18), h = (int32_t)u32(22);
    if (u16(26) != 1 || u16(28) != 24 || u32(30) != 0) return fail("BMP: not uncompressed 24-bit");
    if (w <= 0 || h == 0 || w > 65535 || h > 65535 || h < -65535) return fail("BMP: bad dimensions");
    const uint32_t ah = (uint32_t)(h < 0 ? -h : h);
    const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
    if (off < 14 + dib || (size_t)off + stride * ah > size) return fail("BMP: pixel array exceeds the file");
    if (width) *width = (uint32_t)w;
    if (height) *height = ah;
    if (pixel_offset) *pixel_offset = off;
    if (bottom_up) *bottom_up = h > 0;
    return 0;
}

// Request bytes in, features out: header check on the host (54 bytes), the pixel arrays go up as they are and the device does
// BGR -> RGB, the row flip, ToTensor / Normalize / .half() (the whole of clip_server.py:131-146 for the client's own format).
int mse_siglip_encode_bmp(mse_siglip* m, const uint8_t* const* bmps, const size_t* sizes, int batch, int normalize, float* out_f32,
                          uint16_t* out_f16) {
    if (!m || !bmps || !sizes) return fail("null engine or image list");
    std::lock_guard<std::recursive_mutex> call_lock(m->call_mu);
    if (batch <= 0 || batch > m->max_batch) return fail("siglip: batch exceeds max_batch");
    const mse_siglip_config& c = m->cfg;
    if (c.in_chans != 3) return fail("siglip: BMP input needs a 3-channel model");
    const int S = c.img_size;
    const size_t row_stride = ((size_t)S * 3 + 3) & ~(size_t)3, img_stride = row_stride * S;
    // staging: behind the fp16 image (2 bytes per element of a 4-byte-per-element allocation), flags after the pixels
    uint8_t* u8 = reinterpret_cast<uint8_t*>(m->img_dev) + 2 * (size_t)m->max_batch * 3 * S * S;
    const size_t room = 2 * (size_t)m->max_batch * 3 * S * S;
    if ((size_t)batch * img_stride + batch > room) return fail("siglip: BMP staging does not fit");
    std::vector<uint8_t> flags(batch);
    for (int b = 0; b < batch; b++) {
        uint32_t w = 0, h = 0, off = 0;
        int bu = 0;
        if (mse_bmp24_info(bmps[b], sizes[b], &w, &h, &off, &bu)) return -1;
        if ((int)w != S || (int)h != S) return fail("BMP: image is not " + std::to_string(S) + " x " + std::to_string(S));
        flags[b] = (uint8_t)bu;
        MSE_HIP_TRY(hipMemcpyAsync(u8 + (size_t)b * img_stride, bmps[b] + off, img_stride, hipMemcpyHostToDevice, m->stream));
    }
    uint8_t* flags_dev = u8 + (size_t)batch * img_stride;
    MSE_HIP_TRY(hipMemcpyAsync(flags_dev, flags.data(), batch, hipMemcpyHostToDevice, m->stream));
    MSE_HIP_TRY(hipStreamSynchronize(m->stream));   // `flags` is a stack-owned source
    if (launch_bmp24_to_nchw_f16(u8, img_stride, (int)row_stride, flags_dev, m->img_dev, batch, S, S, m->stream)) return -1;
    return mse_siglip_encode_image(m, m->img_dev, 1, 1, batch, normalize, out_f32, out_f16);
}

int mse_siglip_encode_image(mse_siglip* m, const void* images, int dtype, int on_device, int batch, int normalize,
                            float* out_f32, uint16_t* out_f16) {
    if (!m) return fail("null engine");
    if (!m->finalized) return fail("siglip: call mse_siglip_finalize after loading the weights");
    std::lock_guard<std::recursive_mutex> call_lock(m->call_mu);
    if (batch <= 0 || batch > m->max_batch) return fail("siglip: batch exceeds max_batch");  // clip_server.py:139
    if (dtype != 0 && dtype != 1) return fail("siglip: dtype must be 0 (f32) or 1 (f16)");
    hipStream_t st = m->stream;
    const mse_siglip_config& c = m->cfg;
    const size_t img_elems = (size_t)batch * c.in_chans * c.img_size * c.img_size;
    const void* img = images;
    if (!on_device) {
        MSE_HIP_TRY(hipMemcpyAsync(m->img_dev, images, img_elems * (dtype ? 2 : 4), hipMemcpyHostToDevice, st));
        img = m->img_dev;
    }
    const int D = m->D, T = m->tokens, TS = m->n_pad, DP = m->dp;
    const int gelu_tanh = c.gelu_tanh;
    const size_t img_stride = (size_t)c.in_chans * c.img_size * c.img_size * (dtype ? 2 : 4);
    // The trunk (patch embedding .. MAP-head pooling) of images [b0, b0 + batch) on stream st: every activation buffer is indexed by
    // image or by token row b * n_pad + t, so a sub-batch that starts at a multiple of 8 images (= 23 whole 256-row blocks) is just
    // an offset into each of them.
    // Up to SMALL_BATCH images (<= 3072 token rows) take the small-batch GEMM kernels (siglip_kernels.hip "Mid-size GEMM") with
    // LayerNorm as a pass of its own: the 256-row tiles of the batch kernels put 15-200 workgroups on 256 CUs (7.6 ms for ONE image,
    // 3.3 with these).  Decided by the CALL's batch, so every row of a larger batch still runs the same arithmetic whatever its
    // sub-batch; an image encoded alone agrees with the same image inside a batch of more than SMALL_BATCH to bf16 rounding (both
    // within the oracle's tolerance), not bit for bit.  MSE_SIGLIP_NOSMALL=1 (read when the engine is created) keeps the batch kernels
    // for every size.
    const int sk = (!m->no_small && batch <= mse_siglip::SMALL_BATCH) ? 1 : 0;
    auto trunk = [&](int b0, int batch, hipStream_t st) -> int {
        const size_t r0 = (size_t)b0 * TS, bh0 = (size_t)b0 * m->H;
        const void* v_img = reinterpret_cast<const char*>(img) + (size_t)b0 * img_stride;
        uint16_t* v_patches = m->patches + r0 * m->kpe_pad;
        uint16_t *v_x = m->x + r0 * D, *v_h = m->h + r0 * D, *v_dlt = m->dlt + r0 * DP, *v_mlp_h = m->mlp_h + r0 * m->mlp_pad;
        uint16_t* v_kvb = m->kvb + r0 * 2 * D;
        float* v_ln_stats = m->ln_stats ? m->ln_stats + 2 * r0 : nullptr;
        float* v_ln_part = m->ln_part ? m->ln_part + 2 * r0 : nullptr;
        uint16_t* v_qb = m->qb + bh0 * m->n_pad * m->dh_pad;
        uint16_t* v_kb = m->kb + bh0 * m->n_pad * attention_k_stride();
        uint16_t* v_vtb = m->vtb + bh0 * m->dv_pad * m->n_pad;
        float* v_pool_a = m->pool_a + (size_t)b0 * D;
        const int M = batch * TS;   // rows incl. the (finite, never read as keys) padding rows of every image
        const int Mp = (int)round_up(M, 256);
        // PatchEmbedder + PositionalEmbeddings (model.py:57-80,122)
        if (launch_patchify(v_img, dtype, batch, c.in_chans, c.img_size, c.img_size, c.patch_size, m->kpe_pad, TS, v_patches, st)) return -1;
        {
            GemmLaunch g; g.x = v_patches; g.w = m->wpe; g.bias = m->bpe; g.M = Mp; g.N = D; g.K = m->kpe_pad; g.m_valid = M;
            g.out_bf16 = v_x; g.ldo = D; g.ldr = D; g.pos = m->pos; g.tokens = TS;   // writes the fp16 residual stream
            g.skinny = sk;
            if (launch_gemm(GEMM_EPI_PATCH, g, st)) return -1;
        }
        const bool fused = !sk && m->fused && gemm_fused_ok(Mp, D, m->mlp_pad, m->H, m->dh, TS, m->n_pad, M);
        if (fused && launch_row_stats(v_x, D, D, (size_t)Mp, c.eps, v_ln_stats, st)) return -1;
        for (int i = 0; fused && i < c.depth; i++) {  // Encoder1DBlock (model.py:26-44) with LN1 / LN2 folded into the GEMMs around them
            const Block& b = m->blocks[i];
            {
                GemmLaunch g; g.x = v_x; g.w = b.wqkv16; g.bias = b.bqkv2; g.csum = b.cqkv; g.ln_stats = v_ln_stats;
                g.M = Mp; g.N = 3 * D; g.K = D; g.m_valid = M; g.tokens = TS;
                g.q = v_qb; g.k = v_kb; g.vt = v_vtb; g.heads = m->H; g.dh = m->dh; g.dh_pad = m->dh_pad; g.n_pad = m->n_pad;
                g.dv_pad = m->dv_pad; g.kdh_pad = attention_k_stride();
                if (launch_gemm_fused(GEMM_EPI_QKV, g, st)) return -1;
            }
            if (launch_attention(v_qb, v_kb, v_vtb, batch, m->H, T, m->n_pad, m->dh, m->dh_pad, m->dv_pad, v_h, D, TS, st)) return -1;
            {
                GemmLaunch g; g.x = v_h; g.w = b.wproj; g.bias = b.bproj; g.M = Mp; g.N = DP; g.K = D; g.m_valid = M;
                g.xres = v_x; g.ldr = D; g.part = v_ln_part; g.part_rows = m->m_pad; g.n_valid = D; g.sink = m->sink;
                if (launch_gemm_fused(GEMM_EPI_RESID_LN, g, st)) return -1;   // x += attention branch, statistics for LN2
            }
            if (launch_ln_finalize(v_ln_part, m->m_pad, D / 64, (size_t)Mp, c.eps, v_ln_stats, st)) return -1;
            {
                GemmLaunch g; g.x = v_x; g.w = b.w116; g.bias = b.b12; g.csum = b.c1; g.ln_stats = v_ln_stats;
                g.M = Mp; g.N = m->mlp_pad; g.K = D; g.m_valid = M; g.out_bf16 = v_mlp_h; g.ldo = m->mlp_pad; g.gelu_tanh = gelu_tanh;
                if (launch_gemm_fused(GEMM_EPI_GELU, g, st)) return -1;
            }
            {
                GemmLaunch g; g.x = v_mlp_h; g.w = b.w2; g.bias = b.b2; g.M = Mp; g.N = DP; g.K = m->mlp_pad; g.m_valid = M;
                g.xres = v_x; g.ldr = D; g.part = v_ln_part; g.part_rows = m->m_pad; g.n_valid = D; g.sink = m->sink;
                if (launch_gemm_fused(GEMM_EPI_RESID_LN, g, st)) return -1;   // x += MLP branch, statistics for the next LN1
            }
            if (i + 1 < c.depth && launch_ln_finalize(v_ln_part, m->m_pad, D / 64, (size_t)Mp, c.eps, v_ln_stats, st)) return -1;
        }
        // one image: fc2 (K = 4352 for 1152 columns) is split four ways along K across workgroups; its partial sums and bias are added
        // by the LayerNorm that consumes the branch (siglip_kernels.hip gemm_small_ksplit)
        const int ksp = sk ? gemm_small_ksplit(M, D, m->mlp_pad) : 1;
        LnDelta fc2_delta;   // what the LayerNorm after an fc2 adds to x (bias filled in per block)
        if (ksp > 1) { fc2_delta.parts = m->kparts; fc2_delta.n_parts = ksp; fc2_delta.part_stride = (size_t)gemm_small_ksplit_rows(M) * D; fc2_delta.ldp = D; }
        else { fc2_delta.bf16 = v_dlt; fc2_delta.ldd = DP; }
        for (int i = 0; !fused && i < c.depth; i++) {  // Encoder1DBlock (model.py:26-44)
            const Block& b = m->blocks[i];
            // x += (fc2 output of the previous block), then LayerNorm
            LnDelta d1;
            if (i) { d1 = fc2_delta; d1.bias = m->blocks[i - 1].b2; }
            if (launch_layernorm_d(v_x, 1, D, d1, b.ln1_g, b.ln1_b, c.eps, D, M, v_h, D, nullptr, st)) return -1;
            {
                GemmLaunch g; g.x = v_h; g.w = b.wqkv; g.bias = b.bqkv; g.M = Mp; g.N = 3 * D; g.K = D; g.m_valid = M; g.tokens = TS;
                g.q = v_qb; g.k = v_kb; g.vt = v_vtb; g.heads = m->H; g.dh = m->dh; g.dh_pad = m->dh_pad; g.n_pad = m->n_pad;
                g.dv_pad = m->dv_pad; g.kdh_pad = attention_k_stride(); g.skinny = sk;
                if (launch_gemm(GEMM_EPI_QKV, g, st)) return -1;
            }
            if (launch_attention(v_qb, v_kb, v_vtb, batch, m->H, T, m->n_pad, m->dh, m->dh_pad, m->dv_pad, v_h, D, TS, st)) return -1;
            {
                GemmLaunch g; g.x = v_h; g.w = b.wproj; g.bias = b.bproj; g.M = Mp; g.N = sk ? D : DP; g.K = D; g.m_valid = M;
                g.out_bf16 = v_dlt; g.ldo = DP;   // residual branch: added to x by the next LayerNorm (columns >= D are padding)
                g.skinny = sk;
                if (launch_gemm(GEMM_EPI_BF16, g, st)) return -1;
            }
            if (launch_layernorm(v_x, 1, D, v_dlt, DP, b.ln2_g, b.ln2_b, c.eps, D, M, v_h, D, nullptr, st)) return -1;   // x += attention branch
            {
                GemmLaunch g; g.x = v_h; g.w = b.w1; g.bias = b.b1; g.M = Mp; g.N = m->mlp_pad; g.K = D; g.m_valid = M;
                g.out_bf16 = v_mlp_h; g.ldo = m->mlp_pad; g.gelu_tanh = gelu_tanh; g.skinny = sk;
                if (launch_gemm(GEMM_EPI_GELU, g, st)) return -1;
            }
            {
                GemmLaunch g; g.x = v_mlp_h; g.w = b.w2; g.bias = b.b2; g.M = Mp; g.N = sk ? D : DP; g.K = m->mlp_pad; g.m_valid = M;
                g.out_bf16 = v_dlt; g.ldo = DP;   // residual branch: added to x by the next LayerNorm
                g.skinny = sk;
                if (ksp > 1) { g.kpart = m->kparts; g.kpart_stride = fc2_delta.part_stride; g.ksplit = ksp; g.ldr = D; }
                if (launch_gemm(ksp > 1 ? GEMM_EPI_PART : GEMM_EPI_BF16, g, st)) return -1;
            }
        }
        {
            LnDelta df;
            if (c.depth && !fused) { df = fc2_delta; df.bias = m->blocks[c.depth - 1].b2; }
            if (launch_layernorm_d(v_x, 1, D, df, m->lnf_g, m->lnf_b, c.eps, D, M, v_h, D, nullptr, st)) return -1;  // model.py:50,55
        }
        // MAPHead (model.py:82-111)
        {
            GemmLaunch g; g.x = v_h; g.w = m->wkv; g.bias = m->bkv; g.M = Mp; g.N = 2 * D; g.K = D; g.m_valid = M;
            g.out_bf16 = v_kvb; g.ldo = 2 * D; g.skinny = sk;
            if (launch_gemm(GEMM_EPI_BF16, g, st)) return -1;
        }
        if (launch_pool_attention(v_kvb, 2 * D, m->qlat, batch, m->H, m->dh, T, TS, v_pool_a, D, st)) return -1;
        return 0;
    };
    // Sub-batches on separate streams: a persistent GEMM's last round leaves most CUs idle (736 row blocks over 256 CUs), and another
    // sub-batch's next kernel takes them.  MSE_SIGLIP_STREAMS=1 keeps the whole batch on one stream.
    const int parts = std::min(m->n_side + 1, batch / 16);
    const int per = parts > 1 ? (int)round_up((size_t)(batch + parts - 1) / parts, 8) : batch;
    if (parts > 1 && per < batch) {
        MSE_HIP_TRY(hipEventRecord(m->ev_fork, st));
        int used = 0;
        for (int b0 = per; b0 < batch; b0 += per, used++) {
            MSE_HIP_TRY(hipStreamWaitEvent(m->side[used], m->ev_fork, 0));
            if (trunk(b0, std::min(per, batch - b0), m->side[used])) return -1;
            MSE_HIP_TRY(hipEventRecord(m->ev_join[used], m->side[used]));
        }
        if (trunk(0, per, st)) return -1;
        for (int i = 0; i < used; i++) MSE_HIP_TRY(hipStreamWaitEvent(st, m->ev_join[i], 0));
    } else if (trunk(0, batch, st)) {
        return -1;
    }
    // proj, LayerNorm, MLP with residual on the matrix cores: the batch is one (or a few) 256-row block of the same GEMM kernels
    // (rows >= batch are zero padding; the fp32 accumulating epilogue builds pool_o = proj, then pool_o += fc2)
    {
        const int Bp = (int)round_up((size_t)batch, 256);
        if (launch_f32_to_bf16_pad(m->pool_a, batch, D, D, m->pool_a16, Bp, D, st)) return -1;
        MSE_HIP_TRY(hipMemsetAsync(m->pool_o, 0, (size_t)Bp * D * 4, st));
        GemmLaunch g; g.x = m->pool_a16; g.w = m->wpp; g.bias = m->bpp; g.M = Bp; g.N = D; g.K = D; g.m_valid = batch;
        g.resid = m->pool_o; g.ldr = D; g.skinny = sk;
        if (launch_gemm(GEMM_EPI_RESID, g, st)) return -1;
        if (launch_layernorm(m->pool_o, 0, D, nullptr, 0, m->lnp_g, m->lnp_b, c.eps, D, batch, m->pool_ln16, D, nullptr, st)) return -1;
        GemmLaunch g1; g1.x = m->pool_ln16; g1.w = m->wp1; g1.bias = m->bp1; g1.M = Bp; g1.N = m->mlp_pad; g1.K = D; g1.m_valid = batch;
        g1.out_bf16 = m->pool_h16; g1.ldo = m->mlp_pad; g1.gelu_tanh = gelu_tanh; g1.skinny = sk;
        if (launch_gemm(GEMM_EPI_GELU, g1, st)) return -1;
        GemmLaunch g2; g2.x = m->pool_h16; g2.w = m->wp2; g2.bias = m->bp2; g2.M = Bp; g2.N = D; g2.K = m->mlp_pad; g2.m_valid = batch;
        g2.resid = m->pool_o; g2.ldr = D; g2.skinny = sk;
        if (launch_gemm(GEMM_EPI_RESID, g2, st)) return -1;
    }
    // features /= norm (clip_server.py:115); fp16 rows are what the server serialises (clip_server.py:166)
    if (launch_l2norm(m->pool_o, D, D, batch, normalize, m->out_f32, m->out_f16, st)) return -1;
    if (out_f32) MSE_HIP_TRY(hipMemcpyAsync(out_f32, m->out_f32, (size_t)batch * D * 4, hipMemcpyDeviceToHost, st));
    if (out_f16) MSE_HIP_TRY(hipMemcpyAsync(out_f16, m->out_f16, (size_t)batch * D * 2, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    m->last_batch = batch;
    return 0;
}

// device pointer to the [batch][emb] fp32 (which = 0) / fp16 (which = 1) result of the last call
const void* mse_siglip_output_device(const mse_siglip* m, int which) { return m ? (which ? (const void*)m->out_f16 : (const void*)m->out_f32) : nullptr; }

void* mse_siglip_stream(const mse_siglip* m) { return m ? (void*)m->stream : nullptr; }

// developer hook: average ms of the fc1-shaped GEMM (bias + GELU) over `iters` launches with ablation `abl`
namespace {
// developer data for the GEMM timing hook: bf16 values uniform in (-1, 1) from a counter hash (MSE_GEMM_RANDOM=1);
// constant operands let the chip clock higher than real activations do
__global__ void fill_random_bf16_kernel(uint16_t* p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float f = (float)(int32_t)h * (1.0f / 2147483648.0f);
        p[i] = (uint16_t)(__float_as_uint(f) >> 16);
    }
}
}  // namespace

int mse_debug_gemm_ms(int M, int N, int K, int abl, int iters, float* ms_out) {
    if (M % 256 || N % 256 || K % 64) return fail("debug gemm: M, N multiples of 256, K of 64");
    DevBuf x, w, bias, out;
    if (x.ensure((size_t)M * K * 2) || w.ensure((size_t)N * K * 2) || bias.ensure((size_t)N * 4) || out.ensure((size_t)M * N * 2)) return -1;
    MSE_HIP_TRY(hipMemset(x.p, 0x3c, (size_t)M * K * 2));   // bf16 0x3c3c ~ 0.0115
    MSE_HIP_TRY(hipMemset(w.p, 0x3c, (size_t)N * K * 2));
    MSE_HIP_TRY(hipMemset(bias.p, 0, (size_t)N * 4));
    if (MSE_DEV_KNOB("MSE_GEMM_RANDOM")) {   // developer library: operands with random mantissas
        hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(4096), dim3(256), 0, nullptr, x.as<uint16_t>(), (size_t)M * K, 1u);
        hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(4096), dim3(256), 0, nullptr, w.as<uint16_t>(), (size_t)N * K, 2u);
        MSE_HIP_TRY(hipGetLastError());
    }
    GemmLaunch g; g.x = x.as<uint16_t>(); g.w = w.as<uint16_t>(); g.bias = bias.as<float>(); g.M = M; g.N = N; g.K = K;
    g.m_valid = M; g.out_bf16 = out.as<uint16_t>(); g.ldo = N;
    hipEvent_t e0, e1;
    MSE_HIP_TRY(hipEventCreate(&e0)); MSE_HIP_TRY(hipEventCreate(&e1));
    if (launch_gemm256_ablation(abl, g, nullptr)) return -1;
    MSE_HIP_TRY(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++) if (launch_gemm256_ablation(abl, g, nullptr)) return -1;
    MSE_HIP_TRY(hipEventRecord(e1, nullptr));
    MSE_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0; MSE_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

// developer / test hook for the small-batch GEMM kernels: `rows` real rows of a [rows][K] x [N][K]^T product with the bias (epi 0) or
// bias + GELU (epi 1) epilogue, run by `variant` (0 = the large-batch kernels, 1 = chosen by size, 2 = K-split skinny, 3 = 64 x 64
// tiles, 4 = 128 x 128 tiles).  ms_out: average over `iters` launches, each with its OWN weight matrix out of a ring larger than
// the last-level cache (weights stream from HBM in a forward pass); n_diff (optional, two words): output elements that differ from
// the large-batch kernels', and those that differ by more than two bf16 steps (must be 0: the kernels differ in summation order only).
int mse_debug_gemm_small(int rows, int N, int K, int epi, int variant, int iters, float* ms_out, uint64_t* n_diff) {
    if (rows <= 0 || N % 128 || K % 64 || iters <= 0 || (epi != 0 && epi != 1) || variant < 0 || variant > 4)
        return fail("debug gemm small: rows > 0, N % 128 == 0, K % 64 == 0, epi 0 / 1, variant 0..4");
    const int M = (int)round_up((size_t)rows, 256);
    const size_t wbytes = (size_t)N * K * 2;
    const int ring = (int)std::min<size_t>(64, std::max<size_t>(1, ((size_t)768 << 20) / wbytes));
    DevBuf x, w, bias, out, ref;
    if (x.ensure((size_t)M * K * 2) || w.ensure(wbytes * ring) || bias.ensure((size_t)N * 4) || out.ensure((size_t)M * N * 2) ||
        ref.ensure((size_t)M * N * 2)) return -1;
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(4096), dim3(256), 0, nullptr, x.as<uint16_t>(), (size_t)M * K, 1u);
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3(4096), dim3(256), 0, nullptr, w.as<uint16_t>(), (size_t)N * K * ring, 2u);
    MSE_HIP_TRY(hipGetLastError());
    {
        std::vector<float> hb(N);
        for (int n = 0; n < N; n++) hb[n] = 0.01f * (float)((n * 37) % 101 - 50);
        MSE_HIP_TRY(hipMemcpy(bias.p, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    }
    MSE_HIP_TRY(hipMemset(out.p, 0, (size_t)M * N * 2));
    MSE_HIP_TRY(hipMemset(ref.p, 0, (size_t)M * N * 2));
    auto run = [&](int var, int slot, uint16_t* dst) -> int {
        GemmLaunch g; g.x = x.as<uint16_t>(); g.w = w.as<uint16_t>() + (size_t)slot * N * K; g.bias = bias.as<float>();
        g.M = M; g.N = N; g.K = K; g.m_valid = rows; g.out_bf16 = dst; g.ldo = N; g.skinny = var;
        return launch_gemm(epi ? GEMM_EPI_GELU : GEMM_EPI_BF16, g, nullptr);
    };
    if (n_diff) {
        if (run(0, 0, ref.as<uint16_t>()) || run(variant, 0, out.as<uint16_t>())) return -1;
        MSE_HIP_TRY(hipDeviceSynchronize());
        std::vector<uint16_t> ha((size_t)M * N), hb((size_t)M * N);
        MSE_HIP_TRY(hipMemcpy(ha.data(), ref.p, ha.size() * 2, hipMemcpyDeviceToHost));
        MSE_HIP_TRY(hipMemcpy(hb.data(), out.p, hb.size() * 2, hipMemcpyDeviceToHost));
        uint64_t d = 0, far = 0;
        auto f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; };
        for (size_t r = 0; r < (size_t)rows; r++)
            for (size_t n = 0; n < (size_t)N; n++) {
                const uint16_t p = ha[r * N + n], q = hb[r * N + n];
                if (p == q) continue;
                d++;
                const float x = f(p), y = f(q);
                if (!(fabsf(x - y) <= 0.0157f * std::max(fabsf(x), fabsf(y)) + 1e-3f)) far++;   // more than two bf16 steps apart (or not finite)
            }
        n_diff[0] = d;
        n_diff[1] = far;
    }
    hipEvent_t e0, e1;
    MSE_HIP_TRY(hipEventCreate(&e0)); MSE_HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < std::min(ring, 4); i++) if (run(variant, i, out.as<uint16_t>())) return -1;
    MSE_HIP_TRY(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++) if (run(variant, i % ring, out.as<uint16_t>())) return -1;
    MSE_HIP_TRY(hipEventRecord(e1, nullptr));
    MSE_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0; MSE_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

// test hook: the residual stream ([batch*tokens][emb] fp32) as left by the last forward (after the last block)
int mse_siglip_debug_residual(mse_siglip* m, float* out) {
    if (!m || !m->last_batch) return fail("siglip: no forward has run");
    std::vector<uint16_t> h16((size_t)m->last_batch * m->tokens * m->D);
    MSE_HIP_TRY(hipMemcpy2D(h16.data(), (size_t)m->tokens * m->D * 2, m->x, (size_t)m->n_pad * m->D * 2, (size_t)m->tokens * m->D * 2,
                            m->last_batch, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < h16.size(); e++) {   // fp16 -> fp32 (exact)
        _Float16 hv;
        memcpy(&hv, &h16[e], 2);
        out[e] = (float)hv;
    }
    return 0;
}

}  // extern "C"
