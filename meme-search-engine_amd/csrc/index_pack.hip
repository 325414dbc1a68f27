// Index packing pieces of dump_processor, next to quantize_batch (pq.hip):
//
//   mse_score_model_*        src/score_model.rs:13-32   score_batch: down . silu(up . x^T + bias) * d_emb / d_hidden
//   mse_descriptor_buckets   src/dump_processor.rs:483-491: each score channel is inverted through its CDF by
//                            `binary_search_by(|x| x.partial_cmp(score))` and stored as one byte
//
// The reference multiplies with candle (summation order not restated: "parity unpinned") and takes silu from libm's
// expf; here every dot product is 64 interleaved fused partial sums (k = l mod 64) added in lane order and
// silu(x) = x / (1 + exp(-x)) with the device's expf, so oracle <-> HIP parity is to tolerance (1e-5 relative), not bit exact.  The bucket search is integer
// logic and is bit exact: it replays core::slice::binary_search_by (Rust 1.7x: halve `size`, keep `base`).
#include "../../include/mse.h"
#include "runtime.h"
#include <new>

using namespace mse;

struct mse_score_model {
    float *up = nullptr, *bias = nullptr, *down = nullptr;  // device
    size_t d_emb = 0, d_hidden = 0, out_ch = 0;
    DevBuf x, h, y;
    std::mutex mu;
};

namespace {

// one wave per (row b, unit n): y[b][n] = act(sum_k w[n][k] * x[b][k] + bias[n]) * scale
__global__ __launch_bounds__(256) void dense_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, int K,
                                                         const float* __restrict__ bias, int N, size_t B, int silu, float scale,
                                                         float* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const size_t wid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wid >= B * (size_t)N) return;
    const size_t b = wid / N;
    const int n = (int)(wid % N);
    const float* xr = x + b * ldx;
    const float* wr = w + (size_t)n * K;
    // lane l owns k = l, l + 64, ...; the 64 partial sums are then added in lane order (the oracle does the same)
    float part = 0.0f;
    for (int k = lane; k < K; k += 64) part = fmaf(wr[k], xr[k], part);
    float s = 0.0f;
    for (int l = 0; l < 64; l++) s += __shfl(part, l);
    if (lane == 0) {
        if (bias) s += bias[n];
        if (silu) s = s / (1.0f + expf(-s));
        y[b * ldy + n] = s * scale;
    }
}

__global__ void descriptor_buckets_kernel(const float* __restrict__ cdfs, int n_desc, int cdf_len, const float* __restrict__ scores,
                                          size_t n, uint8_t* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * (size_t)n_desc) return;
    const int j = (int)(idx % n_desc);
    const float* cdf = cdfs + (size_t)j * cdf_len;
    const float s = scores[idx];
    // core::slice::binary_search_by with cmp(x) = x.partial_cmp(&s)
    size_t size = (size_t)cdf_len, base = 0;
    if (size == 0) { out[idx] = 0; return; }
    while (size > 1) {
        const size_t half = size / 2, mid = base + half;
        base = (cdf[mid] > s) ? base : mid;
        size -= half;
    }
    const float c = cdf[base];
    const size_t r = (c == s) ? base : base + (c < s ? 1 : 0);
    out[idx] = (uint8_t)r;
}

}  // namespace

extern "C" {

mse_score_model* mse_score_model_load(const float* up_proj, const float* bias, const float* down_proj, size_t d_emb,
                                      size_t d_hidden, size_t out_channels) {
    if (!up_proj || !bias || !down_proj || !d_emb || !d_hidden || !out_channels) { fail("score_model_load: bad argument"); return nullptr; }
    mse_score_model* m = new (std::nothrow) mse_score_model();
    if (!m) { fail("out of host memory"); return nullptr; }
    m->d_emb = d_emb; m->d_hidden = d_hidden; m->out_ch = out_channels;
    bool ok = hipMalloc((void**)&m->up, d_hidden * d_emb * 4) == hipSuccess && hipMalloc((void**)&m->bias, d_hidden * 4) == hipSuccess &&
              hipMalloc((void**)&m->down, out_channels * d_hidden * 4) == hipSuccess;
    ok = ok && hipMemcpy(m->up, up_proj, d_hidden * d_emb * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(m->bias, bias, d_hidden * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(m->down, down_proj, out_channels * d_hidden * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { mse_score_model_free(m); fail("score_model_load: device allocation/copy failed"); return nullptr; }
    return m;
}

void mse_score_model_free(mse_score_model* m) {
    if (!m) return;
    if (m->up) (void)hipFree(m->up);
    if (m->bias) (void)hipFree(m->bias);
    if (m->down) (void)hipFree(m->down);
    delete m;
}

size_t mse_score_model_output_channels(const mse_score_model* m) { return m ? m->out_ch : 0; }

int mse_score_model_score_batch(mse_score_model* m, const float* input, size_t batch, float* out) {
    if (!m || !input || !out) return fail("score_batch: null argument");
    if (batch == 0) return 0;
    std::lock_guard<std::mutex> g(m->mu);
    if (m->x.ensure(batch * m->d_emb * 4) || m->h.ensure(batch * m->d_hidden * 4) || m->y.ensure(batch * m->out_ch * 4)) return -1;
    MSE_HIP_TRY(hipMemcpy(m->x.p, input, batch * m->d_emb * 4, hipMemcpyHostToDevice));
    const float scale = (float)m->d_emb / (float)m->d_hidden;   // score_model.rs:16
    size_t waves = batch * m->d_hidden;
    hipLaunchKernelGGL(dense_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, nullptr, m->x.as<float>(), (int)m->d_emb, m->up,
                       (int)m->d_emb, m->bias, (int)m->d_hidden, batch, 1, 1.0f, m->h.as<float>(), (int)m->d_hidden);
    MSE_HIP_TRY(hipGetLastError());
    waves = batch * m->out_ch;
    hipLaunchKernelGGL(dense_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, nullptr, m->h.as<float>(), (int)m->d_hidden,
                       m->down, (int)m->d_hidden, (const float*)nullptr, (int)m->out_ch, batch, 0, scale, m->y.as<float>(), (int)m->out_ch);
    MSE_HIP_TRY(hipGetLastError());
    MSE_HIP_TRY(hipMemcpy(out, m->y.p, batch * m->out_ch * 4, hipMemcpyDeviceToHost));
    return 0;
}

int mse_descriptor_buckets(const float* cdfs, size_t n_desc, size_t cdf_len, const float* scores, size_t n, uint8_t* out) {
    if (!cdfs || !scores || !out) return fail("descriptor_buckets: null argument");
    if (cdf_len > 255) return fail("descriptor_buckets: a CDF has at most 255 entries (the bucket is one byte)");
    if (n == 0 || n_desc == 0) return 0;
    DevBuf c, s, o;
    if (c.ensure(n_desc * cdf_len * 4 + 4) || s.ensure(n * n_desc * 4) || o.ensure(n * n_desc)) return -1;
    MSE_HIP_TRY(hipMemcpy(c.p, cdfs, n_desc * cdf_len * 4, hipMemcpyHostToDevice));
    MSE_HIP_TRY(hipMemcpy(s.p, scores, n * n_desc * 4, hipMemcpyHostToDevice));
    const size_t total = n * n_desc;
    hipLaunchKernelGGL(descriptor_buckets_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, c.as<float>(), (int)n_desc,
                       (int)cdf_len, s.as<float>(), n, o.as<uint8_t>());
    MSE_HIP_TRY(hipGetLastError());
    MSE_HIP_TRY(hipMemcpy(out, o.p, total, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
