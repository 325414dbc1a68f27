// Internal launch interface of the SigLIP kernels (siglip_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <algorithm>

namespace mse {
namespace siglip {

enum { GEMM_EPI_BF16 = 0, GEMM_EPI_GELU = 1, GEMM_EPI_RESID = 2, GEMM_EPI_PATCH = 3, GEMM_EPI_QKV = 4, GEMM_EPI_RESID_LN = 5, GEMM_EPI_PART = 6 };

struct GemmLaunch {
    const uint16_t* x = nullptr;   // [M][K] bf16
    const uint16_t* w = nullptr;   // [N][K] bf16
    const float* bias = nullptr;   // [N]
    int M = 0, N = 0, K = 0, m_valid = 0;
    uint16_t* out_bf16 = nullptr; int ldo = 0;
    float* resid = nullptr; int ldr = 0;
    const float* pos = nullptr; int tokens = 0;
    uint16_t *q = nullptr, *k = nullptr, *vt = nullptr;
    int heads = 0, dh = 0, dh_pad = 0, n_pad = 0, dv_pad = 0;
    int kdh_pad = 0;               // k row stride in elements (attention_k_stride()); 0 = dh_pad
    int gelu_tanh = 0;
    int skinny = 0;                // launch_gemm: few rows may take the small-batch kernels: 1 = chosen by size (K-split skinny <= 512 rows,
                                   // 64 x 64 / 128 x 128 tiles <= 3072 rows); 2 / 3 / 4 force one of them (developer timing)
    hipStream_t side = nullptr;    // launch_gemm: the 128-column remainder launch runs here, beside the full column tiles (ev_fork / ev_join order it)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // LayerNorm-fused launches (launch_gemm_fused)
    const float* ln_stats = nullptr;   // consumers (QKV, GELU): (mean, 1/std) per row of x, which is then the FP16 residual stream
    const float* csum = nullptr;       // consumers: sum_k w'[n][k]; `w` holds the fp16 gamma-folded weights, `bias` the beta-folded bias
    uint16_t* xres = nullptr;          // RESID_LN: fp16 residual stream [M][ldr], x += acc + bias in place
    float* part = nullptr;             // RESID_LN: [n_valid / 64][part_rows][2] statistics of the 64-column groups
    size_t part_rows = 0;
    int n_valid = 0;                   // RESID_LN: real output columns (the rest of N is tile padding)
    void* sink = nullptr;              // RESID_LN: >= 2 KiB of scratch
    // GEMM_EPI_PART (few rows, long K): K split `ksplit` ways across workgroups, split s stores raw fp32 sums to
    // kpart[s * kpart_stride + m * ldr + n] (no bias); the consuming LayerNorm adds them (LnDelta.parts)
    float* kpart = nullptr;
    size_t kpart_stride = 0;
    int ksplit = 0;
};

// What a LayerNorm adds to x before normalising (and writes back): the bf16 output of the preceding GEMM, or that GEMM's K-split
// partial sums (n_parts fp32 slabs, part_stride elements apart, rows of ldp) plus its bias.
struct LnDelta {
    const uint16_t* bf16 = nullptr; int ldd = 0;
    const float* parts = nullptr; int n_parts = 0; size_t part_stride = 0; int ldp = 0; const float* bias = nullptr;
};

int attention_k_stride();   // row stride (elements) launch_attention expects of the K buffer

int gemm_bm();
int gemm_bn();
int gemm_bk();
int launch_gemm(int epi, const GemmLaunch& g, hipStream_t st);
// K ranges a GEMM of `rows` rows with a LayerNorm behind it should be split into (1 = none: use the bf16 epilogue), and the rows a
// slab of its partial sums must hold (kpart_stride >= that * ldr)
int gemm_small_ksplit(int rows, int N, int K);
int gemm_small_ksplit_rows(int rows);
int gemm_small_ksplit_short(int rows, int N, int K);   // the short-K projection of one text: K ranges (1 = none)
int launch_gemm256_ablation(int abl, const GemmLaunch& g, hipStream_t st);
// LayerNorm folded into the GEMMs on either side of it (siglip_kernels.hip, "Fused LayerNorm"): epi = GEMM_EPI_QKV / GEMM_EPI_GELU
// (consumers) or GEMM_EPI_RESID_LN (producer); gemm_fused_ok says whether a geometry can run them
bool gemm_fused_ok(int M, int D, int mlp_pad, int heads, int dh, int tokens_stride, int n_pad, int m_valid);
int launch_gemm_fused(int epi, const GemmLaunch& g, hipStream_t st);
int launch_ln_fold(const uint16_t* w, int n_rows, int K, const float* gamma, const float* beta, const float* bias, uint16_t* w16, float* csum,
                   float* bias2, hipStream_t st);
int launch_row_stats(const uint16_t* x_f16, int ldx, int width, size_t rows, float eps, float* stats, hipStream_t st);
int launch_ln_finalize(const float* part, size_t part_rows, int groups, size_t rows, float eps, float* stats, hipStream_t st);
// delta (optional, bf16 [rows][ldd]): x += delta is applied and written back before normalising
// x: fp32 rows, or fp16 rows (x_is_f16: the towers' residual stream)
int launch_layernorm(void* x, int x_is_f16, int ldx, const uint16_t* delta, int ldd, const float* gamma, const float* beta, float eps,
                     int width, size_t rows, uint16_t* out, int ldo, float* out_f32, hipStream_t st);
// token rows of image b are rows b * tstride + t (t < tokens; the rest of the stride is padding)
int launch_layernorm_d(void* x, int x_is_f16, int ldx, const LnDelta& delta, const float* gamma, const float* beta, float eps,
                       int width, size_t rows, uint16_t* out, int ldo, float* out_f32, hipStream_t st);
int launch_patchify(const void* img, int is_f16, int B, int C, int H, int W, int P, int k_pad, int tstride, uint16_t* out,
                    hipStream_t st);
int launch_attention(const uint16_t* q, const uint16_t* k, const uint16_t* vt, int B, int heads, int tokens, int n_pad, int dh,
                     int dh_pad, int dv_pad, uint16_t* out, int ldo, int tstride, hipStream_t st);
// u8 [B][H][W][C] -> fp16 [B][C][H][W], x / 127.5 - 1
int launch_bmp24_to_nchw_f16(const uint8_t* in, size_t img_stride, int row_stride, const uint8_t* flags, void* out, int B, int H, int W,
                             hipStream_t st);
int launch_rgb8_to_nchw_f16(const uint8_t* in, void* out, int B, int C, int H, int W, hipStream_t st);
// must be applied once to a (zero-initialised) Vt buffer before launch_attention is used on it
int launch_vt_ones_row(uint16_t* vt, size_t n_mats, int dh, int dv_pad, int n_pad, hipStream_t st);
int launch_pool_attention(const uint16_t* kv, int ldkv, const float* qlat, int B, int heads, int dh, int tokens, int tstride,
                          float* out, int ldo, hipStream_t st);
int launch_small_linear(const float* x, int ldx, const uint16_t* w, int ldw, const float* bias, int K, int N, int B, int act,
                        const float* res, int ldres, float* y, int ldy, hipStream_t st);
int launch_l2norm(const float* x, int ldx, int width, int B, int normalize, float* out_f32, uint16_t* out_f16, hipStream_t st);
int launch_embed_tokens(const int64_t* tokens, const float* tok_emb, const float* pos, int vocab, int ctx, int D, size_t rows,
                        uint16_t* x_f16, hipStream_t st);
int launch_f32_to_bf16_pad(const float* in, int rows, int cols, int ld_in, uint16_t* out, int rows_pad, int cols_pad,
                           hipStream_t st);

}  // namespace siglip
}  // namespace mse
