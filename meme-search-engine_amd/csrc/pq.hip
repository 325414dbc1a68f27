// OPQ/PQ codec kernels: diskann::vector::ProductQuantizer (diskann/src/vector.rs:308-406), the
// descriptor bias of the disk search (src/query_disk_index.rs:135-142) and rank counting for the
// evaluator (src/query_disk_index.rs:271-273).
#include "common.h"
#include "kernels.h"

namespace mse {
namespace {

// apply_transform (vector.rs:320-329): out[j][i] = sum_k T[i][k] * x[j][k].
// The reference delegates to matrixmultiply::sgemm, whose blocking (and so its summation order) is
// not restated by the reference: "parity unpinned".  This kernel and the oracle both accumulate k
// ascending with one fused multiply-add per term, so they agree bit-for-bit with each other.
constexpr int TT = 16, TK = 32;
__global__ __launch_bounds__(TT* TT) void pq_transform_kernel(const float* __restrict__ T, int d,
                                                              const float* __restrict__ x, size_t n,
                                                              float* __restrict__ out) {
    __shared__ float Ts[TT][TK + 1];
    __shared__ float Xs[TT][TK + 1];
    const int ti = threadIdx.x % TT, tj = threadIdx.x / TT;
    const size_t i0 = (size_t)blockIdx.x * TT, j0 = (size_t)blockIdx.y * TT;
    float acc = 0.0f;
    for (int k0 = 0; k0 < d; k0 += TK) {
        for (int e = threadIdx.x; e < TT * TK; e += TT * TT) {
            const int r = e / TK, c = e % TK;
            Ts[r][c] = (i0 + r < (size_t)d && k0 + c < d) ? T[(i0 + r) * d + k0 + c] : 0.0f;
            Xs[r][c] = (j0 + r < n && k0 + c < d) ? x[(j0 + r) * d + k0 + c] : 0.0f;
        }
        __syncthreads();
        const int kmax = d - k0 < TK ? d - k0 : TK;
        for (int c = 0; c < kmax; c++) acc = fmaf(Ts[ti][c], Xs[tj][c], acc);
        __syncthreads();
    }
    if (i0 + ti < (size_t)d && j0 + tj < n) out[(j0 + tj) * d + i0 + ti] = acc;
}

// preprocess_query table (vector.rs:373-381): lut[i*C + j] = (f32) sum_u t[i*dpc+u] * c_j[i*dpc+u]
// with the sum carried in f64 (simsimd's f32 dot returns f64; its lane order is CPU dependent, so
// "parity unpinned"; products of two f32 are exact in f64, hence index-order f64 adds here).
__global__ void pq_lut_kernel(const float* __restrict__ centroids, int n_centroids, int d, int dpc,
                              const float* __restrict__ t, float* __restrict__ lut) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_chunks = d / dpc;
    if (idx >= n_chunks * n_centroids) return;
    const int i = idx / n_centroids, j = idx % n_centroids;
    double s = 0.0;
    for (int u = 0; u < dpc; u++) s += (double)t[i * dpc + u] * (double)centroids[(size_t)j * d + i * dpc + u];
    lut[idx] = (float)s;
}

// quantize_batch (vector.rs:345-361): per sub-space the centroid with the largest inner product,
// strict `>` from -inf so the FIRST maximum wins.
__global__ void pq_quantize_kernel(const float* __restrict__ centroids, int n_centroids, int d, int dpc,
                                   const float* __restrict__ t, size_t n, uint8_t* __restrict__ codes) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n_chunks = d / dpc;
    if (idx >= n * (size_t)n_chunks) return;
    const size_t v = idx / n_chunks;
    const int i = (int)(idx % n_chunks);
    const float* tv = t + v * d + i * dpc;
    float best = -__builtin_inff();
    int code = 0;
    for (int c = 0; c < n_centroids; c++) {
        const float* cv = centroids + (size_t)c * d + i * dpc;
        float s = 0.0f;
        for (int u = 0; u < dpc; u++) s = fmaf(tv[u], cv[u], s);
        if (s > best) { best = s; code = c; }
    }
    codes[idx] = (uint8_t)code;
}

// asymmetric_dot_product (vector.rs:387-405) with the table in LDS: per vector, s = 0; for chunk i
// ascending: s += lut[i][code_i] (plain fp32 adds, the reference's order), then `(s * 2^32) as i64`;
// optional descriptor bias added afterwards in i64 (src/query_disk_index.rs:135-142,202).
// ids == nullptr: vector p is row p of `codes` (full scan); otherwise row ids[p] (gathered).
__global__ __launch_bounds__(256) void pq_adc_kernel(const float* __restrict__ lut, int n_chunks, int n_centroids,
                                                     const uint8_t* __restrict__ codes, size_t n_codes,
                                                     const uint32_t* __restrict__ ids, size_t n,
                                                     const uint8_t* __restrict__ desc, int n_desc,
                                                     const float* __restrict__ scales, int64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_lut[];
    const int lut_n = n_chunks * n_centroids;
    for (int e = threadIdx.x; e < lut_n; e += blockDim.x) s_lut[e] = lut[e];
    __syncthreads();
    const bool vec16 = (n_chunks % 16) == 0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        size_t row = ids ? (size_t)ids[p] : p;
        const bool ok = row < n_codes;
        if (!ok) row = 0;
        const uint8_t* cp = codes + row * (size_t)n_chunks;
        float s = 0.0f;
        if (vec16) {
            for (int i0 = 0; i0 < n_chunks; i0 += 16) {
                const uint4 w = *reinterpret_cast<const uint4*>(cp + i0);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) {
                        const int i = i0 + a * 4 + bb;
                        s = add_rn(s, s_lut[i * n_centroids + ((ww[a] >> (8 * bb)) & 0xff)]);
                    }
            }
        } else {
            for (int i = 0; i < n_chunks; i++) s = add_rn(s, s_lut[i * n_centroids + cp[i]]);
        }
        int64_t r = scale_dot_result(s);
        if (desc && scales) {
            for (int j = 0; j < n_desc; j++) r += scale_dot_result(scales[j] * (float)desc[row * (size_t)n_desc + j]);
        }
        out[p] = ok ? r : INT64_MIN;
    }
}

// out[p] += descriptor_product(scales, ids[p])   (exact re-score path, query_disk_index.rs:169-170)
__global__ void add_descriptor_kernel(const uint32_t* __restrict__ ids, size_t n, const uint8_t* __restrict__ desc,
                                      int n_desc, size_t n_codes, const float* __restrict__ scales,
                                      int64_t* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t id = ids[p];
    if (id == ID_NONE || id >= n_codes) return;
    int64_t r = 0;
    for (int j = 0; j < n_desc; j++) r += scale_dot_result(scales[j] * (float)desc[(size_t)id * n_desc + j]);
    out[p] += r;
}

__global__ void f32_to_f16_kernel(const float* __restrict__ in, size_t n, uint16_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const _Float16 h = (_Float16)in[i];  // v_cvt_f16_f32: round to nearest even (half::f16::from_f32)
        out[i] = __builtin_bit_cast(uint16_t, h);
    }
}

// rank of each target row in the (score desc, id asc) order = number of rows that precede it
constexpr int RANK_MAX_TARGETS = 1024;
__global__ __launch_bounds__(256) void rank_kernel(const int64_t* __restrict__ scores, size_t n,
                                                   const uint32_t* __restrict__ targets, int m,
                                                   unsigned long long* __restrict__ counts) {
    __shared__ int64_t t_score[RANK_MAX_TARGETS];
    __shared__ uint32_t t_id[RANK_MAX_TARGETS];
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        const uint32_t id = targets[j];
        t_id[j] = id;
        t_score[j] = id < n ? scores[id] : INT64_MAX;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t n_round = (n + 63) / 64 * 64;
    for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_round;
         row += (size_t)gridDim.x * blockDim.x) {
        const bool valid = row < n;
        const int64_t s = valid ? scores[row] : INT64_MIN;
        for (int j = 0; j < m; j++) {
            const bool before = valid && (s > t_score[j] || (s == t_score[j] && row < (size_t)t_id[j]));
            const unsigned long long mask = __ballot(before);
            if (lane == 0 && mask) atomicAdd(&counts[j], (unsigned long long)__popcll(mask));
        }
    }
}

}  // namespace

int launch_pq_transform(const float* T, int d, const float* x, size_t n, float* out, hipStream_t stream) {
    if (n == 0) return 0;
    dim3 grid((d + TT - 1) / TT, (unsigned)((n + TT - 1) / TT));
    hipLaunchKernelGGL(pq_transform_kernel, grid, dim3(TT * TT), 0, stream, T, d, x, n, out);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_pq_lut(const float* centroids, int n_centroids, int d, int dpc, const float* t, float* lut,
                  hipStream_t stream) {
    const int total = (d / dpc) * n_centroids;
    hipLaunchKernelGGL(pq_lut_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, centroids, n_centroids, d, dpc, t,
                       lut);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_pq_quantize(const float* centroids, int n_centroids, int d, int dpc, const float* t, size_t n,
                       uint8_t* codes, hipStream_t stream) {
    const size_t total = n * (size_t)(d / dpc);
    if (total == 0) return 0;
    hipLaunchKernelGGL(pq_quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, centroids,
                       n_centroids, d, dpc, t, n, codes);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_pq_adc(const float* lut, int n_chunks, int n_centroids, const uint8_t* codes, size_t n_codes,
                  const uint32_t* ids, size_t n, const uint8_t* desc, int n_desc, const float* scales, int64_t* out,
                  int n_cu, hipStream_t stream) {
    if (n == 0) return 0;
    const size_t lds = (size_t)n_chunks * n_centroids * 4;
    if (lds > 160 * 1024) return fail("PQ table does not fit LDS");
    if (lds > 64 * 1024) {
        MSE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pq_adc_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    size_t blocks = (n + 255) / 256;
    const size_t cap = (size_t)n_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(pq_adc_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, lut, n_chunks, n_centroids, codes,
                       n_codes, ids, n, desc, n_desc, scales, out);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_add_descriptor(const uint32_t* ids, size_t n, const uint8_t* desc, int n_desc, size_t n_codes,
                          const float* scales, int64_t* out, hipStream_t stream) {
    if (n == 0 || !desc || !scales) return 0;
    hipLaunchKernelGGL(add_descriptor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ids, n, desc,
                       n_desc, n_codes, scales, out);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_f32_to_f16(const float* in, size_t n, uint16_t* out, hipStream_t stream) {
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, n, out);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int rank_max_targets() { return RANK_MAX_TARGETS; }
int launch_rank(const int64_t* scores, size_t n, const uint32_t* targets, int m, unsigned long long* counts, int n_cu,
                hipStream_t stream) {
    if (m == 0 || n == 0) return 0;
    if (m > RANK_MAX_TARGETS) return fail("rank: too many targets per call");
    size_t blocks = (n + 255) / 256;
    const size_t cap = (size_t)n_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rank_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, scores, n, targets, m, counts);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mse
