// Exact-order f16 dot-product kernels: bit-identical to the reference's AVX2 kernel.
//
// Reference: diskann/src/vector.rs:255-306 (fast_dot_noprefetch) / :192-252 (fast_dot).
// The reference keeps 4 accumulators x 8 lanes of fp32; element e of the vectors feeds
// accumulator (e mod 32) / 8, lane e mod 8, by one fused multiply-add per 32-element step,
// and reduces them with a fixed tree (:295-303).  Here FOUR GPU lanes share one base row:
// lane part `a` (= lane & 3) owns accumulator a, i.e. the 16-byte group [32t+8a, 32t+8a+8)
// of every step t, and runs the same sequential chain of fused multiply-adds
// (v_fma_mix_f32: exact f16->f32 widening + one rounding).  The reduction tree is replayed
// with two cross-lane exchanges.  A wave covers 16 rows; a 16-byte load per lane makes each
// row's step a contiguous 64-byte read.
//
// HBM-bound: 2*d bytes per row per pass; the queries (<= 8 per pass) live in LDS.
#include "common.h"
#include "kernels.h"

namespace mse {

namespace {

template <int QB, bool QF32>
struct QAcc {
    float v[QB][8];
};

// one 32-element step for one query: 8 chained FMAs on this lane's accumulator group
template <bool QF32>
__device__ __forceinline__ void step_fma(float (&acc)[8], const uint4& x, const char* qptr) {
    if constexpr (!QF32) {
        const uint4 q = *reinterpret_cast<const uint4*>(qptr);
        acc[0] = fma_h_lo(x.x, q.x, acc[0]);
        acc[1] = fma_h_hi(x.x, q.x, acc[1]);
        acc[2] = fma_h_lo(x.y, q.y, acc[2]);
        acc[3] = fma_h_hi(x.y, q.y, acc[3]);
        acc[4] = fma_h_lo(x.z, q.z, acc[4]);
        acc[5] = fma_h_hi(x.z, q.z, acc[5]);
        acc[6] = fma_h_lo(x.w, q.w, acc[6]);
        acc[7] = fma_h_hi(x.w, q.w, acc[7]);
    } else {
        const float4 q0 = *reinterpret_cast<const float4*>(qptr);
        const float4 q1 = *reinterpret_cast<const float4*>(qptr + 16);
        acc[0] = fma_hf_lo(x.x, q0.x, acc[0]);
        acc[1] = fma_hf_hi(x.x, q0.y, acc[1]);
        acc[2] = fma_hf_lo(x.y, q0.z, acc[2]);
        acc[3] = fma_hf_hi(x.y, q0.w, acc[3]);
        acc[4] = fma_hf_lo(x.z, q1.x, acc[4]);
        acc[5] = fma_hf_hi(x.z, q1.y, acc[5]);
        acc[6] = fma_hf_lo(x.w, q1.z, acc[6]);
        acc[7] = fma_hf_hi(x.w, q1.w, acc[7]);
    }
}

// The reference's reduction (vector.rs:295-303) across the four lanes of a row.
// Every lane of the quad returns the final value.
__device__ __forceinline__ float reduce_a1(const float (&acc)[8]) {
    float v[8];
#pragma unroll
    for (int l = 0; l < 8; l++) v[l] = add_rn(acc[l], __shfl_xor(acc[l], 1));  // acc1+acc2 | acc3+acc4
    const float p0 = add_rn(v[0], v[1]), p1 = add_rn(v[2], v[3]);               // hadd pairs
    const float p2 = add_rn(v[4], v[5]), p3 = add_rn(v[6], v[7]);
    const float first = add_rn(p0, p2);   // parts 0,1: s0   parts 2,3: s2
    const float second = add_rn(p1, p3);  // parts 0,1: s1   parts 2,3: s3
    const float of = __shfl_xor(first, 2), os = __shfl_xor(second, 2);
    const bool low = ((threadIdx.x & 2) == 0);
    const float s0 = low ? first : of, s1 = low ? second : os;
    const float s2 = low ? of : first, s3 = low ? os : second;
    return add_rn(add_rn(add_rn(s0, s1), s2), s3);
}

constexpr int ROWS_PER_WAVE = 16;
constexpr int WAVES = 4;
constexpr int ROWS_PER_BLOCK = ROWS_PER_WAVE * WAVES;

// scores[j * score_stride + row] for j < nq_valid.  queries: [QB][d] (f16 bits or f32), rows
// beyond nq_valid must be readable (the host pads the staging buffer).
template <int QB, bool QF32>
__global__ __launch_bounds__(256) void scan_exact_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                         const void* __restrict__ queries, int nq_valid,
                                                         int64_t* __restrict__ scores, size_t score_stride,
                                                         float* __restrict__ fscores) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QEL = QF32 ? 4 : 2;
    const int qbytes = d * QEL;
    // stage queries
    {
        const uint4* src = reinterpret_cast<const uint4*>(queries);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const int n16 = QB * qbytes / 16;
        for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane & 3, rw = lane >> 2;
    const int steps = d / 32;
    const size_t row_bytes = (size_t)d * 2;
    const int qstep = 32 * QEL;       // query bytes per step
    const int qpart = part * 8 * QEL; // this lane's group inside the step

    for (size_t rb = (size_t)blockIdx.x * ROWS_PER_BLOCK + wave * ROWS_PER_WAVE; rb < n_rows;
         rb += (size_t)gridDim.x * ROWS_PER_BLOCK) {
        size_t row = rb + rw;
        const bool valid = row < n_rows;
        if (!valid) row = n_rows - 1;
        const uint4* xp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + row * row_bytes) + part;

        float acc[QB][8];
#pragma unroll
        for (int j = 0; j < QB; j++)
#pragma unroll
            for (int l = 0; l < 8; l++) acc[j][l] = 0.0f;

        int t = 0;
        for (; t + 4 <= steps; t += 4) {
            uint4 x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) x[u] = xp[(t + u) * 4];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int j = 0; j < QB; j++) step_fma<QF32>(acc[j], x[u], smem + j * qbytes + (t + u) * qstep + qpart);
        }
        for (; t < steps; t++) {
            const uint4 x = xp[t * 4];
#pragma unroll
            for (int j = 0; j < QB; j++) step_fma<QF32>(acc[j], x, smem + j * qbytes + t * qstep + qpart);
        }
#pragma unroll
        for (int j = 0; j < QB; j++) {
            const float f = reduce_a1(acc[j]);
            if (valid && part == 0 && j < nq_valid) {
                if (scores) scores[(size_t)j * score_stride + row] = scale_dot_result(f);
                if (fscores) fscores[(size_t)j * score_stride + row] = f;
            }
        }
    }
}

// Gather-and-score (diskann/src/lib.rs:201-207; src/query_disk_index.rs:168-169): pair p scores
// row ids[p] against query qidx[p] (or query p / per_query when qidx == nullptr).
template <bool QF32>
__global__ __launch_bounds__(256) void score_rows_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                         const void* __restrict__ queries,
                                                         const uint32_t* __restrict__ ids, size_t n_pairs,
                                                         size_t pairs_per_query, int64_t* __restrict__ out,
                                                         float* __restrict__ fout) {
    constexpr int QEL = QF32 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int part = lane & 3;
    const size_t quad = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const size_t nquads = ((size_t)gridDim.x * blockDim.x) >> 2;
    const int steps = d / 32;
    const size_t n_round = (n_pairs + 15) / 16 * 16;  // keep whole waves in the shuffles
    for (size_t p = quad; p < n_round; p += nquads) {
        const bool valid = p < n_pairs;
        const size_t pp = valid ? p : n_pairs - 1;
        uint32_t id = ids[pp];
        const bool id_ok = id < n_rows;
        if (!id_ok) id = 0;
        const size_t qi = pp / pairs_per_query;
        const uint4* xp = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + (size_t)id * d * 2) + part;
        const char* qp = reinterpret_cast<const char*>(queries) + qi * (size_t)d * QEL + part * 8 * QEL;
        float acc[8];
#pragma unroll
        for (int l = 0; l < 8; l++) acc[l] = 0.0f;
        for (int t = 0; t < steps; t++) {
            const uint4 x = xp[t * 4];
            step_fma<QF32>(acc, x, qp + (size_t)t * 32 * QEL);
        }
        const float f = reduce_a1(acc);
        if (valid && part == 0) {
            if (out) out[p] = id_ok ? scale_dot_result(f) : INT64_MIN;
            if (fout) fout[p] = id_ok ? f : -3.402823466e+38f;
        }
    }
}

template <int QB, bool QF32>
int launch_scan(const uint16_t* base, size_t n_rows, int d, const void* q, int nq_valid, int64_t* scores,
                size_t stride, float* fscores, int grid, hipStream_t stream) {
    const size_t lds = (size_t)QB * d * (QF32 ? 4 : 2);
    hipLaunchKernelGGL((scan_exact_kernel<QB, QF32>), dim3(grid), dim3(256), lds, stream, base, n_rows, d, q, nq_valid,
                       scores, stride, fscores);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

int exact_scan_block() { return 8; }

// queries_dev: [qb_padded][d] where qb_padded = next of {1,2,4,8} >= nq (caller pads with zeros).
int launch_scan_exact(const uint16_t* base, size_t n_rows, int d, const void* queries_dev, int nq, bool q_is_f32,
                      int64_t* scores, size_t score_stride, float* fscores, int n_cu, hipStream_t stream) {
    if (n_rows == 0 || nq == 0) return 0;
    if (d % 64 != 0 || d <= 0 || d > D_MAX) return fail("vector width must be a positive multiple of 64");
    if (nq > 8) return fail("launch_scan_exact handles at most 8 queries per pass");
    size_t blocks_needed = (n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    size_t cap = (size_t)n_cu * 8;
    int grid = (int)(blocks_needed < cap ? blocks_needed : cap);
    const int qb = nq <= 1 ? 1 : nq <= 2 ? 2 : nq <= 4 ? 4 : 8;
#define MSE_CASE(QB)                                                                                             \
    case QB:                                                                                                     \
        return q_is_f32 ? launch_scan<QB, true>(base, n_rows, d, queries_dev, nq, scores, score_stride, fscores, \
                                                grid, stream)                                                    \
                        : launch_scan<QB, false>(base, n_rows, d, queries_dev, nq, scores, score_stride, fscores, \
                                                 grid, stream);
    switch (qb) {
        MSE_CASE(1)
        MSE_CASE(2)
        MSE_CASE(4)
        MSE_CASE(8)
    }
#undef MSE_CASE
    return fail("unreachable");
}

int launch_score_rows(const uint16_t* base, size_t n_rows, int d, const void* queries_dev, bool q_is_f32,
                      const uint32_t* ids_dev, size_t n_pairs, size_t pairs_per_query, int64_t* out, float* fout,
                      hipStream_t stream) {
    if (n_pairs == 0) return 0;
    if (d % 64 != 0 || d <= 0 || d > D_MAX) return fail("vector width must be a positive multiple of 64");
    if (pairs_per_query == 0) return fail("pairs_per_query must be positive");
    size_t quads = (n_pairs + 15) / 16 * 16;
    size_t blocks = (quads * 4 + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    if (q_is_f32)
        hipLaunchKernelGGL(score_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, base, n_rows, d,
                           queries_dev, ids_dev, n_pairs, pairs_per_query, out, fout);
    else
        hipLaunchKernelGGL(score_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, base, n_rows, d,
                           queries_dev, ids_dev, n_pairs, pairs_per_query, out, fout);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mse
