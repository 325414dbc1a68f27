// The disk-index beam search of query-disk-index with every array HBM-resident, plus the two small selectors
// around it (shard / entry-point choice, medioid).
//
//   mse_disk_greedy_search  src/query_disk_index.rs:144-212  (+ :83-97 next_several_unvisited, :101-104, :135-142)
//   mse_select_shard        src/query_disk_index.rs:254-256,447-450
//   mse_medioid             diskann/src/lib.rs:52-68 (+ `dot`, diskann/src/vector.rs:49-52)
//
// The reference fetches one 4 KiB record per visited node from NVMe (io_uring, a whole beam in flight) and
// gathers PQ codes from an mmap.  Here the record vectors (mse_base), PQ codes and descriptor bytes (mse_codes)
// live in HBM; one beam iteration is ONE batched submission: exact fast_dot of the beam's nodes, ADC of every
// neighbour that became fresh in this iteration, descriptor bias for both, one device->host copy.  The
// traversal (NeighbourBuffer, visited sets) stays on the host and is replayed in the reference's order.
#include "../../include/mse.h"
#include "runtime.h"
#include <cstdlib>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <vector>

using namespace mse;

namespace {

__global__ void shard_keys_kernel(const float* __restrict__ centroids, int n_shards, int d, const float* __restrict__ q,
                                  int64_t* __restrict__ keys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_shards) return;
    double acc = 0.0;
    for (int k = 0; k < d; k++) acc = acc + (double)centroids[(size_t)s * d + k] * (double)q[k];
    keys[s] = scale_dot_result_f64(acc);
}

// running mean of lib.rs:55-58: c += (row - c) * (1 / (i + 1)) row after row, in f32 -- a sequential chain per component.
// One thread per component carries the chain; what the first version got wrong was feeding it: every step waited for its own
// 2-byte global load (384 ns per row, 38 s at 1e8 rows).  Here a workgroup of 1024 threads owns 128 components: all threads
// fetch the next 64 rows' 256-byte slices (16 bytes each) into registers while 128 of them walk the current 64 rows out of LDS,
// so the chain (three dependent f32 operations per row) is the only thing on the critical path.  Same operations in the same
// order as before (and as the oracle): identical result.
constexpr int CEN_COMP = 128, CEN_ROWS = 64;
__global__ __launch_bounds__(1024) void centroid_kernel(const uint16_t* __restrict__ base, size_t n, int d, uint16_t* __restrict__ mean) {
    __shared__ __attribute__((aligned(16))) float tile[2][CEN_ROWS][CEN_COMP];   // rows already widened to f32 (exact)
    __shared__ float wt[2][CEN_ROWS];                                              // 1 / (i + 1), IEEE f32 division, per row
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * CEN_COMP;
    const int lrow = tid >> 4, piece = tid & 15;                       // loader role: row of the chunk, 16-byte piece of the slice
    const bool piece_ok = k0 + piece * 8 < d;                           // d is a multiple of 8 (checked by the caller)
    auto fetch = [&](size_t chunk) -> uint4 {
        const size_t r = chunk * CEN_ROWS + lrow;
        if (r < n && piece_ok) return *reinterpret_cast<const uint4*>(base + r * d + k0 + piece * 8);
        return uint4{0u, 0u, 0u, 0u};
    };
    const size_t n_chunks = (n + CEN_ROWS - 1) / CEN_ROWS;
    uint4 next = fetch(0);
    float c = 0.0f;
    for (size_t chunk = 0; chunk < n_chunks; chunk++) {
        const int buf = (int)(chunk & 1);
        {
            const uint32_t w4[4] = {next.x, next.y, next.z, next.w};
            float* dst = &tile[buf][lrow][piece * 8];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                dst[2 * j] = __half2float(__ushort_as_half((unsigned short)(w4[j] & 0xffffu)));
                dst[2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(w4[j] >> 16)));
            }
            if (tid < CEN_ROWS) wt[buf][tid] = 1.0f / (float)(chunk * CEN_ROWS + tid + 1);
        }
        __syncthreads();                                                // also: everybody is done with tile[buf] of two chunks ago
        if (chunk + 1 < n_chunks) next = fetch(chunk + 1);
        if (tid < CEN_COMP) {
            const size_t r0 = chunk * CEN_ROWS;
            const int rows = (int)(n - r0 < (size_t)CEN_ROWS ? n - r0 : (size_t)CEN_ROWS);
            for (int rr = 0; rr < rows; rr++) {
                const float diff = tile[buf][rr][tid] - c;
                const float step = diff * wt[buf][rr];
                c = c + step;
            }
        }
    }
    if (tid < CEN_COMP && k0 + tid < d) reinterpret_cast<__half*>(mean)[k0 + tid] = __float2half_rn(c);
}

// `dot` (vector.rs:49-52) in the order the oracle states for simsimd: exact products, f64 sum in index order
__global__ void dot_f64_rows_kernel(const uint16_t* __restrict__ base, size_t n, int d, const uint16_t* __restrict__ y,
                                    int64_t* __restrict__ out) {
    extern __shared__ float ys[];
    for (int k = threadIdx.x; k < d; k += blockDim.x) ys[k] = __half2float(reinterpret_cast<const __half*>(y)[k]);
    __syncthreads();
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint4* row = reinterpret_cast<const uint4*>(base + r * d);
    double acc = 0.0;
    for (int c = 0; c < d / 8; c++) {
        const uint4 v = row[c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float lo = __half2float(__ushort_as_half((unsigned short)(w[j] & 0xffffu)));
            const float hi = __half2float(__ushort_as_half((unsigned short)(w[j] >> 16)));
            acc = acc + (double)lo * (double)ys[c * 8 + 2 * j];
            acc = acc + (double)hi * (double)ys[c * 8 + 2 * j + 1];
        }
    }
    out[r] = scale_dot_result_f64(acc);
}

// bit (i, j) = [ dot_f32(row ids[i], row ids[j]) > threshold ] for j < i; one wave per (i, block of 64 j); the dot is a
// k-ascending fp32 FMA chain over the widened f16 rows (the order the oracle states for the reference's sgemm)
__global__ __launch_bounds__(256) void sim_bits_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                       const uint32_t* __restrict__ ids, int n, float threshold,
                                                       unsigned long long* __restrict__ bits, int words) {
    const int lane = threadIdx.x & 63;
    const size_t wid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = (int)(wid / words), w = (int)(wid % words);
    if (i >= n) return;
    if (w * 64 >= i) { if (lane == 0) bits[(size_t)i * words + w] = 0ull; return; }
    const int j = w * 64 + lane;
    bool hit = false;
    if (j < i) {
        const __half* a = reinterpret_cast<const __half*>(base) + (size_t)ids[i] * d;
        const __half* b = reinterpret_cast<const __half*>(base) + (size_t)ids[j] * d;
        float s = 0.0f;
        for (int k = 0; k < d; k++) s = fmaf(__half2float(a[k]), __half2float(b[k]), s);
        hit = s > threshold;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) bits[(size_t)i * words + w] = m;
}

// The same bits from a register-tiled kernel: a workgroup computes a 64 x 64 tile of the (lower-triangular) similarity matrix, a
// thread a 4 x 4 block; the gathered rows are widened to f32 into k-major LDS tiles, one ds_read_b128 feeds four operands.  Every
// (i, j) dot is still the chain s = fmaf(a_k, b_k, s) for k ascending, so the comparison against the threshold is unchanged
// (the first kernel spent its time in per-lane 2-byte loads: 3.3 ms for 1000 visited rows, 40 ms for 4000).
constexpr int SB_T = 64, SB_K = 32;
__device__ __forceinline__ void sim_bits_tile(const uint16_t* __restrict__ base, int d, const uint32_t* __restrict__ ids, int n, float threshold,
                                              unsigned long long* __restrict__ bits, int words) {
    __shared__ __attribute__((aligned(16))) float As[SB_K][SB_T + 4];   // rows i0 .. i0+63
    __shared__ __attribute__((aligned(16))) float Bs[SB_K][SB_T + 4];   // rows j0 .. j0+63
    __shared__ unsigned long long mask[SB_T];
    // tile (bi, bj) with bj <= bi from a linear index over the lower triangle
    int bi = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) bi++;
    while (bi * (bi + 1) / 2 > (int)blockIdx.x) bi--;
    const int bj = (int)blockIdx.x - bi * (bi + 1) / 2;
    const int i0 = bi * SB_T, j0 = bj * SB_T;
    if (i0 >= n) return;   // (batched form: a list shorter than the grid was sized for; uniform for the workgroup)
    const int tid = threadIdx.x, ti = tid & 15, tj = tid >> 4;
    if (tid < SB_T) mask[tid] = 0ull;
    float acc[4][4];   // [a: row i][b: row j]
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0f;
    const __half* hb = reinterpret_cast<const __half*>(base);
    for (int k0 = 0; k0 < d; k0 += SB_K) {
#pragma unroll
        for (int r = 0; r < (SB_T * SB_K) / 256; r++) {
            const int e = tid + 256 * r;
            const int kk = e % SB_K, row = e / SB_K;
            const bool kin = k0 + kk < d;
            As[kk][row] = (kin && i0 + row < n) ? __half2float(hb[(size_t)ids[i0 + row] * d + k0 + kk]) : 0.0f;
            Bs[kk][row] = (kin && j0 + row < n) ? __half2float(hb[(size_t)ids[j0 + row] * d + k0 + kk]) : 0.0f;
        }
        __syncthreads();
        const int kmax = d - k0 < SB_K ? d - k0 : SB_K;
        for (int kk = 0; kk < kmax; kk++) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[kk][ti * 4]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[kk][tj * 4]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = fmaf(av[a], bv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int i = i0 + ti * 4 + a;
        unsigned long long m = 0ull;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int j = j0 + tj * 4 + b;
            if (i < n && j < i && acc[a][b] > threshold) m |= 1ull << (tj * 4 + b);
        }
        if (m) atomicOr(&mask[ti * 4 + a], m);
    }
    __syncthreads();
    if (tid < SB_T && i0 + tid < n) bits[(size_t)(i0 + tid) * words + bj] = mask[tid];
}
__global__ __launch_bounds__(256) void sim_bits_tiled_kernel(const uint16_t* __restrict__ base, int d, const uint32_t* __restrict__ ids,
                                                             int n, float threshold, unsigned long long* __restrict__ bits, int words) {
    sim_bits_tile(base, d, ids, n, threshold, bits, words);
}
// The request path's runtime de-duplication INSIDE the call (src/query_disk_index.rs:482-527), for every query of a batch: the same
// similarity bits over the query's visited records (blockIdx.y = query; lists are [cap] wide, n_visited[q] long) ...
__global__ __launch_bounds__(256) void sim_bits_batch_kernel(const uint16_t* __restrict__ base, int d, const uint32_t* __restrict__ vis_ids, size_t cap,
                                                             const uint32_t* __restrict__ n_visited, float threshold, unsigned long long* __restrict__ bits,
                                                             int words) {
    const size_t q = blockIdx.y;
    const int n = (int)min((size_t)n_visited[q], cap);
    sim_bits_tile(base, d, vis_ids + q * cap, n, threshold, bits + q * cap * (size_t)words, words);
}
// ... and the greedy keep-first filter in visit order (:514-527), one wave per query: lane w holds word w of the "kept" set; a record
// that resembles an already kept one leaves the list (id none, score minimal: the select that follows skips it)
__global__ __launch_bounds__(64) void dedup_filter_batch_kernel(const unsigned long long* __restrict__ bits, int words, size_t cap,
                                                                const uint32_t* __restrict__ n_visited, uint32_t* __restrict__ vis_ids,
                                                                long long* __restrict__ vis_scores) {
    const size_t q = blockIdx.x;
    const int lane = threadIdx.x, n = (int)min((size_t)n_visited[q], cap);
    const unsigned long long* b = bits + q * cap * (size_t)words;
    unsigned long long kept = 0ull;
    unsigned long long row = lane < words && n > 0 ? b[lane] : 0ull;
    for (int i = 0; i < n; i++) {
        const unsigned long long next = (lane < words && i + 1 < n) ? b[(size_t)(i + 1) * words + lane] : 0ull;   // in flight during the vote
        const bool dropped = __ballot((row & kept) != 0ull) != 0ull;
        if (dropped) {
            if (lane == 0) { vis_ids[q * cap + i] = 0xffffffffu; vis_scores[q * cap + i] = (long long)INT64_MIN; }
        } else if (lane == i / 64) {
            kept |= 1ull << (i % 64);
        }
        row = next;
    }
}

}  // namespace

namespace mse {
// scratch_bytes(nq, cap): what `bits` must hold; cap <= 4096 (64 words of 64 records)
// The similarity bits of a query are cap x ceil(cap / 64) words: a coalesced pass of 1024 queries at search_list 1024 (cap 2112) would
// need 570 MB per worker searcher.  The batch is therefore run in query chunks that fit a fixed scratch budget.
static constexpr size_t DEDUP_SCRATCH_BUDGET = (size_t)128 << 20;
static size_t dedup_chunk_queries(size_t nq, size_t cap) {
    const size_t per_query = cap * ((cap + 63) / 64) * 8;
    if (per_query == 0) return nq;
    return std::max<size_t>(1, std::min(nq, DEDUP_SCRATCH_BUDGET / per_query));
}
size_t dedup_batch_scratch_bytes(size_t nq, size_t cap) { return dedup_chunk_queries(nq, cap) * cap * ((cap + 63) / 64) * 8; }
int launch_dedup_batch(const uint16_t* base, int d, uint32_t* vis_ids, long long* vis_scores, size_t cap, const uint32_t* n_visited, size_t nq,
                       float threshold, void* bits, hipStream_t st) {
    const int words = (int)((cap + 63) / 64);
    if (words > 64) return fail("request-path de-duplication: more than 4096 visited records per query");
    if (nq == 0 || cap == 0) return 0;
    const size_t chunk = std::min<size_t>(dedup_chunk_queries(nq, cap), 65535);   // (gridDim.y)
    for (size_t q0 = 0; q0 < nq; q0 += chunk) {   // chunks run one after the other on the stream: they share the scratch
        const size_t m = std::min(chunk, nq - q0);
        MSE_HIP_TRY(hipMemsetAsync(bits, 0, m * cap * (size_t)words * 8, st));   // words above the diagonal are never written
        hipLaunchKernelGGL(sim_bits_batch_kernel, dim3((unsigned)(words * (words + 1) / 2), (unsigned)m), dim3(256), 0, st, base, d, vis_ids + q0 * cap, cap,
                           n_visited + q0, threshold, static_cast<unsigned long long*>(bits), words);
        MSE_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(dedup_filter_batch_kernel, dim3((unsigned)m), dim3(64), 0, st, static_cast<const unsigned long long*>(bits), words, cap,
                           n_visited + q0, vis_ids + q0 * cap, vis_scores + q0 * cap);
        MSE_HIP_TRY(hipGetLastError());
    }
    return 0;
}
}  // namespace mse

extern "C" {

int mse_dedup_visited(mse_searcher* s, const uint32_t* ids, size_t n, float threshold, uint8_t* keep) {
    if (!s || !s->base || !ids || !keep) return fail("dedup_visited: null argument");
    if (n == 0) return 0;
    if (n > 65535) return fail("dedup_visited: more than 65535 visited rows");
    const mse_base* b = s->base;
    for (size_t i = 0; i < n; i++)
        if (ids[i] >= b->n) return fail("dedup_visited: id outside the index");
    const int words = (int)((n + 63) / 64);
    hipStream_t st = s->stream;
    if (s->cand_ids.ensure(n * 4) || s->misc.ensure(n * (size_t)words * 8)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(s->cand_ids.p, ids, n * 4, hipMemcpyHostToDevice, st));
    static const bool old_sim = MSE_DEV_KNOB("MSE_DEDUP_OLD");
    if (old_sim) {
        const size_t waves = n * (size_t)words;
        hipLaunchKernelGGL(sim_bits_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, b->dev, b->n, (int)b->d,
                           s->cand_ids.as<uint32_t>(), (int)n, threshold, s->misc.as<unsigned long long>(), words);
    } else {
        // words above the diagonal are never written by the tiled kernel: clear the whole array first
        MSE_HIP_TRY(hipMemsetAsync(s->misc.p, 0, n * (size_t)words * 8, st));
        const unsigned nb = (unsigned)words;
        hipLaunchKernelGGL(sim_bits_tiled_kernel, dim3(nb * (nb + 1) / 2), dim3(256), 0, st, b->dev, (int)b->d,
                           s->cand_ids.as<uint32_t>(), (int)n, threshold, s->misc.as<unsigned long long>(), words);
    }
    MSE_HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> bits(n * (size_t)words), kept((size_t)words, 0ull);
    MSE_HIP_TRY(hipMemcpyAsync(bits.data(), s->misc.p, bits.size() * 8, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < n; i++) {   // greedy keep-first filter in visit order (:514-527)
        bool dropped = false;
        for (int w = 0; w < words && !dropped; w++) dropped = (bits[i * words + w] & kept[w]) != 0ull;
        keep[i] = dropped ? 0 : 1;
        if (!dropped) kept[i / 64] |= 1ull << (i % 64);
    }
    return 0;
}

int mse_select_shard(const float* centroids, size_t n_shards, size_t d, const float* query, size_t* shard_out) {
    if (!centroids || !query || !shard_out) return fail("select_shard: null argument");
    if (n_shards == 0) return fail("select_shard: no shards");
    DevBuf c, q, k;
    if (c.ensure(n_shards * d * 4) || q.ensure(d * 4) || k.ensure(n_shards * 8)) return -1;
    MSE_HIP_TRY(hipMemcpy(c.p, centroids, n_shards * d * 4, hipMemcpyHostToDevice));
    MSE_HIP_TRY(hipMemcpy(q.p, query, d * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(shard_keys_kernel, dim3((unsigned)((n_shards + 63) / 64)), dim3(64), 0, nullptr, c.as<float>(),
                       (int)n_shards, (int)d, q.as<float>(), k.as<int64_t>());
    MSE_HIP_TRY(hipGetLastError());
    std::vector<int64_t> keys(n_shards);
    MSE_HIP_TRY(hipMemcpy(keys.data(), k.p, n_shards * 8, hipMemcpyDeviceToHost));
    size_t best = 0;
    for (size_t s = 1; s < n_shards; s++)
        if (keys[s] >= keys[best]) best = s;  // position_max_by_key: the LAST maximum
    *shard_out = best;
    return 0;
}

int mse_medioid(const mse_base* b, uint32_t* id_out) {
    if (!b || !id_out) return fail("medioid: null argument");
    if (b->n == 0) return fail("medioid: empty vector list");
    const size_t n = b->n;
    const int d = (int)b->d;
    if (d % 8) return fail("medioid: d must be a multiple of 8");
    DevBuf mean, keys;
    if (mean.ensure((size_t)d * 2) || keys.ensure(n * 8)) return -1;
    hipLaunchKernelGGL(centroid_kernel, dim3((unsigned)((d + CEN_COMP - 1) / CEN_COMP)), dim3(1024), 0, nullptr, b->dev, n, d,
                       mean.as<uint16_t>());
    MSE_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(dot_f64_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), (size_t)d * 4, nullptr, b->dev, n, d,
                       mean.as<uint16_t>(), keys.as<int64_t>());
    MSE_HIP_TRY(hipGetLastError());
    std::vector<int64_t> h(n);
    MSE_HIP_TRY(hipMemcpy(h.data(), keys.p, n * 8, hipMemcpyDeviceToHost));
    size_t best = 0;
    for (size_t r = 1; r < n; r++)
        if (h[r] >= h[best]) best = r;       // Iterator::max_by keeps the last maximum
    *id_out = (uint32_t)best;
    return 0;
}

int mse_disk_greedy_search(mse_searcher* s, mse_pq* pq, const mse_codes* c, const uint32_t* adj, const uint32_t* deg,
                           size_t max_deg, const uint8_t* has_url, uint32_t start, const uint16_t* query, const float* lut,
                           const float* scales, int disable_pq, size_t beamwidth, mse_nb* buf, uint32_t* visited_ids,
                           int64_t* visited_scores, size_t visited_cap, size_t* n_visited_out, size_t* cmps_out,
                           size_t* pq_cmps_out) {
    if (!s || !s->base || !pq || !c || !buf || !adj || !deg || !query || !lut) return fail("disk_greedy_search: null argument");
    const mse_base* b = s->base;
    const size_t n = b->n, d = b->d;
    if (c->n != n) return fail("disk_greedy_search: codes and vectors differ in length");
    if (c->code_size != pq->n_chunks) return fail("disk_greedy_search: code size does not match the quantiser");
    if (start >= n) return fail("disk_greedy_search: start node out of range");
    if (beamwidth == 0) return fail("disk_greedy_search: beamwidth must be positive");
    hipStream_t st = s->stream;
    const bool bias = scales && c->n_desc && c->desc;
    const size_t lut_bytes = pq->n_chunks * pq->n_centroids * 4;
    const size_t max_batch = beamwidth * (max_deg + 1);
    // device scratch of the searcher: [query | lut | scales], ids, scores
    const size_t q_off = 0, lut_off = (d * 2 + 255) & ~(size_t)255, sc_off = lut_off + ((lut_bytes + 255) & ~(size_t)255);
    if (s->q_stage.ensure(std::max<size_t>(sc_off + c->n_desc * 4 + 256, 8 * d * 2)) || s->cand_ids.ensure(max_batch * 4) ||
        s->cand_scores.ensure(max_batch * 8))
        return -1;
    char* stage = s->q_stage.as<char>();
    MSE_HIP_TRY(hipMemcpyAsync(stage + q_off, query, d * 2, hipMemcpyHostToDevice, st));
    MSE_HIP_TRY(hipMemcpyAsync(stage + lut_off, lut, lut_bytes, hipMemcpyHostToDevice, st));
    if (bias) MSE_HIP_TRY(hipMemcpyAsync(stage + sc_off, scales, c->n_desc * 4, hipMemcpyHostToDevice, st));
    const float* lut_dev = reinterpret_cast<const float*>(stage + lut_off);
    const float* scales_dev = bias ? reinterpret_cast<const float*>(stage + sc_off) : nullptr;
    uint32_t* ids_dev = s->cand_ids.as<uint32_t>();
    int64_t* sc_dev = s->cand_scores.as<int64_t>();

    std::vector<uint8_t> visited_adjacent(n, 0), visited(n, 0);
    std::vector<uint32_t> batch;          // [beam nodes | fresh neighbours of this iteration]
    std::vector<size_t> seg_end;          // pre-buffer length after each beam node
    std::vector<int64_t> sc;
    batch.reserve(max_batch);
    sc.resize(max_batch);
    size_t cmps = 0, pq_cmps = 0, n_visited = 0;
    mse_nb_clear(buf);
    mse_nb_insert(buf, start, 0);          // :153 -- the entry point enters with score 0
    visited_adjacent[start] = 1;
    for (;;) {
        batch.clear();
        seg_end.clear();
        uint32_t p;
        while (batch.size() < beamwidth && mse_nb_next_unvisited(buf, &p)) batch.push_back(p);   // :83-97
        const size_t npts = batch.size();
        if (npts == 0) break;
        for (size_t j = 0; j < npts; j++) {   // :184-188 for every node of the beam, in fetch order
            const uint32_t pt = batch[j];
            for (uint32_t e = 0; e < deg[pt]; e++) {
                const uint32_t nb = adj[(size_t)pt * max_deg + e];
                if (nb >= n) return fail("graph edge points outside the index");
                if (!visited_adjacent[nb]) { visited_adjacent[nb] = 1; batch.push_back(nb); }
            }
            seg_end.push_back(batch.size() - npts);
        }
        const size_t npre = batch.size() - npts;
        MSE_HIP_TRY(hipMemcpyAsync(ids_dev, batch.data(), batch.size() * 4, hipMemcpyHostToDevice, st));
        const size_t n_exact = disable_pq ? batch.size() : npts;
        if (launch_score_rows(b->dev, n, (int)d, stage + q_off, false, ids_dev, n_exact, n_exact, sc_dev, nullptr, st)) return -1;
        if (bias && launch_add_descriptor(ids_dev, n_exact, c->desc, (int)c->n_desc, n, scales_dev, sc_dev, st)) return -1;
        if (!disable_pq && npre &&
            launch_pq_adc(lut_dev, (int)pq->n_chunks, (int)pq->n_centroids, c->codes, n, ids_dev + npts, npre,
                          bias ? c->desc : nullptr, (int)c->n_desc, scales_dev, sc_dev + npts, s->n_cu, st))
            return -1;
        MSE_HIP_TRY(hipMemcpyAsync(sc.data(), sc_dev, batch.size() * 8, hipMemcpyDeviceToHost, st));
        MSE_HIP_TRY(hipStreamSynchronize(st));
        for (size_t j = 0; j < npts; j++) {   // replay :165-208
            const uint32_t pt = batch[j];
            cmps++;
            if (!visited[pt]) {
                visited[pt] = 1;
                if (!has_url || has_url[pt]) {
                    if (n_visited < visited_cap) {
                        if (visited_ids) visited_ids[n_visited] = pt;
                        if (visited_scores) visited_scores[n_visited] = sc[j];
                    }
                    n_visited++;
                }
            }
            // neighbour_pre_buffer is cleared per beam iteration, not per node (:157): node j re-inserts everything
            // gathered so far in this iteration
            for (size_t i = 0; i < seg_end[j]; i++) {
                mse_nb_insert(buf, batch[npts + i], sc[npts + i]);
                if (!disable_pq) pq_cmps++;
            }
        }
    }
    if (n_visited_out) *n_visited_out = n_visited;
    if (cmps_out) *cmps_out = cmps;
    if (pq_cmps_out) *pq_cmps_out = pq_cmps;
    return 0;
}

}  // extern "C"
