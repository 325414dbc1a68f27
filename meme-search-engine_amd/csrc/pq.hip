// OPQ/PQ codec kernels: diskann::vector::ProductQuantizer (diskann/src/vector.rs:308-406), the
// descriptor bias of the disk search (src/query_disk_index.rs:135-142) and rank counting for the
// evaluator (src/query_disk_index.rs:271-273).
#include "common.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

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

// Register-tiled form for batches (index packing transforms every record): a thread owns a 4 x 4 block of outputs, a workgroup
// 64 (i) x 64 (j); operand tiles are stored k-major in LDS so that one ds_read_b128 delivers T[i..i+3][k] (or x[j..j+3][k]) and
// 16 fused multiply-adds follow.  Every output still accumulates k = 0 .. d-1 in ascending order with one fmaf per term, so the
// result is bit-identical to the kernel above (and to the oracle); 2 LDS values per 16 FMAs instead of 2 per FMA.
constexpr int T4_B = 64, T4_K = 32;
__global__ __launch_bounds__(256) void pq_transform_tiled_kernel(const float* __restrict__ T, int d, const float* __restrict__ x, size_t n,
                                                                 float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Ts[T4_K][T4_B + 4];
    __shared__ __attribute__((aligned(16))) float Xs[T4_K][T4_B + 4];
    const int tid = threadIdx.x;
    const int ti = tid & 15, tj = tid >> 4;
    const size_t i0 = (size_t)blockIdx.x * T4_B, j0 = (size_t)blockIdx.y * T4_B;
    float acc[4][4];   // [b: vector j][a: output i]
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++) acc[b][a] = 0.0f;
    for (int k0 = 0; k0 < d; k0 += T4_K) {
#pragma unroll
        for (int r = 0; r < (T4_B * T4_K) / 256; r++) {
            const int e = tid + 256 * r;
            const int kk = e % T4_K, row = e / T4_K;           // consecutive threads walk k: 128-byte runs of one row
            const bool kin = k0 + kk < d;
            Ts[kk][row] = (kin && i0 + row < (size_t)d) ? T[(i0 + row) * d + k0 + kk] : 0.0f;
            Xs[kk][row] = (kin && j0 + row < n) ? x[(j0 + row) * d + k0 + kk] : 0.0f;
        }
        __syncthreads();
        const int kmax = d - k0 < T4_K ? d - k0 : T4_K;
        for (int kk = 0; kk < kmax; kk++) {
            const float4 a4 = *reinterpret_cast<const float4*>(&Ts[kk][ti * 4]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Xs[kk][tj * 4]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int b = 0; b < 4; b++)
#pragma unroll
                for (int a = 0; a < 4; a++) acc[b][a] = fmaf(av[a], bv[b], acc[b][a]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const size_t j = j0 + tj * 4 + b;
        if (j >= n) continue;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const size_t i = i0 + ti * 4 + a;
            if (i < (size_t)d) out[j * d + i] = acc[b][a];
        }
    }
}

// The same product for ONE vector (the query path): thread i owns out[i] and runs the identical k-ascending chain of fused
// multiply-adds, reading the TRANSPOSED matrix Tt[k][i] so that a wave's loads are contiguous (the tiled kernel above keeps
// 16 of its 256 threads busy when n = 1).
__global__ __launch_bounds__(64) void pq_transform_vec_kernel(const float* __restrict__ Tt, int d, const float* __restrict__ x,
                                                              float* __restrict__ out) {
    extern __shared__ float xs[];
    for (int k = threadIdx.x; k < d; k += blockDim.x) xs[k] = x[k];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    const float* col = Tt + i;
    float acc = 0.0f;
    int k = 0;
    // 64 loads in flight per thread: the chain is latency-bound (a miss costs ~0.5 us and only 18 waves are at work)
    for (; k + 64 <= d; k += 64) {
        float t[64];
#pragma unroll
        for (int u = 0; u < 64; u++) t[u] = col[(size_t)(k + u) * d];
#pragma unroll
        for (int u = 0; u < 64; u++) acc = fmaf(t[u], xs[k + u], acc);
    }
    for (; k + 8 <= d; k += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = col[(size_t)(k + u) * d];
#pragma unroll
        for (int u = 0; u < 8; u++) acc = fmaf(t[u], xs[k + u], acc);
    }
    for (; k < d; k++) acc = fmaf(col[(size_t)k * d], xs[k], acc);
    out[i] = acc;
}

// preprocess_query table (vector.rs:373-381): lut[i*C + j] = (f32) sum_u t[i*dpc+u] * c_j[i*dpc+u]
// with the sum carried in f64 (simsimd's f32 dot returns f64; its lane order is CPU dependent, so
// "parity unpinned"; products of two f32 are exact in f64, hence index-order f64 adds here).
__global__ void pq_lut_kernel(const float* __restrict__ centroids, int n_centroids, int d, int dpc,
                              const float* __restrict__ t, float* __restrict__ lut) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_chunks = d / dpc;
    if (idx >= n_chunks * n_centroids) return;
    t += (size_t)blockIdx.y * d;                               // one table per query (blockIdx.y)
    lut += (size_t)blockIdx.y * n_chunks * n_centroids;
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

// The same arithmetic, organised for throughput (index packing encodes every record: src/dump_processor.rs:470,523).  The
// kernel above gives one thread one (vector, sub-space) pair and lets it stream 256 x dpc centroid values from global memory:
// 6.9 ms per 8192-row batch, 1 % of the fp32 rate.  Here a workgroup takes ONE sub-space and 256 x VPT vectors: the 256 x DPC
// centroid block sits in LDS (18 KiB at DPC = 18), a thread keeps its VPT vectors' DPC transformed values in registers and
// walks the centroids in ascending order -- every LDS value is a wave-wide broadcast and feeds VPT fused multiply-adds.  Each
// dot is the same chain of DPC fmaf in ascending u, the comparison the same strict `>` in ascending c: bit-identical codes.
template <int DPC, int VPT>
__global__ __launch_bounds__(256) void pq_quantize_tiled_kernel(const float* __restrict__ centroids, int n_centroids, int d,
                                                                const float* __restrict__ t, size_t n, uint8_t* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) float cs[];   // [n_centroids][DPC]
    const int n_chunks = d / DPC;
    const int chunk = blockIdx.y;
    for (int e = threadIdx.x; e < n_centroids * DPC; e += blockDim.x)
        cs[e] = centroids[(size_t)(e / DPC) * d + chunk * DPC + e % DPC];
    __syncthreads();
    const size_t v0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VPT;
    float tv[VPT][DPC];
#pragma unroll
    for (int j = 0; j < VPT; j++) {
        const size_t v = v0 + j < n ? v0 + j : (n ? n - 1 : 0);
#pragma unroll
        for (int u = 0; u < DPC; u++) tv[j][u] = t[v * d + chunk * DPC + u];
    }
    float best[VPT];
    int code[VPT];
#pragma unroll
    for (int j = 0; j < VPT; j++) { best[j] = -__builtin_inff(); code[j] = 0; }
    for (int c = 0; c < n_centroids; c++) {
        float cv[DPC];
#pragma unroll
        for (int u = 0; u < DPC; u++) cv[u] = cs[c * DPC + u];
#pragma unroll
        for (int j = 0; j < VPT; j++) {
            float s = 0.0f;
#pragma unroll
            for (int u = 0; u < DPC; u++) s = fmaf(tv[j][u], cv[u], s);
            if (s > best[j]) { best[j] = s; code[j] = c; }
        }
    }
#pragma unroll
    for (int j = 0; j < VPT; j++)
        if (v0 + j < n) codes[(v0 + j) * n_chunks + chunk] = (uint8_t)code[j];
}

// asymmetric_dot_product (vector.rs:387-405) with the table in LDS: per vector, s = 0; for chunk i
// ascending: s += lut[i][code_i] (plain fp32 adds, the reference's order), then `(s * 2^32) as i64`;
// optional descriptor bias added afterwards in i64 (src/query_disk_index.rs:135-142,202).
// ids == nullptr: vector p is row p of `codes` (full scan); otherwise row ids[p] (gathered).
__global__ __launch_bounds__(1024) void pq_adc_kernel(const float* __restrict__ lut, int n_chunks, int n_centroids,
                                                     const uint8_t* __restrict__ codes, size_t n_codes,
                                                     const uint32_t* __restrict__ ids, size_t n,
                                                     const uint8_t* __restrict__ desc, int n_desc,
                                                     const float* __restrict__ scales, int64_t* __restrict__ out,
                                                     size_t q_stride /* blockIdx.y = query: ids / out advance by this, lut by one table */
#ifdef MSE_DEV_KERNELS
                                                     , int g_adc_variant   // developer library: timing probes beside a running scan (nothing written): see launch_pq_adc
#endif
                                                     ) {
    extern __shared__ __attribute__((aligned(16))) float s_lut[];
    const int lut_n = n_chunks * n_centroids;
    lut += (size_t)blockIdx.y * lut_n;
    if (ids) ids += (size_t)blockIdx.y * q_stride;
    out += (size_t)blockIdx.y * q_stride;
    // The product codec (64 chunks) with gathered ids: a thread takes its candidates four at a time and has every load of a
    // batch in flight before it uses any -- ids first, before the table copy; then the four 64-byte code rows and descriptor
    // words; the barrier that publishes the table falls between issue and use.  As the tail of a batched scan this kernel runs on
    // the CUs the next scan leaves free while that scan saturates HBM, and every dependent round trip costs microseconds there:
    // one candidate at a time (13 round trips per workgroup) took 750-930 us for 8 x 32 768 candidates, 31 us on an idle device
    // (profiles/r04_pq_timeline.txt).
    constexpr int PER = 4;
    const bool use_desc = desc && scales;
    const bool fast = n_chunks == 64 && ids && (!use_desc || n_desc == 4);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t base = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rid[PER];
    bool okv[PER];
    auto load_ids = [&](size_t b) {
#pragma unroll
        for (int j = 0; j < PER; j++) {
            const size_t p = b + (size_t)j * stride;
            rid[j] = p < n ? ids[p] : 0xffffffffu;
        }
    };
    uint4 w[PER][4];
    uint32_t dw[PER];
    auto load_codes = [&]() {
#pragma unroll
        for (int j = 0; j < PER; j++) {
            okv[j] = (size_t)rid[j] < n_codes;
            const size_t row = okv[j] ? (size_t)rid[j] : 0;
            const uint4* cp = reinterpret_cast<const uint4*>(codes + row * 64);
#pragma unroll
            for (int a = 0; a < 4; a++) w[j][a] = cp[a];
            dw[j] = use_desc ? *reinterpret_cast<const uint32_t*>(desc + row * 4) : 0u;
        }
    };
    float sc4[4] = {0.f, 0.f, 0.f, 0.f};
    if (fast) {
        load_ids(base);
        if (use_desc)
            for (int j = 0; j < 4; j++) sc4[j] = scales[j];
    }
    if ((lut_n & 3) == 0 && (reinterpret_cast<uintptr_t>(lut) & 15) == 0) {     // 16 bytes per thread and step
        const float4* src = reinterpret_cast<const float4*>(lut);
        float4* dst = reinterpret_cast<float4*>(s_lut);
        for (int e = threadIdx.x; e < lut_n / 4; e += blockDim.x) dst[e] = src[e];
    } else {
        for (int e = threadIdx.x; e < lut_n; e += blockDim.x) s_lut[e] = lut[e];
    }
#ifdef MSE_DEV_KERNELS
    if (g_adc_variant == 1) { __syncthreads(); if (s_lut[threadIdx.x] == 1234.5f) out[0] = 1; return; }          // table copy only
    if (g_adc_variant == 2 && fast) {                                                                            // neighbouring rows instead of gathered ones
#pragma unroll
        for (int j = 0; j < PER; j++) rid[j] = (uint32_t)((base + (size_t)j * stride) % n_codes);
    }
    if (g_adc_variant == 3) {                                                                                    // no table copy wait, no sums: ids + code rows only
        if (fast) load_codes();
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) x ^= w[j][0].x ^ w[j][1].y ^ w[j][2].z ^ w[j][3].w ^ dw[j];
        if (x == 0x12345678u) out[0] = 1;
        return;
    }
#endif
    if (fast) load_codes();
    __syncthreads();
    if (fast) {
        for (;;) {
            const size_t next = base + (size_t)PER * stride;
            const bool more = next < n;
            bool okc[PER];
#pragma unroll
            for (int j = 0; j < PER; j++) okc[j] = okv[j];
            if (more) load_ids(next);   // the next batch's ids travel while this one is summed (its code rows follow below)
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const size_t p = base + (size_t)j * stride;
                float sum = 0.0f;
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const uint32_t ww[4] = {w[j][a].x, w[j][a].y, w[j][a].z, w[j][a].w};
#pragma unroll
                    for (int b4 = 0; b4 < 4; b4++)
#pragma unroll
                        for (int bb = 0; bb < 4; bb++) {
                            const int i = a * 16 + b4 * 4 + bb;
                            sum = add_rn(sum, s_lut[i * n_centroids + ((ww[b4] >> (8 * bb)) & 0xff)]);
                        }
                }
                int64_t r = scale_dot_result(sum);
                if (use_desc) {
#pragma unroll
                    for (int q = 0; q < 4; q++) r += scale_dot_result(sc4[q] * (float)((dw[j] >> (8 * q)) & 0xff));
                }
                if (p < n) out[p] = okc[j] ? r : INT64_MIN;
            }
            if (!more) break;
            base = next;
            load_codes();
        }
        return;
    }
    const bool vec16 = (n_chunks % 16) == 0;
    for (size_t p = base; p < n; p += stride) {
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

// Full ADC scan (64 chunks x 256 centroids, the reference's 64 x 8-bit codec): same arithmetic as pq_adc_kernel
// -- one lane owns one vector and adds its 64 table entries in chunk order -- but
//   * 16 waves per workgroup share ONE copy of the 64 KiB table (4 waves per SIMD hide the LDS gather latency);
//   * codes arrive by LDS-DMA as whole 4 KiB groups of 64 vectors (coalesced 1 KiB requests instead of 64 lanes
//     x 16 B at a 64-byte stride), each lane then reads its own 64-byte row from the wave's private staging area;
//     the DMA permutes the four 16-byte pieces of row r by (r >> 3) & 3 so that those ds_read_b128 are conflict free;
//   * the next group's DMA is issued as soon as the rows are in registers.
// Bound by the LDS gather rate (64 random 4-byte reads per vector), roughly at par with the HBM rate of the codes.
constexpr int PQS_WAVES = 16;
constexpr int PQS_LUT_BYTES = 64 * 256 * 4;
constexpr int PQS_STAGE = 4096 + 256;   // 64 code rows + 64 x 4 descriptor bytes
constexpr int PQS_LDS = PQS_LUT_BYTES + PQS_WAVES * PQS_STAGE;

// (M0 is written and consumed inside one asm statement; this kernel issues all of its DMA through these two helpers and has no
// other use of M0 -- see the note at dma16_s in siglip_kernels.hip)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void pq_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);   // wave-uniform by construction; pin it to an SGPR
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void pq_dma4(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop

// GMAX: instead of one i64 per vector (8 B written per 68 B read, and read again by the selection), the wave keeps only
// the maximum of its 64 scores: out[group] (0.125 B per vector).  The r best vectors lie inside the r best groups by
// (maximum desc, group asc) -- a group ranked ahead of v's group holds a vector that precedes v, by score or, on a tie,
// by its lower id -- so the caller re-scores those r x 64 vectors (pq_adc_kernel, same arithmetic) and selects among them.
// LDS byte offset of table entry `code` (ENTRY_SHIFT = log2 of the entry size) for byte BB of the code word w: (byte BB of w) << shift
// in ONE instruction (SDWA byte select); hipcc emits v_bfe_u32 + v_lshl_add_u32, and the scan kernels are VALU-bound next to their
// LDS gathers (profiles/r03_pmc_pq_scan.txt).
template <int BB> __device__ __forceinline__ uint32_t code_offset(uint32_t w, uint32_t shift) {
    uint32_t r;
    if constexpr (BB == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(shift), "v"(w));
    else if constexpr (BB == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(shift), "v"(w));
    else if constexpr (BB == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(shift), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(shift), "v"(w));
    return r;
}
// an LDS location by its byte address (no symbol: hipcc adds the -- zero -- address of the dynamic LDS array with a v_add per gather)
template <typename T> __device__ __forceinline__ const __attribute__((address_space(3))) T* lds_at(uint32_t addr) {
    return reinterpret_cast<const __attribute__((address_space(3))) T*>((uintptr_t)addr);
}
// the scan kernels address their table from LDS address 0: they declare no static LDS, so the dynamic array starts there; anything else
// must stop the kernel, not return wrong maxima
__device__ __forceinline__ void require_lds_base_zero(const void* smem_base) {
    if ((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)smem_base != 0u) __builtin_trap();
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// the 16 table entries of code words w[4 * Q .. 4 * Q + 3] (chunks 16 Q .. 16 Q + 15): all offsets, then all gathers
template <typename T, int ENTRY> __device__ __forceinline__ void gather16(const uint32_t (&w)[16], int Q, uint32_t shift, T (&e)[16]) {
    uint32_t off[16];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        off[a * 4 + 0] = code_offset<0>(w[Q * 4 + a], shift);
        off[a * 4 + 1] = code_offset<1>(w[Q * 4 + a], shift);
        off[a * 4 + 2] = code_offset<2>(w[Q * 4 + a], shift);
        off[a * 4 + 3] = code_offset<3>(w[Q * 4 + a], shift);
    }
#pragma unroll
    for (int c = 0; c < 16; c++) e[c] = *lds_at<T>(off[c] + (Q * 16 + c) * 256 * ENTRY);
}

template <bool GMAX>
__global__ __launch_bounds__(PQS_WAVES * 64) void pq_scan64_kernel(const float* __restrict__ lut, const uint8_t* __restrict__ codes,
                                                                  size_t n, const uint8_t* __restrict__ desc /* [n][4] or null */,
                                                                  const float* __restrict__ scales, int64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_lut = reinterpret_cast<float*>(smem);
    require_lds_base_zero(smem);
    for (int e = threadIdx.x; e < 64 * 256; e += blockDim.x) s_lut[e] = lut[e];
    float sc[4] = {0.f, 0.f, 0.f, 0.f};
    if (desc)
        for (int j = 0; j < 4; j++) sc[j] = scales[j];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* stage = smem + PQS_LUT_BYTES + wave * PQS_STAGE;
    const uint32_t stage_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)stage;
    // DMA instruction j covers rows 16j .. 16j+15 of the group: lane -> row 16j + lane/4, LDS slot lane & 3 <- piece slot ^ f(row)
    uint32_t voff[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = 16 * j + (lane >> 2);
        voff[j] = (uint32_t)(r * 64 + (((lane & 3) ^ ((r >> 3) & 3)) * 16));
    }
    const size_t ngroups = (n + 63) / 64;
    const size_t stride = (size_t)gridDim.x * PQS_WAVES;
    size_t grp = (size_t)blockIdx.x * PQS_WAVES + wave;
    // every VMEM operation of the loop but the result store is an LDS-DMA issued here (descriptor bytes included), so
    // the waits can be counted by hand: the 4 (+1) DMAs of a group are older than the previous group's store
    auto issue = [&](size_t gi) {
        const uint8_t* base = codes + gi * 4096;   // the allocations carry slack for the last, partial group
#pragma unroll
        for (int j = 0; j < 4; j++) pq_dma16(base, voff[j], stage_lds + j * 1024);
        if (desc) pq_dma4(desc + gi * 256, (uint32_t)(lane * 4), stage_lds + 4096);
    };
    if (grp < ngroups) issue(grp);
    const int rsw = (lane >> 3) & 3;
    bool first = true;
    for (; grp < ngroups; grp += stride) {
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // the one younger operation is the previous result store
        first = false;
        uint4 w4[4];
#pragma unroll
        for (int p = 0; p < 4; p++) w4[p] = *reinterpret_cast<const uint4*>(stage + lane * 64 + ((p ^ rsw) * 16));
        const uint32_t dw = desc ? *reinterpret_cast<const uint32_t*>(stage + 4096 + lane * 4) : 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // rows are in registers: the staging area may be refilled
        if (grp + stride < ngroups) issue(grp + stride);
        const uint32_t w[16] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w,
                                w4[2].x, w4[2].y, w4[2].z, w4[2].w, w4[3].x, w4[3].y, w4[3].z, w4[3].w};
        float s = 0.0f;
        uint32_t sh2 = 2;
        asm volatile("" : "+v"(sh2));   // the shift amount lives in a VGPR (SDWA takes no literal)
        // the table starts at LDS address 0 (checked above): an entry's byte offset IS its address, the chunk goes into the ds_read's immediate
#pragma unroll
        for (int Q = 0; Q < 4; Q++) {
            float e[16];
            gather16<float, 4>(w, Q, sh2, e);
#pragma unroll
            for (int c = 0; c < 16; c++) s = add_rn(s, e[c]);     // chunk order 0 .. 63 (asymmetric_dot_product, vector.rs:387-405)
        }
        const size_t v = grp * 64 + lane;
        int64_t r = scale_dot_result(s);
        if (desc) {
#pragma unroll
            for (int j = 0; j < 4; j++) r += scale_dot_result(sc[j] * (float)((dw >> (8 * j)) & 0xffu));
        }
        if (GMAX) {
            if (v >= n) r = INT64_MIN;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const int64_t other = __shfl_xor(r, o);
                r = other > r ? other : r;
            }
            if (lane == 0) out[grp] = r;      // one store per group either way: the counted waits above are unchanged
        } else if (v < n) {
            out[v] = r;
        }
    }
}

// Two queries per pass over the codes (round 3).  The one-query scan above is bound by its LDS gathers and the VALU work around
// them (profiles/r03_pmc_pq_scan.txt: LDS 79 % busy, 46 % of the kernel's cycles are bank-conflict cycles of the 64 random 4-byte
// reads per vector; VALU 70 %), and all of that buys ONE query.  Here the table holds both queries' entries side by side --
// s_lut2[chunk * 256 + code] = {q0, q1} (128 KiB) -- so the same 64 gathers per vector, now ds_read_b64, and the same pass over the
// codes serve two queries: per query, half the LDS cycles and half the HBM bytes.  Each query's sum is still 64 sequential f32 adds
// in chunk order (asymmetric_dot_product, vector.rs:387-405) and the descriptor bias is added after the conversion, so both group
// maxima are bit-identical to pq_scan64_kernel<true>'s for that query (tests: test_group_maxima_equal_the_gathered_scores).
// No LDS is left for staging the codes (128 of 160 KiB are table): every lane loads its own 64-byte code row straight from global
// memory (four 16-byte loads at a 64-byte stride; lane pairs share a 128-byte line, and the four loads of a wave touch the same 32
// lines back to back), one group ahead.  NW = 12 waves (register-limited; 8 waves 8 % slower, 16 spill).  A form with the round-2
// LDS-DMA staging (8 waves x 4 KiB) measured 3 % slower at 2e7 codes and was dropped.
constexpr int PQ2_LUT_BYTES = 64 * 256 * 8;
constexpr int PQ2_WAVES = 12;
template <int NW>
__global__ __launch_bounds__(NW * 64) void pq_scan64x2_kernel(const float* __restrict__ lut0, const float* __restrict__ lut1,
                                                              const uint8_t* __restrict__ codes, size_t n,
                                                              const uint8_t* __restrict__ desc, const float* __restrict__ scales,
                                                              int64_t* __restrict__ out0, int64_t* __restrict__ out1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* s_lut2 = reinterpret_cast<float2*>(smem);
    require_lds_base_zero(smem);
    for (int e = threadIdx.x; e < 64 * 256; e += blockDim.x) s_lut2[e] = make_float2(lut0[e], lut1[e]);
    float sc[4] = {0.f, 0.f, 0.f, 0.f};
    if (desc)
        for (int j = 0; j < 4; j++) sc[j] = scales[j];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t ngroups = (n + 63) / 64;
    const size_t stride = (size_t)gridDim.x * NW;
    size_t grp = (size_t)blockIdx.x * NW + wave;
    auto load_rows = [&](size_t gi, uint4 (&w4)[4], uint32_t& dw) {
        const size_t v = gi * 64 + lane;            // the allocations carry slack for the last, partial group
        typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
        const u32x4v* row = reinterpret_cast<const u32x4v*>(codes + v * 64);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const u32x4v t = __builtin_nontemporal_load(row + p);
            w4[p] = uint4{t.x, t.y, t.z, t.w};
        }
        dw = (desc && v < n) ? reinterpret_cast<const uint32_t*>(desc)[v] : 0u;
    };
    uint4 nx[4];
    uint32_t dw_next = 0;
    if (grp < ngroups) load_rows(grp, nx, dw_next);
    for (; grp < ngroups; grp += stride) {
        uint4 w4[4] = {nx[0], nx[1], nx[2], nx[3]};
        const uint32_t dw = dw_next;
        if (grp + stride < ngroups) load_rows(grp + stride, nx, dw_next);
        const uint32_t w[16] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w,
                                w4[2].x, w4[2].y, w4[2].z, w4[2].w, w4[3].x, w4[3].y, w4[3].z, w4[3].w};
        float s0 = 0.0f, s1 = 0.0f;
        uint32_t sh3 = 3;
        asm volatile("" : "+v"(sh3));   // the shift amount lives in a VGPR (SDWA takes no literal)
#pragma unroll
        for (int Q = 0; Q < 4; Q++) {
            f32x2 e[16];
            gather16<f32x2, 8>(w, Q, sh3, e);
#pragma unroll
            for (int c = 0; c < 16; c++) { s0 = add_rn(s0, e[c].x); s1 = add_rn(s1, e[c].y); }   // chunk order, per query
        }
        const size_t v = grp * 64 + lane;
        int64_t r0 = scale_dot_result(s0), r1 = scale_dot_result(s1);
        if (desc) {
            int64_t bias = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) bias += scale_dot_result(sc[j] * (float)((dw >> (8 * j)) & 0xffu));
            r0 += bias; r1 += bias;
        }
        if (v >= n) { r0 = INT64_MIN; r1 = INT64_MIN; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int64_t o0 = __shfl_xor(r0, o), o1 = __shfl_xor(r1, o);
            r0 = o0 > r0 ? o0 : r0;
            r1 = o1 > r1 ? o1 : r1;
        }
        if (lane == 0) { out0[grp] = r0; out1[grp] = r1; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Four queries per pass: an INTEGER nomination scan under a certificate (round 3).
// The flat scan only nominates groups of 64 vectors; the nominated vectors are re-scored in the reference's arithmetic by
// pq_adc_kernel and the answer is taken among those.  So the scan itself need not reproduce the f32 sums -- it needs a score that
// brackets them.  Per query the table is quantised to 12 bits against ONE step delta (so that integer sums are comparable):
//     e[c][v] = round((lut[c][v] - lo_c) / delta) in [0, 4095],  delta = max_c (hi_c - lo_c) / 4095
// and the four descriptor terms sc_j * v (query_disk_index.rs:135-142) become four more chunks of the same table (<= 16383 each).
// Four queries' entries sit side by side in 8 bytes, so ONE ds_read_b64 gather serves four queries, and their sums ride in packed
// 16-bit lanes (v_pk_add_u16: 16 x 4095 < 2^16, widened to 32 bits every 16 chunks): per query a quarter of the LDS cycles and of the
// HBM bytes of the one-query scan.  With S = the integer sum of a vector and C = sum_c lo_c, the real-valued ADC score X satisfies
// |X - (delta S + C)| <= 34 delta (68 roundings of at most delta / 2), and the reference-order i64 score R satisfies
// |R / 2^32 - X| <= 64 u A + bias rounding + 5 truncations (u = 2^-24, A = sum_c max |lut[c]|): together eps (pq4_table_kernel).
// Certificate (pq4_certify_kernel): nominate R' groups by their maximum S, re-score them exactly, take the exact top-r; every vector
// outside the nominated groups has S <= g = the (R'+1)-th best group maximum, hence R / 2^32 <= delta g + C + eps.  If the r-th exact
// score is strictly above that, no excluded vector belongs to the top r.  Otherwise the query is repeated through the exact scan.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int PQ4_CHUNKS = 68;                     // 64 code chunks + 4 descriptor bytes
// LDS image of the four queries' table (round 4): the entry of (code v, chunk c) lies at byte (c / 32) * 65536 + v * 256 +
// (c % 32) * 8 -- code-major, so that the bank pair of a ds_read_b64 is c mod 32 whatever the code, and with the code in byte 1
// of the address, so that ONE v_perm_b32 assembles an address from a code byte and a lane constant -- and the entry of
// (descriptor value v, byte d) at 131072 + v * 40 + d * 8.  An entry is 8 bytes, two per query: the value's low 6 bits and the
// bits above them (<= 63 for the 12-bit code entries, <= 255 for the 14-bit descriptor entries), each XOR 0x80 (the matrix
// core reads the bytes as signed: stored value = byte - 128).
// Eight queries per pass: a code entry holds one byte per query; a DESCRIPTOR entry keeps 14 bits per query in 16 bytes (bytes 0-7
// the low 6 bits, bytes 8-15 the bits above, rows of 80 bytes) and its MFMA uses a second B operand with weights 1 / 64 -- a
// descriptor term spans 255 |scale|, often ten times a chunk's range, and 8 bits for it would set the step for everything.
constexpr int PQ4_CODE_BYTES = 256 * 64 * 8;
template <int NQ> struct Pq4Layout {
    static constexpr int DESC_ENTRY = NQ == 4 ? 8 : 16;
    static constexpr int DESC_STRIDE = NQ == 4 ? 40 : 80;
    static constexpr int TABLE_BYTES = PQ4_CODE_BYTES + 256 * DESC_STRIDE;   // 138 KiB / 148 KiB
};
constexpr int PQ4_TABLE_BYTES = Pq4Layout<8>::TABLE_BYTES;   // the larger image: what callers reserve
constexpr int PQ4_WAVES = 16;

// per (chunk, query): minimum and maximum over the 256 table entries; a query with a NaN / infinite entry is flagged
__global__ __launch_bounds__(64) void pq4_minmax_kernel(const float* __restrict__ luts /* [NQ][64 * 256] */, int n_valid,
                                                       float* __restrict__ lohi /* [NQ][64][2] */, int* __restrict__ bad /* [NQ], zeroed */) {
    const int c = blockIdx.x, j = blockIdx.y, t = threadIdx.x;
    if (j >= n_valid) return;
    const float4 x = reinterpret_cast<const float4*>(luts + ((size_t)j * 64 + c) * 256)[t];
    const bool b = !(fabsf(x.x) <= 3.0e38f) || !(fabsf(x.y) <= 3.0e38f) || !(fabsf(x.z) <= 3.0e38f) || !(fabsf(x.w) <= 3.0e38f);
    float lo = fminf(fminf(x.x, x.y), fminf(x.z, x.w)), hi = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if (__ballot(b) != 0ull && t == 0) atomicOr(&bad[j], 1);
    if (t == 0) { lohi[(j * 64 + c) * 2] = lo; lohi[(j * 64 + c) * 2 + 1] = hi; }
}

// block c (0 .. 63: code chunks, 64 .. 67: descriptor bytes), thread v (code / descriptor value): the NQ queries' entries of
// (v, c), one 8-byte store.  Every block works out the queries' parameters for itself (64-term loops in double, in the same
// order as ever: the numbers are those of round 3's single-workgroup kernel); block 0 publishes them.
// NQ = 4: 12-bit code entries (14-bit descriptor entries), two bytes per query: the value's low 6 bits and the bits above.
// NQ = 8: 8-bit code entries, one byte per query (delta = the widest chunk range / 255); descriptor entries keep 14 bits (16 bytes).
template <int NQ>
__global__ __launch_bounds__(256) void pq4_quant_kernel(const float* __restrict__ luts, const float* __restrict__ scales, int n_valid,
                                                       const float* __restrict__ lohi, const int* __restrict__ bad,
                                                       unsigned long long* __restrict__ table, Pq4Params* __restrict__ params) {
    constexpr double CODE_MAX = NQ == 4 ? 4095.0 : 255.0, DESC_MAX = 16383.0;
    const int c = blockIdx.x, t = threadIdx.x;
    __shared__ double s_inv[NQ];
    __shared__ float s_lo[NQ];
    if (t < NQ) {
        const int j = t;
        double inv = 0.0;
        float lo_c = 0.0f;
        Pq4Params P{0.0, 0.0, 0.0, 0};
        if (j < n_valid) {
            double range = 0.0, a_sum = 0.0, c_sum = 0.0;
            for (int cc = 0; cc < 64; cc++) {
                const double lo = (double)lohi[(j * 64 + cc) * 2], hi = (double)lohi[(j * 64 + cc) * 2 + 1];
                range = fmax(range, hi - lo);
                a_sum += fmax(fabs(lo), fabs(hi));
                c_sum += lo;
            }
            int is_bad = bad[j];
            double delta = range / CODE_MAX, bias_err = 0.0, bias_abs = 0.0;
            if (scales) {
                for (int d = 0; d < 4; d++) {
                    const double sc = (double)scales[d];
                    delta = fmax(delta, fabs(sc) * 255.0 / DESC_MAX);
                    c_sum += fmin(0.0, sc * 255.0);                       // lo of descriptor chunk d
                    bias_err += fabs(sc) * 255.0 * 5.9604644775390625e-8;  // rounding of the f32 product sc * v
                    bias_abs += fabs(sc) * 255.0;
                    if (!(fabs(sc) <= 3.0e38)) is_bad = 1;
                }
            }
            // `(x * 2^32) as i64` saturates from |x| = 2^31 on: the bound below assumes it never does (unreachable for
            // normalised embeddings; such a query takes the exact scan)
            if (!(a_sum * 1.001 + bias_abs < 1073741824.0)) is_bad = 1;
            if (!(delta > 1e-300)) delta = 1e-300;
            inv = 1.0 / delta;
            // 34 delta (68 entry roundings) + f32 summation + bias products + five i64 truncations, with a relative cushion for the
            // double arithmetic of this bound itself
            const double eps = (34.0 * delta + 64.0 * 5.9604644775390625e-8 * 1.01 * a_sum + bias_err + 5.0 / 4294967296.0) * (1.0 + 1e-9) + 1e-300;
            P = Pq4Params{delta, c_sum, eps, is_bad ? 0 : 1};
            if (c < 64) lo_c = lohi[(j * 64 + c) * 2];
        }
        s_inv[j] = inv;
        s_lo[j] = lo_c;
        if (c == 0) params[j] = P;
    }
    __syncthreads();
    unsigned long long e8 = 0ull, e8hi = 0ull;
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        double q = 0.0;
        if (j < n_valid) {
            if (c < 64) {
                q = ((double)luts[((size_t)j * 64 + c) * 256 + t] - (double)s_lo[j]) * s_inv[j];
                q = q < 0.0 ? 0.0 : (q > CODE_MAX ? CODE_MAX : q);
            } else if (scales) {
                const double sc = (double)scales[c - 64];
                q = (sc * (double)t - fmin(0.0, sc * 255.0)) * s_inv[j];
                q = q < 0.0 ? 0.0 : (q > DESC_MAX ? DESC_MAX : q);
            }
        }
        const unsigned long long e = (unsigned long long)(uint16_t)__double2int_rn(q);
        if constexpr (NQ == 4) {
            e8 |= (((e & 63ull) | ((e >> 6) << 8)) ^ 0x8080ull) << (16 * j);
        } else if (c < 64) {
            e8 |= (e ^ 0x80ull) << (8 * j);
        } else {
            e8 |= ((e & 63ull) ^ 0x80ull) << (8 * j);
            e8hi |= ((e >> 6) ^ 0x80ull) << (8 * j);
        }
    }
    if (c < 64) {
        table[((size_t)(c >> 5) * 65536 + (size_t)t * 256 + (size_t)(c & 31) * 8) / 8] = e8;
    } else {
        const size_t off = (size_t)PQ4_CODE_BYTES + (size_t)t * Pq4Layout<NQ>::DESC_STRIDE + (size_t)(c - 64) * Pq4Layout<NQ>::DESC_ENTRY;
        table[off / 8] = e8;
        if constexpr (NQ == 8) table[off / 8 + 1] = e8hi;
    }
}

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));

// out[g * NQ + j] = max over the 64 vectors of group g of query j's integer sum (0 past the end of the codes).
//
// Round 4: the sums are taken by the matrix cores and the gathers are bank-conflict free.
//   * A wave handles 16 vectors at a time; lane (v = lane % 16, g = lane / 16) owns bytes 16g .. 16g+15 of vector v's code row,
//     i.e. chunks 16g .. 16g+15, and gathers their sixteen 8-byte entries (four queries each).
//   * The integer sum is order-free, so WHICH of its sixteen chunks a lane fetches at step s is free: lane (v, g) takes chunk
//     16g + (s + v) % 16.  With the table code-major (bank pair = chunk % 32) the 32 lanes of a ds_read_b64 group then hit 32
//     different bank pairs at every step, whatever the codes (round 3: 2.4-way conflicts on random codes, 46 % of the kernel's
//     cycles).  The lane's 16 code bytes arrive by ONE dwordx4 load (a wave covers 1 KiB contiguously) and are rotated left by v
//     bytes in registers -- two select stages for the dwords, four v_alignbyte_b32 for the bytes -- so that the byte for step s
//     sits at the static position s.  (Loading the dwords in rotated order instead -- 16 scattered dword loads per lane and
//     group -- made the texture addresser the bottleneck: TA busy 75 %, kernel 20 % slower than round 3's.)
//   * A gather address is ONE instruction: v_perm_b32 puts the code byte into byte 1 and the lane's chunk offset (a per-lane
//     constant per step) into bytes 0 and 2.
//   * Two gathered entries are sixteen signed bytes = one lane's A operand of v_mfma_i32_16x16x64_i8 (row v, any 16 of the 64
//     k positions).  B[k][q] = 1 for k % 8 == 2q (low 6 bits) and 64 for k % 8 == 2q + 1 (the bits above): column q of the
//     16 x 16 result IS query q's integer sum (minus a constant for the signed bytes) -- 8 MFMAs (+1 for the descriptor entries)
//     replace 2 x 68 packed adds per vector, and nothing is left to combine afterwards.
//   VALU instructions per 64 vectors: 340 (round 3) -> 344 (first matrix-core form: two-instruction addresses, low/high byte
//   columns combined by DPP) -> this form; LDS cycles per gather instruction 5.8 -> 2.3.
//   NQ = 8 (round 4): the pass is HBM-bound with every on-chip unit under 55 %, so the same instruction stream serves EIGHT
//   queries when an entry holds one byte per query (8-bit tables, B = identity): twice the queries per byte of codes read.  The
//   coarser step (delta x 16) widens the certificate's band; the caller nominates more groups for it (api_pq.hip).
template <int NW, int NQ>
__global__ __launch_bounds__(NW * 64) void pq_scan64x4_kernel(const uint4* __restrict__ table, const uint8_t* __restrict__ codes, size_t n,
                                                             const uint8_t* __restrict__ desc, uint32_t* __restrict__ out, size_t n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    require_lds_base_zero(smem);
    {
        uint4* s_tab = reinterpret_cast<uint4*>(smem);
        for (int e = threadIdx.x; e < Pq4Layout<NQ>::TABLE_BYTES / 16; e += blockDim.x) s_tab[e] = table[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int v = lane & 15, g = lane >> 4, a = v >> 2;
    const uint32_t b = (uint32_t)(v & 3);
    // chunk 16g + (s + v) % 16 as address bytes: byte 2 = chunk / 32, byte 0 = (chunk % 32) * 8; byte 1 will be the code
    uint32_t choff[16];
#pragma unroll
    for (int s = 0; s < 16; s++) choff[s] = (uint32_t)(((g >> 1) << 16) | (128 * (g & 1) + 8 * ((s + v) & 15)));
    // B[k][col], lane (col = v, k block g) holds k = 16g .. 16g+15.  NQ = 4: k % 8 == 2 col -> 1, == 2 col + 1 -> 64 (col < 4);
    // NQ = 8: k % 8 == col -> 1 (col < 8)
    v4i32 bop = {0, 0, 0, 0};
    if constexpr (NQ == 4) {
        if (v < 4) {
            const int w = 0x4001 << (16 * (v & 1));
            if (v < 2) { bop.x = w; bop.z = w; } else { bop.y = w; bop.w = w; }
        }
    } else {
        if (v < 8) {
            const int w = 1 << (8 * (v & 3));
            if (v < 4) { bop.x = w; bop.z = w; } else { bop.y = w; bop.w = w; }
        }
    }
    // this lane's 16 bytes inside a 16-vector block of code rows (1 KiB): the wave's dwordx4 loads cover the block contiguously
    const uint32_t roff = (uint32_t)(v * 64 + 16 * g);
    const bool a1 = (a & 1) != 0, a2 = (a & 2) != 0;
    const uint32_t doff = (uint32_t)(PQ4_CODE_BYTES + Pq4Layout<NQ>::DESC_ENTRY * g);   // descriptor entry of byte g: + value * DESC_STRIDE
    // B operand of the descriptor MFMA when an entry is 16 bytes (NQ = 8): k % 16 == col -> 1 (low 6 bits), == 8 + col -> 64
    v4i32 bop_d = bop;
    if constexpr (NQ == 8) {
        bop_d = v4i32{0, 0, 0, 0};
        if (v < 8) {
            const int w1 = 1 << (8 * (v & 3)), w64 = 64 << (8 * (v & 3));
            if (v < 4) { bop_d.x = w1; bop_d.z = w64; } else { bop_d.y = w1; bop_d.w = w64; }
        }
    }
    const int n_entries = desc ? 68 : 64;
    const int bias = NQ == 4 ? 128 * 65 * n_entries : 128 * 64 + (desc ? 4 * 128 * 65 : 0);
    const size_t stride = (size_t)gridDim.x * NW;
    size_t grp = (size_t)blockIdx.x * NW + wave;
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    u32x4v nx[4];
    uint32_t dnx[4];
    auto load_rows = [&](size_t gi) {       // the allocations carry slack for the last, partial group
        const uint8_t* base = codes + gi * 4096 + roff;
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
            nx[sb] = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(base + sb * 1024));
            const size_t vec = gi * 64 + sb * 16 + v;
            dnx[sb] = (desc && vec < n) ? reinterpret_cast<const uint32_t*>(desc)[vec] : 0u;
        }
    };
    if (grp < n_groups) load_rows(grp);
    for (; grp < n_groups; grp += stride) {
        u32x4v x[4];
        uint32_t dw[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { x[i] = nx[i]; dw[i] = dnx[i]; }
        if (grp + stride < n_groups) load_rows(grp + stride);
        int rows[16];
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {      // two blocks of 16 vectors at a time: two independent accumulator chains
            uint32_t rr[2][4];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                // rotate the lane's 16 bytes left by v = 4a + b bytes: dwords by a (two select stages), bytes by b (funnel shifts)
                const u32x4v xx = x[2 * pr + u];
                const uint32_t z0 = a1 ? xx.y : xx.x, z1 = a1 ? xx.z : xx.y, z2 = a1 ? xx.w : xx.z, z3 = a1 ? xx.x : xx.w;
                const uint32_t w[4] = {a2 ? z2 : z0, a2 ? z3 : z1, a2 ? z0 : z2, a2 ? z1 : z3};
#pragma unroll
                for (int i = 0; i < 4; i++) rr[u][i] = __builtin_amdgcn_alignbyte(w[(i + 1) & 3], w[i], b);
            }
            v4i32 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                u32x2v e[2][8];
#pragma unroll
                for (int t = 0; t < 8; t++)
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int s = 8 * h + t;
                        // address bytes: [0] chunk offset, [1] code byte s of the rotated row, [2] chunk half, [3] zero
                        const uint32_t addr = __builtin_amdgcn_perm(rr[u][s >> 2], choff[s], 0x0c020000u | ((4u + (s & 3)) << 8));
                        e[u][t] = *lds_at<u32x2v>(addr);
                    }
#pragma unroll
                for (int t = 0; t < 8; t += 2)
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const v4i32 av = {(int)e[u][t].x, (int)e[u][t].y, (int)e[u][t + 1].x, (int)e[u][t + 1].y};
                        acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bop, acc[u], 0, 0, 0);
                    }
            }
            if (desc) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint32_t dv = (dw[2 * pr + u] >> (8 * g)) & 0xffu;
                    if constexpr (NQ == 4) {
                        const u32x2v e0 = *lds_at<u32x2v>(dv * (uint32_t)Pq4Layout<NQ>::DESC_STRIDE + doff);
                        const v4i32 av = {(int)e0.x, (int)e0.y, 0, 0};
                        acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bop, acc[u], 0, 0, 0);
                    } else {
                        const v4i32 av = *lds_at<v4i32>(dv * (uint32_t)Pq4Layout<NQ>::DESC_STRIDE + doff);
                        acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bop_d, acc[u], 0, 0, 0);
                    }
                }
            }
            // acc[u][i] = column v of row 4g + i of block 2 pr + u: in lanes v = q < 4 query q's sum minus `bias`
#pragma unroll
            for (int u = 0; u < 2; u++) {
                rows[(2 * pr + u) * 4 + 0] = acc[u].x; rows[(2 * pr + u) * 4 + 1] = acc[u].y;
                rows[(2 * pr + u) * 4 + 2] = acc[u].z; rows[(2 * pr + u) * 4 + 3] = acc[u].w;
            }
        }
        if (grp * 64 + 64 > n) {     // the last, partial group: vectors past the end count as all-zero sums
#pragma unroll
            for (int j = 0; j < 16; j++)
                if (grp * 64 + (j >> 2) * 16 + 4 * g + (j & 3) >= n) rows[j] = -bias;
        }
        int best = max(rows[0], rows[1]);
#pragma unroll
        for (int j = 2; j < 16; j += 2) best = max(best, max(rows[j], rows[j + 1]));     // v_max3_i32
        // lanes (v = q, g = 0 .. 3) hold the maxima of their rows: the maximum over the four g by two row swaps (gfx950's
        // v_permlane32_swap / v_permlane16_swap), then lanes q < NQ store the group's NQ results side by side: out is group-major,
        // [n_groups][NQ] (round 4's first form read them out with 4 NQ v_readlane and stored them from one lane: 265 -> VALU
        // instructions per group at NQ = 8)
        typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
        const u32x2s h = __builtin_amdgcn_permlane32_swap((uint32_t)best, (uint32_t)best, false, false);
        best = max((int)h.x, (int)h.y);
        const u32x2s r2 = __builtin_amdgcn_permlane16_swap((uint32_t)best, (uint32_t)best, false, false);
        best = max((int)r2.x, (int)r2.y);
        if (lane < NQ) out[grp * NQ + lane] = (uint32_t)(best + bias);
    }
}

// flag[0] = 1 when the exact top-r found among the nominated groups is provably the exact top-r of ALL vectors (see the header)
// one thread per query j: group_keys [nq][n_sel], top_ids / top_scores [nq][top_stride]
__global__ void pq4_certify_kernel(const Pq4Params* __restrict__ params, const uint32_t* __restrict__ group_keys /* best first */,
                                   int n_nominated, int n_sel, const uint32_t* __restrict__ top_ids, const int64_t* __restrict__ top_scores,
                                   size_t top_stride, int r, int nq, int* __restrict__ flag) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nq) return;
    int ok = params[j].ok;
    if (ok && n_sel > n_nominated) {     // a best excluded group exists
        const double ub = params[j].delta * (double)group_keys[(size_t)j * n_sel + n_nominated] + params[j].c + params[j].eps;
        // the r-th best nominated vector must exist and beat every excluded one strictly
        ok = top_ids[j * top_stride + r - 1] != ID_NONE && (double)top_scores[j * top_stride + r - 1] > ub * 4294967296.0;
    }
    flag[j] = ok;
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

__global__ void f16_to_f32_kernel(const uint16_t* __restrict__ in, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (float)__builtin_bit_cast(_Float16, in[i]);   // exact widening (half::f16::to_f32)
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
    static const bool old_t = MSE_DEV_KNOB("MSE_PQ_OLDTRANSFORM");
    // the register-tiled kernel puts 64 x 64 outputs on a workgroup: a batch of 32-64 query vectors is 18 workgroups on a 256-CU
    // part (262 us); below 256 vectors the one-output-per-thread kernel (16 x 16 outputs per workgroup, 144+ workgroups) is the
    // faster one.  Same sums in the same order either way.
    if (n >= 256 && !old_t) {
        dim3 grid4((d + T4_B - 1) / T4_B, (unsigned)((n + T4_B - 1) / T4_B));
        hipLaunchKernelGGL(pq_transform_tiled_kernel, grid4, dim3(256), 0, stream, T, d, x, n, out);
        MSE_HIP_TRY(hipGetLastError());
        return 0;
    }
    dim3 grid((d + TT - 1) / TT, (unsigned)((n + TT - 1) / TT));
    hipLaunchKernelGGL(pq_transform_kernel, grid, dim3(TT * TT), 0, stream, T, d, x, n, out);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_pq_transform_vec(const float* Tt, int d, const float* x, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(pq_transform_vec_kernel, dim3((d + 63) / 64), dim3(64), (size_t)d * 4, stream, Tt, d, x, out);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_pq_lut(const float* centroids, int n_centroids, int d, int dpc, const float* t, float* lut,
                  hipStream_t stream) {
    return launch_pq_lut_batch(centroids, n_centroids, d, dpc, t, 1, lut, stream);
}
int launch_pq_lut_batch(const float* centroids, int n_centroids, int d, int dpc, const float* t, size_t nq, float* lut,
                        hipStream_t stream) {
    const int total = (d / dpc) * n_centroids;
    for (size_t q0 = 0; q0 < nq; q0 += 65535) {
        const size_t m = nq - q0 < 65535 ? nq - q0 : 65535;
        hipLaunchKernelGGL(pq_lut_kernel, dim3((total + 255) / 256, (unsigned)m), dim3(256), 0, stream, centroids, n_centroids, d, dpc,
                           t + q0 * d, lut + q0 * total);
        MSE_HIP_TRY(hipGetLastError());
    }
    return 0;
}
int launch_pq_quantize(const float* centroids, int n_centroids, int d, int dpc, const float* t, size_t n,
                       uint8_t* codes, hipStream_t stream) {
    const size_t total = n * (size_t)(d / dpc);
    if (total == 0) return 0;
    static const bool old_q = MSE_DEV_KNOB("MSE_PQ_OLDQUANT");
    if (dpc == 18 && n_centroids <= 256 && !old_q) {       // the reference's codec shape (aopq_train.py:9-13)
        constexpr int VPT = 4;
        const size_t per_block = 256 * VPT;
        dim3 grid((unsigned)((n + per_block - 1) / per_block), (unsigned)(d / dpc));
        hipLaunchKernelGGL((pq_quantize_tiled_kernel<18, VPT>), grid, dim3(256), (size_t)n_centroids * 18 * 4, stream, centroids,
                           n_centroids, d, t, n, codes);
        MSE_HIP_TRY(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(pq_quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, centroids,
                       n_centroids, d, dpc, t, n, codes);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
bool pq_scan_gmax_supported(int n_chunks, int n_centroids, const uint8_t* desc, int n_desc, const float* scales) {
    return n_chunks == 64 && n_centroids == 256 && (!(desc && scales) || n_desc == 4);
}
// workgroups of a flat scan: every CU but one per XCD (8 XCDs on MI355X)
static size_t scan_cus(int n_cu) { return n_cu >= 64 ? (size_t)(n_cu - 8) : (size_t)n_cu; }   // (16, 24, 32 spare CUs: no gain, measured)

// group maxima of a full scan: gmax[g] = max ADC score (+ descriptor bias) of vectors 64g .. 64g+63 (INT64_MIN past the end)
int launch_pq_scan_gmax(const float* lut, const uint8_t* codes, size_t n, const uint8_t* desc, const float* scales,
                        int64_t* gmax, int n_cu, hipStream_t stream) {
    if (n == 0) return 0;
    MSE_DYN_LDS(pq_scan64_kernel<true>, PQS_LDS);
    const size_t groups = (n + 63) / 64;
    // one 133-KiB workgroup per CU, on all but ONE CU PER XCD: the kernels of another query's tail (radix selects, ~50 KiB of LDS
    // each) can then run beside a scan instead of queueing behind all of its workgroups.  Workgroups go to the eight XCDs round
    // robin, so the spare CUs must be spread the same way: with 252 workgroups on 256 CUs XCDs 0-3 were full, and a four-workgroup
    // select whose workgroups landed there waited for the whole scan (1.99 ms instead of 0.07, profiles/r04_pq_scan_stats.txt).
    const size_t cus = scan_cus(n_cu);
    const unsigned blocks = (unsigned)std::min<size_t>((groups + PQS_WAVES - 1) / PQS_WAVES, cus);
    hipLaunchKernelGGL(pq_scan64_kernel<true>, dim3(blocks), dim3(PQS_WAVES * 64), PQS_LDS, stream, lut, codes, n,
                       (desc && scales) ? desc : nullptr, scales, gmax);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

// the same for TWO queries in one pass over the codes (pq_scan64x2_kernel); scales are shared by the two queries (one request)
int launch_pq_scan_gmax2(const float* lut0, const float* lut1, const uint8_t* codes, size_t n, const uint8_t* desc,
                         const float* scales, int64_t* gmax0, int64_t* gmax1, int n_cu, hipStream_t stream) {
    if (n == 0) return 0;
    const size_t groups = (n + 63) / 64;
    const size_t cus = scan_cus(n_cu);   // one CU per XCD stays free for the previous pair's tail (see above)
    MSE_DYN_LDS(pq_scan64x2_kernel<PQ2_WAVES>, PQ2_LUT_BYTES);
    const unsigned blocks = (unsigned)std::min<size_t>((groups + PQ2_WAVES - 1) / PQ2_WAVES, cus);
    hipLaunchKernelGGL(pq_scan64x2_kernel<PQ2_WAVES>, dim3(blocks), dim3(PQ2_WAVES * 64), PQ2_LUT_BYTES, stream, lut0, lut1, codes, n,
                       (desc && scales) ? desc : nullptr, scales, gmax0, gmax1);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- four queries per pass (integer nomination under a certificate; header above pq4_table_kernel) ----
size_t pq4_table_bytes() { return PQ4_TABLE_BYTES + 8 * 64 * 2 * 4 + 64; }   // LDS image + the table build's scratch
// nq = 4 (12-bit tables) or 8 (8-bit tables): luts [nq][64 * 256], params [nq]
int launch_pq4_table(const float* luts, const float* scales, int n_valid, void* table, Pq4Params* params, hipStream_t stream, int nq) {
    if (nq != 4 && nq != 8) return fail("pq table: 4 or 8 queries per pass");
    // scratch behind the table image: [8][64][2] f32 minima / maxima, [8] i32 flags (pq4_table_bytes() reserves it)
    char* tail = reinterpret_cast<char*>(table) + PQ4_TABLE_BYTES;
    float* lohi = reinterpret_cast<float*>(tail);
    int* bad = reinterpret_cast<int*>(tail + 8 * 64 * 2 * 4);
    MSE_HIP_TRY(hipMemsetAsync(bad, 0, 32, stream));
    hipLaunchKernelGGL(pq4_minmax_kernel, dim3(64, nq), dim3(64), 0, stream, luts, n_valid, lohi, bad);
    if (nq == 4)
        hipLaunchKernelGGL(pq4_quant_kernel<4>, dim3(PQ4_CHUNKS), dim3(256), 0, stream, luts, scales, n_valid, lohi, bad,
                           reinterpret_cast<unsigned long long*>(table), params);
    else
        hipLaunchKernelGGL(pq4_quant_kernel<8>, dim3(PQ4_CHUNKS), dim3(256), 0, stream, luts, scales, n_valid, lohi, bad,
                           reinterpret_cast<unsigned long long*>(table), params);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_pq_scan_gmax4(const void* table, const uint8_t* codes, size_t n, const uint8_t* desc, uint32_t* gmax, int n_cu,
                         hipStream_t stream, int nq) {
    if (n == 0) return 0;
    if (nq != 4 && nq != 8) return fail("pq scan: 4 or 8 queries per pass");
    const size_t groups = (n + 63) / 64;
    const size_t cus = scan_cus(n_cu);
    const unsigned blocks = (unsigned)std::min<size_t>((groups + PQ4_WAVES - 1) / PQ4_WAVES, cus);
    if (nq == 4) {
        MSE_DYN_LDS((pq_scan64x4_kernel<PQ4_WAVES, 4>), Pq4Layout<4>::TABLE_BYTES);
        hipLaunchKernelGGL((pq_scan64x4_kernel<PQ4_WAVES, 4>), dim3(blocks), dim3(PQ4_WAVES * 64), Pq4Layout<4>::TABLE_BYTES, stream,
                           reinterpret_cast<const uint4*>(table), codes, n, desc, gmax, groups);
    } else {
        MSE_DYN_LDS((pq_scan64x4_kernel<PQ4_WAVES, 8>), Pq4Layout<8>::TABLE_BYTES);
        hipLaunchKernelGGL((pq_scan64x4_kernel<PQ4_WAVES, 8>), dim3(blocks), dim3(PQ4_WAVES * 64), Pq4Layout<8>::TABLE_BYTES, stream,
                           reinterpret_cast<const uint4*>(table), codes, n, desc, gmax, groups);
    }
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_pq4_certify(const Pq4Params* params, const uint32_t* group_keys, int n_nominated, int n_sel, const uint32_t* top_ids,
                       const int64_t* top_scores, size_t top_stride, int r, int nq, int* flag, hipStream_t stream) {
    hipLaunchKernelGGL(pq4_certify_kernel, dim3(1), dim3(64), 0, stream, params, group_keys, n_nominated, n_sel, top_ids, top_scores,
                       top_stride, r, nq, flag);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_pq_adc(const float* lut, int n_chunks, int n_centroids, const uint8_t* codes, size_t n_codes,
                  const uint32_t* ids, size_t n, const uint8_t* desc, int n_desc, const float* scales, int64_t* out,
                  int n_cu, hipStream_t stream, int nq, size_t q_stride) {
    if (n == 0 || nq <= 0) return 0;
    if (nq > 1 && !ids) return fail("pq_adc: several queries per launch need gathered ids");
    const bool desc_ok = !(desc && scales) || n_desc == 4;
    static const bool old_scan = MSE_DEV_KNOB("MSE_PQ_OLDSCAN");
    if (!ids && n_chunks == 64 && n_centroids == 256 && desc_ok && n == n_codes && !old_scan) {
        MSE_DYN_LDS(pq_scan64_kernel<false>, PQS_LDS);
        const size_t groups = (n + 63) / 64;
        const unsigned blocks = (unsigned)std::min<size_t>((groups + PQS_WAVES - 1) / PQS_WAVES, (size_t)n_cu);
        hipLaunchKernelGGL(pq_scan64_kernel<false>, dim3(blocks), dim3(PQS_WAVES * 64), PQS_LDS, stream, lut, codes, n,
                           (desc && scales) ? desc : nullptr, scales, out);
        MSE_HIP_TRY(hipGetLastError());
        return 0;
    }
    const size_t lds = (size_t)n_chunks * n_centroids * 4;
    if (lds > 160 * 1024) return fail("PQ table does not fit LDS");
    if (lds > 64 * 1024) {
        MSE_DYN_LDS(pq_adc_kernel, lds);
    }
    // 1024-thread workgroups: every workgroup first copies the whole table (up to 64 KiB+) into LDS, so few fat workgroups beat
    // many thin ones -- above all when this kernel is part of a scan's tail and only the CUs a concurrent scan leaves free (one per
    // XCD) are there to take them: 300 workgroups of 256 threads took 1.3 ms there, 30 us on an idle device
    // (profiles/r04_pq_scan_stats.txt)
    const int threads = n >= 1024 ? 1024 : 256;
    size_t blocks = (n + threads - 1) / threads;
    const size_t cap = (size_t)n_cu * 2;
    if (blocks > cap) blocks = cap;
    // several queries per launch = the tail of a batched scan, beside which the NEXT scan is resident on all but scan_cus()'s
    // spare CUs (one per XCD).  A workgroup of this kernel (1024 threads, ~100 VGPRs, 64 KiB of LDS) cannot share a CU with a scan
    // workgroup, and a dispatch whose workgroups do not ALL fit into what is free at once does not trickle through the spare CUs: it
    // waits for the scan to end (scripts/native/coresidency_probe.hip: 8 such workgroups beside a 248-workgroup resident kernel
    // finish in 6 us, 64 of them in 2.4 ms = when the resident kernel ends; profiles/r04_pq_timeline.txt: 64 workgroups took
    // 750-950 us here, 8 take ~100).  So: as many workgroups as there are spare CUs, each looping over its share.
    if (nq > 1) {
        const size_t spare = std::max<size_t>((size_t)n_cu - scan_cus(n_cu), 8);
        blocks = std::min(blocks, std::max<size_t>(spare / (size_t)nq, 1));
    }
#ifdef MSE_DEV_KERNELS
    // MSE_PQ_ADC_VARIANT=v: one EXTRA launch of timing variant v (1 table copy only, 2 neighbouring rows instead of gathered ones,
    // 3 ids + code rows only) ahead of the real one, so that a kernel trace shows what each part costs beside a running scan
    static const int variant = getenv("MSE_PQ_ADC_VARIANT") ? atoi(getenv("MSE_PQ_ADC_VARIANT")) : 0;
    if (variant > 0)
        hipLaunchKernelGGL(pq_adc_kernel, dim3((unsigned)blocks, (unsigned)nq), dim3(threads), lds, stream, lut, n_chunks, n_centroids, codes,
                           n_codes, ids, n, desc, n_desc, scales, out, q_stride, variant);
    hipLaunchKernelGGL(pq_adc_kernel, dim3((unsigned)blocks, (unsigned)nq), dim3(threads), lds, stream, lut, n_chunks, n_centroids, codes,
                       n_codes, ids, n, desc, n_desc, scales, out, q_stride, 0);
#else
    hipLaunchKernelGGL(pq_adc_kernel, dim3((unsigned)blocks, (unsigned)nq), dim3(threads), lds, stream, lut, n_chunks, n_centroids, codes,
                       n_codes, ids, n, desc, n_desc, scales, out, q_stride);
#endif
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
int launch_f16_to_f32(const uint16_t* in, size_t n, float* out, hipStream_t stream) {
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, n, out);
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
