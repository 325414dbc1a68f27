// Batched brute-force scan on the matrix cores: candidate stage of the batched top-k.
//
// Reference work being replaced: the N x fast_dot loop of the brute-force evaluator
// (src/query_disk_index.rs:263-269) run for a whole batch of queries at once.  The matrix core sums
// the 1152 products in its own order, so these scores are NOT the reference's bit pattern; they only
// nominate candidate row groups.  Every candidate row is then re-scored in the reference order
// (scan_exact.hip) and a certificate (topk.hip::finalize_kernel) proves that no row outside the
// candidate groups can belong to the exact top-k; otherwise the caller widens the candidate set.
//
// Kernel: C[row, query] = sum_k X[row,k] * Q[query,k] with v_mfma_f32_32x32x16_f16.
//   - a wave owns 32 base rows (MFMA A operand) x 128 queries (4 column tiles, B operand);
//     lane (i = lane&31, h = lane>>5) holds row i; per 64-element K block it loads 64 contiguous
//     bytes of that row straight from HBM into VGPRs (no LDS round trip for the streamed operand)
//     and feeds MFMA step s with bytes [h*64 + s*16, +16).  The contraction index is permuted the
//     same way on the query side, which a dot product does not care about.
//   - queries are re-tiled once per search into K-block-major, XOR-swizzled 16 KiB tiles
//     (pack_queries_kernel) so that a tile is a linear copy into LDS and ds_read_b128 of the B
//     fragments is bank-conflict free; tiles are double buffered, one barrier per K block.
//   - epilogue per 32-row tile: max over the 32 rows of each query column -> group_max[group][q]
//     (the level-0 array of the selection tournament).  4 bytes written per 32*2304 bytes read.
// Roofline: HBM.  Algorithmic bytes = 2*d per base row per pass of <= 128 queries.
#include "common.h"
#include "kernels.h"

namespace mse {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int BN = 128;          // queries per pass
constexpr int KB = 64;           // contraction elements per K block
constexpr int WAVES = 4;
constexpr int TILE_ROWS = 32 * WAVES;
constexpr int QT_SLOTS = BN * 8; // uint4 slots per query tile (16 KiB)

// packed[kb][q][slot ^ ((q>>1)&7)] = 16-byte slot `slot` of K block kb of query q
__global__ void pack_queries_kernel(const uint16_t* __restrict__ queries, int d, uint4* __restrict__ packed) {
    const int nkb = d / KB;
    const int total = nkb * QT_SLOTS;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int kb = idx / QT_SLOTS;
        const int r = idx % QT_SLOTS;
        const int q = r >> 3, slot_sw = r & 7;
        const int slot = slot_sw ^ ((q >> 1) & 7);
        packed[idx] = *reinterpret_cast<const uint4*>(queries + (size_t)q * d + kb * KB + slot * 8);
    }
}

__device__ __forceinline__ half8 as_half8(const uint4& v) { return __builtin_bit_cast(half8, v); }

__global__ __launch_bounds__(256, 2) void scan_mfma_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                           const uint4* __restrict__ packed, float* __restrict__ gmax,
                                                           int nq_pad, size_t n_tiles) {
    __shared__ uint4 lds[2][QT_SLOTS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int nkb = d / KB;
    const int row_u4 = d / 8;  // uint4 per row
    const int swz = (i >> 1) & 7;
    const size_t n_groups = (n_rows + 31) / 32;

    size_t tile = blockIdx.x;
    if (tile >= n_tiles) return;

    auto row_ptr = [&](size_t t) -> const uint4* {
        size_t row = t * TILE_ROWS + wave * 32 + i;
        if (row >= n_rows) row = n_rows - 1;
        return reinterpret_cast<const uint4*>(base) + row * (size_t)row_u4 + h * 4;
    };

    // prologue: first query tile into LDS buffer 0, first X block into registers
    uint4 qreg[4];
#pragma unroll
    for (int u = 0; u < 4; u++) qreg[u] = packed[tid + 256 * u];
    const uint4* xp = row_ptr(tile);
    uint4 xcur[4], xnext[4];
#pragma unroll
    for (int s = 0; s < 4; s++) xcur[s] = xp[s];
#pragma unroll
    for (int u = 0; u < 4; u++) lds[0][tid + 256 * u] = qreg[u];
    __syncthreads();

    int buf = 0;
    while (true) {
        float16v acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[ct][r] = 0.0f;

        const size_t next_tile = tile + gridDim.x;
        const bool has_next_tile = next_tile < n_tiles;
        const uint4* xp_next = has_next_tile ? row_ptr(next_tile) : xp;

        for (int kb = 0; kb < nkb; kb++) {
            const bool last = kb + 1 == nkb;
            const int nkbi = last ? 0 : kb + 1;
            // prefetch: next X block (this tile, or the first block of the next tile) and next query tile
            const uint4* src = last ? xp_next : xp;
#pragma unroll
            for (int s = 0; s < 4; s++) xnext[s] = src[nkbi * 8 + s];
#pragma unroll
            for (int u = 0; u < 4; u++) qreg[u] = packed[(size_t)nkbi * QT_SLOTS + tid + 256 * u];

            const uint4* qt = lds[buf];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const half8 a = as_half8(xcur[s]);
                const int slot = (h * 4 + s) ^ swz;
#pragma unroll
                for (int ct = 0; ct < 4; ct++) {
                    const half8 b = as_half8(qt[(ct * 32 + i) * 8 + slot]);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[ct], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) lds[buf ^ 1][tid + 256 * u] = qreg[u];
            __syncthreads();
            buf ^= 1;
#pragma unroll
            for (int s = 0; s < 4; s++) xcur[s] = xnext[s];
        }

        // epilogue: per query column, max over this wave's 32 rows
        const size_t group = tile * WAVES + wave;
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
            float m = acc[ct][0];
#pragma unroll
            for (int r = 1; r < 16; r++) m = fmaxf(m, acc[ct][r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            if (h == 0 && group < n_groups) gmax[group * (size_t)nq_pad + ct * 32 + i] = m;
        }

        if (!has_next_tile) break;
        tile = next_tile;
        xp = xp_next;
    }
}

}  // namespace

int mfma_query_tile() { return BN; }

size_t mfma_packed_bytes(int d) { return (size_t)(d / KB) * QT_SLOTS * sizeof(uint4); }

// packed_scratch: mfma_packed_bytes(d) bytes of device scratch owned by the caller (per searcher)
int launch_scan_mfma(const uint16_t* base, size_t n_rows, int d, const uint16_t* queries_dev, int nq_pad,
                     void* packed_scratch, float* group_max, int n_cu, hipStream_t stream, hipEvent_t ev_begin,
                     hipEvent_t ev_end) {
    if (n_rows == 0) return 0;
    if (d % 64 != 0 || d <= 0 || d > D_MAX) return fail("vector width must be a positive multiple of 64");
    if (nq_pad != BN) return fail("scan_mfma: query tile must be padded to 128");
    uint4* packed = reinterpret_cast<uint4*>(packed_scratch);
    hipLaunchKernelGGL(pack_queries_kernel, dim3(64), dim3(256), 0, stream, queries_dev, d, packed);
    if (ev_begin) MSE_HIP_TRY(hipEventRecord(ev_begin, stream));
    const size_t n_tiles = (n_rows + TILE_ROWS - 1) / TILE_ROWS;
    size_t grid = (size_t)n_cu * 2;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(scan_mfma_kernel, dim3((unsigned)grid), dim3(256), 0, stream, base, n_rows, d, packed, group_max,
                       nq_pad, n_tiles);
    MSE_HIP_TRY(hipGetLastError());
    if (ev_end) MSE_HIP_TRY(hipEventRecord(ev_end, stream));
    return 0;
}

}  // namespace mse
