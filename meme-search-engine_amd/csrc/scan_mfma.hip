// Batched brute-force scan on the matrix cores: candidate stage of the batched top-k.
//
// Reference work being replaced: the N x fast_dot loop of the brute-force evaluator
// (src/query_disk_index.rs:263-269) run for a whole batch of queries at once.  The matrix core sums
// the 1152 products in its own order, so these scores are NOT the reference's bit pattern; they only
// nominate candidate row groups.  Every candidate row is then re-scored in the reference order
// (scan_exact.hip) and a certificate (topk.hip::finalize_kernel) proves that no row outside the
// candidate groups can belong to the exact top-k; otherwise the caller widens the candidate set.
//
// Kernel: C[row, query] = sum_k X[row,k] * Q[query,k] with v_mfma_f32_16x16x32_f16.
//   - a wave owns 32 base rows (two 16-row MFMA tiles, A operand) x 128 queries (eight 16-query column
//     tiles, B operand; each B fragment read from LDS feeds two MFMAs).
//   - base rows go HBM -> VGPR directly (no LDS round trip for the streamed operand).  The 16x16x32 A
//     layout puts FOUR lanes on one matrix row (lane = row + 16*g, g = 8-element k group), so one
//     global_load_dwordx4 reads 64 contiguous bytes from each of 16 rows.  (The 32x32x16 shape puts two
//     lanes on a row, 32 B per row per instruction: measured ~4.0 TB/s at every prefetch depth and
//     occupancy; this shape: 4.4-4.7 TB/s at 4 waves per workgroup.)
//   - queries are re-tiled once per search into K-block-major, XOR-swizzled 16 KiB tiles
//     (pack_queries_kernel) so that a tile is a linear copy into LDS and ds_read_b128 of the B fragments
//     is bank-conflict free; tiles are double buffered in LDS, one barrier per 64-element K block.
//     A workgroup of W waves shares one tile for 32*W rows: L2->LDS query traffic is 128/(32*W) of the
//     HBM traffic, and it competes with the HBM stream for the CU's outstanding-miss slots, so W = 8.
//   - epilogue per 32-row group: max over the rows of each query column -> group_max[group][q]
//     (the level-0 array of the selection tournament).  4 bytes written per 32*2304 bytes read.
// Roofline: HBM.  Algorithmic bytes = 2*d per base row per pass of <= 128 queries.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace mse {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 128;           // queries per pass
constexpr int KB = 64;            // contraction elements per K block
constexpr int QT_SLOTS = BN * 8;  // 16-byte slots per query tile (16 KiB)

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

__device__ __forceinline__ half8 as_half8(const u32x4& v) { return __builtin_bit_cast(half8, v); }

// ---- hand-issued global loads ------------------------------------------------------------------------
// hipcc schedules an ordinary global load next to its first use; in this loop that is BELOW the MFMA block
// of the same iteration, which leaves the load no flight time.  sched_barrier / sched_group_barrier /
// volatile / asm memory clobbers either do not move the loads or cost more than they give (serialised
// ds_read->MFMA, vmcnt(0) after every volatile load, spills).  So every VMEM load of the main loop is
// issued by inline asm in program order and the vmcnt waits are counted by hand: per K block each lane
// issues QL query-tile loads, then 4 base-row loads, always in that order.  Loads return in order, so
// "at most N outstanding" identifies which ones have landed; the epilogue's stores share the counter
// but can only make a wait stricter.  `vm_wait<N>` carries the registers it guards as in/out operands, so
// the compiler cannot move their first use above the wait (cdna_hip_programming.md 5.7).
template <int OFF> __device__ __forceinline__ void gload(u32x4& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(dst) : "v"(p), "n"(OFF));
}
template <int N> __device__ __forceinline__ void vm_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N> __device__ __forceinline__ void vm_wait(u32x4& a, u32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N> __device__ __forceinline__ void vm_wait(u32x4& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int N, int QL> __device__ __forceinline__ void vm_wait_q(u32x4 (&q)[QL]) {
    if constexpr (QL == 4) vm_wait<N>(q[0], q[1], q[2], q[3]);
    else if constexpr (QL == 2) vm_wait<N>(q[0], q[1]);
    else vm_wait<N>(q[0]);
}

// P = depth of the register ring of X K-blocks (P-1 blocks in flight beyond the one being consumed);
// nkb must be a multiple of P so that ring slots are compile-time constants.  W = waves per workgroup.
template <int P, int W>
__global__ __launch_bounds__(W * 64) void scan_mfma_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                           const uint4* __restrict__ packed_ro,
                                                           float* __restrict__ gmax, int nq_pad, size_t n_tiles) {
    constexpr int THREADS = W * 64;
    constexpr int TILE_ROWS = 32 * W;
    constexpr int QL = QT_SLOTS / THREADS;  // query-tile slots copied per thread: 4 / 2 / 1
    static_assert(QL == 4 || QL == 2 || QL == 1, "W must be 4, 8 or 16");
    __shared__ u32x4 lds[2][QT_SLOTS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nkb = d / KB;
    const int row_u4 = d / 8;  // 16-byte units per row
    const int swz = (i >> 1) & 7;
    const size_t n_groups = (n_rows + 31) / 32;
    const u32x4* packed = reinterpret_cast<const u32x4*>(packed_ro) + tid;  // this lane's first slot in a tile

    size_t tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // pointer to this lane's 16-byte piece of K block 0 for its row in row tile rt
    auto row_ptr = [&](size_t t, int rt) -> const u32x4* {
        size_t row = t * TILE_ROWS + wave * 32 + rt * 16 + i;
        if (row >= n_rows) row = n_rows - 1;
        return reinterpret_cast<const u32x4*>(base) + row * (size_t)row_u4 + g;
    };
    // one K block for this lane: [row tile][k step] -> bytes ks*64 + g*16 of the block
    auto load_block = [&](u32x4(&dst)[4], const u32x4* p0, const u32x4* p1, int kblock) {
        const u32x4* a0 = p0 + (size_t)kblock * 8;
        const u32x4* a1 = p1 + (size_t)kblock * 8;
        gload<0>(dst[0], a0);
        gload<64>(dst[1], a0);
        gload<0>(dst[2], a1);
        gload<64>(dst[3], a1);
    };
    // this thread's QL slots of query tile kblock (slot index tid + THREADS*u)
    auto load_qtile = [&](u32x4(&dst)[QL], int kblock) {
        const u32x4* src = packed + (size_t)kblock * QT_SLOTS;
        if constexpr (QL == 4) {  // byte offsets 0 / 4096 / 8192 / 12288 (13-bit signed immediates)
            gload<-4096>(dst[0], src + 256);
            gload<0>(dst[1], src + 256);
            gload<-4096>(dst[2], src + 768);
            gload<0>(dst[3], src + 768);
        } else if constexpr (QL == 2) {  // 0 / 8192
            gload<0>(dst[0], src);
            gload<0>(dst[1], src + 512);
        } else {
            gload<0>(dst[0], src);
        }
    };

    // prologue: query tile 0 into LDS buffer 0; X blocks 0..P-2 of the first tile into the ring.
    // The counted waits in the loop assume the steady-state issue pattern, so drain the prologue completely.
    u32x4 qreg[QL];
    load_qtile(qreg, 0);
    const u32x4* xp0 = row_ptr(tile, 0);
    const u32x4* xp1 = row_ptr(tile, 1);
    u32x4 xr[P][4];
#pragma unroll
    for (int j = 0; j < P - 1; j++) load_block(xr[j], xp0, xp1, j % nkb);
    vm_wait_q<0, QL>(qreg);
#pragma unroll
    for (int j = 0; j < P - 1; j++) vm_wait<0>(xr[j][0], xr[j][1], xr[j][2], xr[j][3]);
#pragma unroll
    for (int u = 0; u < QL; u++) lds[0][tid + THREADS * u] = qreg[u];
    __syncthreads();

    int buf = 0;
    while (true) {
        float4v acc[2][8];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[rt][ct][r] = 0.0f;

        const size_t next_tile = tile + gridDim.x;
        const bool has_next_tile = next_tile < n_tiles;
        const u32x4* xn0 = has_next_tile ? row_ptr(next_tile, 0) : xp0;
        const u32x4* xn1 = has_next_tile ? row_ptr(next_tile, 1) : xp1;

        for (int kb0 = 0; kb0 < nkb; kb0 += P) {
#pragma unroll
            for (int j = 0; j < P; j++) {
                const int kb = kb0 + j;
                // ---- issue this iteration's loads: next query tile (QL), then the X block P-1 steps ahead (4)
                load_qtile(qreg, kb + 1 == nkb ? 0 : kb + 1);
                {
                    const int kf = kb + P - 1;
                    const bool wrap = kf >= nkb;
                    load_block(xr[(j + P - 1) % P], wrap ? xn0 : xp0, wrap ? xn1 : xp1, wrap ? kf - nkb : kf);
                }
                // the X block consumed now was issued P-1 iterations ago: everything issued after it may stay in flight
                vm_wait<(QL + 4) * (P - 1)>(xr[j][0], xr[j][1], xr[j][2], xr[j][3]);

                // B fragments: software pipelined two column tiles ahead of the MFMAs that consume them
                const u32x4* qt = lds[buf] + i * 8;
                const int slot_a = g ^ swz, slot_b = (4 + g) ^ swz;  // k step 0 / 1
                u32x4 bq[2][2];
                bq[0][0] = qt[0 * 128 + slot_a];
                bq[0][1] = qt[1 * 128 + slot_a];
#pragma unroll
                for (int t = 0; t < 8; t++) {  // t = ks*4 + column-tile pair
                    const int ks = t >> 2, cp = t & 3;
                    if (t + 1 < 8) {
                        const int ks2 = (t + 1) >> 2, cp2 = (t + 1) & 3;
                        bq[(t + 1) & 1][0] = qt[(cp2 * 2) * 128 + (ks2 ? slot_b : slot_a)];
                        bq[(t + 1) & 1][1] = qt[(cp2 * 2 + 1) * 128 + (ks2 ? slot_b : slot_a)];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the reads for t+1 ahead of the MFMAs of t
                    const half8 a0 = as_half8(xr[j][ks]);
                    const half8 a1 = as_half8(xr[j][2 + ks]);
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const half8 b = as_half8(bq[t & 1][e]);
                        const int ct = cp * 2 + e;
                        acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, acc[0][ct], 0, 0, 0);
                        acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, acc[1][ct], 0, 0, 0);
                    }
                }
                // query tile for the next K block: only the 4 X loads issued after it may remain outstanding
                vm_wait_q<4, QL>(qreg);
#pragma unroll
                for (int u = 0; u < QL; u++) lds[buf ^ 1][tid + THREADS * u] = qreg[u];
                __syncthreads();
                buf ^= 1;
            }
        }

        // epilogue: per query column, max over this wave's 32 rows (2 row tiles x 4 accumulator rows x 4 lane groups)
        const size_t group = tile * W + wave;
#pragma unroll
        for (int ct = 0; ct < 8; ct++) {
            float m = fmaxf(fmaxf(acc[0][ct][0], acc[0][ct][1]), fmaxf(acc[0][ct][2], acc[0][ct][3]));
            m = fmaxf(m, fmaxf(fmaxf(acc[1][ct][0], acc[1][ct][1]), fmaxf(acc[1][ct][2], acc[1][ct][3])));
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            if (g == 0 && group < n_groups) gmax[group * (size_t)nq_pad + ct * 16 + i] = m;
        }

        if (!has_next_tile) break;
        tile = next_tile;
        xp0 = xn0;
        xp1 = xn1;
    }
    // drain the loads still in flight before the wave ends (their destination registers die with it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int P, int W>
void launch_variant(size_t grid, hipStream_t stream, const uint16_t* base, size_t n_rows, int d, const uint4* packed,
                    float* group_max, int nq_pad) {
    const size_t n_tiles = (n_rows + 32 * W - 1) / (32 * W);
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL((scan_mfma_kernel<P, W>), dim3((unsigned)grid), dim3(W * 64), 0, stream, base, n_rows, d, packed,
                       group_max, nq_pad, n_tiles);
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
    const int nkb = d / KB;
    // developer knobs (defaults are the shipped configuration)
    static const int env_p = getenv("MSE_SCAN_P") ? atoi(getenv("MSE_SCAN_P")) : 0;
    static const int env_w = getenv("MSE_SCAN_W") ? atoi(getenv("MSE_SCAN_W")) : 8;
    static const int env_wg = getenv("MSE_SCAN_WG") ? atoi(getenv("MSE_SCAN_WG")) : 1;
    const size_t grid = (size_t)n_cu * env_wg;
    int P = nkb % 3 == 0 ? 3 : nkb % 2 == 0 ? 2 : 1;
    if (env_p && nkb % env_p == 0) P = env_p;
#define MSE_LAUNCH(PP, WW) launch_variant<PP, WW>(grid, stream, base, n_rows, d, packed, group_max, nq_pad)
#define MSE_LAUNCH_W(WW)                                                             \
    do {                                                                             \
        if (P == 3) MSE_LAUNCH(3, WW); else if (P == 2) MSE_LAUNCH(2, WW); else MSE_LAUNCH(1, WW); \
    } while (0)
    if (env_w == 4) MSE_LAUNCH_W(4);
    else if (env_w == 16) MSE_LAUNCH_W(16);
    else MSE_LAUNCH_W(8);
#undef MSE_LAUNCH_W
#undef MSE_LAUNCH
    MSE_HIP_TRY(hipGetLastError());
    if (ev_end) MSE_HIP_TRY(hipEventRecord(ev_end, stream));
    return 0;
}

}  // namespace mse
