// Batched brute-force scan on the matrix cores: candidate stage of the batched top-k.
//
// Reference work being replaced: the N x fast_dot loop of the brute-force evaluator
// (src/query_disk_index.rs:263-269) run for a whole batch of queries at once.  The matrix core sums
// the 1152 products in its own order, so these scores are NOT the reference's bit pattern; they only
// nominate candidate row groups.  Every candidate row is then re-scored in the reference order
// (scan_exact.hip) and a certificate (topk.hip::finalize_kernel) proves that no row outside the
// candidate groups can belong to the exact top-k; otherwise the caller widens the candidate set.
//
// Kernel: C[row, query] = sum_k X[row,k] * Q[query,k] with v_mfma_f32_16x16x32_f16, 8 waves per workgroup,
// one workgroup per CU (128 / 160 KiB of LDS), persistent grid-stride over 256-row tiles.  BN = queries per
// pass over the rows: 128, or 256 (round 2: the same HBM stream serves twice the queries; the matrix cores go
// from ~35 % to ~70 % busy, which is what the HBM-bound 128-query pass left idle).
//   - a wave owns 32 base rows (two 16-row MFMA tiles, A operand) x BN queries (BN/16 16-query column
//     tiles, B operand; each B fragment read from LDS feeds two MFMAs; 64 or 128 accumulator registers).
//   - base rows are streamed HBM -> LDS by the DMA path (global_load_lds, no VGPR staging).  Each wave DMAs
//     exactly the 32 rows x 128 B it will consume into its OWN ring of S stages (4 KiB each), so the
//     streamed operand needs no barrier: an LDS-DMA is ordered for a ds_read only by the issuing wave's
//     vmcnt, and that wait is counted by hand (the loads of a K block are always issued in the same order).
//     One DMA instruction moves 8 rows x 128 B, i.e. whole cache lines.  The LDS image of a DMA is
//     lane-linear, so the bank-conflict swizzle is applied to the SOURCE address: LDS slot (row, s) holds
//     global 16-byte piece s ^ f(row), f(row) = (row >> 1) & 7, and the A-fragment ds_read_b128 applies the
//     same involution (measured SQ_LDS_BANK_CONFLICT = 0).
//   - queries are re-tiled once per search into K-block-major, identically swizzled BN x 128 B tiles
//     (pack_queries_kernel); a tile is DMA'd linearly into a double buffer (BN/64 instructions per wave), one
//     barrier per 64-element K block.  The tile is shared by 256 rows, so L2->LDS query traffic is BN/256 of the
//     HBM traffic (it competes with the HBM stream for the CU's outstanding-miss slots).
//   - epilogue per 32-row group: max over the rows of each query column -> group_max[group][q]
//     (the level-0 array of the selection tournament).  4 bytes written per 32*2304 bytes read.
// Roofline: HBM.  Algorithmic bytes = 2*d per base row per pass of <= 128 queries; PMC FETCH_SIZE (x2 on
// gfx950, KiB) = 1.007 x that.
//
// Measured on MI355X at 1e7 x 1152 (profiles/): earlier variants of this kernel, kept here as a record of
// what did not work --  32x32x16 MFMA with lane = row (32 B per row per load instruction): 3.9-4.2 TB/s at
// every prefetch depth and occupancy; 16x16x32 with register-staged rows (64 B per row per instruction):
// 4.4-4.7 TB/s at 4 waves, 5.0-5.2 TB/s at 8 waves per workgroup; non-temporal loads on partial lines: -25 %;
// this DMA variant: 5.3-5.6 TB/s.  Ordinary (compiler-scheduled) loads are sunk next to their first use by
// hipcc, which leaves them no flight time; hence hand-issued DMA + counted waits.
//
// Round 2, the 256-query pass (1e7 x 1152, random rows, scan kernel alone; scripts/scan_ablate.py, MSE_SCAN_ABL):
//   128 queries 3.95-4.1 ms (HBM-bound, 5.6-5.8 TB/s) | 256 queries 5.65-5.75 ms = 4.0 TB/s, 1.04 PFLOP/s: +40 % queries/s.
//   Ablations of the 256-query kernel: MFMAs only 3.6 ms | X DMA only 3.9 | query-tile DMA only 0.83 | everything but the
//   MFMAs 4.1 | everything but the X DMA 4.5.  The parts do not hide behind each other the way the instruction counts
//   suggest, and re-timing them does not help: DMA pieces in one burst 5.90, one piece between MFMA groups (kept) 5.72,
//   bursts of the two waves of a SIMD half an iteration apart 6.30, a two-group ping-pong kernel (one wave per SIMD on the
//   matrix core while the other issues pieces, two barriers per K block) 6.06, nt loads for the rows: no change.  In-kernel
//   s_memtime stamps put the clock at 1.64 GHz under this load (2176 of 3400 cycles per K block are matrix-core cycles),
//   and the same kernel on all-zero rows -- same traffic, no operand toggling -- runs in 4.7 ms (MFMAs only: 3.0 ms):
//   the pass is bound by the chip's power budget (DVFS), not by a schedule.  LDS-DMA issue is not the limit either
//   (scripts/microbench/dma_rate.hip: 137 GB/s per CU from L2 with >= 4 waves, 7.5 ns per 1 KiB piece).  What would lower
//   the energy per row: wave tiles of 64 rows x 128 queries (a third fewer B-fragment LDS reads), not tried.
//   PMC counters (profiles/r02_pmc_scan_1e7.txt): 1.53 GHz and 62 % MFMA-busy at 256 queries, 1.84 GHz and 38 % at 128; both
//   passes pull ~19 B per cycle per CU through the vector L1 (rows + query tiles), which suggested that path as the bound --
//   but a 128-query kernel with 64 rows per wave (512-row tiles: a query tile fetched once per 512 rows, 17 % fewer L1 bytes
//   per row, 2-stage ring) ran 4.33 vs 4.20 ms at 1e7 rows and 42.0 vs 41.0 ms at 1e8: not that either.
//   Also without effect: v_mfma_f32_32x32x16_f16 over the same LDS images (half the MFMA instructions, one fragment read per
//   MFMA of twice the size; conflict-free with the same swizzle): 5.97 vs 5.80 ms; a 2-stage row ring instead of 3 (MSE_SCAN_S=2):
//   5.96 vs 5.95 ms at 256 queries, 4.25 vs 4.17 at 128 -- the pass is not limited by rows in flight; fragment reads issued by hand (asm ds_read_b128 + counted lgkmcnt waits instead of the
//   `s_waitcnt lgkmcnt(0)` hipcc puts in front of every MFMA group): 5.95 vs 6.06 ms on the same box, within noise.
//   Per K block (32 KiB of rows + 32 KiB of query tile through the CU's load path) the pass takes 2.1 us where the rows alone
//   take 1.4 us and the MFMAs alone 1.3 us; at 128 queries (48 KiB per K block) it takes 1.45 us.
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace mse {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 64;            // contraction elements per K block
constexpr int W = 8;              // waves per workgroup
constexpr int TILE_ROWS = 32 * W;

// packed[kb][q][slot ^ ((q>>1)&7)] = 16-byte slot `slot` of K block kb of query q   (bn queries per tile)
__global__ void pack_queries_kernel(const uint16_t* __restrict__ queries, int d, int bn, uint4* __restrict__ packed) {
    const int nkb = d / KB;
    const int QT_SLOTS = bn * 8;
    const int total = nkb * QT_SLOTS;
    queries += (size_t)blockIdx.y * bn * d;   // batched passes: pass y = the next bn query rows, the next `total` slots
    packed += (size_t)blockIdx.y * total;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int kb = idx / QT_SLOTS;
        const int r = idx % QT_SLOTS;
        const int q = r >> 3, slot_sw = r & 7;
        const int slot = slot_sw ^ ((q >> 1) & 7);
        packed[idx] = *reinterpret_cast<const uint4*>(queries + (size_t)q * d + kb * KB + slot * 8);
    }
}

__device__ __forceinline__ half8 as_half8(const u32x4& v) { return __builtin_bit_cast(half8, v); }

// s_waitcnt vmcnt(N) placed by hand: the "memory" clobber keeps LDS reads below it, the sched_barrier keeps
// register-only MFMAs from being hoisted above it (cdna_hip_programming.md 5.4 rule 18).
template <int N> __device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// one LDS-DMA instruction: 16 bytes per lane, LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void dma16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// S = stages in each wave's ring of X K-blocks (S-1 blocks in flight beyond the one being consumed);
// nkb must be a multiple of S so that stage indices are compile-time constants.
// NCT = 16-query column tiles per pass (8: 128 queries, 16: 256 queries)
// ABL (developer ablations, MSE_SCAN_ABL; results are then meaningless, only the timing is of interest):
// bit 0 no MFMAs, bit 1 no query-tile DMA, bit 2 no X DMA, bit 3 no B-fragment LDS reads
template <int S, int NCT, int ABL = 0>
__global__ __launch_bounds__(W * 64) void scan_mfma_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                           const uint4* __restrict__ packed_ro,
                                                           float* __restrict__ gmax, int nq_pad, size_t n_tiles, uint32_t y_packed, uint32_t y_cols) {
    constexpr int BN = NCT * 16;
    constexpr int QT_BYTES = BN * 128;       // one query tile: BN x 64 f16
    constexpr int QI = BN / 64;              // DMA instructions per wave per query tile
    constexpr bool ILV = S >= 2 && !(ABL & 16);   // DMA pieces interleaved with the MFMA groups (ABL bit 4: burst, as in round 1)
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ALL LDS in one object (5.x trap (a))
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int nkb = d / KB;
    const size_t row_bytes = (size_t)d * 2;
    const int swz = (i >> 1) & 7;
    const size_t n_groups = (n_rows + 31) / 32;
    char* const qbase = smem;                                     // [2][BN x 128 B] query tiles
    char* const xbase = smem + 2 * QT_BYTES + wave * (S * 4096);  // this wave's ring: S stages x 4 KiB
    gmax += (size_t)blockIdx.y * y_cols;   // batched passes (launch_scan_mfma n_pass > 1): pass y has its own query tiles and columns
    const char* const packed = reinterpret_cast<const char*>(packed_ro + (size_t)blockIdx.y * y_packed) + (size_t)(wave * QI * 64 + lane) * 16;

    size_t tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // global source of this lane for DMA instruction u of a stage: row 8u + (lane>>3), piece (lane&7) ^ f(row).
    // Rows past the end are clamped to the last row: they only duplicate a row of the same (last) group.
    auto src_ptr = [&](size_t t, int u) -> const char* {
        const int r = 8 * u + (lane >> 3);
        size_t row = t * TILE_ROWS + wave * 32 + r;
        if (row >= n_rows) row = n_rows - 1;
        const int piece = (lane & 7) ^ ((r >> 1) & 7);
        return reinterpret_cast<const char*>(base) + row * row_bytes + piece * 16;
    };
    auto dma_x = [&](const char* (&rp)[4], int kblock, int stage) {
        char* dst = xbase + stage * 4096;
        if (ABL & 4) return;
#pragma unroll
        for (int u = 0; u < 4; u++) dma16(rp[u] + (size_t)kblock * 128, dst + u * 1024);
    };
    auto dma_q = [&](int kblock, int qb) {
        const char* src = packed + (size_t)kblock * QT_BYTES;
        char* dst = qbase + qb * QT_BYTES + wave * (QI * 1024);
        if (ABL & 2) return;
#pragma unroll
        for (int u = 0; u < QI; u++) dma16(src + u * 1024, dst + u * 1024);
    };

    const char* rp[4];
    const char* rn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) rp[u] = src_ptr(tile, u);
    // prologue: query tile 0 and X blocks 0..S-2; drained completely so the counted waits start in steady state
    dma_q(0, 0);
#pragma unroll
    for (int j = 0; j < S - 1; j++) dma_x(rp, j % nkb, j);
    vm_wait<0>();
    __builtin_amdgcn_s_barrier();

    int buf = 0;
    while (true) {
        float4v acc[2][NCT];
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < NCT; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[rt][ct][r] = 0.0f;

        const size_t next_tile = tile + gridDim.x;
        const bool has_next_tile = next_tile < n_tiles;
#pragma unroll
        for (int u = 0; u < 4; u++) rn[u] = has_next_tile ? src_ptr(next_tile, u) : rp[u];

        for (int kb0 = 0; kb0 < nkb; kb0 += S) {
#pragma unroll
            for (int j = 0; j < S; j++) {
                const int kb = kb0 + j;
                // ---- DMA pieces of this iteration, always in this order: next query tile (QI), then the X block S-1 steps
                // ahead (4).  ILV: one piece at a time BETWEEN the MFMA groups below -- a burst of QI + 4 pieces at the top
                // costs each wave 100-185 cycles of issue per piece while both waves of the SIMD sit in the same phase and
                // the matrix core idles (measured: 256-query pass 5.94 ms burst vs MFMA-only 3.63 / DMA-only 4.58 at 1e7 rows).
                const int kq = kb + 1 == nkb ? 0 : kb + 1;
                const int kf = kb + S - 1;
                const bool x_next = kf >= nkb;                                 // first blocks of the next tile
                const size_t xoff = (size_t)(kf >= nkb ? kf - nkb : kf) * 128;
                char* const xdst = xbase + ((j + S - 1) % S) * 4096;
                const char* const qsrc = packed + (size_t)kq * QT_BYTES;
                char* const qdst = qbase + (buf ^ 1) * QT_BYTES + wave * (QI * 1024);
                auto issue_piece = [&](int p) {
                    if (p < QI) { if (!(ABL & 2)) dma16(qsrc + p * 1024, qdst + p * 1024); }
                    else if (!(ABL & 4)) dma16((x_next ? rn[p - QI] : rp[p - QI]) + xoff, xdst + (p - QI) * 1024);
                };
                if (!ILV) {
#pragma unroll
                    for (int p = 0; p < QI + 4; p++) issue_piece(p);
                    // the X block consumed now was issued S-1 iterations ago: everything issued after it may fly on.
                    // (DMAs return in order; the epilogue's stores share the counter and only make a wait stricter.)
                    vm_wait<(QI + 4) * (S - 1)>();
                } else if (S == 2) {
                    vm_wait<0>();   // the block consumed now was the last thing issued in the previous iteration
                }
                // ILV, S >= 3: the block consumed now was issued BEFORE the query pieces the previous iteration's closing
                // wait already covered, so it has landed.

                const u32x4* xs = reinterpret_cast<const u32x4*>(xbase + j * 4096) + i * 8;
                const u32x4* qt = reinterpret_cast<const u32x4*>(qbase + buf * QT_BYTES) + i * 8;
                const int slot_a = g ^ swz, slot_b = (4 + g) ^ swz;  // k step 0 / 1 of the K block
                // B fragments are software pipelined one column-tile pair ahead of the MFMAs that consume them
                u32x4 bq[2][2];
                bq[0][0] = qt[0 * 128 + slot_a];
                bq[0][1] = qt[1 * 128 + slot_a];
                const half8 a00 = as_half8(xs[slot_a]), a10 = as_half8(xs[128 + slot_a]);  // row tile 0 / 1, k step 0
                const half8 a01 = as_half8(xs[slot_b]), a11 = as_half8(xs[128 + slot_b]);  // k step 1
                constexpr int NCP = NCT / 2;   // column-tile pairs per k step
#pragma unroll
                for (int t = 0; t < 2 * NCP; t++) {  // t = ks*NCP + column-tile pair
                    const int ks = t / NCP, cp = t % NCP;
                    if (t + 1 < 2 * NCP) {
                        const int ks2 = (t + 1) / NCP, cp2 = (t + 1) % NCP;
                        if (!(ABL & 8)) {
                            bq[(t + 1) & 1][0] = qt[(cp2 * 2) * 128 + (ks2 ? slot_b : slot_a)];
                            bq[(t + 1) & 1][1] = qt[(cp2 * 2 + 1) * 128 + (ks2 ? slot_b : slot_a)];
                        } else {
                            bq[(t + 1) & 1][0] = bq[t & 1][1];
                            bq[(t + 1) & 1][1] = bq[t & 1][0];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the reads for t+1 ahead of the MFMAs of t
                    if (ILV) {
                        // pieces spread evenly over the 2*NCP MFMA groups: piece p goes before group 1 + p*STEP
                        constexpr int STEP = (2 * NCP) / (QI + 4);
                        if (t >= 1 && (t - 1) % STEP == 0 && (t - 1) / STEP < QI + 4) {
                            issue_piece((t - 1) / STEP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    const half8 a0 = ks ? a01 : a00;
                    const half8 a1 = ks ? a11 : a10;
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const half8 b = as_half8(bq[t & 1][e]);
                        const int ct = cp * 2 + e;
                        if (!(ABL & 1)) {
                            acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, acc[0][ct], 0, 0, 0);
                            acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, acc[1][ct], 0, 0, 0);
                        } else {
                            asm volatile("" ::"v"(b), "v"(a0), "v"(a1));
                        }
                    }
                }
                // next query tile: this wave's share has landed once only the 4 X DMAs issued after it remain;
                // the barrier then makes every wave's share visible to all.  It also separates this iteration's
                // reads of tile `buf` from the DMA that overwrites it two iterations later.
                vm_wait<4>();
                __builtin_amdgcn_s_barrier();
                buf ^= 1;
            }
        }

        // epilogue: per query column, max over this wave's 32 rows (2 row tiles x 4 accumulator rows x 4 lane groups)
        const size_t group = tile * W + wave;
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) {
            float m = fmaxf(fmaxf(acc[0][ct][0], acc[0][ct][1]), fmaxf(acc[0][ct][2], acc[0][ct][3]));
            m = fmaxf(m, fmaxf(fmaxf(acc[1][ct][0], acc[1][ct][1]), fmaxf(acc[1][ct][2], acc[1][ct][3])));
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            if (g == 0 && group < n_groups) gmax[group * (size_t)nq_pad + ct * 16 + i] = m;
        }

        if (!has_next_tile) break;
        tile = next_tile;
#pragma unroll
        for (int u = 0; u < 4; u++) rp[u] = rn[u];
    }
    vm_wait<0>();  // nothing may still be writing this workgroup's LDS when it is released
}


// Round 3: 2-D wave split of the 256-query pass.  The 8 waves form 4 row groups x 2 query halves; a wave owns 64 rows x 128
// queries (4 x 8 MFMA tiles, the same 128 accumulator registers).  Per K block a wave now reads 8 A + 16 B fragments from LDS
// (24 KiB) instead of 4 + 32 (36 KiB), and every B fragment feeds four MFMAs.  The two waves of a row group share one ring
// (S stages x 64 rows x 128 B); each DMAs the 32 rows of its own half.  No extra barrier: the wait + barrier that publishes
// the next query tile at the end of iteration kb also covers X block kb+1 (issued one iteration earlier, and older in the
// in-order vmcnt queue than the query pieces the wait is counted for).
// MF = 16: v_mfma_f32_16x16x32_f16; MF = 32: v_mfma_f32_32x32x16_f16 (2 x 4 tiles of 32 x 32, 16 accumulators each) over the
// same LDS images.
typedef float float16v __attribute__((ext_vector_type(16)));

#ifdef MSE_DEV_KERNELS
// developer profile of the 2-D kernel (PROF = 1): shader cycles a wave spends [0] issuing a K block (barrier release -> closing wait),
// [1] in the closing vmcnt wait, [2] at the barrier; [3] = K blocks counted.  s_memtime stamps sit where lgkmcnt is drained anyway.
__device__ unsigned long long g_scan_prof[4];
__device__ __forceinline__ uint32_t memtime() { return (uint32_t)__builtin_amdgcn_s_memtime(); }   // deltas fit 32 bits
#endif

template <int S, int MF, int PROF = 0>
__global__ __launch_bounds__(W * 64) void scan_mfma2d_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                             const uint4* __restrict__ packed_ro,
                                                             float* __restrict__ gmax, int nq_pad, size_t n_tiles, uint32_t y_packed, uint32_t y_cols) {
    constexpr int BN = 256;
    constexpr int QT_BYTES = BN * 128;
    constexpr int QI = BN / 64;
    constexpr int RG_BYTES = 64 * 128;       // one K block of a row group
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int rg = wave >> 1, qh = wave & 1;
    const int nkb = d / KB;
    const size_t row_bytes = (size_t)d * 2;
    const size_t n_groups = (n_rows + 31) / 32;
    char* const qbase = smem;
    char* const xbase = smem + 2 * QT_BYTES + rg * (S * RG_BYTES);
    gmax += (size_t)blockIdx.y * y_cols;   // batched passes (launch_scan_mfma n_pass > 1): pass y has its own query tiles and columns
    const char* const packed = reinterpret_cast<const char*>(packed_ro + (size_t)blockIdx.y * y_packed) + (size_t)(wave * QI * 64 + lane) * 16;

    size_t tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // this wave's DMA share of a K block: rows qh*32 + 8u + (lane>>3) of the row group, u = 0..3
    auto src_ptr = [&](size_t t, int u) -> const char* {
        const int r = qh * 32 + 8 * u + (lane >> 3);
        size_t row = t * TILE_ROWS + rg * 64 + r;
        if (row >= n_rows) row = n_rows - 1;
        const int piece = (lane & 7) ^ ((r >> 1) & 7);
        return reinterpret_cast<const char*>(base) + row * row_bytes + piece * 16;
    };
    const char* rp[4];
    const char* rn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) rp[u] = src_ptr(tile, u);
    {
        char* qdst = qbase + wave * (QI * 1024);
#pragma unroll
        for (int u = 0; u < QI; u++) dma16(packed + u * 1024, qdst + u * 1024);
#pragma unroll
        for (int j = 0; j < S - 1; j++)
#pragma unroll
            for (int u = 0; u < 4; u++) dma16(rp[u] + (size_t)(j % nkb) * 128, xbase + j * RG_BYTES + (qh * 4 + u) * 1024);
    }
    vm_wait<0>();
    __builtin_amdgcn_s_barrier();

    // fragment addressing: MF = 16: lane (i = lane & 15, g = lane >> 4): row / query i of a 16-tile, 16-byte k slot 4*ks + g;
    //                      MF = 32: lane (i = lane & 31, g = lane >> 5): row / query i of a 32-tile, k slot 2*ks + g (ks = 0..3)
    const int i = MF == 16 ? (lane & 15) : (lane & 31);
    const int g = MF == 16 ? (lane >> 4) : (lane >> 5);
    const int swz = (i >> 1) & 7;
    constexpr int NKS = MF == 16 ? 2 : 4;          // k steps per K block
    constexpr int NRT = 64 / MF, NCTW = 128 / MF;  // row / column tiles of the wave
    constexpr int NACC = MF == 16 ? 4 : 16;
    typedef typename std::conditional<MF == 16, float4v, float16v>::type accv;

    int buf = 0;
#ifdef MSE_DEV_KERNELS
    uint32_t pt_issue = 0, pt_wait = 0, pt_bar = 0, pt_n = 0, pt_last = 0;
    if constexpr (PROF == 1) pt_last = memtime();
#endif
    while (true) {
        accv acc[NRT][NCTW];
#pragma unroll
        for (int rt = 0; rt < NRT; rt++)
#pragma unroll
            for (int ct = 0; ct < NCTW; ct++)
#pragma unroll
                for (int r = 0; r < NACC; r++) acc[rt][ct][r] = 0.0f;

        const size_t next_tile = tile + gridDim.x;
        const bool has_next_tile = next_tile < n_tiles;
#pragma unroll
        for (int u = 0; u < 4; u++) rn[u] = has_next_tile ? src_ptr(next_tile, u) : rp[u];

        for (int kb0 = 0; kb0 < nkb; kb0 += S) {
#pragma unroll
            for (int j = 0; j < S; j++) {
                const int kb = kb0 + j;
                const int kq = kb + 1 == nkb ? 0 : kb + 1;
                const int kf = kb + S - 1;
                const bool x_next = kf >= nkb;
                const size_t xoff = (size_t)(kf >= nkb ? kf - nkb : kf) * 128;
                char* const xdst = xbase + ((j + S - 1) % S) * RG_BYTES + qh * 4096;
                const char* const qsrc = packed + (size_t)kq * QT_BYTES;
                char* const qdst = qbase + (buf ^ 1) * QT_BYTES + wave * (QI * 1024);
                auto issue_piece = [&](int p) {
                    if (p < QI) dma16(qsrc + p * 1024, qdst + p * 1024);
                    else dma16((x_next ? rn[p - QI] : rp[p - QI]) + xoff, xdst + (p - QI) * 1024);
                };
                if (S == 2) vm_wait<0>();

                const u32x4* xs = reinterpret_cast<const u32x4*>(xbase + j * RG_BYTES) + i * 8;
                const u32x4* qt = reinterpret_cast<const u32x4*>(qbase + buf * QT_BYTES) + (qh * 128 + i) * 8;
                auto slot = [&](int ks) { return (MF == 16 ? (4 * ks + g) : (2 * ks + g)) ^ swz; };
                half8 a[2][NRT];   // A fragments of k step ks live in a[ks & 1]; the next k step's are read while this one multiplies
#pragma unroll
                for (int rt = 0; rt < NRT; rt++) a[0][rt] = as_half8(xs[rt * (MF * 8) + slot(0)]);
                constexpr int NT = NKS * NCTW;   // B fragments per K block: 16 in both forms
                u32x4 bq[3];
                bq[0] = qt[0 * (MF * 8) + slot(0)];
                bq[1] = qt[1 * (MF * 8) + slot(0)];
#pragma unroll
                for (int t = 0; t < NT; t++) {   // t = ks * NCTW + ct
                    const int ks = t / NCTW, ct = t % NCTW;
                    if (t + 2 < NT) bq[(t + 2) % 3] = qt[((t + 2) % NCTW) * (MF * 8) + slot((t + 2) / NCTW)];
                    if (ct == NCTW - 2 && ks + 1 < NKS) {
#pragma unroll
                        for (int rt = 0; rt < NRT; rt++) a[(ks + 1) & 1][rt] = as_half8(xs[rt * (MF * 8) + slot(ks + 1)]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        constexpr int STEP = NT / (QI + 4);
                        if (t >= 1 && (t - 1) % STEP == 0 && (t - 1) / STEP < QI + 4) {
                            issue_piece((t - 1) / STEP);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    const half8 b = as_half8(bq[t % 3]);
                    if constexpr (PROF == 3) __builtin_amdgcn_s_setprio(1);   // developer variant: MFMA groups at raised issue priority
#pragma unroll
                    for (int rt = 0; rt < NRT; rt++) {
                        if constexpr (MF == 16) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks & 1][rt], b, acc[rt][ct], 0, 0, 0);
                        else acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][rt], b, acc[rt][ct], 0, 0, 0);
                    }
                    if constexpr (PROF == 3) __builtin_amdgcn_s_setprio(0);
                }
#ifdef MSE_DEV_KERNELS
                if constexpr (PROF == 1) {
                    const uint32_t t0 = memtime();
                    vm_wait<4>();
                    const uint32_t t1 = memtime();
                    __builtin_amdgcn_s_barrier();
                    const uint32_t t2 = memtime();
                    pt_issue += t0 - pt_last; pt_wait += t1 - t0; pt_bar += t2 - t1; pt_n++; pt_last = t2;
                } else
#endif
                {
                    vm_wait<4>();
#ifdef MSE_DEV_KERNELS
                    if constexpr (PROF != 2)   // PROF = 2: timing ablation WITHOUT the barrier (racy, results meaningless)
#endif
                    __builtin_amdgcn_s_barrier();
                }
                buf ^= 1;
            }
        }

        // epilogue: per query column, max over each 32-row group of this wave (two groups)
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const size_t group = tile * W + rg * 2 + p;
#pragma unroll
            for (int ct = 0; ct < NCTW; ct++) {
                float m;
                if constexpr (MF == 16) {
                    m = fmaxf(fmaxf(acc[2 * p][ct][0], acc[2 * p][ct][1]), fmaxf(acc[2 * p][ct][2], acc[2 * p][ct][3]));
                    m = fmaxf(m, fmaxf(fmaxf(acc[2 * p + 1][ct][0], acc[2 * p + 1][ct][1]), fmaxf(acc[2 * p + 1][ct][2], acc[2 * p + 1][ct][3])));
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                } else {
                    m = acc[p][ct][0];
#pragma unroll
                    for (int r = 1; r < 16; r++) m = fmaxf(m, acc[p][ct][r]);
                    m = fmaxf(m, __shfl_xor(m, 32));
                }
                if (g == 0 && group < n_groups) gmax[group * (size_t)nq_pad + qh * 128 + ct * MF + i] = m;
            }
        }

        if (!has_next_tile) break;
        tile = next_tile;
#pragma unroll
        for (int u = 0; u < 4; u++) rp[u] = rn[u];
#ifdef MSE_DEV_KERNELS
        if constexpr (PROF == 1) pt_last = memtime();   // the epilogue is not charged to the next K block
#endif
    }
    vm_wait<0>();
#ifdef MSE_DEV_KERNELS
    if constexpr (PROF == 1) {
        if (lane == 0) {
            atomicAdd(&g_scan_prof[0], (unsigned long long)pt_issue); atomicAdd(&g_scan_prof[1], (unsigned long long)pt_wait);
            atomicAdd(&g_scan_prof[2], (unsigned long long)pt_bar); atomicAdd(&g_scan_prof[3], (unsigned long long)pt_n);
        }
    }
#endif
}


#ifdef MSE_DEV_KERNELS
// Round 3, third form: the 2-D tiling with the DMA ISSUE specialised by wave.  Waves 0-3 issue all row traffic (wave j streams the
// 64 rows of row group j: 8 pieces per K block), waves 4-7 all query-tile traffic (8 pieces each).  Every wave still multiplies its
// 64 x 128 tile.  Why: a wave's loads retire in order, so in the mixed form the closing wait for the next query tile (L2 hits
// issued a few hundred cycles ago) is also a wait for the row pieces in front of them, which pins the row prefetch to ~1.25 K blocks
// of flight time.  A row wave's queue holds only row pieces: block kb+2 is issued from the START of iteration kb and must land by
// the end of iteration kb+1 (~1.9 K blocks of flight), with the same 3-stage ring; a query wave's queue holds only L2 hits.
// Addresses: per lane two 64-bit pointers per tile (row lane>>3 of the row group, even / odd piece swizzle); a piece adds a
// wave-uniform offset (8u rows + the K block).  Needs n_rows % 256 == 0: the launcher sends a ragged tail through scan_mfma_kernel.
// MEASURED (1e7 rows, same box, answers equal the exact-order kernel): 5.98-6.01 ms against 5.59-5.60 ms for the mixed 2-D form --
// the pass is not bound by the flight time of the row prefetch.  Kept in the developer library only.

template <int S>
__global__ __launch_bounds__(W * 64) void scan_mfma2s_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                             const uint4* __restrict__ packed_ro,
                                                             float* __restrict__ gmax, int nq_pad, size_t n_tiles, uint32_t y_packed, uint32_t y_cols) {
    static_assert(S == 3, "the specialised schedule is written for the 3-stage ring");
    constexpr int BN = 256;
    constexpr int QT_BYTES = BN * 128;
    constexpr int RG_BYTES = 64 * 128;
    constexpr int NP = 8;                    // DMA pieces per wave per K block (rows: 64 x 128 B; queries: 1/4 of the 32 KiB tile)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave >> 1, qh = wave & 1;
    const bool row_wave = wave < 4;          // wave-uniform
    const int w4 = wave & 3;
    const int nkb = d / KB;
    const size_t row_bytes = (size_t)d * 2;
    char* const qbase = smem;
    char* const xring = smem + 2 * QT_BYTES;                      // [4 row groups][S][8 KiB]
    char* const xbase = xring + rg * (S * RG_BYTES);              // the ring this wave READS
    char* const xfill = xring + w4 * (S * RG_BYTES);              // the ring a row wave FILLS (row group = wave)
    char* const qfill = qbase + w4 * (NP * 1024);                 // a query wave's quarter of a query tile
    gmax += (size_t)blockIdx.y * y_cols;
    const char* const packed = reinterpret_cast<const char*>(packed_ro + (size_t)blockIdx.y * y_packed) + (size_t)w4 * (NP * 1024) + lane * 16;

    size_t tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // row wave j, piece u: row 8u + (lane >> 3) of row group j, 16-byte piece (lane & 7) ^ f(row), f(r) = (r >> 1) & 7 =
    // ((lane >> 4) + 4 (u & 1)) & 7: the odd pieces' lane offset is the even one with bit 6 flipped (row_bytes is a multiple of 128)
    const size_t lane_off = (size_t)(lane >> 3) * row_bytes + (size_t)(((lane & 7) ^ (lane >> 4)) * 16);
    const char* const rows0 = reinterpret_cast<const char*>(base) + (size_t)w4 * 64 * row_bytes;
    auto tile_ptr = [&](size_t t) { return rows0 + t * TILE_ROWS * row_bytes + lane_off; };
    const char* lp = tile_ptr(tile);   // even pieces; odd pieces: the same address with bit 6 flipped
    const char* ln = lp;
    const size_t piece_rows = 8 * row_bytes;
    auto dma_row_piece = [&](const char* p, int u, size_t koff, char* dst) {
        const char* q = (u & 1) ? reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(p) ^ 64u) : p;
        dma16(q + (u * piece_rows + koff), dst + u * 1024);
    };
    if (row_wave) {
#pragma unroll
        for (int j = 0; j < S - 1; j++)
#pragma unroll
            for (int u = 0; u < NP; u++) dma_row_piece(lp, u, (size_t)(j % nkb) * 128, xfill + j * RG_BYTES);
    } else {
#pragma unroll
        for (int u = 0; u < NP; u++) dma16(packed + u * 1024, qfill + u * 1024);
    }
    vm_wait<0>();
    __builtin_amdgcn_s_barrier();

    const int i = lane & 15, g = lane >> 4;
    const int swz = (i >> 1) & 7;
    int buf = 0;
    while (true) {
        float4v acc[4][8];
#pragma unroll
        for (int rt = 0; rt < 4; rt++)
#pragma unroll
            for (int ct = 0; ct < 8; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[rt][ct][r] = 0.0f;

        const size_t next_tile = tile + gridDim.x;
        const bool has_next_tile = next_tile < n_tiles;
        ln = has_next_tile ? tile_ptr(next_tile) : lp;   // the last tile re-reads its own first blocks

        for (int kb0 = 0; kb0 < nkb; kb0 += S) {
#pragma unroll
            for (int j = 0; j < S; j++) {
                const int kb = kb0 + j;
                const int kq = kb + 1 == nkb ? 0 : kb + 1;
                const int kf = kb + S - 1;
                const char* const xp = kf >= nkb ? ln : lp;
                const size_t xoff = (size_t)(kf >= nkb ? kf - nkb : kf) * 128;
                char* const xdst = xfill + ((j + S - 1) % S) * RG_BYTES;
                const char* const qsrc = packed + (size_t)kq * QT_BYTES;
                char* const qdst = qfill + (buf ^ 1) * QT_BYTES;
                auto issue_piece = [&](int p) {
                    if (row_wave) dma_row_piece(xp, p, xoff, xdst);
                    else dma16(qsrc + p * 1024, qdst + p * 1024);
                };

                const u32x4* xs = reinterpret_cast<const u32x4*>(xbase + j * RG_BYTES) + i * 8;
                const u32x4* qt = reinterpret_cast<const u32x4*>(qbase + buf * QT_BYTES) + (qh * 128 + i) * 8;
                auto slot = [&](int ks) { return (4 * ks + g) ^ swz; };
                half8 a[2][4];
#pragma unroll
                for (int rt = 0; rt < 4; rt++) a[0][rt] = as_half8(xs[rt * 128 + slot(0)]);
                u32x4 bq[3];
                bq[0] = qt[0 * 128 + slot(0)];
                bq[1] = qt[1 * 128 + slot(0)];
#pragma unroll
                for (int t = 0; t < 16; t++) {   // t = ks * 8 + ct
                    const int ks = t / 8, ct = t % 8;
                    if (t + 2 < 16) bq[(t + 2) % 3] = qt[((t + 2) % 8) * 128 + slot((t + 2) / 8)];
                    if (ct == 6 && ks == 0) {
#pragma unroll
                        for (int rt = 0; rt < 4; rt++) a[1][rt] = as_half8(xs[rt * 128 + slot(1)]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (t >= 1 && (t - 1) % 2 == 0 && (t - 1) / 2 < NP) {
                        issue_piece((t - 1) / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const half8 b = as_half8(bq[t % 3]);
#pragma unroll
                    for (int rt = 0; rt < 4; rt++) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ks][rt], b, acc[rt][ct], 0, 0, 0);
                }
                // row wave: everything but the 8 pieces of block kb+2 just issued has landed -> block kb+1 is in LDS;
                // query wave: tile kb+1 complete.  The barrier publishes both to all waves.
                if (row_wave) vm_wait<NP>(); else vm_wait<0>();
                __builtin_amdgcn_s_barrier();
                buf ^= 1;
            }
        }

#pragma unroll
        for (int p = 0; p < 2; p++) {
            const size_t group = tile * W + rg * 2 + p;
#pragma unroll
            for (int ct = 0; ct < 8; ct++) {
                float m = fmaxf(fmaxf(acc[2 * p][ct][0], acc[2 * p][ct][1]), fmaxf(acc[2 * p][ct][2], acc[2 * p][ct][3]));
                m = fmaxf(m, fmaxf(fmaxf(acc[2 * p + 1][ct][0], acc[2 * p + 1][ct][1]), fmaxf(acc[2 * p + 1][ct][2], acc[2 * p + 1][ct][3])));
                m = fmaxf(m, __shfl_xor(m, 16));
                m = fmaxf(m, __shfl_xor(m, 32));
                if (g == 0) gmax[group * (size_t)nq_pad + qh * 128 + ct * 16 + i] = m;
            }
        }
        if (!has_next_tile) break;
        tile = next_tile;
        lp = ln;
    }
    vm_wait<0>();
}
#endif  // MSE_DEV_KERNELS

// batched passes of one launch_scan_mfma call (set by it around its launches; launch_scan_mfma is not re-entered concurrently with
// different values on one thread: thread_local)
thread_local unsigned g_pass_count = 1;
thread_local uint32_t g_pass_packed = 0, g_pass_cols = 0;

template <typename K>
int launch_kernel(K kernel, size_t lds, size_t grid, hipStream_t stream, const uint16_t* base, size_t n_rows, int d,
                  const uint4* packed, float* group_max, int nq_pad) {
    const size_t n_tiles = (n_rows + TILE_ROWS - 1) / TILE_ROWS;
    if (grid > n_tiles) grid = n_tiles;
    MSE_DYN_LDS(kernel, lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid, g_pass_count), dim3(W * 64), lds, stream, base, n_rows, d, packed, group_max, nq_pad, n_tiles,
                       g_pass_packed, g_pass_cols);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

template <int S, int NCT, int ABL>
int launch_variant(size_t grid, hipStream_t stream, const uint16_t* base, size_t n_rows, int d, const uint4* packed,
                   float* group_max, int nq_pad) {
    const size_t lds = 2 * (size_t)(NCT * 16 * 128) + (size_t)W * S * 4096;  // S = 3: 128 KiB at 128 queries, 160 KiB at 256
    return launch_kernel(scan_mfma_kernel<S, NCT, ABL>, lds, grid, stream, base, n_rows, d, packed, group_max, nq_pad);
}

#ifdef MSE_DEV_KERNELS
template <int S, int NCT, int ABL>
int launch_variant(size_t grid, hipStream_t stream, const uint16_t* base, size_t n_rows, int d, const uint4* packed,
                   float* group_max, int nq_pad);

template <int S>
int launch_2s(size_t grid, hipStream_t stream, const uint16_t* base, size_t n_rows, int d, const uint4* packed,
              float* group_max, int nq_pad) {
    const size_t lds = 2 * (size_t)(256 * 128) + (size_t)4 * S * 8192;
    const size_t n_full = n_rows / TILE_ROWS * TILE_ROWS;
    if (n_full && launch_kernel(scan_mfma2s_kernel<S>, lds, grid, stream, base, n_full, d, packed, group_max, nq_pad)) return -1;
    if (n_full < n_rows)   // ragged tail (< 256 rows): one workgroup of the row-clamping kernel
        return launch_variant<S, 16, 0>(1, stream, base + n_full * (size_t)d, n_rows - n_full, d, packed,
                                        group_max + (n_full / 32) * (size_t)nq_pad, nq_pad);
    return 0;
}
#endif

template <int S, int MF, int PROF = 0>
int launch_2d(size_t grid, hipStream_t stream, const uint16_t* base, size_t n_rows, int d, const uint4* packed,
              float* group_max, int nq_pad) {
    const size_t lds = 2 * (size_t)(256 * 128) + (size_t)4 * S * 8192;
    return launch_kernel(scan_mfma2d_kernel<S, MF, PROF>, lds, grid, stream, base, n_rows, d, packed, group_max, nq_pad);
}

}  // namespace

// largest pass: 320 queries on the two-stage ring (the K-block count must be even), else 256
int mfma_query_tile(int d) { return (d / KB) % 2 == 0 ? 320 : 256; }
// padded query count of a pass of nq queries: 128, 192 (three-stage ring only: the K-block count must divide by 3) or 256
int mfma_pad(int nq, int d) {
    if (nq <= 128) return 128;
    if (nq <= 192 && (d / KB) % 3 == 0) return 192;
    if (nq <= 256 || (d / KB) % 2 != 0) return 256;
    return 320;
}

size_t mfma_packed_bytes(int d) { return (size_t)(d / KB) * 320 * 128; }

// packed_scratch: mfma_packed_bytes(d) bytes of device scratch owned by the caller (per searcher)
int launch_scan_mfma(const uint16_t* base, size_t n_rows, int d, const uint16_t* queries_dev, int nq_pad,
                     void* packed_scratch, float* group_max, int n_cu, hipStream_t stream, hipEvent_t ev_begin,
                     hipEvent_t ev_end, int gm_stride, int n_pass) {
    if (n_rows == 0) return 0;
    if (n_pass < 1 || n_pass > 65535) return fail("scan_mfma: 1..65535 passes per launch");
    if (n_pass > 1 && !gm_stride) return fail("scan_mfma: batched passes need a common group-maximum stride");
    // the kernels use their last integer argument only as the row stride of group_max: a pass may write its nq_pad columns into a
    // wider array (several passes side by side, api.hip mfma_pass over a small base)
    const int gs = gm_stride ? gm_stride : nq_pad;
    if (gs < nq_pad) return fail("scan_mfma: group-maximum stride below the pass width");
    if (d % 64 != 0 || d <= 0 || d > D_MAX) return fail("vector width must be a positive multiple of 64");
    if (nq_pad != 128 && nq_pad != 256 && !(nq_pad == 192 && (d / KB) % 3 == 0) && !(nq_pad == 320 && (d / KB) % 2 == 0))
        return fail("scan_mfma: query tile must be padded to 128, 192 (K blocks divisible by 3), 256 or 320 (K blocks even)");
    uint4* packed = reinterpret_cast<uint4*>(packed_scratch);
    // n_pass > 1: queries_dev holds n_pass x nq_pad rows, packed_scratch n_pass x mfma_packed_bytes(d), and pass y writes columns
    // [y * nq_pad, (y + 1) * nq_pad) of group_max -- ONE launch; a small base (a few row tiles) then fills the chip with its passes
    // side by side instead of running them one after the other on a few CUs
    hipLaunchKernelGGL(pack_queries_kernel, dim3(64, (unsigned)n_pass), dim3(256), 0, stream, queries_dev, d, nq_pad, packed);
    struct PassScope {
        PassScope(unsigned n, uint32_t pk, uint32_t cols) { g_pass_count = n; g_pass_packed = pk; g_pass_cols = cols; }
        ~PassScope() { g_pass_count = 1; g_pass_packed = 0; g_pass_cols = 0; }
    } pass_scope((unsigned)n_pass, (uint32_t)((size_t)(d / KB) * nq_pad * 8), (uint32_t)nq_pad);
    if (ev_begin) MSE_HIP_TRY(hipEventRecord(ev_begin, stream));
    const int nkb = d / KB;
    const int S = nkb % 3 == 0 ? 3 : nkb % 2 == 0 ? 2 : 1;   // ring depth: 3 when the K-block count allows it
    const size_t grid = (size_t)n_cu;  // one workgroup per CU
    int rc = -2;
#ifdef MSE_DEV_KERNELS
    // Developer build only (make dev -> libmse_hip_dev.so, scripts/scan_ablate.py): timing ablations whose RESULTS ARE WRONG
    // and alternative tilings.  None of this is compiled into the product library.
    const int abl = getenv("MSE_SCAN_ABL") ? atoi(getenv("MSE_SCAN_ABL")) : 0;
    const int v2d = getenv("MSE_SCAN_2D") ? atoi(getenv("MSE_SCAN_2D")) : -1;
    if (nq_pad == 128 && S == 3 && abl == 16) rc = launch_variant<3, 8, 16>(grid, stream, base, n_rows, d, packed, group_max, gs);
    else if (nq_pad == 256 && S == 3 && abl) {   // timing ablations of the round-2 kernel
        switch (abl) {
            case 1: rc = launch_variant<3, 16, 1>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            case 2: rc = launch_variant<3, 16, 2>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            case 4: rc = launch_variant<3, 16, 4>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            case 6: rc = launch_variant<3, 16, 6>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            case 8: rc = launch_variant<3, 16, 8>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            case 9: rc = launch_variant<3, 16, 9>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            case 14: rc = launch_variant<3, 16, 14>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            case 16: rc = launch_variant<3, 16, 16>(grid, stream, base, n_rows, d, packed, group_max, gs); break;
            default: break;
        }
    }
    else if (nq_pad == 256 && S == 3 && v2d == 0) rc = launch_variant<3, 16, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
    else if (nq_pad == 256 && S == 3 && v2d == 161) rc = launch_2d<3, 16, 1>(grid, stream, base, n_rows, d, packed, group_max, gs);
    else if (nq_pad == 256 && S == 3 && v2d == 162) rc = launch_2d<3, 16, 2>(grid, stream, base, n_rows, d, packed, group_max, gs);
    else if (nq_pad == 256 && S == 3 && v2d == 163) rc = launch_2d<3, 16, 3>(grid, stream, base, n_rows, d, packed, group_max, gs);
    else if (nq_pad == 256 && S == 3 && v2d == 17) rc = launch_2s<3>(grid, stream, base, n_rows, d, packed, group_max, gs);
    else if (nq_pad == 256 && S == 3 && v2d == 32) rc = launch_2d<3, 32>(grid, stream, base, n_rows, d, packed, group_max, gs);
    if (rc != -2) {
        if (rc) return rc;
        if (ev_end) MSE_HIP_TRY(hipEventRecord(ev_end, stream));
        return 0;
    }
#endif
    if (nq_pad == 320) {
        // 20 column tiles per wave (32 rows x 320 queries, 160 accumulator registers of 252 used: 24 tiles spill and run 2.6x
        // slower), two-stage row ring: 2 x 40 KiB of query tiles + 8 x 2 x 4 KiB = 144 KiB of LDS.
        rc = launch_variant<2, 20, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
    } else if (nq_pad == 192) {
        // 12 column tiles on the one-dimensional wave split (32 rows x 192 queries per wave, 96 accumulators): the point between
        // the HBM-bound 128-query pass and the power-bound 256-query pass (profiles/r04_scan_variants.txt)
        rc = launch_variant<3, 12, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
    } else if (nq_pad == 128) {
        if (S == 3) rc = launch_variant<3, 8, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
        else if (S == 2) rc = launch_variant<2, 8, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
        else rc = launch_variant<1, 8, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
    } else {
        if (S == 3) rc = launch_2d<3, 16>(grid, stream, base, n_rows, d, packed, group_max, gs);
        else if (S == 2) rc = launch_variant<2, 16, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
        else rc = launch_variant<1, 16, 0>(grid, stream, base, n_rows, d, packed, group_max, gs);
    }
    if (rc) return rc;
    if (ev_end) MSE_HIP_TRY(hipEventRecord(ev_end, stream));
    return 0;
}

}  // namespace mse

#ifdef MSE_DEV_KERNELS
// developer library only (not in include/mse.h): read and clear the counters of the profiled 2-D scan (MSE_SCAN_2D=161)
extern "C" __attribute__((visibility("default"))) int mse_dev_scan_prof(unsigned long long out[4]) {
    unsigned long long z[4] = {0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mse::g_scan_prof), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mse::g_scan_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
