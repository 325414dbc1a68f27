// Deterministic top-k selection over score arrays: a tournament of group maxima plus an exact
// radix select, ordering candidates by (score descending, id ascending).
//
// The reference ranks brute-force scores with a full sort (src/query_disk_index.rs:271) and
// FAISS keeps a heap (src/main.rs:900); both only define "the k largest".  Ties among equal
// scores are unspecified there (sort_unstable) and fixed here as lower id first.
//
// Structure: level 0 is the per-row score array (or per-32-row group maxima from the MFMA scan).
// `reduce_max` builds levels of 256-way group maxima until a level is small; `select` picks the k
// best entries of the top level, then of the children of those, and so on down.  The k best rows
// always lie inside the k best groups of every level (a group ranked ahead of row r's group holds
// a row ranked ahead of r), so the descent is exact, not heuristic.
//
// `select` = one workgroup per query; 96-bit composite key (sortable score, ~id) is unique per
// candidate, so an MSB-first byte-wise radix select finds the exact k-th composite, and the
// selected set is {composite >= threshold}.  No data-dependent early exits: cost depends only on
// the candidate count.
#include "common.h"
#include "kernels.h"
#include <type_traits>
#include <algorithm>

namespace mse { int device_cu_count(); }   // api.hip

namespace mse {
namespace {

__device__ __forceinline__ uint64_t key_of(int64_t v) { return sortable_i64(v); }
__device__ __forceinline__ uint64_t key_of(uint64_t v) { return v; }
__device__ __forceinline__ uint32_t key_of(float v) { return sortable_f32_bits(__float_as_uint(v)); }
__device__ __forceinline__ uint32_t key_of(uint32_t v) { return v; }

template <typename T> struct KeyT;
template <> struct KeyT<int64_t> { using type = uint64_t; };
template <> struct KeyT<uint64_t> { using type = uint64_t; };
template <> struct KeyT<float> { using type = uint32_t; };
template <> struct KeyT<uint32_t> { using type = uint32_t; };

template <typename K> __device__ __forceinline__ K wave_max(K v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        K other = __shfl_xor(v, o);
        v = other > v ? other : v;
    }
    return v;
}

// one wave per output group (F must be 256), a bounded number of waves walking the (query, group) pairs.  The grid is capped
// (launch_reduce_t) so that this kernel never holds more than half of a CU's wave slots: as the first kernel of a batched PQ scan's
// tail it is dispatched at the same moment as the NEXT scan (other stream), whose 16-wave workgroups need four free slots on every
// SIMD of a CU -- 12 000 four-wave workgroups churning through all 32 slots kept them out for the kernel's whole 140 us, and a
// scan workgroup that starts late ends late (static partition): 1.40 ms per scan instead of 1.22 (profiles/r04_pq_timeline.txt).
// Element-strided input (group-major [n_in][nq], in_estride = nq): consecutive waves take the queries of ONE group run, so the
// cache lines they share are fetched once per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void reduce_max_kernel(const T* __restrict__ in, size_t in_stride, size_t n_in,
                                                         typename KeyT<T>::type* __restrict__ out, size_t out_stride,
                                                         size_t n_out, int nq, size_t in_estride) {
    using K = typename KeyT<T>::type;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t total = n_out * (size_t)nq;
    for (size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < total; wave += n_waves) {
        size_t q, g;
        if (in_estride > 1) { q = wave % (size_t)nq; g = wave / (size_t)nq; }
        else { q = wave / n_out; g = wave % n_out; }
        const T* src = in + q * in_stride;
        K best = 0;
        const size_t lo = g * TOPK_FANOUT;
        T v[TOPK_FANOUT / 64];
#pragma unroll
        for (int u = 0; u < TOPK_FANOUT / 64; u++) {   // all loads of the run first
            const size_t i = lo + (size_t)u * 64 + lane;
            v[u] = i < n_in ? src[i * in_estride] : T(0);
        }
#pragma unroll
        for (int u = 0; u < TOPK_FANOUT / 64; u++) {
            const size_t i = lo + (size_t)u * 64 + lane;
            if (i < n_in) {
                const K k = key_of(v[u]);
                best = k > best ? k : best;
            }
        }
        best = wave_max(best);
        if (lane == 0) out[q * out_stride + g] = best;
    }
}

// group-major float input [n_in][nq_pad] (as written by the MFMA scan): thread = (query, out group)
__global__ __launch_bounds__(256) void reduce_max_gq_kernel(const float* __restrict__ in, int nq_pad, size_t n_in,
                                                            uint32_t* __restrict__ out, size_t out_stride,
                                                            size_t n_out, int nq) {
    const int q = blockIdx.y * blockDim.x + threadIdx.x;
    const size_t g = blockIdx.x;
    if (q >= nq || g >= n_out) return;
    const size_t lo = g * TOPK_FANOUT;
    const size_t hi = lo + TOPK_FANOUT < n_in ? lo + TOPK_FANOUT : n_in;
    // eight independent loads in flight per thread (the plain loop issued them one at a time: 4.0 of a possible ~5.5 TB/s over the
    // 3.2 GB of group maxima of a 1e8-row, 256-query pass)
    uint32_t b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = lo;
    for (; i + 8 <= hi; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t k = key_of(in[(i + u) * nq_pad + q]);
            b[u] = k > b[u] ? k : b[u];
        }
    }
    for (; i < hi; i++) {
        const uint32_t k = key_of(in[i * nq_pad + q]);
        b[0] = k > b[0] ? k : b[0];
    }
    uint32_t best = b[0];
#pragma unroll
    for (int u = 1; u < 8; u++) best = b[u] > best ? b[u] : best;
    out[(size_t)q * out_stride + g] = best;
}

struct Composite {
    uint64_t hi;  // sortable score key (u32 keys occupy the top 32 bits)
    uint32_t lo;  // ~id, so that a lower id is a larger composite
};
__device__ __forceinline__ bool ge(const Composite& a, const Composite& b) {
    return a.hi > b.hi || (a.hi == b.hi && a.lo >= b.lo);
}
__device__ __forceinline__ bool gt(const Composite& a, const Composite& b) {
    return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo);
}
// digit positions 11..4 = bytes of hi (MSB first), 3..0 = bytes of lo
__device__ __forceinline__ int digit_of(const Composite& c, int pos) {
    return pos >= 4 ? (int)((c.hi >> (8 * (pos - 4))) & 0xff) : (int)((c.lo >> (8 * pos)) & 0xff);
}
__device__ __forceinline__ bool match_above(const Composite& c, const Composite& p, int pos) {
    // digits strictly above `pos` equal
    if (pos >= 11) return true;
    if (pos >= 4) {
        const int sh = 8 * (pos - 4 + 1);
        return (c.hi >> sh) == (p.hi >> sh);
    }
    if (c.hi != p.hi) return false;
    if (pos == 3) return true;
    const int sh = 8 * (pos + 1);
    return (c.lo >> sh) == (p.lo >> sh);
}

constexpr int SEL_THREADS = 1024;
constexpr int SEL_LIST_CAP = 4096;

template <typename T>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(SelectArgs a, size_t in_estride) {
    using K = typename KeyT<T>::type;
    constexpr bool K32 = sizeof(K) == 4;
    __shared__ uint32_t s_parents[TOPK_KMAX];
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_sel_hi[TOPK_KMAX];
    __shared__ uint32_t s_sel_lo[TOPK_KMAX];
    __shared__ uint64_t s_prefix_hi;
    __shared__ uint32_t s_prefix_lo;
    __shared__ uint32_t s_need, s_count, s_valid, s_flag, s_list_n;
    __shared__ unsigned long long s_or_hi, s_and_hi;   // OR / AND of all valid score keys: digits on which they agree need no pass
    __shared__ int s_next_pos;
    __shared__ uint32_t s_list[SEL_LIST_CAP];

    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const T* in = reinterpret_cast<const T*>(a.in) + (size_t)q * a.in_stride;
    const bool use_list = a.list_ids != nullptr;
    const bool use_par = !use_list && a.parents != nullptr;
    const uint32_t* list_ids = use_list ? a.list_ids + (size_t)q * a.list_stride : nullptr;
    const T* list_keys = use_list ? reinterpret_cast<const T*>(a.list_keys) + (size_t)q * a.list_stride : nullptr;
    size_t M;
    if (use_list) M = a.n_list;
    else if (use_par) M = a.n_par * (size_t)a.fanout;
    else M = a.n_in;

    if (use_par)
        for (size_t i = tid; i < a.n_par; i += SEL_THREADS) s_parents[i] = a.parents[(size_t)q * a.par_stride + i];
    if (tid == 0) { s_prefix_hi = 0; s_prefix_lo = 0; s_count = 0; s_valid = 0; s_or_hi = 0ull; s_and_hi = ~0ull; }
    __syncthreads();

    const int fshift = (a.fanout > 0 && (a.fanout & (a.fanout - 1)) == 0) ? __ffs(a.fanout) - 1 : -1;
    const uint64_t floor_hi = a.floor_hi ? a.floor_hi[q] : 0ull;
    auto load = [&](size_t c, Composite& out) -> bool {
        uint32_t id;
        K key;
        if (use_list) {
            size_t ci = c, ii = c;
            if (a.list_chunk) {
                const size_t ch = c / a.list_chunk, r = c % a.list_chunk;
                ci = ch * a.list_chunk_stride + r;
                ii = a.list_id_chunk_stride ? ch * a.list_id_chunk_stride + r : ci;
            }
            id = list_ids[ii];
            if (id == ID_NONE) return false;
            key = key_of(list_keys[ci]);
        } else if (use_par) {
            // fanout is a power of two in every caller (256 / 32): shift and mask instead of two integer divisions per candidate
            const size_t pi = fshift >= 0 ? (c >> fshift) : c / a.fanout;
            const size_t ci = fshift >= 0 ? (c & (size_t)(a.fanout - 1)) : c % a.fanout;
            const uint32_t p = s_parents[pi];
            if (p == ID_NONE) return false;
            const size_t child = (size_t)p * a.fanout + ci;
            if (child >= a.n_in) return false;
            id = (uint32_t)child;
            key = key_of(in[child * in_estride]);
        } else {
            id = (uint32_t)c;
            key = key_of(in[c * in_estride]);
        }
        out.hi = K32 ? ((uint64_t)key << 32) : (uint64_t)key;
        out.lo = ~id;
        return out.hi >= floor_hi;
    };

    // MSB-first radix select on the composite.  The first pass also counts the valid candidates.  As soon as the
    // chosen bucket holds at most SEL_LIST_CAP candidates their positions are gathered into an LDS list and the
    // remaining passes walk that list instead of all M candidates; a pass whose bucket holds exactly the number still
    // needed ends the search (the lower digits of the threshold stay zero).  Both shortcuts are functions of the
    // multiset of composites only, so the result is as deterministic as before.
    bool list_mode = false, full_list = false;
    if (a.floor_hi) {
        // A floor is given (the k-th key of the level above: at least k candidates reach it, and usually not many more): gather
        // the candidates that reach it ONCE -- if they fit the LDS list every radix pass and the final collection walk that list
        // instead of all M candidates (the 77 k children of 301 nominated PQ groups hold a few hundred keys above the floor).
        if (tid == 0) s_list_n = 0;
        __syncthreads();
        Composite c;
        for (size_t i = tid; i < M; i += SEL_THREADS)
            if (load(i, c)) {
                const uint32_t slot = atomicAdd(&s_list_n, 1u);
                if (slot < (uint32_t)SEL_LIST_CAP) s_list[slot] = (uint32_t)i;
            }
        __syncthreads();
        if (s_list_n <= (uint32_t)SEL_LIST_CAP) { list_mode = true; full_list = true; }
        __syncthreads();
    }
    for (int pos = 11; pos >= 0; pos--) {
        if (K32 && pos >= 4 && pos < 8) continue;  // low half of `hi` is zero for 32-bit keys
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const Composite prefix{s_prefix_hi, s_prefix_lo};
        Composite c;
        if (list_mode && pos == 11) {
            unsigned long long o = 0ull, n = ~0ull;
            const uint32_t ln = s_list_n;
            for (uint32_t j = tid; j < ln; j += SEL_THREADS)
                if (load(s_list[j], c)) { atomicAdd(&s_hist[digit_of(c, pos)], 1u); o |= c.hi; n &= c.hi; }
            if (o | ~n) { atomicOr(&s_or_hi, o); atomicAnd(&s_and_hi, n); }
        } else if (list_mode) {
            const uint32_t ln = s_list_n;
            for (uint32_t j = tid; j < ln; j += SEL_THREADS)
                if (load(s_list[j], c) && match_above(c, prefix, pos)) atomicAdd(&s_hist[digit_of(c, pos)], 1u);
        } else if (pos == 11) {
            unsigned long long o = 0ull, n = ~0ull;
            for (size_t i = tid; i < M; i += SEL_THREADS)
                if (load(i, c)) { atomicAdd(&s_hist[digit_of(c, pos)], 1u); o |= c.hi; n &= c.hi; }
            if (o | ~n) { atomicOr(&s_or_hi, o); atomicAnd(&s_and_hi, n); }
        } else {
            for (size_t i = tid; i < M; i += SEL_THREADS)
                if (load(i, c) && match_above(c, prefix, pos)) atomicAdd(&s_hist[digit_of(c, pos)], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // the bucket holding the need-th largest digit, found by wave 0: lane l owns buckets 4l .. 4l+3, a suffix sum over
            // the lanes replaces the serial walk over 256 LDS counters (8 us per pass when one thread does it)
            const uint32_t h0 = s_hist[4 * tid], h1 = s_hist[4 * tid + 1], h2 = s_hist[4 * tid + 2], h3 = s_hist[4 * tid + 3];
            uint32_t suf = h0 + h1 + h2 + h3;            // becomes the inclusive suffix sum over lanes >= tid
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t other = __shfl_down(suf, o);
                if (tid + o < 64) suf += other;
            }
            const uint32_t total = __shfl(suf, 0);
            const uint32_t above = suf - (h0 + h1 + h2 + h3);   // candidates in buckets of higher lanes
            uint32_t need = s_need;
            int flag = 0;
            if (pos == 11) {
                need = total < (uint32_t)a.k ? total : (uint32_t)a.k;
                if (total <= (uint32_t)a.k) flag = 2;   // take everything
            }
            // exactly one lane has above < need <= suf (lane 0 takes the walk's default, bucket 0, if none does).  That lane works
            // out its answer in registers; lane 0 fetches it by shuffle and is the ONLY lane that stores to the shared state
            // (two lanes storing to s_flag / s_need / s_next_pos relied on the order of divergent LDS stores within a wave).
            const bool mine = flag == 0 && ((above < need && need <= suf) || (tid == 0 && need > suf));
            uint32_t w_need = need, w_h = 0;
            int w_b = 0;
            if (mine) {
                uint32_t cum = above;
                int b = 4 * tid + 3;
                const uint32_t hh[4] = {h0, h1, h2, h3};
                for (; b > 4 * tid; b--) {
                    if (cum + hh[b - 4 * tid] >= need) break;
                    cum += hh[b - 4 * tid];
                }
                w_need = need - cum;
                w_b = b;
                w_h = hh[b - 4 * tid];
            }
            const unsigned long long winners = __ballot(mine);
            const int src = winners ? (int)__ffsll((long long)winners) - 1 : 0;
            const uint32_t r_need = __shfl(w_need, src), r_h = __shfl(w_h, src);
            const int r_b = __shfl(w_b, src);
            if (tid == 0) {
                int next_pos = pos - 1, f = flag;
                uint32_t new_need = need;
                if (pos == 11) s_valid = total;
                if (winners) {
                    new_need = r_need;
                    if (pos >= 4) s_prefix_hi |= (uint64_t)r_b << (8 * (pos - 4));
                    else s_prefix_lo |= (uint32_t)r_b << (8 * pos);
                    if (pos == 11) {
                        // score digits on which every valid candidate agrees (i64 scores below 2^33 share their top bytes)
                        // are copied into the threshold without a pass
                        const unsigned long long diff = s_or_hi ^ s_and_hi;
                        int np = 10;
                        while (np >= 4 && ((diff >> (8 * (np - 4))) & 0xffull) == 0ull) {
                            s_prefix_hi |= ((s_or_hi >> (8 * (np - 4))) & 0xffull) << (8 * (np - 4));
                            np--;
                        }
                        next_pos = np;
                    }
                    if (r_h == r_need) f = 1;                                                   // bucket taken whole: done
                    else if (!list_mode && pos > 0 && r_h <= (uint32_t)SEL_LIST_CAP) f = 3;     // gather the bucket
                }
                s_need = new_need;
                s_next_pos = next_pos;
                s_flag = f;
            }
        }
        __syncthreads();
        const uint32_t flag = s_flag;
        const int next_pos = s_next_pos;
        if (flag == 1 || flag == 2) break;
        if (flag == 3) {
            if (tid == 0) s_list_n = 0;
            __syncthreads();
            const Composite np{s_prefix_hi, s_prefix_lo};
            for (size_t i = tid; i < M; i += SEL_THREADS)
                if (load(i, c) && match_above(c, np, pos - 1)) {
                    const uint32_t slot = atomicAdd(&s_list_n, 1u);
                    if (slot < (uint32_t)SEL_LIST_CAP) s_list[slot] = (uint32_t)i;
                }
            __syncthreads();
            list_mode = true;
        }
        pos = next_pos + 1;   // the loop's pos-- lands on next_pos
    }
    __syncthreads();
    const bool take_all = s_valid <= (uint32_t)a.k;
    const Composite thr{take_all ? 0ull : s_prefix_hi, take_all ? 0u : s_prefix_lo};
    if (a.kth_hi_out && tid == 0) a.kth_hi_out[q] = take_all ? 0ull : s_prefix_hi;   // a lower bound of the k-th key when the search ended early
    {
        Composite c;
        if (full_list) {      // the list holds every valid candidate
            const uint32_t ln = s_list_n;
            for (uint32_t j = tid; j < ln; j += SEL_THREADS)
                if (load(s_list[j], c) && ge(c, thr)) {
                    const uint32_t p = atomicAdd(&s_count, 1u);
                    if (p < (uint32_t)TOPK_KMAX) { s_sel_hi[p] = c.hi; s_sel_lo[p] = c.lo; }
                }
        } else {
            for (size_t i = tid; i < M; i += SEL_THREADS)
                if (load(i, c) && ge(c, thr)) {
                    const uint32_t p = atomicAdd(&s_count, 1u);
                    if (p < (uint32_t)TOPK_KMAX) { s_sel_hi[p] = c.hi; s_sel_lo[p] = c.lo; }
                }
        }
    }
    __syncthreads();
    const uint32_t n_sel = s_count < (uint32_t)a.k ? s_count : (uint32_t)a.k;
    uint32_t* out_ids = a.out_ids + (size_t)q * a.out_stride;
    T* out_keys = a.out_keys ? reinterpret_cast<T*>(a.out_keys) + (size_t)q * a.out_stride : nullptr;
    // rank sort (composites are unique)
    for (uint32_t i = tid; i < n_sel; i += SEL_THREADS) {
        const Composite me{s_sel_hi[i], s_sel_lo[i]};
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n_sel; j++) rank += gt(Composite{s_sel_hi[j], s_sel_lo[j]}, me) ? 1u : 0u;
        out_ids[rank] = ~me.lo;
        if (out_keys) {
            if constexpr (sizeof(T) == 8) {
                if constexpr (std::is_same<T, int64_t>::value) out_keys[rank] = unsortable_i64(me.hi);
                else out_keys[rank] = (T)me.hi;
            } else {
                const uint32_t k32 = (uint32_t)(me.hi >> 32);
                if constexpr (std::is_same<T, float>::value) out_keys[rank] = __uint_as_float(unsortable_f32_bits(k32));
                else out_keys[rank] = (T)k32;
            }
        }
    }
    for (uint32_t i = n_sel + tid; i < (uint32_t)a.k; i += SEL_THREADS) {
        out_ids[i] = ID_NONE;
        if (out_keys) {
            if constexpr (std::is_same<T, int64_t>::value) out_keys[i] = INT64_MIN;
            else if constexpr (std::is_same<T, float>::value) out_keys[i] = -__builtin_inff();
            else out_keys[i] = 0;
        }
    }
}

__global__ void expand_groups_kernel(const uint32_t* __restrict__ parents, size_t par_stride, size_t n_par, int group,
                                     size_t n_rows, uint32_t* __restrict__ ids, size_t ids_stride, int nq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_q = n_par * (size_t)group;
    if (i >= per_q * (size_t)nq) return;
    const size_t q = i / per_q, r = i % per_q;
    const uint32_t p = parents[q * par_stride + r / group];
    uint32_t id = ID_NONE;
    if (p != ID_NONE) {
        const size_t row = (size_t)p * group + (r % group);
        if (row < n_rows) id = (uint32_t)row;
    }
    ids[q * ids_stride + r] = id;
}

__global__ void finalize_kernel(const uint32_t* __restrict__ sel_ids, const int64_t* __restrict__ sel_scores,
                                size_t sel_stride, int k, int nq, uint64_t id_offset, int64_t* __restrict__ out_scores,
                                uint32_t* __restrict__ out_ids, size_t out_stride, const float* __restrict__ group_keys,
                                size_t gk_stride, int kg, size_t n_groups, const float* __restrict__ eps,
                                float* __restrict__ margin) {
    const int q = blockIdx.x;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint32_t id = sel_ids[(size_t)q * sel_stride + i];
        out_ids[(size_t)q * out_stride + i] = id == ID_NONE ? ID_NONE : (uint32_t)(id + id_offset);
        out_scores[(size_t)q * out_stride + i] = id == ID_NONE ? INT64_MIN : sel_scores[(size_t)q * sel_stride + i];
    }
    if (threadIdx.x == 0 && margin) {
        // Certificate for the approximate (MFMA) candidate stage: every row outside the kg selected
        // groups has approximate score <= g = key of the worst selected group; its exact score is
        // <= g + eps.  If the exact k-th best score exceeds that, no excluded row can enter the top k.
        float m;
        if ((size_t)kg >= n_groups) {
            m = __builtin_inff();  // every group was re-scored exactly
        } else {
            const float g = group_keys[(size_t)q * gk_stride + (kg - 1)];
            const uint32_t idk = sel_ids[(size_t)q * sel_stride + (k - 1)];
            if (idk == ID_NONE) {
                m = -__builtin_inff();
            } else {
                const float sk = (float)((double)sel_scores[(size_t)q * sel_stride + (k - 1)] / 4294967296.0);
                m = sk - (g + eps[q]);
            }
        }
        margin[q] = m;
    }
}

// the same certificate for f32 keys (the flat index keeps FAISS's float distances): margin[q] = k-th exact key - (g + eps)
__global__ void margin_f32_kernel(const uint32_t* __restrict__ sel_ids, const float* __restrict__ sel_keys, size_t sel_stride, int k,
                                  const float* __restrict__ group_keys, size_t gk_stride, int kg, size_t n_groups,
                                  const float* __restrict__ eps, float* __restrict__ margin, int nq) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float m;
    if ((size_t)kg >= n_groups) {
        m = __builtin_inff();
    } else if (sel_ids[(size_t)q * sel_stride + (k - 1)] == ID_NONE) {
        m = -__builtin_inff();
    } else {
        m = sel_keys[(size_t)q * sel_stride + (k - 1)] - (group_keys[(size_t)q * gk_stride + (kg - 1)] + eps[q]);
    }
    margin[q] = m;
}

template <typename T>
int launch_reduce_t(const void* in, size_t in_stride, size_t n_in, void* out, size_t out_stride, size_t n_out, int nq,
                    hipStream_t stream, size_t in_estride) {
    const size_t waves = n_out * (size_t)nq;
    // at most four 4-wave workgroups per CU (16 of its 32 wave slots: see the kernel)
    const size_t blocks = std::min<size_t>((waves + 3) / 4, (size_t)mse::device_cu_count() * 4);
    hipLaunchKernelGGL(reduce_max_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       reinterpret_cast<const T*>(in), in_stride, n_in,
                       reinterpret_cast<typename KeyT<T>::type*>(out), out_stride, n_out, nq, in_estride);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

int launch_reduce_max(KeyKind kind, const void* in, size_t in_stride, size_t n_in, void* out, size_t out_stride,
                      size_t n_out, int nq, hipStream_t stream, size_t in_estride) {
    if (n_out == 0 || nq == 0) return 0;
    switch (kind) {
        case KEY_I64: return launch_reduce_t<int64_t>(in, in_stride, n_in, out, out_stride, n_out, nq, stream, in_estride);
        case KEY_U64: return launch_reduce_t<uint64_t>(in, in_stride, n_in, out, out_stride, n_out, nq, stream, in_estride);
        case KEY_F32: return launch_reduce_t<float>(in, in_stride, n_in, out, out_stride, n_out, nq, stream, in_estride);
        case KEY_U32: return launch_reduce_t<uint32_t>(in, in_stride, n_in, out, out_stride, n_out, nq, stream, in_estride);
    }
    return fail("reduce_max: bad key kind");
}

int launch_reduce_max_gq(const float* in, int nq_pad, size_t n_in, uint32_t* out, size_t out_stride, size_t n_out,
                         int nq, hipStream_t stream) {
    if (n_out == 0 || nq == 0) return 0;
    if (n_out > 0x7fffffffull) return fail("reduce_max_gq: grid too large");
    const int threads = 64;
    hipLaunchKernelGGL(reduce_max_gq_kernel, dim3((unsigned)n_out, (unsigned)((nq + threads - 1) / threads)),
                       dim3(threads), 0, stream, in, nq_pad, n_in, out, out_stride, n_out, nq);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_select_strided(const SelectArgs& a, size_t in_estride, hipStream_t stream) {
    if (a.nq == 0 || a.k == 0) return 0;
    if (a.k > TOPK_KMAX) return fail("select: k exceeds TOPK_KMAX");
    if (a.parents && a.n_par > (size_t)TOPK_KMAX) return fail("select: too many parents");
    switch (a.kind) {
        case KEY_I64: hipLaunchKernelGGL(select_kernel<int64_t>, dim3(a.nq), dim3(SEL_THREADS), 0, stream, a, in_estride); break;
        case KEY_U64: hipLaunchKernelGGL(select_kernel<uint64_t>, dim3(a.nq), dim3(SEL_THREADS), 0, stream, a, in_estride); break;
        case KEY_F32: hipLaunchKernelGGL(select_kernel<float>, dim3(a.nq), dim3(SEL_THREADS), 0, stream, a, in_estride); break;
        case KEY_U32: hipLaunchKernelGGL(select_kernel<uint32_t>, dim3(a.nq), dim3(SEL_THREADS), 0, stream, a, in_estride); break;
    }
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_select(const SelectArgs& a, hipStream_t stream) { return launch_select_strided(a, 1, stream); }

int launch_expand_groups(const uint32_t* parents, size_t par_stride, size_t n_par, int group, size_t n_rows,
                         uint32_t* ids, size_t ids_stride, int nq, hipStream_t stream) {
    const size_t total = n_par * (size_t)group * (size_t)nq;
    if (total == 0) return 0;
    hipLaunchKernelGGL(expand_groups_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, parents,
                       par_stride, n_par, group, n_rows, ids, ids_stride, nq);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- helpers of the per-query widening (api.hip mfma_pass): the few queries whose certificate failed are carried on as a compact set
__global__ void gather_columns_kernel(const float* __restrict__ in, int nq_pad, size_t n_groups, const uint32_t* __restrict__ idx, int nb,
                                      float* __restrict__ out, int nbp) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t g = t / (size_t)nbp;
    const int j = (int)(t % (size_t)nbp);
    if (g >= n_groups) return;
    out[t] = j < nb ? in[g * (size_t)nq_pad + idx[j]] : -__builtin_inff();
}
__global__ void gather_rows16_kernel(const uint4* __restrict__ in, size_t row_u4, const uint32_t* __restrict__ idx, int nb, uint4* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nb * row_u4) return;
    const size_t j = t / row_u4, c = t % row_u4;
    out[t] = in[(size_t)idx[j] * row_u4 + c];
}
__global__ void scatter_topk_kernel(const uint32_t* __restrict__ idx, const uint8_t* __restrict__ take, int nb, int k, const int64_t* __restrict__ src_s,
                                    const uint32_t* __restrict__ src_i, int64_t* __restrict__ dst_s, uint32_t* __restrict__ dst_i, size_t dst_stride) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * k) return;
    const int j = t / k, i = t % k;
    if (take && !take[j]) return;
    dst_s[(size_t)idx[j] * dst_stride + i] = src_s[t];
    dst_i[(size_t)idx[j] * dst_stride + i] = src_i[t];
}

// dst row idx[j] = src row j (rows of k elements of 4 bytes) for the j with take[j] != 0 (take == nullptr: all)
__global__ void scatter_rows4_kernel(const uint32_t* __restrict__ idx, const uint8_t* __restrict__ take, int nb, int k, const uint32_t* __restrict__ src,
                                     uint32_t* __restrict__ dst, size_t dst_stride) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb * k) return;
    const int j = t / k, i = t % k;
    if (take && !take[j]) return;
    dst[(size_t)idx[j] * dst_stride + i] = src[t];
}
int launch_scatter_rows4(const uint32_t* idx, const uint8_t* take, int nb, int k, const void* src, void* dst, size_t dst_stride, hipStream_t stream) {
    if (nb == 0 || k == 0) return 0;
    hipLaunchKernelGGL(scatter_rows4_kernel, dim3((unsigned)((nb * k + 255) / 256)), dim3(256), 0, stream, idx, take, nb, k,
                       reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(dst), dst_stride);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_margin_f32(const uint32_t* sel_ids, const float* sel_keys, size_t sel_stride, int k, int nq, const float* group_keys,
                      size_t gk_stride, int kg, size_t n_groups, const float* eps, float* margin, hipStream_t stream) {
    if (nq == 0 || k == 0) return 0;
    hipLaunchKernelGGL(margin_f32_kernel, dim3((nq + 63) / 64), dim3(64), 0, stream, sel_ids, sel_keys, sel_stride, k, group_keys,
                       gk_stride, kg, n_groups, eps, margin, nq);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gather_columns(const float* in, int nq_pad, size_t n_groups, const uint32_t* idx, int nb, float* out, int nbp, hipStream_t stream) {
    const size_t total = n_groups * (size_t)nbp;
    if (total == 0) return 0;
    hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, nq_pad, n_groups, idx, nb, out, nbp);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_gather_rows16(const void* in, size_t row_bytes, const uint32_t* idx, int nb, void* out, hipStream_t stream) {
    const size_t total = (size_t)nb * (row_bytes / 16);
    if (total == 0) return 0;
    hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<const uint4*>(in),
                       row_bytes / 16, idx, nb, reinterpret_cast<uint4*>(out));
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_scatter_topk(const uint32_t* idx, const uint8_t* take, int nb, int k, const int64_t* src_s, const uint32_t* src_i, int64_t* dst_s,
                        uint32_t* dst_i, size_t dst_stride, hipStream_t stream) {
    if (nb == 0 || k == 0) return 0;
    hipLaunchKernelGGL(scatter_topk_kernel, dim3((unsigned)((nb * k + 255) / 256)), dim3(256), 0, stream, idx, take, nb, k, src_s, src_i, dst_s,
                       dst_i, dst_stride);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_finalize(const uint32_t* sel_ids, const int64_t* sel_scores, size_t sel_stride, int k, int nq,
                    uint64_t id_offset, int64_t* out_scores, uint32_t* out_ids, size_t out_stride,
                    const float* group_keys, size_t gk_stride, int kg, size_t n_groups, const float* eps,
                    float* margin, hipStream_t stream) {
    if (nq == 0 || k == 0) return 0;
    hipLaunchKernelGGL(finalize_kernel, dim3(nq), dim3(256), 0, stream, sel_ids, sel_scores, sel_stride, k, nq,
                       id_offset, out_scores, out_ids, out_stride, group_keys, gk_stride, kg, n_groups, eps, margin);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}


// hand-over of a shard's results as a packed block: out_sc[i] = sc[i], out_ids[i] = ids[i] + id_offset; an empty slot (id ID_NONE, or
// sc == nullptr: nothing to hand over) becomes (INT64_MIN, ID_NONE)
__global__ void block_finish_kernel(const int64_t* __restrict__ sc, const uint32_t* __restrict__ ids, size_t n, unsigned long long id_offset,
                                    int64_t* __restrict__ out_sc, uint32_t* __restrict__ out_ids) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = sc ? ids[i] : ID_NONE;
    out_sc[i] = id == ID_NONE ? INT64_MIN : sc[i];
    out_ids[i] = id == ID_NONE ? ID_NONE : (uint32_t)(id + id_offset);
}
int launch_block_finish(const int64_t* sc, const uint32_t* ids, size_t n, uint64_t id_offset, int64_t* out_sc, uint32_t* out_ids, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(block_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sc, ids, n, (unsigned long long)id_offset, out_sc, out_ids);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace mse
