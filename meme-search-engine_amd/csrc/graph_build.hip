// Vamana graph build on the device (diskann/src/lib.rs:183-389, driven by src/generate_index_shard.rs:85-133) and the
// batched GPU-resident form of the in-RAM greedy search (lib.rs:183-211).
//
// Everything the reference keeps in `Scratch` (lib.rs:157-175) lives on the device, one workgroup per point:
//   NeighbourBuffer (lib.rs:74-155)          three sorted arrays in LDS (capacity <= 1024)
//   visited (HashSet<u32>)                   one bit per node in HBM, per workgroup slot
//   neighbour_pre_buffer                     LDS (<= 64 ids)
//   visited_list = robust_prune candidates   HBM while it grows (thousands of entries), then its best maxc entries, found by
//                                            a value cut and sorted once in LDS (bitonic, 2048-entry window)
//   robust_prune_scratch_buffer              LDS list of the candidates still alive (exact p_star walk), or per-block masks
//                                            (candidate-major walk on the matrix cores)
// Every score that leaves a kernel or orders a list is fast_dot_noprefetch's value (exact_dot.h), every comparison is on
// the i64 fixed-point scores, and the sequential parts (list inserts, the prune walk, the back edges of one list) give the
// reference's result, so the built graph is bit-identical to the oracle's for the same insertion order and batch size
// (include/mse.h says what a batch is; batch = 1 is the reference's single-threaded loop).  Where a product between two
// candidates is only needed to decide `(alpha * s) >> 16 >= score` (both prunes), it may come from MFMA tiles: the decision
// is taken only if it holds across the MFMA error bound, otherwise the exact dot decides (wg_robust_prune_mfma,
// backedge_gram_kernel; the all-exact routes wg_robust_prune / backedge_kernel remain for other widths and for testing).
//
// Kernels per batch: graph_search_kernel (one 128-lane workgroup per point: search + merge_existing_neighbours),
// prune_kernel (one workgroup per point), apply_lists_kernel, then the back edges grouped per touched list on the host
// (counting sort) and applied by backedge_gram_kernel (one wave per list).  Exact scoring is one quad of lanes per row,
// the row fetched six 16-byte pieces per lane ahead: a row costs 2304 B of HBM/L2 traffic.
#include "../../include/mse.h"
#include "exact_dot.h"
#include "visited_set.h"
#include "runtime.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace mse;

namespace {

constexpr int GB_THREADS = 256;
constexpr int GS_THREADS = 128;   // search kernel: two waves per query, so that eight queries share a CU at its register budget
constexpr int GB_LMAX = 1024;   // search list
constexpr int GB_RMAX = 64;     // degree bound
constexpr int GB_CMAX = 1024;   // maxc
constexpr int GB_WIN = 2048;    // sort window
constexpr long long GB_MIN = (long long)0x8000000000000000ull;   // i64::MIN marks a discarded candidate (lib.rs:269)

__device__ __forceinline__ bool cand_before(long long sa, uint32_t pa, long long sb, uint32_t pb) {
    return sa > sb || (sa == sb && pa < pb);
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ long long shfl_i64(long long v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src), hi = (uint32_t)__shfl((int)(uint32_t)((unsigned long long)v >> 32), src);
    return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long readlane_i64(long long v, int src) {   // src uniform
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), src);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Bitonic sort of N (power of two, <= GB_WIN) candidates, best first: score descending, earlier position first among
// equal scores (sort_unstable_by_key(-score), lib.rs:233, restated as a stable sort -- see the oracle's note).
__device__ void wg_sort(long long* sc, uint32_t* id, uint32_t* pos, int N) {
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < N / 2; t += GB_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
                const bool up = (i & k) == 0;
                const long long si = sc[i], sl = sc[l];
                const uint32_t pi = pos[i], pl = pos[l];
                if ((si != sl || pi != pl) && cand_before(sl, pl, si, pi) == up) {
                    sc[i] = sl; sc[l] = si;
                    pos[i] = pl; pos[l] = pi;
                    const uint32_t ii = id[i];
                    id[i] = id[l]; id[l] = ii;
                }
            }
            __syncthreads();
        }
}

// The best min(total, want) (want <= GB_CMAX) of the candidate list (vi, vs)[0..total) in HBM, sorted into c_*[0..).  Returns
// how many are valid (at least min(total, want)).
// A list that fits the 2048-entry window is sorted whole.  A longer one (a search at L = 192 leaves about 7000) is first
// cut down by value: 4096 equal buckets between the smallest and the largest score, a histogram, and the bucket that holds
// the want-th best score -- every entry in that bucket or above is gathered (position kept) and the survivors are sorted
// once.  The cut only removes entries that cannot be among the best `want`, so the result equals a full stable sort.
// If ties crowd more than a window's worth of entries into the cut, the list is fed through the window 1024 at a time.
__device__ int wg_best_candidates(const uint32_t* vi, const long long* vs, int total, long long* c_sc, uint32_t* c_id, uint32_t* c_pos,
                                  int want = GB_CMAX) {
    const int tid = threadIdx.x;
    __shared__ long long s_mn, s_mx;
    __shared__ int s_thr, s_count;
    __shared__ uint32_t s_part[GB_THREADS];
    if (total > GB_WIN) {
        long long mn = 0x7fffffffffffffffll, mx = GB_MIN;
        for (int e = tid; e < total; e += GB_THREADS) {
            const long long v = vs[e];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const long long a = shfl_i64(mn, (threadIdx.x & 63) ^ o), b = shfl_i64(mx, (threadIdx.x & 63) ^ o);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if (tid == 0) { s_mn = 0x7fffffffffffffffll; s_mx = GB_MIN; s_count = 0; }
        uint32_t* hist = reinterpret_cast<uint32_t*>(c_sc);   // 4096 bins = the window's score array
        for (int e = tid; e < 4096; e += GB_THREADS) hist[e] = 0u;
        __syncthreads();
        if ((tid & 63) == 0) { atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
        __syncthreads();
        mn = s_mn;
        const unsigned long long range = (unsigned long long)s_mx - (unsigned long long)mn;
        int sh = 0;
        while ((range >> sh) >= (1ull << 40)) sh++;
        const unsigned long long den = (range >> sh) + 1ull;
        auto bucket = [&](long long v) -> int { return (int)(((((unsigned long long)v - (unsigned long long)mn) >> sh) * 4096ull) / den); };
        for (int e = tid; e < total; e += GB_THREADS) atomicAdd(&hist[bucket(vs[e])], 1u);
        __syncthreads();
        // bucket of the want-th best: the largest b with (entries in buckets >= b) >= want
        uint32_t mine = 0;
        for (int x = 0; x < 16; x++) mine += hist[tid * 16 + x];
        s_part[tid] = mine;
        __syncthreads();
        if (tid == 0) {
            uint32_t above = 0;
            int t = GB_THREADS - 1;
            while (t > 0 && above + s_part[t] < (uint32_t)want) { above += s_part[t]; t--; }
            int b = t * 16 + 15;
            while (b > t * 16 && above + hist[b] < (uint32_t)want) { above += hist[b]; b--; }
            s_thr = b;
            uint32_t cnt = above;           // entries in buckets > b so far; add bucket b itself
            s_part[0] = cnt + hist[b];
        }
        __syncthreads();
        const int thr = s_thr;
        const int kept = (int)s_part[0];
        __syncthreads();
        if (kept <= GB_WIN) {
            for (int e = tid; e < total; e += GB_THREADS) {
                const long long v = vs[e];
                if (bucket(v) >= thr) {
                    const int slot = atomicAdd(&s_count, 1);
                    c_sc[slot] = v; c_id[slot] = vi[e]; c_pos[slot] = (uint32_t)e;
                }
            }
            __syncthreads();
            int N = 2;
            while (N < kept) N <<= 1;
            for (int e = kept + tid; e < N; e += GB_THREADS) { c_sc[e] = GB_MIN; c_id[e] = 0xffffffffu; c_pos[e] = 0xffffffffu; }
            __syncthreads();
            wg_sort(c_sc, c_id, c_pos, N);
            return kept < GB_CMAX ? kept : GB_CMAX;
        }
    }
    const int first = total < GB_WIN ? total : GB_WIN;
    int N = 2;
    while (N < first) N <<= 1;
    for (int e = tid; e < N; e += GB_THREADS) {
        const bool in = e < first;
        c_sc[e] = in ? vs[e] : GB_MIN;
        c_id[e] = in ? vi[e] : 0xffffffffu;
        c_pos[e] = in ? (uint32_t)e : 0xffffffffu;
    }
    __syncthreads();
    wg_sort(c_sc, c_id, c_pos, N);
    for (int off = GB_WIN; off < total; off += GB_WIN / 2) {
        for (int e = tid; e < GB_WIN / 2; e += GB_THREADS) {
            const int src = off + e;
            const bool in = src < total;
            c_sc[GB_WIN / 2 + e] = in ? vs[src] : GB_MIN;
            c_id[GB_WIN / 2 + e] = in ? vi[src] : 0xffffffffu;
            c_pos[GB_WIN / 2 + e] = in ? (uint32_t)src : 0xffffffffu;
        }
        __syncthreads();
        wg_sort(c_sc, c_id, c_pos, GB_WIN);
    }
    return total < GB_CMAX ? total : GB_CMAX;
}

struct PruneParams {
    const uint16_t* base; int d;
    uint32_t qb; long long alpha, qalpha; int r, saturate;
    uint32_t n; uint32_t* err;
    long long eps_fix;   // > 0: candidate products may be taken from the matrix cores (see wg_robust_prune_mfma); 0: exact dots only
};

// robust_prune (lib.rs:227-285) after its sort/truncate: candidates c_*[0..nc) best first.  Called by the whole
// workgroup; returns the new list's length, ids in s_neigh.  s_live: room for nc u16; s_cnt: one LDS int.
__device__ int wg_robust_prune(const PruneParams& pp, uint32_t p, int nc, long long* c_sc, const uint32_t* c_id, uint16_t* s_star,
                               uint16_t* s_live, uint32_t* s_neigh, int* s_cnt) {
    const int tid = threadIdx.x, d = pp.d;
    int nn = 0, ci = 0;
    __syncthreads();
    while (nn < pp.r && ci < nc) {
        const uint32_t p_star = c_id[ci];
        const long long p_star_score = c_sc[ci];
        ci++;
        if (p_star == p || p_star_score == GB_MIN) continue;   // :241 (uniform: every lane reads the same LDS words)
        if (tid == 0) { s_neigh[nn] = p_star; *s_cnt = 0; }
        nn++;
        __syncthreads();
        // :250-255 -- note the range starts one past the candidate behind p_star (candidate_index was already advanced)
        for (int i = ci + 1 + tid; i < nc; i += GB_THREADS)
            if (c_sc[i] != GB_MIN) s_live[atomicAdd(s_cnt, 1)] = (uint16_t)i;
        if (p_star >= pp.n) { if (tid == 0) atomicOr(pp.err, 16u); break; }
        for (int e = tid; e < d / 8; e += GB_THREADS)
            reinterpret_cast<uint4*>(s_star)[e] = reinterpret_cast<const uint4*>(pp.base + (size_t)p_star * d)[e];
        __syncthreads();
        const int nlive = *s_cnt;
        for (int e0 = 0; e0 < nlive; e0 += GB_THREADS / 4) {   // :257-271, one lane quad per surviving candidate
            const int e = e0 + (tid >> 2);
            const int i = s_live[e < nlive ? e : nlive - 1];
            uint32_t p_prime = c_id[i];
            if (p_prime >= pp.n) { atomicOr(pp.err, 32u); p_prime = 0; }
            const float f = quad_fast_dot_f32(pp.base + (size_t)p_prime * d, s_star, d);
            if (e < nlive && (tid & 3) == 0) {
                const long long a = p_prime >= pp.qb ? pp.qalpha : pp.alpha;
                const long long scaled = (long long)((unsigned long long)a * (unsigned long long)scale_dot_result(f)) >> 16;
                if (scaled >= c_sc[i]) c_sc[i] = GB_MIN;
            }
        }
        __syncthreads();
    }
    if (pp.saturate || p >= pp.qb) {   // :275-284
        if (tid < 64) {
            uint32_t mine = tid < nn ? s_neigh[tid] : 0u;
            for (int i = 0; i < nc && nn < pp.r; i++) {
                const uint32_t id = c_id[i];
                if (__ballot(tid < nn && mine == id)) continue;
                if (tid == nn) mine = id;
                nn++;
            }
            if (tid < nn) s_neigh[tid] = mine;
            if (tid == 0) *s_cnt = nn;
        }
        __syncthreads();
        nn = *s_cnt;
    }
    __syncthreads();
    return nn;
}

struct GraphArgs {
    const uint16_t* base; uint32_t n; int d;
    const uint32_t* adj; const uint32_t* deg; int r;
    const uint32_t* points;      // build: the batch's points; search: the start node per query
    const uint16_t* queries;     // search only
    uint32_t medioid, qb; int base_only;
    int L, maxc, saturate; long long alpha, qalpha;
    uint32_t* bitmap; size_t bm_words; int hash_bits;   // visited set per workgroup slot (visited_set.h): bm_words u32 each
    uint32_t* vl_ids; long long* vl_sc; uint32_t vl_cap;
    uint32_t* out_ids; long long* out_sc; uint32_t* out_len; uint32_t* out_dist;
    uint32_t* err;   // bit 0: an edge points outside the graph; bit 1: visited list overflow
};

// LDS of the search kernel: the query row, the search list (sized by L, so the usual L = 192 leaves room for eight
// workgroups per CU) and the pre-buffer; of the prune kernel: the p_star row, the sort window, the prune state.
inline size_t search_lds_bytes(int d, int L) { return (size_t)((d * 2 + 15) & ~15) + (size_t)L * 16 + 64 * 16; }
inline size_t prune_lds_bytes(int d) { return (size_t)((d * 2 + 15) & ~15) + GB_WIN * 16 + 64 * 4 + GB_CMAX * 2; }

// greedy_search (lib.rs:183-211), one workgroup per query.  BUILD: the query is point p's own vector, the start is the
// medioid, the visited list is kept in HBM and merge_existing_neighbours (:215-221) is appended to it -- the candidate
// list robust_prune starts from; out_dist = its length.  !BUILD: an outside query; the buffer is the output.
template <bool BUILD>
// (Round 4 probe: a shallower row prefetch -- three pieces per lane in flight instead of six -- brings the kernel from 126 to 78-92
// registers and 10-12 workgroups per CU instead of 8; the build rate at 1e7 rows stayed within +-3 % at batch 4096 / 6144
// (scripts/graph_occupancy_probe.sh): the build is not limited by the searches in flight.  Kept as it was.)
__global__ __launch_bounds__(GS_THREADS) void graph_search_kernel(GraphArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dq = (a.d * 2 + 15) & ~15;
    uint16_t* s_q = reinterpret_cast<uint16_t*>(smem);
    char* p0 = smem + dq;
    long long* nb_sc = reinterpret_cast<long long*>(p0); p0 += (size_t)a.L * 8;
    long long* pre_sc = reinterpret_cast<long long*>(p0); p0 += 64 * 8;
    uint32_t* nb_id = reinterpret_cast<uint32_t*>(p0); p0 += (size_t)a.L * 4;
    uint32_t* nb_vis = reinterpret_cast<uint32_t*>(p0); p0 += (size_t)a.L * 4;
    uint32_t* pre_id = reinterpret_cast<uint32_t*>(p0); p0 += 64 * 4;
    int* s_rank = reinterpret_cast<int*>(p0);
    __shared__ int s_len, s_next, s_npre, s_cnt, s_pt, s_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t bi = blockIdx.x;
    const int cap = a.L, d = a.d;
    const uint32_t p = BUILD ? a.points[bi] : 0u;
    const uint32_t start = BUILD ? a.medioid : a.points[bi];
    const bool base_only = BUILD ? p >= a.qb : a.base_only != 0;
    uint32_t* bm = a.bitmap + bi * a.bm_words;
    uint32_t* vl_i = a.vl_ids + bi * (size_t)a.vl_cap;
    long long* vl_s = a.vl_sc + bi * (size_t)a.vl_cap;

    {
        const uint16_t* qsrc = BUILD ? a.base + (size_t)p * d : a.queries + bi * d;
        for (int e = tid; e < d / 8; e += GS_THREADS) reinterpret_cast<uint4*>(s_q)[e] = reinterpret_cast<const uint4*>(qsrc)[e];
    }
    __syncthreads();
    // NeighbourBuffer::next_unvisited (lib.rs:93-107) by wave 0, 64 visited flags per step (round 6: one lane walking the flags paid an
    // LDS round trip per entry); s_pt = the node to expand or -1
    auto pop = [&]() {
        const int cur = s_next;
        int pt = -1, nxt = -1;
        if (cur >= 0) {
            const int len = s_len;
            pt = (int)nb_id[cur];
            for (int g0 = cur + 1; g0 < len; g0 += 64) {
                const int idx = g0 + lane;
                const unsigned long long m = __ballot(idx < len && !nb_vis[idx]);
                if (m) { nxt = g0 + __ffsll((long long)m) - 1; break; }
            }
            if (lane == 0) nb_vis[cur] = 1;
        }
        if (lane == 0) { s_next = nxt; s_pt = pt; }
    };
    if (wave == 0) {   // :188-189
        const float f = quad_fast_dot_f32(a.base + (size_t)start * d, s_q, d);
        if (lane == 0) {
            nb_id[0] = start; nb_sc[0] = scale_dot_result(f); nb_vis[0] = 0;
            s_len = 1; s_next = 0; s_abort = 0;
            (void)visited_insert(bm, a.hash_bits, start);
        }
        pop();
    }

    // One step = expand one node.  Wave 0 owns the list: it takes the node, reads its neighbours and filters them
    // through the visited bits; after the first barrier all four waves score the survivors; after the second, wave 0
    // inserts them and takes the next node while the other waves already wait at the next step's first barrier.
    uint32_t n_vl = 0;   // wave 0: visited_list.len() == counters.distances
    for (;;) {
        if (wave == 0) {   // :194-200
            const int pti = s_abort ? -1 : s_pt;
            if (pti < 0) {
                if (lane == 0) s_npre = -1;
            } else {
                const uint32_t pt = (uint32_t)pti;
                const uint32_t raw = lane < a.r ? a.adj[(size_t)pt * a.r + lane] : 0xffffffffu;   // row and length in one round trip
                int dg = (int)a.deg[pt];
                if (dg > a.r) dg = a.r;
                const uint32_t nb = lane < dg ? raw : 0xffffffffu;
                bool cand = lane < dg;
                if (cand && nb >= a.n) { cand = false; atomicOr(a.err, 1u); }
                for (int l = 0; l < dg; l++) {   // an id listed twice: HashSet::insert accepts the first occurrence only
                    const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)nb, l);
                    if (l < lane && o == nb) cand = false;
                }
                // a query node met while base_vectors_only is dropped whether or not it was seen before, so its bit is not needed
                if (cand && base_only && nb >= a.qb) cand = false;
                bool fresh = false;
                if (cand) fresh = visited_insert(bm, a.hash_bits, nb);
                const unsigned long long m = __ballot(fresh);
                        if (fresh) pre_id[__popcll(m & ((1ull << lane) - 1ull))] = nb;
                if (lane == 0) s_npre = __popcll(m);
            }
        }
        __syncthreads();
        const int npre = s_npre;
        if (npre < 0) break;
        for (int e0 = 0; e0 < npre; e0 += GS_THREADS / 4) {   // :201-204, one lane quad per neighbour
            const int e = e0 + (tid >> 2);
            const uint32_t id = pre_id[e < npre ? e : npre - 1];
            const float f = quad_fast_dot_f32(a.base + (size_t)id * d, s_q, d);
            if (e < npre && (tid & 3) == 0) pre_sc[e] = scale_dot_result(f);
        }
        __syncthreads();

        if (wave == 0 && npre > 0) {
            if (BUILD) {   // :206
                if (n_vl + (uint32_t)npre <= a.vl_cap) {
                    if (lane < npre) { vl_i[n_vl + lane] = pre_id[lane]; vl_s[n_vl + lane] = pre_sc[lane]; }
                } else if (lane == 0) {
                    atomicOr(a.err, 2u);
                }
            }
            n_vl += (uint32_t)npre;
            if (a.hash_bits && n_vl >= (1u << (a.hash_bits - 1)) && lane == 0) {   // table half full: the host repeats with bit maps
                atomicOr(a.err, 4u);
                s_abort = 1;
            }
            int len = s_len, nu = s_next;
            // All newcomers at once.  While no two scores involved are equal, the order of the inserts does not matter:
            // the list ends up as the best `cap` of old and new entries, and next_unvisited as the smaller of its old value
            // and the slot the best newcomer took on arrival (every other insert lands at or behind that slot).  So each lane
            // places one newcomer by two counts -- old entries above it (binary search) and newcomers above it -- and the old
            // entries move up by the number of newcomers that go before them.  A tie anywhere, and the reference's loop is replayed.
            const bool valid = lane < npre;
            const long long my_sc = valid ? pre_sc[lane] : GB_MIN;
            const uint32_t my_id = valid ? pre_id[lane] : 0u;
            int lo = 0, hi = valid ? len : 0;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (nb_sc[mid] > my_sc) lo = mid + 1; else hi = mid;
            }
            bool tie = valid && lo < len && nb_sc[lo] == my_sc;
            int r_new = 0;
            for (int k = 0; k < npre; k++) {
                const long long sk = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(my_sc >> 32), k) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readlane((int)my_sc, k));
                r_new += sk > my_sc;
                tie |= valid && k != lane && sk == my_sc;
            }
            if (!__ballot(tie)) {
                if (valid) s_rank[r_new] = lo;   // old entries above the newcomer of rank r_new; ascending in the rank
                for (int c = (len - 1) / 64; c >= 0; c--) {   // highest entries first: nothing unread is overwritten
                    const int i = c * 64 + lane;
                    const bool act = i < len;
                    long long osc = 0;
                    uint32_t oid = 0, ovis = 0;
                    if (act) { osc = nb_sc[i]; oid = nb_id[i]; ovis = nb_vis[i]; }
                    int a0 = 0, b0 = npre;
                    while (a0 < b0) {
                        const int mid = (a0 + b0) >> 1;
                        if (s_rank[mid] <= i) a0 = mid + 1; else b0 = mid;
                    }
                    const int np = i + a0;
                    if (act && a0 > 0 && np < cap) { nb_sc[np] = osc; nb_id[np] = oid; nb_vis[np] = ovis; }
                }
                const int pos = lo + r_new;
                if (valid && pos < cap) { nb_sc[pos] = my_sc; nb_id[pos] = my_id; nb_vis[pos] = 0; }
                const int first = s_rank[0];
                if (first < cap && (nu < 0 || first < nu)) nu = first;
                len = len + npre < cap ? len + npre : cap;
            } else {
                for (int ii = 0; ii < npre; ii++) {   // NeighbourBuffer::insert (lib.rs:117-147), in order
                    const uint32_t id = pre_id[ii];
                    const long long sc = pre_sc[ii];
                    if (len == cap && nb_sc[len - 1] > sc) continue;
                    // position by counting with 64 lanes; only a run of equal scores makes binary_search_by's probe sequence matter
                    int n_gt = 0, n_eq = 0, eq_pos = -1;
                    for (int b0 = 0; b0 < len; b0 += 64) {
                        const int idx = b0 + lane;
                        const long long v = idx < len ? nb_sc[idx] : 0;
                        n_gt += __popcll(__ballot(idx < len && v > sc));
                        const unsigned long long me = __ballot(idx < len && v == sc);
                        if (me) {
                            if (eq_pos < 0) eq_pos = b0 + __ffsll((long long)me) - 1;
                            n_eq += __popcll(me);
                        }
                    }
                    int loc = 0;
                    if (n_eq == 0) {
                        loc = n_gt;
                    } else if (n_eq == 1) {
                        loc = eq_pos;
                    } else {
                        int size = len, bs = 0;
                        while (size > 1) {
                            const int half = size / 2, mid = bs + half;
                            bs = (sc > nb_sc[mid]) ? bs : mid;
                            size -= half;
                        }
                        const long long c = nb_sc[bs];
                        loc = (sc == c) ? bs : bs + (sc < c ? 1 : 0);
                    }
                    if (loc < len && nb_id[loc] == id) continue;
                    const int newlen = len < cap ? len + 1 : cap;
                    for (int top = newlen - 1; top > loc; top -= 64) {
                        const int idx = top - lane;
                        const bool act = idx > loc;
                        uint32_t mi = 0, mv = 0;
                        long long ms = 0;
                        if (act) { mi = nb_id[idx - 1]; ms = nb_sc[idx - 1]; mv = nb_vis[idx - 1]; }
                        if (act) { nb_id[idx] = mi; nb_sc[idx] = ms; nb_vis[idx] = mv; }
                    }
                    if (lane == 0) { nb_id[loc] = id; nb_sc[loc] = sc; nb_vis[loc] = 0; }
                    len = newlen;
                    if (nu < 0 || loc < nu) nu = loc;
                }
            }
            if (lane == 0) { s_len = len; s_next = nu; }
        }
        if (wave == 0) pop();
    }
    __syncthreads();

    if (!BUILD) {
        const int len = s_len;
        for (int e = tid; e < len; e += GS_THREADS) {
            a.out_ids[bi * a.L + e] = nb_id[e];
            a.out_sc[bi * a.L + e] = nb_sc[e];
        }
        if (tid == 0) { a.out_len[bi] = (uint32_t)len; a.out_dist[bi] = n_vl; }
        return;
    }

    // ---- merge_existing_neighbours (lib.rs:215-221): the point's current list, scored against the point ----
    if (tid == 0) s_cnt = (int)n_vl;
    __syncthreads();
    n_vl = (uint32_t)s_cnt;
    int dg = (int)a.deg[p];
    if (dg > a.r) dg = a.r;
    for (int e0 = 0; e0 < dg; e0 += GS_THREADS / 4) {
        const int e = e0 + (tid >> 2);
        uint32_t id = a.adj[(size_t)p * a.r + (e < dg ? e : dg - 1)];
        if (id >= a.n) { id = 0; atomicOr(a.err, 1u); }
        const float f = quad_fast_dot_f32(a.base + (size_t)id * d, s_q, d);
        if (e < dg && (tid & 3) == 0) {
            if (n_vl + (uint32_t)e < a.vl_cap) { vl_i[n_vl + e] = id; vl_s[n_vl + e] = scale_dot_result(f); }
            else atomicOr(a.err, 2u);
        }
    }
    if (tid == 0) {
        uint32_t total = n_vl + (uint32_t)dg;
        if (total > a.vl_cap) total = a.vl_cap;   // err bit 1 is set; the host repeats the batch with more room
        a.out_dist[bi] = total;
    }
}

// robust_prune's loop, candidate-major, with the products from the matrix cores.
// The reference walks p_star by p_star and discards later candidates (lib.rs:236-272).  Whether candidate i is still alive
// when the walk reaches it depends only on the selected candidates before it: it is dead iff some selected s at index
// <= i - 2 (the candidate right behind a p_star is never tested against it, :240,250) has (alpha * dot(s, i)) >> 16 >= score_i.
// So the walk can go candidate by candidate, sixteen at a time: the four waves take the products of the sixteen candidates
// with the (at most 64) rows selected so far, one 16 x 16 MFMA tile per wave, plus the tile of the sixteen among themselves;
// wave 0 then settles the sixteen in order.  A product decides only if the comparison holds for every value within the MFMA
// error bound (eps_fix, the scan's certificate); inside the bound the exact quad dot decides.  Candidate rows are read once
// instead of once per p_star they survive.  Same result as wg_robust_prune, bit for bit.
// P: [64][16] floats, Q: [16][16] floats, s_selidx: [64] ints (LDS).  d / 32 must be a multiple of 6.
__device__ int wg_robust_prune_mfma(const PruneParams& pp, uint32_t p, int nc, const long long* c_sc, const uint32_t* c_id, float* P, float* Q,
                                    int* s_selidx, uint32_t* s_neigh, int* s_cnt) {
    constexpr int PD = 6;    // k-steps of the candidate rows in flight
    constexpr int TK = 36;   // k-steps per row: this route is built for d = 1152 (the caller checks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g4 = lane >> 4;
    const int d = pp.d, r = pp.r;
    // The selected rows are the A operand of every tile, so they stay in registers: wave w keeps rows 16 w .. 16 w + 15 of the
    // selected set as 36 fragments (144 VGPRs) and fetches a row once, when it is selected.  Only the sixteen candidate rows
    // of a block stream through (B operand): a candidate row is read once per prune instead of once per surviving p_star.
    half8 afrag[TK];
#pragma unroll
    for (int ks = 0; ks < TK; ks++) afrag[ks] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    int have = 0;
    __shared__ unsigned long long s_border[16];   // per candidate of the block: selected rows whose comparison needs the exact dot
    __shared__ int s_kill[16];                    // per candidate: some selected row discards it for certain
    if (tid == 0) *s_cnt = 0;
    uint32_t sink = 0;
    // candidate rows are first touched two blocks before their tiles are computed, so that the fragments come from L2
    auto touch = [&](int b0) {
        const int lines = (d * 2 + 127) / 128;
        for (int e = tid; e < 16 * lines; e += GB_THREADS) {
            const int c = b0 + e / lines;
            if (c < nc) sink ^= *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(pp.base + (size_t)c_id[c] * d) + (size_t)(e % lines) * 128);
        }
    };
    __syncthreads();
    touch(0);
    touch(16);
    for (int b0 = 0; b0 < nc; b0 += 16) {
        const int nn0 = *s_cnt;
        if (nn0 >= r) break;
        touch(b0 + 32);
        if (tid < 16) { s_border[tid] = 0ull; s_kill[tid] = 0; }
        const int ntile = (nn0 + 15) / 16, intra_wave = ntile < 4 ? ntile : 0;
        int avail = nn0 - 16 * wave;
        avail = avail < 0 ? 0 : (avail > 16 ? 16 : avail);
        if (avail > have) {   // rows selected since this wave last looked
            if (i16 >= have && i16 < avail) {
                const uint4* ra = reinterpret_cast<const uint4*>(pp.base + (size_t)s_neigh[16 * wave + i16] * d) + g4;
#pragma unroll
                for (int ks = 0; ks < TK; ks++) afrag[ks] = __builtin_bit_cast(half8, ra[ks * 4]);
            }
            have = avail;
        }
        const bool do_sel = avail > 0, do_intra = wave == intra_wave;
        if (do_sel || do_intra) {
            const int cb = b0 + i16 < nc ? b0 + i16 : nc - 1;
            const uint4* rb = reinterpret_cast<const uint4*>(pp.base + (size_t)c_id[cb] * d) + g4;
            float4v acc_s = {0.0f, 0.0f, 0.0f, 0.0f}, acc_i = {0.0f, 0.0f, 0.0f, 0.0f};
            uint4 fb[PD];
#pragma unroll
            for (int u = 0; u < PD; u++) fb[u] = rb[u * 4];
#pragma unroll
            for (int ks = 0; ks < TK; ks++) {
                const half8 B = __builtin_bit_cast(half8, fb[ks % PD]);
                fb[ks % PD] = rb[(ks + PD < TK ? ks + PD : TK - 1) * 4];   // clamped: always inside the row
                if (do_sel) acc_s = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[ks], B, acc_s, 0, 0, 0);
                if (do_intra) acc_i = __builtin_amdgcn_mfma_f32_16x16x32_f16(B, B, acc_i, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; v++) {
                if (do_sel) P[(16 * wave + 4 * g4 + v) * 16 + i16] = acc_s[v];
                if (do_intra) Q[(4 * g4 + v) * 16 + i16] = acc_i[v];
            }
        }
        __syncthreads();
        // every (selected row, candidate of the block) pair at once: thread = (candidate j, four selected rows)
        {
            const int j = tid >> 4, i = b0 + j;
            if (i < nc) {
                const long long sc = c_sc[i];
                const uint32_t id = c_id[i];
                const long long al = id >= pp.qb ? pp.qalpha : pp.alpha;
                const long long m = ((al * pp.eps_fix) >> 16) + 2;
                unsigned long long bord = 0ull;
                int kill = 0;
                for (int l = tid & 15; l < nn0; l += 16)
                    if (s_selidx[l] <= i - 2) {
                        const long long scaled = (long long)((unsigned long long)al * (unsigned long long)scale_dot_result(P[l * 16 + j])) >> 16;
                        if (scaled - m >= sc) kill = 1;
                        else if (scaled + m >= sc) bord |= 1ull << l;
                    }
                if (kill) atomicOr(&s_kill[j], 1);
                if (bord) atomicOr(&s_border[j], bord);
            }
        }
        __syncthreads();
        if (wave == 0) {   // the sixteen in order; only rows selected inside this block still have to be looked at one by one
            int nn = nn0;
            for (int j = 0; j < 16 && b0 + j < nc && nn < r; j++) {
                const int i = b0 + j;
                const long long sc = c_sc[i];
                const uint32_t id = c_id[i];
                if (sc == GB_MIN) continue;
                const long long al = id >= pp.qb ? pp.qalpha : pp.alpha;
                const long long m = ((al * pp.eps_fix) >> 16) + 2;
                bool kill = s_kill[j] != 0, border = false;
                if (!kill && lane >= nn0 && lane < nn) {
                    const int si = s_selidx[lane];
                    if (si <= i - 2) {
                        const long long scaled = (long long)((unsigned long long)al * (unsigned long long)scale_dot_result(Q[(si - b0) * 16 + j])) >> 16;
                        if (scaled - m >= sc) kill = true;
                        else if (scaled + m >= sc) border = true;
                    }
                }
                bool dead = __ballot(kill) != 0ull;
                unsigned long long bm = dead ? 0ull : (__ballot(border) | s_border[j]);
                while (bm && !dead) {   // inside the error bound: exact dots, sixteen selected rows per pass
                    const int cnt = __popcll(bm), take = cnt < 16 ? cnt : 16;
                    unsigned long long mm = bm;
                    for (int z = (lane >> 2) < take ? (lane >> 2) : 0; z > 0; z--) mm &= mm - 1;
                    const int sl = __ffsll((long long)mm) - 1;
                    const long long s = scale_dot_result(quad_fast_dot_f32(pp.base + (size_t)id * d, pp.base + (size_t)s_neigh[sl] * d, d));
                    const bool hit = (lane >> 2) < take && ((long long)((unsigned long long)al * (unsigned long long)s) >> 16) >= sc;
                    dead = __ballot(hit) != 0ull;
                    for (int q = 0; q < take; q++) bm &= bm - 1;
                }
                if (dead || id == p) continue;
                if (lane == 0) { s_neigh[nn] = id; s_selidx[nn] = i; }
                nn++;
            }
            if (lane == 0) *s_cnt = nn;
        }
        __syncthreads();
    }
    if (sink == 0x9e3779b9u && pp.n == 0xffffffffu) atomicOr(pp.err, 256u);   // keeps the touches alive; never true
    int nn = *s_cnt;
    if (pp.saturate || p >= pp.qb) {   // lib.rs:275-284
        __syncthreads();
        if (tid < 64) {
            uint32_t mine = tid < nn ? s_neigh[tid] : 0u;
            for (int i = 0; i < nc && nn < r; i++) {
                const uint32_t id = c_id[i];
                if (__ballot(tid < nn && mine == id)) continue;
                if (tid == nn) mine = id;
                nn++;
            }
            if (tid < nn) s_neigh[tid] = mine;
            if (tid == 0) *s_cnt = nn;
        }
        __syncthreads();
        nn = *s_cnt;
    }
    __syncthreads();
    return nn;
}

// robust_prune (lib.rs:227-285), one workgroup per point: candidate list b is (ci, cs)[b * stride ..][0..counts[b]),
// the point is points[b]; the new list goes to staging row b.
template <bool MFMA>
__global__ __launch_bounds__(GB_THREADS, 2) void prune_kernel(PruneParams pp, const uint32_t* ci, const long long* cs, size_t stride,
                                                           const uint32_t* counts, const uint32_t* points, int maxc, uint32_t* out_ids,
                                                           uint32_t* out_len) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dq = (pp.d * 2 + 15) & ~15;
    uint16_t* s_star = reinterpret_cast<uint16_t*>(smem);
    char* p0 = smem + dq;
    long long* c_sc = reinterpret_cast<long long*>(p0); p0 += GB_WIN * 8;
    uint32_t* c_id = reinterpret_cast<uint32_t*>(p0); p0 += GB_WIN * 4;
    uint32_t* c_pos = reinterpret_cast<uint32_t*>(p0); p0 += GB_WIN * 4;
    uint32_t* s_neigh = reinterpret_cast<uint32_t*>(p0); p0 += 64 * 4;
    uint16_t* s_live = reinterpret_cast<uint16_t*>(p0);
    __shared__ int s_cnt;
    const size_t b = blockIdx.x;
    int nc = wg_best_candidates(ci + b * stride, cs + b * stride, (int)counts[b], c_sc, c_id, c_pos, maxc);
    if (nc > maxc) nc = maxc;
    int nn;
    if constexpr (MFMA) {   // the upper half of the sort window is free once the candidates are sorted: tiles and selected indices
        float* P = reinterpret_cast<float*>(c_sc + GB_WIN / 2);
        nn = wg_robust_prune_mfma(pp, points[b], nc, c_sc, c_id, P, P + 64 * 16, reinterpret_cast<int*>(P + 64 * 16 + 16 * 16), s_neigh, &s_cnt);
    } else {
        nn = wg_robust_prune(pp, points[b], nc, c_sc, c_id, s_star, s_live, s_neigh, &s_cnt);
    }
    if ((int)threadIdx.x < nn) out_ids[b * pp.r + threadIdx.x] = s_neigh[threadIdx.x];
    if (threadIdx.x == 0) out_len[b] = (uint32_t)nn;
}

__global__ void apply_lists_kernel(uint32_t* adj, uint32_t* deg, int r, const uint32_t* points, const uint32_t* staged, const uint32_t* staged_len,
                                   int nb) {
    const int k = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), e = threadIdx.x & 63;
    if (k >= nb) return;
    const uint32_t p = points[k], len = staged_len[k];
    if ((uint32_t)e < len) adj[(size_t)p * r + e] = staged[(size_t)k * r + e];
    if (e == 0) deg[p] = len;
}

struct BackArgs {
    const uint16_t* base; int d;
    uint32_t* adj; uint32_t* deg; int r;
    const uint32_t* targets; const uint32_t* src_off; const uint32_t* srcs;
    uint32_t qb; int maxc, saturate; long long alpha, qalpha;
    uint32_t n; uint32_t* err;
    const uint32_t* src_cnt;   // sources of target i: srcs[src_off[i] .. src_off[i] + src_cnt[i]); null: .. src_off[i + 1])
};
__device__ __forceinline__ uint32_t back_end(const BackArgs& a, uint32_t i) { return a.src_cnt ? a.src_off[i] + a.src_cnt[i] : a.src_off[i + 1]; }

// ---- back edges grouped by the list they touch, on the device (round 5) ---------------------------------------------
// The batch's new lists name, entry by entry (k = position in the batch, j = position in the list), the lists that receive a back
// edge.  They are applied per TARGET list, its sources in (k, j) order.  Rounds 1-4 grouped them on the host: staged lists down, a
// counting sort through a table of one slot per graph node, three arrays up -- 40-110 ms per batch of 16 384 points at 1e8 nodes
// (400 MB of random reads and writes), about as long as the batch's kernels, with the device idle meanwhile.  Here: an
// open-addressing table keyed by target (sized for the batch, not the graph), counts by atomics, targets compacted with the
// "four or more newcomers" class in front (scheduling only: the lists are independent), source segments placed by an atomic
// running total, filled in arrival order and then SORTED by entry number -- so what each list receives, and in which order, is
// exactly what the host grouping produced.  Only the number of targets travels to the host (the launch size of the back-edge kernel).
struct GroupArgs {
    const uint32_t* stg; const uint32_t* len; const uint32_t* points; int r; uint32_t nb;
    uint32_t* keys; uint32_t* cnt; uint32_t* tidx; int bits;          // the table: target id, entries, index in targets[]
    uint32_t* ent_slot;                                             // [nb * r] table slot of every entry (0xffffffff: none)
    uint32_t* counters;   // [0] targets with >= 4 entries, [1] the others, [2] / [3] their running indices, [4] running source total
    uint32_t *targets, *t_off, *t_cnt, *t_slot, *t_fill, *srcs, *srcs_pts;   // srcs: entry numbers as they arrived; srcs_pts: source points in order
};
__global__ void group_count_kernel(GroupArgs a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.nb * (uint32_t)a.r) return;
    const uint32_t k = e / (uint32_t)a.r, j = e - k * (uint32_t)a.r;
    uint32_t slot = 0xffffffffu;
    if (j < a.len[k]) {
        const uint32_t t = a.stg[e], mask = (1u << a.bits) - 1u;
        uint32_t h = (t * 2654435761u) >> (32 - a.bits);
        for (;;) {   // the table has at least twice as many slots as the batch has entries
            const uint32_t old = atomicCAS(&a.keys[h], 0xffffffffu, t);
            if (old == 0xffffffffu || old == t) break;
            h = (h + 1) & mask;
        }
        atomicAdd(&a.cnt[h], 1u);
        slot = h;
    }
    a.ent_slot[e] = slot;
}
// (one atomic per WAVE and counter: a million lanes adding to two words took 2.7 ms per batch)
__global__ void group_classify_kernel(GroupArgs a) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    const bool used = h < (1u << a.bits) && a.keys[h] != 0xffffffffu;
    const bool big = used && a.cnt[h] >= 4;
    const unsigned long long mb = __ballot(big), ms = __ballot(used && !big);
    if ((threadIdx.x & 63) == 0) {
        if (mb) atomicAdd(&a.counters[0], (uint32_t)__popcll(mb));
        if (ms) atomicAdd(&a.counters[1], (uint32_t)__popcll(ms));
    }
}
__global__ void group_compact_kernel(GroupArgs a) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool used = h < (1u << a.bits) && a.keys[h] != 0xffffffffu;
    const uint32_t c = used ? a.cnt[h] : 0u;
    const bool big = used && c >= 4, small = used && !big;
    const unsigned long long mb = __ballot(big), ms = __ballot(small), below = (1ull << lane) - 1ull;
    // the wave's total of c and every lane's share before it (inclusive scan by shuffles)
    uint32_t incl = c;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const uint32_t total = __shfl(incl, 63);
    uint32_t base_b = 0, base_s = 0, base_o = 0;
    if (lane == 0) {
        if (mb) base_b = atomicAdd(&a.counters[2], (uint32_t)__popcll(mb));
        if (ms) base_s = atomicAdd(&a.counters[3], (uint32_t)__popcll(ms));
        if (total) base_o = atomicAdd(&a.counters[4], total);
    }
    base_b = __shfl(base_b, 0); base_s = __shfl(base_s, 0); base_o = __shfl(base_o, 0);
    if (!used) return;
    const uint32_t idx = big ? base_b + (uint32_t)__popcll(mb & below) : a.counters[0] + base_s + (uint32_t)__popcll(ms & below);
    a.targets[idx] = a.keys[h];
    a.t_cnt[idx] = c;
    a.t_off[idx] = base_o + incl - c;
    a.t_slot[idx] = h;
    a.tidx[h] = idx;
}
__global__ void group_fill_kernel(GroupArgs a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.nb * (uint32_t)a.r) return;
    const uint32_t h = a.ent_slot[e];
    if (h == 0xffffffffu) return;
    const uint32_t idx = a.tidx[h];
    a.srcs[a.t_off[idx] + atomicAdd(&a.t_fill[idx], 1u)] = e;
}
// per target: its entries into (k, j) order (rank = entries of the segment that are smaller; entry numbers are unique), entries ->
// source points in srcs_pts; the table slot and the fill counter go back to empty.  Targets with >= 4 entries (the first n_big)
// get a workgroup each -- a hub of the first batches collects thousands --, the others a thread.
__global__ __launch_bounds__(256) void group_order_big_kernel(GroupArgs a) {
    const uint32_t idx = blockIdx.x, off = a.t_off[idx], c = a.t_cnt[idx];
    const uint32_t* seg = a.srcs + off;
    for (uint32_t i = threadIdx.x; i < c; i += 256) {
        const uint32_t v = seg[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < c; j++) rank += seg[j] < v;
        a.srcs_pts[off + rank] = a.points[v / (uint32_t)a.r];
    }
    if (threadIdx.x == 0) {
        const uint32_t h = a.t_slot[idx];
        a.keys[h] = 0xffffffffu; a.cnt[h] = 0u; a.t_fill[idx] = 0u;
    }
}
__global__ void group_order_small_kernel(GroupArgs a, uint32_t first, uint32_t n_targets) {
    const uint32_t idx = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_targets) return;
    const uint32_t off = a.t_off[idx], c = a.t_cnt[idx];   // c <= 3
    uint32_t v[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    for (uint32_t i = 0; i < c && i < 3; i++) v[i] = a.srcs[off + i];
    if (v[0] > v[1]) { const uint32_t t = v[0]; v[0] = v[1]; v[1] = t; }
    if (v[1] > v[2]) { const uint32_t t = v[1]; v[1] = v[2]; v[2] = t; }
    if (v[0] > v[1]) { const uint32_t t = v[0]; v[0] = v[1]; v[1] = t; }
    for (uint32_t i = 0; i < c && i < 3; i++) a.srcs_pts[off + i] = a.points[v[i] / (uint32_t)a.r];
    const uint32_t h = a.t_slot[idx];
    a.keys[h] = 0xffffffffu; a.cnt[h] = 0u; a.t_fill[idx] = 0u;
}

// Back edges (lib.rs:311-322): workgroup b owns list targets[b] and applies its sources in order.
__global__ __launch_bounds__(GB_THREADS) void backedge_kernel(BackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d, dq = (d * 2 + 15) & ~15;
    uint16_t* s_t = reinterpret_cast<uint16_t*>(smem);
    uint16_t* s_star = reinterpret_cast<uint16_t*>(smem + dq);
    __shared__ long long c_sc[128];
    __shared__ uint32_t c_id[128], c_pos[128], s_neigh[64], cur[64];
    __shared__ uint16_t s_live[128];
    __shared__ int s_cnt, s_len;
    const int tid = threadIdx.x;
    const uint32_t t = a.targets[blockIdx.x];
    for (int e = tid; e < d / 8; e += GB_THREADS) reinterpret_cast<uint4*>(s_t)[e] = reinterpret_cast<const uint4*>(a.base + (size_t)t * d)[e];
    if (tid == 0) s_len = (int)min(a.deg[t], (uint32_t)a.r);
    __syncthreads();
    if (tid < s_len) cur[tid] = a.adj[(size_t)t * a.r + tid];
    __syncthreads();
    PruneParams pp{a.base, d, a.qb, a.alpha, a.qalpha, a.r, a.saturate, a.n, a.err};
    int N = 2;
    while (N < a.r + 1) N <<= 1;
    for (uint32_t si = a.src_off[blockIdx.x], se = back_end(a, blockIdx.x); si < se; si++) {
        const uint32_t p = a.srcs[si];
        const int len = s_len;
        __syncthreads();   // everyone has the length before thread 0 may change it below
        if (len == a.r) {   // :314-318 -- the full list plus the newcomer, scored against the list's owner
            {
                const int e = tid >> 2;
                uint32_t id = cur[e < len ? e : len - 1];
                if (id >= pp.n) { atomicOr(pp.err, 64u); id = 0; }
                const float f = quad_fast_dot_f32(a.base + (size_t)id * d, s_t, d);
                if (e < len && (tid & 3) == 0) { c_sc[e] = scale_dot_result(f); c_id[e] = id; c_pos[e] = (uint32_t)e; }
            }
            if (tid < 64) {
                if (p >= pp.n) atomicOr(pp.err, 128u);
                const float f = quad_fast_dot_f32(a.base + (size_t)(p >= pp.n ? 0 : p) * d, s_t, d);
                if (tid == 0) { c_sc[len] = scale_dot_result(f); c_id[len] = p; c_pos[len] = (uint32_t)len; }
            }
            for (int e = len + 1 + tid; e < N; e += GB_THREADS) { c_sc[e] = GB_MIN; c_id[e] = 0xffffffffu; c_pos[e] = 0xffffffffu; }
            __syncthreads();
            wg_sort(c_sc, c_id, c_pos, N);
            int nc = len + 1;
            if (nc > a.maxc) nc = a.maxc;
            const int nn = wg_robust_prune(pp, t, nc, c_sc, c_id, s_star, s_live, s_neigh, &s_cnt);
            if (tid < nn) cur[tid] = s_neigh[tid];
            if (tid == 0) s_len = nn;
        } else if (tid == 0) {   // :319-321
            bool have = false;
            for (int e = 0; e < len; e++) have |= cur[e] == p;
            if (!have && len < a.r) { cur[len] = p; s_len = len + 1; }
        }
        __syncthreads();
    }
    if (tid < s_len) a.adj[(size_t)t * a.r + tid] = cur[tid];
    if (tid == 0) a.deg[t] = (uint32_t)s_len;
}

// ---- back edges with the candidate products taken from the matrix cores --------------------------------------------
// robust_prune over a full list and one newcomer asks for up to 65*64/2 products between candidates, but only to
// decide `(alpha * s) >> 16 >= score` (lib.rs:266-268).  The 65 x 65 Gram matrix of the candidate rows costs 540
// v_mfma_f32_16x16x32_f16 per list; its entries are not fast_dot's bit pattern, so a decision is taken from them only
// when it holds for every value within the error bound of the MFMA sum (the scan kernel's certificate, DESIGN 3.1:
// |mfma - fast_dot| <= 2.8e-4 * |x| |y|); a comparison inside the bound is settled by the exact dot.  The scores
// against the list's owner, which fix the candidate order and are the right-hand sides, are always exact.  The
// result is therefore the reference's, bit for bit; only the cost changes.
// One WAVE per list (four independent waves per workgroup, no block barriers): exact scores by 16 lane quads per pass;
// the Gram tiles accumulate from fragments read straight from global memory (lane = row i, k-slot g, 16 bytes);
// the matrix, the sorted candidates and the prune state live in the wave's 18 KiB of LDS and in registers.

constexpr int GR_NB = 5;                  // 16-row blocks: up to 80 >= 65 candidates
constexpr int GR_C = GB_RMAX + 1;         // 65 candidates at most
constexpr int GR_GS = 66;                 // row stride of the matrix in LDS
constexpr int GR_WAVE_LDS = GR_C * GR_GS * 4 + 72 * 8 + 72 * 4 + 72 * 4;

struct GramArgs {
    BackArgs b;
    int n_targets;
    long long eps_fix;   // ceil(2^32 * bound on |mfma - fast_dot|) + 1
};

__global__ __launch_bounds__(256) void backedge_gram_kernel(GramArgs ga) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BackArgs& a = ga.b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ti = blockIdx.x * 4 + wave;
    if (ti >= ga.n_targets) return;   // the waves of a workgroup never meet at a barrier
    char* wl = smem + wave * GR_WAVE_LDS;
    float* G = reinterpret_cast<float*>(wl);
    long long* s_sc = reinterpret_cast<long long*>(wl + GR_C * GR_GS * 4);
    uint32_t* s_id = reinterpret_cast<uint32_t*>(wl + GR_C * GR_GS * 4 + 72 * 8);
    uint32_t* s_pos = s_id + 72;
    const int d = a.d, r = a.r, T = d / 32;
    const uint32_t t = a.targets[ti];
    const uint16_t* trow = a.base + (size_t)t * d;
    int len = (int)min(a.deg[t], (uint32_t)r);
    uint32_t cur = lane < len ? a.adj[(size_t)t * r + lane] : 0u;   // lane l holds list entry l
    const int i16 = lane & 15, g4 = lane >> 4;

    for (uint32_t si = a.src_off[ti], se = back_end(a, (uint32_t)ti); si < se; si++) {
        const uint32_t p = a.srcs[si];
        if (len != r) {   // lib.rs:319-321
            if (!__ballot(lane < len && cur == p) && len < r) {
                if (lane == len) cur = p;
                len++;
            }
            continue;
        }
        const int nc0 = r + 1;   // candidate c < r is list entry c, candidate r is the newcomer
        // ---- exact scores against the owner (merge_existing_neighbours, lib.rs:215-221) ----
        long long my_sc = GB_MIN, p_sc = 0;
        for (int c0 = 0; c0 < nc0; c0 += 16) {
            int c = c0 + (lane >> 2);
            if (c > r) c = r;
            uint32_t idc = (uint32_t)__shfl((int)cur, c < r ? c : 0);
            if (c == r) idc = p;
            if (idc >= a.n) { atomicOr(a.err, 64u); idc = 0; }
            const long long s = scale_dot_result(quad_fast_dot_f32(a.base + (size_t)idc * d, trow, d));
            const long long got = shfl_i64(s, ((lane - c0) & 15) * 4);
            if (lane >= c0 && lane < c0 + 16 && lane < r) my_sc = got;
            if (r >= c0 && r < c0 + 16) p_sc = shfl_i64(s, (r - c0) * 4);
        }
        // ---- Gram matrix of the candidate rows on the matrix cores ----
        {
            const uint4* rp[GR_NB];
#pragma unroll
            for (int b = 0; b < GR_NB; b++) {
                int c = 16 * b + i16;
                if (c > r) c = r;
                uint32_t idc = (uint32_t)__shfl((int)cur, c < r ? c : 0);
                if (c == r) idc = p;
                if (idc >= a.n) idc = 0;
                rp[b] = reinterpret_cast<const uint4*>(a.base + (size_t)idc * d) + g4;
            }
            const int nb = (nc0 + 15) / 16;
            float4v acc[15];
#pragma unroll
            for (int x = 0; x < 15; x++) acc[x] = (float4v){0.0f, 0.0f, 0.0f, 0.0f};
            uint4 fa[GR_NB], fb[GR_NB];
#pragma unroll
            for (int b = 0; b < GR_NB; b++) fa[b] = b < nb ? rp[b][0] : make_uint4(0, 0, 0, 0);
            for (int ks = 0; ks < T; ks += 2) {
                const int k1 = ks + 1 < T ? ks + 1 : T - 1, k2 = ks + 2 < T ? ks + 2 : T - 1;   // clamped look-ahead: always inside the row
#pragma unroll
                for (int b = 0; b < GR_NB; b++) fb[b] = b < nb ? rp[b][k1 * 4] : make_uint4(0, 0, 0, 0);
                {
                    int x = 0;
#pragma unroll
                    for (int I = 0; I < GR_NB; I++)
#pragma unroll
                        for (int J = I; J < GR_NB; J++, x++)
                            if (J < nb)
                                acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fa[I]), __builtin_bit_cast(half8, fa[J]), acc[x], 0, 0, 0);
                }
#pragma unroll
                for (int b = 0; b < GR_NB; b++) fa[b] = b < nb ? rp[b][k2 * 4] : make_uint4(0, 0, 0, 0);
                if (ks + 1 < T) {
                    int x = 0;
#pragma unroll
                    for (int I = 0; I < GR_NB; I++)
#pragma unroll
                        for (int J = I; J < GR_NB; J++, x++)
                            if (J < nb)
                                acc[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, fb[I]), __builtin_bit_cast(half8, fb[J]), acc[x], 0, 0, 0);
                }
            }
            // tile (I, J): this lane holds rows 16 I + 4 g + v, column 16 J + i; both triangles are written
            int x = 0;
#pragma unroll
            for (int I = 0; I < GR_NB; I++)
#pragma unroll
                for (int J = I; J < GR_NB; J++, x++)
                    if (J < nb) {
                        const int col = 16 * J + i16;
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const int row = 16 * I + 4 * g4 + v;
                            if (row < nc0 && col < nc0) { G[row * GR_GS + col] = acc[x][v]; G[col * GR_GS + row] = acc[x][v]; }
                        }
                    }
        }
        // ---- candidate order: score descending, earlier position first (lib.rs:233), by counting ----
        {
            int rank = 0, rank_p = 0;
            for (int m = 0; m < r; m++) {
                const long long sm = readlane_i64(my_sc, m);
                rank += (sm > my_sc || (sm == my_sc && m < lane)) ? 1 : 0;
                rank_p += sm >= p_sc ? 1 : 0;          // the newcomer is the last position: ties go before it
            }
            rank += p_sc > my_sc ? 1 : 0;
            if (lane < r) { s_sc[rank] = my_sc; s_id[rank] = cur; s_pos[rank] = (uint32_t)lane; }
            if (lane == 0) { s_sc[rank_p] = p_sc; s_id[rank_p] = p; s_pos[rank_p] = (uint32_t)r; }
        }
        int nc = nc0 < a.maxc ? nc0 : a.maxc;
        // sorted entry i = lane in registers; entry 64 (only when nc0 == 65) is kept by every lane
        long long e_sc = lane < nc ? s_sc[lane] : GB_MIN;
        const uint32_t e_id = lane < nc ? s_id[lane] : 0u, e_pos = lane < nc ? s_pos[lane] : 0u;
        const bool has64 = nc > 64;
        long long x_sc = has64 ? s_sc[64] : GB_MIN;
        const uint32_t x_id = has64 ? s_id[64] : 0u, x_pos = has64 ? s_pos[64] : 0u;
        // ---- robust_prune's loop (lib.rs:236-272) ----
        uint32_t mine = 0u;   // lane l: new list entry l
        int nn = 0, ci = 0;
        while (nn < r && ci < nc) {
            const uint32_t p_star = ci < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)e_id, ci) : x_id;
            const uint32_t pos_star = ci < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)e_pos, ci) : x_pos;
            const long long ps = ci < 64 ? readlane_i64(e_sc, ci) : x_sc;
            ci++;
            if (p_star == t || ps == GB_MIN) continue;
            if (lane == nn) mine = p_star;
            nn++;
            // candidates ci + 1 .. nc - 1 that are still alive (the one right behind p_star escapes, as in the reference)
            const float* grow = G + pos_star * GR_GS;
            bool border = false, border64 = false;
            if (lane >= ci + 1 && lane < nc && lane < 64 && e_sc != GB_MIN) {
                const long long al = e_id >= a.qb ? a.qalpha : a.alpha;
                const long long scaled = (long long)((unsigned long long)al * (unsigned long long)scale_dot_result(grow[e_pos])) >> 16;
                const long long m = ((al * ga.eps_fix) >> 16) + 2;
                if (scaled - m >= e_sc) e_sc = GB_MIN;
                else if (scaled + m >= e_sc) border = true;
            }
            if (has64 && 64 >= ci + 1 && x_sc != GB_MIN) {   // uniform
                const long long al = x_id >= a.qb ? a.qalpha : a.alpha;
                const long long scaled = (long long)((unsigned long long)al * (unsigned long long)scale_dot_result(grow[x_pos])) >> 16;
                const long long m = ((al * ga.eps_fix) >> 16) + 2;
                if (scaled - m >= x_sc) x_sc = GB_MIN;
                else if (scaled + m >= x_sc) border64 = true;
            }
            // comparisons inside the error bound: the exact dot decides, sixteen candidates per pass (one lane quad each)
            unsigned long long bm = __ballot(border);
            while (bm) {
                const int cnt = __popcll(bm), take = cnt < 16 ? cnt : 16;
                unsigned long long mm = bm;
                for (int z = (lane >> 2) < take ? (lane >> 2) : 0; z > 0; z--) mm &= mm - 1;
                const int j = __ffsll((long long)mm) - 1;   // this quad's candidate (quads past `take` repeat the first)
                const uint32_t idj = (uint32_t)__shfl((int)e_id, j);
                const long long s = scale_dot_result(quad_fast_dot_f32(a.base + (size_t)idj * d, a.base + (size_t)p_star * d, d));
                const long long al = idj >= a.qb ? a.qalpha : a.alpha;
                const int drop = ((long long)((unsigned long long)al * (unsigned long long)s) >> 16) >= shfl_i64(e_sc, j) ? 1 : 0;
                for (int q = 0; q < take; q++) {
                    const int jq = __builtin_amdgcn_readlane(j, q * 4), dq = __builtin_amdgcn_readlane(drop, q * 4);
                    if (dq && lane == jq) e_sc = GB_MIN;
                    bm &= bm - 1;
                }
            }
            if (border64) {
                const long long s = scale_dot_result(quad_fast_dot_f32(a.base + (size_t)x_id * d, a.base + (size_t)p_star * d, d));
                const long long al = x_id >= a.qb ? a.qalpha : a.alpha;
                if (((long long)((unsigned long long)al * (unsigned long long)s) >> 16) >= x_sc) x_sc = GB_MIN;
            }
        }
        if (a.saturate || t >= a.qb) {   // lib.rs:275-284
            for (int i = 0; i < nc && nn < r; i++) {
                const uint32_t id = i < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)e_id, i) : x_id;
                if (__ballot(lane < nn && mine == id)) continue;
                if (lane == nn) mine = id;
                nn++;
            }
        }
        cur = lane < nn ? mine : 0u;
        len = nn;
    }
    if (lane < len) a.adj[(size_t)t * r + lane] = cur;
    if (lane == 0) a.deg[t] = (uint32_t)len;
}

struct StitchArgs {
    const uint16_t* base; int d;
    uint32_t* adj; uint32_t* deg; int r;
    const uint32_t* targets; const uint32_t* src_off; const uint32_t* srcs;
    int max_add;
};

// robust_stitch's second half (lib.rs:348-373): workgroup b owns base node targets[b]; for each query that it pointed
// at, in order, the query's out-neighbours are scored against the base node and the best new ones appended.
__global__ __launch_bounds__(GB_THREADS) void stitch_kernel(StitchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = a.d;
    uint16_t* s_t = reinterpret_cast<uint16_t*>(smem);
    __shared__ long long c_sc[64];
    __shared__ uint32_t c_id[64], c_pos[64];
    const int tid = threadIdx.x;
    const uint32_t b = a.targets[blockIdx.x];
    for (int e = tid; e < d / 8; e += GB_THREADS) reinterpret_cast<uint4*>(s_t)[e] = reinterpret_cast<const uint4*>(a.base + (size_t)b * d)[e];
    int len = (int)min(a.deg[b], (uint32_t)a.r);   // tracked by wave 0
    uint32_t mine = (tid < 64 && tid < len) ? a.adj[(size_t)b * a.r + tid] : 0u;
    __syncthreads();
    for (uint32_t si = a.src_off[blockIdx.x]; si < a.src_off[blockIdx.x + 1]; si++) {
        const uint32_t q = a.srcs[si];
        const int qn = (int)min(a.deg[q], (uint32_t)a.r);
        if (qn == 0) continue;
        int N = 2;
        while (N < qn) N <<= 1;
        {
            const int e = tid >> 2;
            const uint32_t id = a.adj[(size_t)q * a.r + (e < qn ? e : qn - 1)];
            const float f = quad_fast_dot_f32(a.base + (size_t)id * d, s_t, d);
            if ((tid & 3) == 0 && e < N) {
                const bool in = e < qn;
                c_sc[e] = in ? scale_dot_result(f) : GB_MIN; c_id[e] = in ? id : 0xffffffffu; c_pos[e] = in ? (uint32_t)e : 0xffffffffu;
            }
        }
        __syncthreads();
        wg_sort(c_sc, c_id, c_pos, N);   // :359
        if (tid < 64) {   // :361-371
            int added = 0;
            for (int i = 0; i < qn; i++) {
                if (added >= a.max_add || len >= a.r) break;
                const uint32_t id = c_id[i];
                if (__ballot(tid < len && mine == id)) continue;
                if (tid == len) mine = id;
                len++;
                added++;
            }
        }
        __syncthreads();
    }
    if (tid < 64) {
        if (tid < len) a.adj[(size_t)b * a.r + tid] = mine;
        if (tid == 0) a.deg[b] = (uint32_t)len;
    }
}

__device__ __forceinline__ uint32_t philox_first(uint32_t c0, uint32_t c2, uint32_t k0, uint32_t k1) {
    uint32_t c1 = 0, c3 = 0;
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

__global__ void random_fill_kernel(uint32_t* adj, uint32_t* deg, uint32_t n, int stride, int r, uint32_t seed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* l = adj + (size_t)i * stride;
    uint32_t len = deg[i];
    for (uint32_t k = 0; len < (uint32_t)r && k < (1u << 20); k++) {
        const uint32_t next = (uint32_t)(((uint64_t)philox_first(k, i, seed, 0xF111u) * (uint64_t)n) >> 32);
        bool have = false;
        for (uint32_t e = 0; e < len; e++) have |= l[e] == next;
        if (!have) l[len++] = next;
    }
    deg[i] = len;
}

__global__ void check_graph_kernel(const uint32_t* adj, const uint32_t* deg, uint32_t n, int stride, uint32_t* err) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * stride) return;
    const uint32_t node = (uint32_t)(i / stride), e = (uint32_t)(i % stride);
    if (e == 0 && deg[node] > (uint32_t)stride) atomicOr(err, 1u);
    if (e < deg[node] && adj[i] >= n) atomicOr(err, 1u);
}

int check_config(const mse_searcher* s, const mse_graph* g, const mse_build_config* cfg, const char* who) {
    if (!s || !s->base || !g || !cfg) return fail(std::string(who) + ": null argument");
    const mse_base* b = s->base;
    if (g->n != b->n) return fail(std::string(who) + ": graph and vectors differ in length");
    if (b->n >= 0xffffffffull) return fail(std::string(who) + ": too many vectors");   // lib.rs:288
    if (cfg->r == 0 || cfg->r > GB_RMAX || cfg->r != g->max_deg) return fail(std::string(who) + ": r must be 1..64 and equal the graph's stride");
    if (cfg->l == 0 || cfg->l > GB_LMAX) return fail(std::string(who) + ": l must be 1..1024");
    if (cfg->maxc == 0 || cfg->maxc > GB_CMAX) return fail(std::string(who) + ": maxc must be 1..1024");
    if (b->d % 32 || b->d > 4096) return fail(std::string(who) + ": vector width must be a multiple of 32");
    return 0;
}

int check_graph(const mse_graph* g, hipStream_t st, const char* who) {
    DevBuf e;
    if (e.ensure(4)) return -1;
    MSE_HIP_TRY(hipMemsetAsync(e.p, 0, 4, st));
    const size_t total = g->n * g->max_deg;
    hipLaunchKernelGGL(check_graph_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g->adj, g->deg, (uint32_t)g->n, (int)g->max_deg, e.as<uint32_t>());
    MSE_HIP_TRY(hipGetLastError());
    uint32_t err = 0;
    MSE_HIP_TRY(hipMemcpyAsync(&err, e.p, 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    if (err) return fail(std::string(who) + ": the graph has an edge outside 0..n or a list longer than its stride");
    return 0;
}

// Bound, in the fixed-point scale, on |MFMA f16 sum - fast_dot| for two base rows: 2.8e-4 * (largest row norm)^2, the
// allowance the scan's certificate uses.  *eps_fix = 0 when no bound can be stated (the exact kernels run).
int mfma_bound(const mse_base* b, const mse_build_config* cfg, hipStream_t st, long long* eps_fix) {
    *eps_fix = 0;
    if (b->d % 64 || cfg->alpha <= 0 || cfg->alpha > (1 << 20) || cfg->query_alpha <= 0 || cfg->query_alpha > (1 << 20)) return 0;
    if (ensure_base_norm(b, st)) return -1;
    uint32_t bits[3] = {0, 0, 0};
    MSE_HIP_TRY(hipMemcpy(bits, b->norm_bits_dev, 12, hipMemcpyDeviceToHost));
    float mx, sub, ab;
    memcpy(&mx, &bits[0], 4); memcpy(&sub, &bits[1], 4); memcpy(&ab, &bits[2], 4);
    const char* sc = getenv("MSE_GRAM_EPS_SCALE");   // test hook: widen (or zero) the band in which the exact dot decides
    // accumulation error of the MFMA sum + the products a matrix core that flushes f16 subnormal inputs would drop
    const double bound = (2.8e-4 * (double)mx * (double)mx + 2.0002 * (double)sub * (double)ab) * (sc ? atof(sc) : 1.0);
    if (!(bound == bound) || bound > 1e6) return 0;
    *eps_fix = (long long)ceil(bound * 4294967296.0) + 1;
    return 0;
}

template <typename K> int set_lds(K kernel) {
    MSE_DYN_LDS(kernel, 96 * 1024);
    return 0;
}

}  // namespace

extern "C" {

mse_graph* mse_graph_new(size_t n, size_t max_deg) {
    if (n == 0 || max_deg == 0) { fail("graph_new: bad argument"); return nullptr; }
    mse_graph* g = new (std::nothrow) mse_graph();
    if (!g) { fail("out of host memory"); return nullptr; }
    g->n = n; g->max_deg = max_deg;
    bool ok = hipMalloc((void**)&g->adj, n * max_deg * 4) == hipSuccess && hipMalloc((void**)&g->deg, n * 4) == hipSuccess;
    ok = ok && hipMemset(g->adj, 0, n * max_deg * 4) == hipSuccess && hipMemset(g->deg, 0, n * 4) == hipSuccess;
    if (!ok) { mse_graph_free(g); fail("graph_new: device allocation failed"); return nullptr; }
    return g;
}

int mse_graph_to_host(const mse_graph* g, uint32_t* adj, uint32_t* deg) {
    if (!g || !adj || !deg) return fail("graph_to_host: null argument");
    MSE_HIP_TRY(hipDeviceSynchronize());
    MSE_HIP_TRY(hipMemcpy(adj, g->adj, g->n * g->max_deg * 4, hipMemcpyDeviceToHost));
    MSE_HIP_TRY(hipMemcpy(deg, g->deg, g->n * 4, hipMemcpyDeviceToHost));
    return 0;
}

size_t mse_graph_len(const mse_graph* g) { return g ? g->n : 0; }
size_t mse_graph_max_degree(const mse_graph* g) { return g ? g->max_deg : 0; }

int mse_graph_random_fill(mse_graph* g, uint32_t seed, size_t r) {
    if (!g) return fail("graph_random_fill: null argument");
    if (r == 0 || r > g->max_deg) return fail("graph_random_fill: r must be 1..max_deg");
    if (g->n >= 0xffffffffull) return fail("graph_random_fill: too many nodes");
    hipLaunchKernelGGL(random_fill_kernel, dim3((unsigned)((g->n + 255) / 256)), dim3(256), 0, 0, g->adj, g->deg, (uint32_t)g->n, (int)g->max_deg, (int)r, seed);
    MSE_HIP_TRY(hipGetLastError());
    MSE_HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int mse_build_graph(mse_searcher* s, mse_graph* g, const uint32_t* order, size_t n_order, size_t batch, uint32_t medioid,
                    const mse_build_config* cfg) {
    if (check_config(s, g, cfg, "build_graph")) return -1;
    if (n_order && !order) return fail("build_graph: null argument");
    const mse_base* b = s->base;
    if (medioid >= b->n) return fail("build_graph: medioid out of range");
    for (size_t i = 0; i < n_order; i++)
        if (order[i] >= b->n) return fail("build_graph: point out of range");
    if (n_order == 0) return 0;
    if (batch == 0) batch = 1;
    if (batch > n_order) batch = n_order;
    if (batch > 65536) batch = 65536;
    hipStream_t st = s->stream;
    if (check_graph(g, st, "build_graph")) return -1;
    const int r = (int)cfg->r, d = (int)b->d;
    const size_t words = (b->n + 31) / 32;
    const char* vm = getenv("MSE_VISITED_MODE");   // test hook: "hash" / "bitmap"
    size_t vl_cap = std::max<size_t>(4096, 2 * cfg->l * cfg->r) + cfg->r;
    DevBuf d_order, bm, vli, vls, stg, stg_len, err, cnts, grp;
    // visited sets: bit maps, or hash tables once the index is so large that the tables are the smaller ones (visited_set.h)
    int table_bits = visited_table_bits(std::min<size_t>(b->n, vl_cap));
    bool use_hash = vm ? !strcmp(vm, "hash") : words > ((size_t)1 << table_bits);
    size_t set_words = use_hash ? (size_t)1 << table_bits : words;
    if (cnts.ensure(batch * 4) || d_order.ensure(n_order * 4) || bm.ensure(batch * set_words * 4) || vli.ensure(batch * vl_cap * 4) || vls.ensure(batch * vl_cap * 8) ||
        stg.ensure(batch * r * 4) || stg_len.ensure(batch * 4) || err.ensure(8))
        return -1;
    // back-edge grouping on the device (group_*_kernel): a table of >= 2 slots per entry of a batch + per-entry / per-target arrays
    const size_t n_ent = batch * (size_t)r;
    int gbits = 10;
    while (((size_t)1 << gbits) < 2 * n_ent) gbits++;
    const size_t TS = (size_t)1 << gbits;
    if (grp.ensure(TS * 12 + n_ent * 4 * 8 + 64)) return -1;
    GroupArgs ga{};
    {
        char* q = grp.as<char>();
        ga.keys = reinterpret_cast<uint32_t*>(q); q += TS * 4;
        ga.cnt = reinterpret_cast<uint32_t*>(q); q += TS * 4;
        ga.tidx = reinterpret_cast<uint32_t*>(q); q += TS * 4;
        ga.ent_slot = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.targets = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.t_off = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.t_cnt = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.t_slot = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.t_fill = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.srcs = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.srcs_pts = reinterpret_cast<uint32_t*>(q); q += n_ent * 4;
        ga.counters = reinterpret_cast<uint32_t*>(q);
        ga.bits = gbits; ga.r = r; ga.stg = stg.as<uint32_t>(); ga.len = stg_len.as<uint32_t>();
    }
    MSE_HIP_TRY(hipMemsetAsync(ga.keys, 0xff, TS * 4, st));                 // empty table; the kernels leave it empty again
    MSE_HIP_TRY(hipMemsetAsync(ga.cnt, 0, TS * 4, st));
    MSE_HIP_TRY(hipMemsetAsync(ga.t_fill, 0, n_ent * 4, st));
    MSE_HIP_TRY(hipMemsetAsync(err.p, 0, 8, st));                           // [0] search / prune (reset per try), [1] back edges (sticky)
    MSE_HIP_TRY(hipMemcpyAsync(d_order.p, order, n_order * 4, hipMemcpyHostToDevice, st));
    if (set_lds(graph_search_kernel<true>) || set_lds(prune_kernel<false>) || set_lds(prune_kernel<true>)) return -1;
    const size_t lds = search_lds_bytes(d, (int)cfg->l);
    PruneParams pp{b->dev, d, cfg->query_breakpoint, cfg->alpha, cfg->query_alpha, r, (int)cfg->saturate_graph, (uint32_t)b->n, err.as<uint32_t>(), 0};
    GraphArgs a{};
    a.base = b->dev; a.n = (uint32_t)b->n; a.d = d;
    a.adj = g->adj; a.deg = g->deg; a.r = r;
    a.medioid = medioid; a.qb = cfg->query_breakpoint;
    a.L = (int)cfg->l; a.maxc = (int)cfg->maxc; a.saturate = (int)cfg->saturate_graph; a.alpha = cfg->alpha; a.qalpha = cfg->query_alpha;
    a.out_dist = cnts.as<uint32_t>();
    a.err = err.as<uint32_t>();
    BackArgs ba{};
    ba.base = b->dev; ba.d = d; ba.adj = g->adj; ba.deg = g->deg; ba.r = r;
    ba.n = (uint32_t)b->n; ba.err = err.as<uint32_t>() + 1;
    ba.qb = cfg->query_breakpoint; ba.maxc = (int)cfg->maxc; ba.saturate = (int)cfg->saturate_graph; ba.alpha = cfg->alpha; ba.qalpha = cfg->query_alpha;
    const size_t back_lds = 2 * (size_t)((d * 2 + 15) & ~15);
    // candidate products from the matrix cores wherever the error bound can be stated (finite norms, sane factors)
    long long eps_fix = 0;
    if (mfma_bound(b, cfg, st, &eps_fix)) return -1;
    const bool use_gram = eps_fix > 0 && !getenv("MSE_BUILD_EXACT_BACKEDGE");
    if (use_gram && set_lds(backedge_gram_kernel)) return -1;
    // the candidate-major MFMA walk of the prune keeps the selected rows in registers as 36 fragments: built for d = 1152
    if (eps_fix > 0 && d == 1152 && !getenv("MSE_BUILD_EXACT_PRUNE")) pp.eps_fix = eps_fix;
    for (size_t b0 = 0; b0 < n_order; b0 += batch) {
        const size_t nb = std::min(batch, n_order - b0);
        a.points = d_order.as<uint32_t>() + b0;
        for (;;) {
            a.vl_ids = vli.as<uint32_t>(); a.vl_sc = vls.as<long long>(); a.vl_cap = (uint32_t)vl_cap;
            a.bitmap = bm.as<uint32_t>(); a.bm_words = set_words; a.hash_bits = use_hash ? table_bits : 0;
            MSE_HIP_TRY(hipMemsetAsync(bm.p, use_hash ? 0xff : 0, nb * set_words * 4, st));
            MSE_HIP_TRY(hipMemsetAsync(err.p, 0, 4, st));   // word 0 only
            hipLaunchKernelGGL(graph_search_kernel<true>, dim3((unsigned)nb), dim3(GS_THREADS), lds, st, a);
            MSE_HIP_TRY(hipGetLastError());
            if (pp.eps_fix > 0)
                hipLaunchKernelGGL(prune_kernel<true>, dim3((unsigned)nb), dim3(GB_THREADS), prune_lds_bytes(d), st, pp, a.vl_ids, a.vl_sc, vl_cap,
                                   cnts.as<uint32_t>(), a.points, (int)cfg->maxc, stg.as<uint32_t>(), stg_len.as<uint32_t>());
            else
                hipLaunchKernelGGL(prune_kernel<false>, dim3((unsigned)nb), dim3(GB_THREADS), prune_lds_bytes(d), st, pp, a.vl_ids, a.vl_sc, vl_cap,
                                   cnts.as<uint32_t>(), a.points, (int)cfg->maxc, stg.as<uint32_t>(), stg_len.as<uint32_t>());
            MSE_HIP_TRY(hipGetLastError());
            uint32_t e2[2] = {0, 0};
            MSE_HIP_TRY(hipMemcpyAsync(e2, err.p, 8, hipMemcpyDeviceToHost, st));
            MSE_HIP_TRY(hipStreamSynchronize(st));
            const uint32_t e = e2[0];
            if (e2[1]) return fail("build_graph: internal error " + std::to_string(e2[1]) + " (a candidate id outside the index)");   // the previous batch's back edges
            if (e & 1u) return fail("build_graph: a graph edge points outside the index");
            if (!(e & 6u)) break;
            // a search visited more nodes than there was room for (list or table): repeat the batch with more (the graph is untouched so far)
            if (e & 2u) {
                vl_cap *= 2;
                if (vl_cap > b->n + cfg->r) vl_cap = b->n + cfg->r;
                if (vli.ensure(batch * vl_cap * 4) || vls.ensure(batch * vl_cap * 8)) return -1;
            }
            if (use_hash) {
                table_bits = std::max(table_bits + ((e & 4u) ? 1 : 0), visited_table_bits(std::min<size_t>(b->n, vl_cap)));
                if (words <= ((size_t)1 << table_bits)) use_hash = false;
                set_words = use_hash ? (size_t)1 << table_bits : words;
                if (bm.ensure(batch * set_words * 4)) return -1;
            }
        }
        hipLaunchKernelGGL(apply_lists_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, st, g->adj, g->deg, r, a.points, stg.as<uint32_t>(),
                           stg_len.as<uint32_t>(), (int)nb);
        MSE_HIP_TRY(hipGetLastError());
        // back edges grouped by the list they touch, each group in (position in batch, position in list) order, lists with many
        // newcomers first -- on the device (group_*_kernel above); only the number of lists comes back
        ga.nb = (uint32_t)nb; ga.points = a.points;
        MSE_HIP_TRY(hipMemsetAsync(ga.counters, 0, 32, st));
        const unsigned g_ent = (unsigned)((nb * (size_t)r + 255) / 256), g_tab = (unsigned)(TS / 256);
        hipLaunchKernelGGL(group_count_kernel, dim3(g_ent), dim3(256), 0, st, ga);
        hipLaunchKernelGGL(group_classify_kernel, dim3(g_tab), dim3(256), 0, st, ga);
        hipLaunchKernelGGL(group_compact_kernel, dim3(g_tab), dim3(256), 0, st, ga);
        MSE_HIP_TRY(hipGetLastError());
        uint32_t h_cnt[2] = {0, 0};
        MSE_HIP_TRY(hipMemcpyAsync(h_cnt, ga.counters, 8, hipMemcpyDeviceToHost, st));
        MSE_HIP_TRY(hipStreamSynchronize(st));
        const uint32_t n_targets = h_cnt[0] + h_cnt[1];
        if (n_targets == 0) continue;
        hipLaunchKernelGGL(group_fill_kernel, dim3(g_ent), dim3(256), 0, st, ga);
        if (h_cnt[0]) hipLaunchKernelGGL(group_order_big_kernel, dim3(h_cnt[0]), dim3(256), 0, st, ga);
        if (h_cnt[1]) hipLaunchKernelGGL(group_order_small_kernel, dim3((h_cnt[1] + 255) / 256), dim3(256), 0, st, ga, h_cnt[0], n_targets);
        MSE_HIP_TRY(hipGetLastError());
        ba.targets = ga.targets; ba.src_off = ga.t_off; ba.src_cnt = ga.t_cnt; ba.srcs = ga.srcs_pts;
        if (use_gram) {
            GramArgs gg{ba, (int)n_targets, eps_fix};
            hipLaunchKernelGGL(backedge_gram_kernel, dim3((unsigned)((n_targets + 3) / 4)), dim3(256), 4 * GR_WAVE_LDS, st, gg);
        } else {
            hipLaunchKernelGGL(backedge_kernel, dim3((unsigned)n_targets), dim3(GB_THREADS), back_lds, st, ba);
        }
        MSE_HIP_TRY(hipGetLastError());
        // (no wait here: the back edges' error word is sticky and is read with the next batch's, or below)
    }
    uint32_t e_last[2] = {0, 0};
    MSE_HIP_TRY(hipMemcpyAsync(e_last, err.p, 8, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    if (e_last[1]) return fail("build_graph: internal error " + std::to_string(e_last[1]) + " (a candidate id outside the index)");
    return 0;
}

int mse_robust_stitch(mse_searcher* s, mse_graph* g, const uint32_t* queries_order, const mse_build_config* cfg) {
    if (check_config(s, g, cfg, "robust_stitch")) return -1;
    const mse_base* b = s->base;
    const size_t n = b->n, r = cfg->r;
    const uint32_t qb = cfg->query_breakpoint;
    if (qb >= n) return 0;   // no query nodes: generate_index_shard.rs:129
    if (!queries_order) return fail("robust_stitch: null argument");
    hipStream_t st = s->stream;
    if (check_graph(g, st, "robust_stitch")) return -1;
    const size_t nq = n - qb;
    std::vector<uint32_t> rank(nq, 0xffffffffu);
    for (size_t k = 0; k < nq; k++) {
        if (queries_order[k] < qb || queries_order[k] >= n || rank[queries_order[k] - qb] != 0xffffffffu)
            return fail("robust_stitch: queries_order must list every query node once");
        rank[queries_order[k] - qb] = (uint32_t)k;
    }
    // lib.rs:336-346 is list surgery without arithmetic: done on the host copy of the base nodes' lists
    std::vector<uint32_t> adj((size_t)qb * r), deg(qb);
    MSE_HIP_TRY(hipMemcpyAsync(adj.data(), g->adj, (size_t)qb * r * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(deg.data(), g->deg, (size_t)qb * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    struct Job { uint32_t base, qrank, occ, query; };
    std::vector<Job> jobs;
    for (uint32_t i = 0; i < qb; i++) {
        uint32_t* l = adj.data() + (size_t)i * r;
        uint32_t w = 0, occ = 0;
        for (uint32_t e = 0; e < deg[i]; e++) {
            if (l[e] >= qb) jobs.push_back(Job{i, rank[l[e] - qb], occ++, l[e]});
            else l[w++] = l[e];
        }
        deg[i] = w;
    }
    MSE_HIP_TRY(hipMemcpyAsync(g->adj, adj.data(), (size_t)qb * r * 4, hipMemcpyHostToDevice, st));
    MSE_HIP_TRY(hipMemcpyAsync(g->deg, deg.data(), (size_t)qb * 4, hipMemcpyHostToDevice, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    if (jobs.empty()) return 0;
    // a base node meets its queries in queries_order; a query listed twice by one node is applied twice in a row (:353)
    std::sort(jobs.begin(), jobs.end(), [](const Job& x, const Job& y) {
        if (x.base != y.base) return x.base < y.base;
        if (x.qrank != y.qrank) return x.qrank < y.qrank;
        return x.occ < y.occ;
    });
    std::vector<uint32_t> targets, offs, srcs;
    for (size_t i = 0; i < jobs.size(); i++) {
        if (targets.empty() || targets.back() != jobs[i].base) { targets.push_back(jobs[i].base); offs.push_back((uint32_t)i); }
        srcs.push_back(jobs[i].query);
    }
    offs.push_back((uint32_t)jobs.size());
    DevBuf d_tg, d_off, d_src;
    if (d_tg.ensure(targets.size() * 4) || d_off.ensure(offs.size() * 4) || d_src.ensure(srcs.size() * 4)) return -1;
    MSE_HIP_TRY(hipMemcpyAsync(d_tg.p, targets.data(), targets.size() * 4, hipMemcpyHostToDevice, st));
    MSE_HIP_TRY(hipMemcpyAsync(d_off.p, offs.data(), offs.size() * 4, hipMemcpyHostToDevice, st));
    MSE_HIP_TRY(hipMemcpyAsync(d_src.p, srcs.data(), srcs.size() * 4, hipMemcpyHostToDevice, st));
    StitchArgs sa{};
    sa.base = b->dev; sa.d = (int)b->d; sa.adj = g->adj; sa.deg = g->deg; sa.r = (int)r;
    sa.targets = d_tg.as<uint32_t>(); sa.src_off = d_off.as<uint32_t>(); sa.srcs = d_src.as<uint32_t>();
    sa.max_add = (int)std::min<uint64_t>(cfg->max_add_per_stitch_iter, 1u << 20);
    hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)targets.size()), dim3(GB_THREADS), (size_t)((b->d * 2 + 15) & ~(size_t)15), st, sa);
    MSE_HIP_TRY(hipGetLastError());
    MSE_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int mse_robust_prune(mse_searcher* s, const uint32_t* cand_ids, const int64_t* cand_scores, size_t n_cand, uint32_t p,
                     const mse_build_config* cfg, uint32_t* neigh, size_t* n_neigh) {
    if (!s || !s->base || !cfg || !neigh || !n_neigh || (n_cand && (!cand_ids || !cand_scores))) return fail("robust_prune: null argument");
    const mse_base* b = s->base;
    if (cfg->r == 0 || cfg->r > GB_RMAX) return fail("robust_prune: r must be 1..64");
    if (cfg->maxc == 0 || cfg->maxc > GB_CMAX) return fail("robust_prune: maxc must be 1..1024");
    if (b->d % 32 || b->d > 4096) return fail("robust_prune: vector width must be a multiple of 32");
    if (n_cand > 0x7fffffffull) return fail("robust_prune: too many candidates");
    for (size_t i = 0; i < n_cand; i++)
        if (cand_ids[i] >= b->n) return fail("robust_prune: candidate out of range");
    hipStream_t st = s->stream;
    DevBuf ci, cs, out;
    if (ci.ensure(n_cand * 4 + 16) || cs.ensure(n_cand * 8 + 16) || out.ensure((GB_RMAX + 4) * 4)) return -1;
    if (n_cand) {
        MSE_HIP_TRY(hipMemcpyAsync(ci.p, cand_ids, n_cand * 4, hipMemcpyHostToDevice, st));
        MSE_HIP_TRY(hipMemcpyAsync(cs.p, cand_scores, n_cand * 8, hipMemcpyHostToDevice, st));
    }
    if (set_lds(prune_kernel<false>) || set_lds(prune_kernel<true>)) return -1;
    const uint32_t hdr[3] = {(uint32_t)n_cand, p, 0u};   // counts[0], points[0], error word
    MSE_HIP_TRY(hipMemcpyAsync(out.as<uint32_t>() + GB_RMAX + 1, hdr, 12, hipMemcpyHostToDevice, st));
    PruneParams pp{b->dev, (int)b->d, cfg->query_breakpoint, cfg->alpha, cfg->query_alpha, (int)cfg->r, (int)cfg->saturate_graph, (uint32_t)b->n,
                   out.as<uint32_t>() + GB_RMAX + 3, 0};
    if (b->d == 1152 && !getenv("MSE_BUILD_EXACT_PRUNE") && mfma_bound(b, cfg, st, &pp.eps_fix)) return -1;
    if (pp.eps_fix > 0)
        hipLaunchKernelGGL(prune_kernel<true>, dim3(1), dim3(GB_THREADS), prune_lds_bytes((int)b->d), st, pp, ci.as<uint32_t>(), cs.as<long long>(),
                           (size_t)0, out.as<uint32_t>() + GB_RMAX + 1, out.as<uint32_t>() + GB_RMAX + 2, (int)cfg->maxc, out.as<uint32_t>(),
                           out.as<uint32_t>() + GB_RMAX);
    else
        hipLaunchKernelGGL(prune_kernel<false>, dim3(1), dim3(GB_THREADS), prune_lds_bytes((int)b->d), st, pp, ci.as<uint32_t>(), cs.as<long long>(),
                           (size_t)0, out.as<uint32_t>() + GB_RMAX + 1, out.as<uint32_t>() + GB_RMAX + 2, (int)cfg->maxc, out.as<uint32_t>(),
                           out.as<uint32_t>() + GB_RMAX);
    MSE_HIP_TRY(hipGetLastError());
    uint32_t h[GB_RMAX + 1];
    MSE_HIP_TRY(hipMemcpyAsync(h, out.p, sizeof(h), hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    *n_neigh = h[GB_RMAX];
    for (uint32_t i = 0; i < h[GB_RMAX]; i++) neigh[i] = h[i];
    return 0;
}

static int graph_search_batch_impl(int visited_mode, mse_searcher* s, const mse_graph* g, const uint32_t* starts, const uint16_t* queries,
                                   size_t nq, size_t search_list, int base_vectors_only, uint32_t query_breakpoint, uint32_t* buf_ids,
                                   int64_t* buf_scores, uint32_t* buf_len, uint32_t* n_distances) {
    if (!s || !s->base || !g || !starts || !queries || !buf_ids || !buf_scores || !buf_len || !n_distances)
        return fail("graph_search_batch: null argument");
    if (nq == 0) return 0;
    const mse_base* b = s->base;
    const size_t words = (b->n + 31) / 32;
    const int table_bits = visited_table_bits(std::min<size_t>(b->n, search_list * 192 + 4096));
    const char* vm = getenv("MSE_VISITED_MODE");   // test hook: "hash" / "bitmap"
    const bool use_hash = visited_mode >= 0 ? visited_mode == 1 : (vm ? !strcmp(vm, "hash") : words > ((size_t)1 << table_bits));
    const size_t set_words = use_hash ? (size_t)1 << table_bits : words;
    {   // one visited set per query in flight: long batches go through in pieces that fit the budget
        const size_t per_query = set_words * 4, piece = std::max<size_t>(1, std::max(s->pool[4].cap, visited_budget_bytes()) / per_query);   // pool[4]: the bitmaps already held
        if (nq > piece) {
            for (size_t q0 = 0; q0 < nq; q0 += piece) {
                const size_t m = std::min(piece, nq - q0);
                if (graph_search_batch_impl(visited_mode, s, g, starts + q0, queries + q0 * b->d, m, search_list, base_vectors_only, query_breakpoint,
                                           buf_ids + q0 * search_list, buf_scores + q0 * search_list, buf_len + q0, n_distances + q0))
                    return -1;
            }
            return 0;
        }
    }
    if (g->n != b->n) return fail("graph_search_batch: graph and vectors differ in length");
    if (search_list == 0 || search_list > GB_LMAX) return fail("graph_search_batch: search_list must be 1..1024");
    if (g->max_deg > GB_RMAX) return fail("graph_search_batch: at most 64 neighbours per node");
    if (b->d % 32 || b->d > 4096) return fail("graph_search_batch: vector width must be a multiple of 32");
    for (size_t q = 0; q < nq; q++)
        if (starts[q] >= b->n) return fail("graph_search_batch: start node out of range");
    hipStream_t st = s->stream;
    const size_t d = b->d;
    DevBuf &dq = s->pool[0], &dst = s->pool[3], &bm = s->pool[4], &oi = s->pool[5], &os = s->pool[6], &cnt = s->pool[10];
    if (dq.ensure(nq * d * 2) || dst.ensure(nq * 4) || bm.ensure(nq * set_words * 4) || oi.ensure(nq * search_list * 4) ||
        os.ensure(nq * search_list * 8) || cnt.ensure(nq * 8 + 16))
        return -1;
    MSE_HIP_TRY(hipMemcpyAsync(dq.p, queries, nq * d * 2, hipMemcpyHostToDevice, st));
    MSE_HIP_TRY(hipMemcpyAsync(dst.p, starts, nq * 4, hipMemcpyHostToDevice, st));
    MSE_HIP_TRY(hipMemsetAsync(bm.p, use_hash ? 0xff : 0, nq * set_words * 4, st));
    MSE_HIP_TRY(hipMemsetAsync(cnt.p, 0, nq * 8 + 16, st));
    if (set_lds(graph_search_kernel<false>)) return -1;
    GraphArgs a{};
    a.base = b->dev; a.n = (uint32_t)b->n; a.d = (int)d;
    a.adj = g->adj; a.deg = g->deg; a.r = (int)g->max_deg;
    a.points = dst.as<uint32_t>(); a.queries = dq.as<uint16_t>();
    a.qb = query_breakpoint; a.base_only = base_vectors_only;
    a.L = (int)search_list;
    a.bitmap = bm.as<uint32_t>(); a.bm_words = set_words; a.hash_bits = use_hash ? table_bits : 0;
    a.out_ids = oi.as<uint32_t>(); a.out_sc = os.as<long long>(); a.out_len = cnt.as<uint32_t>(); a.out_dist = cnt.as<uint32_t>() + nq;
    a.err = cnt.as<uint32_t>() + 2 * nq;
    hipLaunchKernelGGL(graph_search_kernel<false>, dim3((unsigned)nq), dim3(GS_THREADS), search_lds_bytes((int)d, (int)search_list), st, a);
    MSE_HIP_TRY(hipGetLastError());
    uint32_t err = 0;
    MSE_HIP_TRY(hipMemcpyAsync(buf_ids, oi.p, nq * search_list * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(buf_scores, os.p, nq * search_list * 8, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(buf_len, a.out_len, nq * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(n_distances, a.out_dist, nq * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(&err, a.err, 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    if (err & 1u) return fail("graph_search_batch: a graph edge points outside the index");
    if (err & 4u)   // a search outgrew its table: the bit maps have room for everything
        return graph_search_batch_impl(0, s, g, starts, queries, nq, search_list, base_vectors_only, query_breakpoint, buf_ids, buf_scores, buf_len,
                                       n_distances);
    return 0;
}

int mse_graph_search_batch(mse_searcher* s, const mse_graph* g, const uint32_t* starts, const uint16_t* queries, size_t nq,
                           size_t search_list, int base_vectors_only, uint32_t query_breakpoint, uint32_t* buf_ids,
                           int64_t* buf_scores, uint32_t* buf_len, uint32_t* n_distances) {
    return graph_search_batch_impl(-1, s, g, starts, queries, nq, search_list, base_vectors_only, query_breakpoint, buf_ids, buf_scores, buf_len,
                                   n_distances);
}

}  // extern "C"
