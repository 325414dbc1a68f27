// GPU-resident form of the disk-index beam search (src/query_disk_index.rs:144-212), batched over queries:
// one workgroup per query, everything the reference keeps in `Scratch` lives on the device --
//   NeighbourBuffer (diskann/src/lib.rs:74-155)      three sorted arrays in LDS (capacity <= 1024)
//   neighbour_pre_buffer, the beam                    LDS
//   visited / visited_adjacent (HashSet<u32>)         one bit per node in HBM, per query
//   QueryLUT (64 x 256 f32)                           LDS, 64 KiB
//   record vectors, PQ codes, descriptors, adjacency  HBM (mse_base, mse_codes, mse_graph)
// and no host round trip happens during a search.  Semantics are the reference's, replayed in its order, including
// the quirks mse_disk_greedy_search documents (entry point inserted with score 0; the pre-buffer is cleared per beam
// iteration, so later nodes of a beam re-insert the earlier nodes' fresh neighbours).  Exact scores use the
// reference's fast_dot order (exact_dot.h), ADC sums are sequential fp32 adds in chunk order: every output is
// bit-identical to the oracle's, query by query.
//
// Work split inside a workgroup (4 waves): the list manipulation is sequential by nature and is done by wave 0 (inserts
// shift the sorted arrays 64 entries at a time); scoring the pre-buffer (64 LDS lookups or one exact dot per entry)
// is spread over all 256 lanes.
#include "../../include/mse.h"
#include "exact_dot.h"
#include "visited_set.h"
#include "runtime.h"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace mse;

namespace {

constexpr int BS_THREADS_MAX = 256;
// calls of at most BS_SMALL_NQ queries run BS_SMALL_WAVES waves per query (latency: scripts/beam_latency_probe.py)
constexpr int BS_SMALL_NQ = 0, BS_SMALL_WAVES = 4;
constexpr int BS_LMAX = 1024;
constexpr int BS_BEAM_MAX = 8;
constexpr int BS_DEG_MAX = 128;   // merged indexes: up to SHARD_SPILL x R neighbours per node
constexpr int BS_DESC_MAX = 8;

struct BeamArgs {
    const uint16_t* base; size_t n; int d;
    const uint8_t* codes; const uint8_t* desc; int n_desc;
    const uint32_t* adj; const uint32_t* deg; int max_deg; const uint8_t* has_url;
    const uint16_t* queries; const float* luts; const float* scales; const uint32_t* starts;
    int beam, L, disable_pq, p_cap;   // p_cap: pre-buffer entries = beam x max_deg rounded up to 64
    uint32_t* bm_adj; uint32_t* bm_vis; size_t bm_words; int hash_bits;   // visited sets (visited_set.h): bm_words u32 per set
    uint32_t* out_ids; long long* out_scores; uint32_t* out_len;
    uint32_t* vis_ids; long long* vis_scores; size_t vis_cap; uint32_t* n_visited;
    uint32_t* cmps; uint32_t* pq_cmps; uint32_t* err;
    unsigned long long* totals;   // optional (mse_searcher_beam_timing): [0] rows scored exactly, [1] nodes fetched, [2] ADC-scored neighbours
    int hash_slots;   // LDS table of a beam iteration's neighbour ids: power of two >= 2 x p_cap
    int fill_vis;   // fused request path: slots of the visited arrays past n_visited are set to (ID_NONE, INT64_MIN) for the device top-k
    // small-batch entry step (entry_top1_rows_kernel): the per-chunk bests [chunk][query] are reduced HERE, by the search's first wave,
    // instead of by a launch of their own (a dependent launch costs ~35 us in a pass of ~280); null = start nodes in `starts`
    const long long* entry_psc; const uint32_t* entry_prow; const uint32_t* entry_ids; int entry_chunks, entry_nq;
};

// THREADS = 256: four waves per query (wave 0 walks the list, all four score and merge) -- needed when the 64 KiB distance table of
// a query sits in LDS (two queries per CU either way) and for the longest lists (4 x THREADS list entries / pre-buffer entries).
// THREADS = 64 (round 4): ONE wave per query, for searches that score their neighbours exactly (no table) with search_list and
// pre-buffer <= 256.  The search is a chain of dependent round trips (82 % of the wave-cycles of the four-wave form were waits,
// profiles/r04_beam_search_pmc.txt) and three of its four waves idle through the sequential parts; one wave per query puts 16
// queries on a CU instead of 3.
// ADC: neighbours scored through the query's distance table (the reference's default); false: exactly (disable_pq).  Two kernels, so
// that each keeps only its own scoring code and its own register budget (the ADC form runs two workgroups per CU whatever it uses:
// its fetched rows' exact scores take all 36 loads of a row in flight at once -- one round trip instead of three, round 6).
template <int THREADS, bool ADC>
__global__ __launch_bounds__(THREADS, THREADS == 64 ? 4 : 1) void beam_search_kernel(BeamArgs a) {
    constexpr bool EXACT = !ADC;   // the call's disable_pq
    constexpr int BS_THREADS = THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lut_bytes = EXACT ? 0 : 65536;   // the distance table is only needed when neighbours are scored by ADC
    float* s_lut = reinterpret_cast<float*>(smem);
    // ADC-scored searches (64 KiB of table per query) score only the few FETCHED nodes exactly: their query stays in global memory
    // (L2-resident, four rows per iteration read it), so that two workgroups fit a CU up to search lists of ~760 (round 6; with the
    // query in LDS and 4-byte visited flags the second workgroup was lost above L = 480: 143 k queries/s at L = 400 against 55 k at 600)
    const int q_bytes = EXACT ? ((a.d * 2 + 15) & ~15) : 0;
    const uint16_t* const s_q = reinterpret_cast<const uint16_t*>(smem + lut_bytes);   // exact scoring: the query in LDS
    const uint16_t* const g_q = a.queries + (size_t)blockIdx.x * a.d;                   // ADC scoring: the query where it lies
    // (two call sites per use, so that each inlined copy of the dot product knows its address space: LDS reads stay ds_read)
    char* p = smem + lut_bytes + q_bytes;
    // the list is sized by this call's search_list and the pre-buffer by its beam width, so that the usual settings
    // (L = 200, beam 4) leave room for two workgroups per CU next to their 64 KiB tables, eight without tables
    const size_t l_cap = (size_t)a.L, p_cap = (size_t)a.p_cap;
    long long* nb_sc = reinterpret_cast<long long*>(p); p += l_cap * 8;
    long long* pre_sc = reinterpret_cast<long long*>(p); p += p_cap * 8;
    uint32_t* nb_id = reinterpret_cast<uint32_t*>(p); p += l_cap * 4;
    uint8_t* nb_vis = reinterpret_cast<uint8_t*>(p); p += (l_cap + 3) & ~(size_t)3;   // one byte per entry
    uint32_t* pre_id = reinterpret_cast<uint32_t*>(p); p += p_cap * 4;
    int* s_rank = reinterpret_cast<int*>(p); p += p_cap * 4;
    uint32_t* s_hash = reinterpret_cast<uint32_t*>(p);   // [hash_slots]: first positions of the ids of one beam iteration's lists
    __shared__ int s_len, s_next, s_npts, s_npre, s_abort, s_nlive;
    __shared__ uint32_t s_pts[BS_BEAM_MAX];
    __shared__ int s_seg[BS_BEAM_MAX];
    __shared__ int s_visok[BS_BEAM_MAX];
    __shared__ long long s_ptsc[BS_BEAM_MAX];
    __shared__ float s_scales[BS_DESC_MAX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t qi = blockIdx.x;
    const int cap = a.L;
    uint32_t* bm_adj = a.bm_adj + qi * a.bm_words;
    uint32_t* bm_vis = a.bm_vis + qi * a.bm_words;
    const bool use_bias = a.scales && a.desc && a.n_desc > 0;

    if (!EXACT)
        for (int e = tid; e < 64 * 256 / 4; e += BS_THREADS)
            reinterpret_cast<float4*>(s_lut)[e] = reinterpret_cast<const float4*>(a.luts + qi * 16384)[e];
    if (EXACT)
        for (int e = tid; e < a.d / 8; e += BS_THREADS)
            reinterpret_cast<uint4*>(smem + lut_bytes)[e] = reinterpret_cast<const uint4*>(a.queries + qi * a.d)[e];
    if (tid < BS_DESC_MAX) s_scales[tid] = (use_bias && tid < a.n_desc) ? a.scales[qi * a.n_desc + tid] : 0.0f;
    uint32_t start_by_entry = 0;
    if (a.entry_psc && wave == 0) {   // the best chunk of the entry step: larger score, lower row on ties
        long long eb = (long long)INT64_MIN;
        uint32_t er = 0xffffffffu;
        for (int c = lane; c < a.entry_chunks; c += 64) {
            const long long sc = a.entry_psc[(size_t)c * a.entry_nq + qi];
            const uint32_t rw = a.entry_prow[(size_t)c * a.entry_nq + qi];
            if (rw != 0xffffffffu && (er == 0xffffffffu || sc > eb || (sc == eb && rw < er))) { eb = sc; er = rw; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const long long sc = __shfl_xor(eb, o);
            const uint32_t rw = __shfl_xor(er, o);
            if (rw != 0xffffffffu && (er == 0xffffffffu || sc > eb || (sc == eb && rw < er))) { eb = sc; er = rw; }
        }
        start_by_entry = a.entry_ids[er == 0xffffffffu ? 0 : er];
    }
    if (tid == 0) {
        const uint32_t start = a.entry_psc ? start_by_entry : a.starts[qi];
        nb_id[0] = start; nb_sc[0] = 0; nb_vis[0] = 0;   // :153 -- the entry point enters with score 0
        s_len = 1; s_next = 0; s_abort = 0;
        (void)visited_insert(bm_adj, a.hash_bits, start);   // :154
    }
    __syncthreads();

    auto bias = [&](uint32_t id) -> long long {   // descriptor_product (:135-142)
        long long r = 0;
        if (use_bias)
            for (int j = 0; j < a.n_desc; j++) r += scale_dot_result(s_scales[j] * (float)a.desc[(size_t)id * a.n_desc + j]);
        return r;
    };

    uint32_t cmps = 0, pq_cmps = 0, n_vis = 0;   // meaningful in thread 0
    uint32_t n_iter = 0, n_replayed = 0;         // measurement only (a.totals): beam iterations, and those that took the sequential insert path
#ifdef MSE_BEAM_PHASES   // probe build (`make phases`, scripts/beam_phase_probe.py): 100 MHz wall-clock ticks per phase of an iteration, thread 0
    unsigned long long ph_t = 0, ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PHASE_STAMP(K) do { if (a.totals && tid == 0) { const unsigned long long now_ = wall_clock64(); ph_acc[K] += now_ - ph_t; ph_t = now_; } } while (0)
#else
#define PHASE_STAMP(K) do { } while (0)
#endif
    uint32_t n_adj = 1;                          // wave 0: ids inserted into visited_adjacent so far
    bool ties = false;                           // this iteration: equal scores inside the list, or met during the replay (see the insert loop)
    for (;;) {
#ifdef MSE_BEAM_PHASES
        if (a.totals && tid == 0) ph_t = wall_clock64();
#endif
        // ---- next_several_unvisited (:83-97 over NeighbourBuffer::next_unvisited, lib.rs:93-107) ----
        // The first `beam` unvisited entries from next_unvisited on, in list order, and the one after them as the new next_unvisited:
        // 64 flags per step by wave 0 (round 6: one thread walking the flags cost 2-4 us of every iteration -- each step an LDS round trip).
        // With exactly scored neighbours an entry's list score IS its exact score + bias (same dot product, same rows), so the
        // record of the visited list needs no second gather of the row -- except for the entry point, which enters with score 0 (:153).
        if (wave == 0) {
            int n = 0, nu = -1, g0 = s_next;
            const int len = s_len;
            bool done = g0 < 0;
            while (!done) {
                const int idx = g0 + lane;
                unsigned long long m = __ballot(idx < len && !nb_vis[idx]);
                while (m) {
                    const int at = g0 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (n < a.beam) {
                        if (lane == 0) { s_pts[n] = nb_id[at]; s_ptsc[n] = nb_sc[at]; nb_vis[at] = 1; }
                        n++;
                    } else {
                        nu = at;
                        done = true;
                        break;
                    }
                }
                if (!done) { g0 += 64; done = g0 >= len; }
            }
            if (lane == 0) { s_next = nu; s_npts = n; }
        }
        __syncthreads();
        PHASE_STAMP(1);
        const int npts = s_npts;
        if (npts == 0 || s_abort) break;

        // ---- everything that needs only the fetched nodes' ids, in flight together (round 4; one node after the other -- degree, then
        // its list, then the set inserts, per node -- was 16 dependent round trips per beam of four):
        //   the nodes' adjacency lists, concatenated node-major into LDS (degree and list entries are independent loads);
        //   their exact scores + bias (:168-170), one lane quad per node;
        //   their own entries in `visited` (HashSet::insert: the first of equal ids wins).
        const int md = a.max_deg, ncat = npts * md;   // <= p_cap
        uint32_t* const s_cat = reinterpret_cast<uint32_t*>(pre_sc);   // [p_cap] neighbour ids (pre_sc is not live here), 0xffffffff = none
        uint32_t* const s_fresh = s_cat + p_cap;                        // [p_cap] 1 = goes into the pre-buffer
        for (int e = tid; e < a.hash_slots; e += BS_THREADS) s_hash[e] = 0xffffffffu;
        // ADC-scored searches: the adjacency entries wait in registers while wave 0 gathers the fetched rows, so that both round trips
        // are in flight together (the latency-bound form: 6.5 us of a 17 us iteration were these two, one after the other)
        constexpr int ADJ_PER = ADC ? (BS_BEAM_MAX * 64 + BS_THREADS - 1) / BS_THREADS : 1;
        [[maybe_unused]] uint32_t nb_r[ADJ_PER], dg_r[ADJ_PER];
        if constexpr (ADC) {
#pragma unroll
            for (int i = 0; i < ADJ_PER; i++) {
                const int e = tid + i * BS_THREADS;
                nb_r[i] = 0xffffffffu; dg_r[i] = 0u;
                if (e < ncat) {
                    const int j = e / md, pos = e - j * md;
                    const uint32_t pt = s_pts[j];
                    dg_r[i] = a.deg[pt];
                    nb_r[i] = a.adj[(size_t)pt * md + pos];
                }
            }
        } else {
            for (int e = tid; e < ncat; e += BS_THREADS) {
                const int j = e / md, pos = e - j * md;
                const uint32_t pt = s_pts[j];
                const uint32_t dg = a.deg[pt];
                uint32_t nb = a.adj[(size_t)pt * md + pos];
                if ((uint32_t)pos >= dg) nb = 0xffffffffu;
                else if (nb >= a.n) { nb = 0xffffffffu; atomicOr(a.err, 1u); }
                s_cat[e] = nb;
            }
        }
        if (tid < npts) {
            const uint32_t pt = s_pts[tid];
            bool first = true;
            for (int i = 0; i < tid; i++) first &= s_pts[i] != pt;
            const bool url = !a.has_url || a.has_url[pt];
            s_visok[tid] = (first && visited_insert(bm_vis, a.hash_bits, pt) && url) ? 1 : 0;
        }
        if (wave == 0 && (!EXACT || n_iter == 0)) {   // (exactly scored searches: s_ptsc was taken from the list above)
            const int qd = lane >> 2;
            const uint32_t pt = s_pts[qd < npts ? qd : npts - 1];
            const float f = EXACT ? quad_fast_dot_f32(a.base + (size_t)pt * a.d, s_q, a.d) : quad_fast_dot_f32<18>(a.base + (size_t)pt * a.d, g_q, a.d);
            if (qd < npts && (lane & 3) == 0) s_ptsc[qd] = scale_dot_result(f) + bias(pt);
        }
        if constexpr (ADC) {
#pragma unroll
            for (int i = 0; i < ADJ_PER; i++) {
                const int e = tid + i * BS_THREADS;
                if (e < ncat) {
                    const int pos = e - (e / md) * md;
                    uint32_t nb = nb_r[i];
                    if ((uint32_t)pos >= dg_r[i]) nb = 0xffffffffu;
                    else if (nb >= a.n) { nb = 0xffffffffu; atomicOr(a.err, 1u); }
                    s_cat[e] = nb;
                }
            }
        }
        __syncthreads();
        PHASE_STAMP(2);

        // ---- fresh neighbours (:171-188).  The reference walks the nodes in fetch order and offers every neighbour to
        // `visited_adjacent`: an id enters the pre-buffer at its FIRST position in the concatenated lists, if the set did not hold it
        // before.  First positions by a small open-addressing table in LDS (slot = smallest position seen for an id) ...
        const uint32_t hmask = (uint32_t)a.hash_slots - 1u;
        for (int e = tid; e < ncat; e += BS_THREADS) {
            const uint32_t id = s_cat[e];
            if (id == 0xffffffffu) continue;
            uint32_t h = (id * 2654435761u) & hmask;
            for (;;) {
                const uint32_t old = atomicCAS(&s_hash[h], 0xffffffffu, (uint32_t)e);
                if (old == 0xffffffffu) break;
                if (s_cat[old] == id) { atomicMin(&s_hash[h], (uint32_t)e); break; }
                h = (h + 1) & hmask;
            }
        }
        __syncthreads();
        PHASE_STAMP(3);
        // ... then ONE round of set inserts for the first positions, all lanes at once
        for (int e = tid; e < ncat; e += BS_THREADS) {
            const uint32_t id = s_cat[e];
            uint32_t fresh = 0;
            if (id != 0xffffffffu) {
                uint32_t h = (id * 2654435761u) & hmask;
                uint32_t v = s_hash[h];
                while (v != 0xffffffffu && s_cat[v] != id) { h = (h + 1) & hmask; v = s_hash[h]; }   // (every id listed was entered above)
                if (v == (uint32_t)e) fresh = visited_insert(bm_adj, a.hash_bits, id) ? 1u : 0u;
            }
            s_fresh[e] = fresh;
        }
        __syncthreads();
        PHASE_STAMP(4);
        // ... and the pre-buffer in list order; s_seg[j] = entries up to and including node j's; the visited list in fetch order
        if (wave == 0) {
            int npre = 0, jb = 0;
            for (int g0 = 0; g0 < ncat; g0 += 64) {
                const int e = g0 + lane;
                const bool f = e < ncat && s_fresh[e];
                const unsigned long long m = __ballot(f);
                if (f) pre_id[npre + __popcll(m & ((1ull << lane) - 1ull))] = s_cat[e];
                while (jb < npts && (jb + 1) * md <= g0 + 64) {   // node jb's list ends inside this group of 64
                    const int cut = (jb + 1) * md - g0;           // 1..64 entries of the group belong to nodes <= jb
                    if (lane == 0) s_seg[jb] = npre + __popcll(cut >= 64 ? m : (m & ((1ull << cut) - 1ull)));
                    jb++;
                }
                npre += __popcll(m);
            }
            if (lane == 0) {
                s_npre = npre;
                s_nlive = 0;
                for (int j = 0; j < npts; j++) {
                    cmps++;
                    if (s_visok[j]) {
                        if (n_vis < a.vis_cap) {
                            a.vis_ids[qi * a.vis_cap + n_vis] = s_pts[j];
                            a.vis_scores[qi * a.vis_cap + n_vis] = s_ptsc[j];
                        }
                        n_vis++;
                    }
                }
            }
            n_adj += (uint32_t)npre;
            if (a.hash_bits && n_adj > (1u << (a.hash_bits - 1)) && lane == 0) {   // table half full: give up, the host repeats with bit maps
                atomicOr(a.err, 4u);
                s_abort = 1;
            }
        }
        __syncthreads();
        PHASE_STAMP(5);

        // ---- scores of the pre-buffer (:189-203): ADC + bias, or exact + bias with disable_pq ----
        const int npre = s_npre;
        if constexpr (ADC) {
            for (int e = tid; e < npre; e += BS_THREADS) {
                const uint32_t id = pre_id[e];
                const uint4* cp = reinterpret_cast<const uint4*>(a.codes + (size_t)id * 64);
                float s = 0.0f;
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) {
                    const uint4 w4 = cp[c4];
                    const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int x = 0; x < 4; x++)
#pragma unroll
                        for (int bb = 0; bb < 4; bb++)
                            s = add_rn(s, s_lut[(c4 * 16 + x * 4 + bb) * 256 + ((w[x] >> (8 * bb)) & 0xff)]);
                }
                pre_sc[e] = scale_dot_result(s) + bias(id);
            }
        } else {
            for (int e0 = 0; e0 < npre; e0 += BS_THREADS / 4) {
                const int e = e0 + (tid >> 2);
                const uint32_t id = pre_id[e < npre ? e : npre - 1];
                const float f = quad_fast_dot_f32(a.base + (size_t)id * a.d, s_q, a.d);
                if (e < npre && (tid & 3) == 0) pre_sc[e] = scale_dot_result(f) + bias(id);
            }
        }
        __syncthreads();
        PHASE_STAMP(6);

        // ---- all newcomers of this beam iteration at once ----
        // While no score that can still enter the list equals another one in play (see `live` below), the
        // order of the inserts does not matter and re-offers change nothing: the list ends up as the best `cap` of old and
        // new entries, and next_unvisited as the smaller of its old value and the slot the best newcomer takes on arrival
        // (every other insert lands at or behind that slot).  Each thread places up to four newcomers by two counts -- old
        // entries above it (binary search) and newcomers above it -- and moves up to four old entries up by the number of
        // newcomers that go before them.  Any equality, and the reference's sequence is replayed below instead.
        bool merged = false;
        if (cap > 0 && npre > 0) {
            const int len = s_len;
            // A newcomer below the worst entry of a FULL list is rejected whatever the order of the inserts (the worst entry only ever
            // rises): equal scores among such newcomers decide nothing.  Only a LIVE newcomer's equalities -- with a list entry or with
            // another live newcomer -- make the order matter.  (With PQ codes this is the common case: neighbours far from the query that
            // share their 64 code bytes have the same ADC score to the last bit.)
            const bool full = len == cap;
            const long long worst0 = full ? nb_sc[len - 1] : (long long)INT64_MIN;
            // Round 6: only the live newcomers are ranked.  Once the list is full -- after the first two or three iterations -- a handful
            // of an iteration's ~150 newcomers can still enter it; ranking every newcomer against every other one was ~1400 LDS reads per
            // wave and iteration (a quarter of the one-wave kernel's time, a third of the four-wave kernel's).  The live ones' scores
            // are compacted into the (idle) first-position table; their order there does not matter: ranks of distinct scores.
            long long* const s_lsc = reinterpret_cast<long long*>(s_hash);   // [<= p_cap] (hash_slots >= 2 p_cap words)
            int lo_[4], rn_[4], slot_[4];
            long long sc_[4];
            bool live_[4];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const int e = tid + h * BS_THREADS;
                lo_[h] = 0; rn_[h] = 0; sc_[h] = 0; slot_[h] = 0; live_[h] = false;
                if (e < npre) {
                    const long long sc = pre_sc[e];
                    sc_[h] = sc;
                    live_[h] = !(full && sc < worst0);
                    if (live_[h]) {
                        slot_[h] = atomicAdd(&s_nlive, 1);
                        s_lsc[slot_[h]] = sc;
                    }
                }
            }
            __syncthreads();
            const int nlive = s_nlive;
            bool tie = false;
#pragma unroll
            for (int h = 0; h < 4; h++) {
                if (live_[h]) {
                    const long long sc = sc_[h];
                    int lo = 0, hi = len;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (nb_sc[mid] > sc) lo = mid + 1; else hi = mid;
                    }
                    tie |= lo < len && nb_sc[lo] == sc;
                    int rn = 0;
                    for (int k = 0; k < nlive; k++) {
                        const long long sk = s_lsc[k];
                        rn += sk > sc ? 1 : 0;
                        tie |= k != slot_[h] && sk == sc;
                    }
                    lo_[h] = lo; rn_[h] = rn;
                }
            }
            // Equal scores that are already INSIDE the list (two ids that tied in an earlier iteration, or a duplicate the re-offer quirk
            // made) do not matter either: the reference's next state depends on the list as it is and on this iteration's offers, and an
            // offer whose own score is unique finds its slot, or its own copy, whatever sits elsewhere in the list.  (Round 6.  The flag
            // used to stick for the rest of the search: with f32 scores on a 2^-24 grid ~40 % of the hard set's searches at L = 200 meet
            // some tie among the ~10 000 neighbours they score, almost always between candidates that never enter the list.)
            const int any_tie = __syncthreads_or(tie ? 1 : 0);
            ties = false;   // what the replay below starts from: no equality has been met in THIS iteration's sequence yet
            if (!any_tie) {
                merged = true;
#pragma unroll
                for (int h = 0; h < 4; h++)
                    if (live_[h]) s_rank[rn_[h]] = lo_[h];
                __syncthreads();
                long long osc[4];
                uint32_t oid[4], ovis[4];
                int onp[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int i = tid + c * BS_THREADS;
                    onp[c] = -1;
                    if (i < len && nlive > 0) {
                        osc[c] = nb_sc[i]; oid[c] = nb_id[i]; ovis[c] = nb_vis[i];
                        int a0 = 0, b0 = nlive;
                        while (a0 < b0) {
                            const int mid = (a0 + b0) >> 1;
                            if (s_rank[mid] <= i) a0 = mid + 1; else b0 = mid;
                        }
                        if (a0 > 0 && i + a0 < cap) onp[c] = i + a0;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (onp[c] >= 0) { nb_sc[onp[c]] = osc[c]; nb_id[onp[c]] = oid[c]; nb_vis[onp[c]] = (uint8_t)ovis[c]; }
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const int e = tid + h * BS_THREADS, pos = lo_[h] + rn_[h];
                    if (live_[h] && pos < cap) { nb_sc[pos] = sc_[h]; nb_id[pos] = pre_id[e]; nb_vis[pos] = 0; }
                }
                if (tid == 0) {
                    if (nlive > 0) {
                        const int first = s_rank[0], nu = s_next;
                        if (first < cap && (nu < 0 || first < nu)) s_next = first;
                    }
                    s_len = len + nlive < cap ? len + nlive : cap;
                    if (!EXACT)
                        for (int j = 0; j < npts; j++) pq_cmps += (uint32_t)s_seg[j];   // every offer counts, re-offers included (:205)
                }
            }
        }

        n_iter++;
        n_replayed += (!merged && cap > 0 && npre > 0) ? 1u : 0u;
        // ---- NeighbourBuffer::insert (lib.rs:117-147) for every (node, pre-buffer entry) pair in the reference's order ----
        if (wave == 0 && !merged) {
            int len = s_len, nu = s_next;
            for (int j = 0; j < npts; j++) {
                const int upto = s_seg[j];
                const int fresh_from = j ? s_seg[j - 1] : 0;   // entries below this index were already offered by an earlier node
                if (!EXACT) pq_cmps += (uint32_t)upto;   // every offer counts, re-offers and rejected ones included (:205)
                if (cap == 0) continue;
                // 64 offers at a time: the ones a full list rejects outright (score below its worst entry -- which only rises, so they are
                // rejected at their turn too) are dropped together; the rest take their turn in the reference's order
                for (int i0 = 0; i0 < upto; i0 += 64) {
                    const int il = i0 + lane;
                    const long long sc_l = il < upto ? pre_sc[il] : 0;
                    const uint32_t id_l = il < upto ? pre_id[il] : 0u;
                    const long long worst_now = len == cap ? nb_sc[len - 1] : (long long)INT64_MIN;
                    unsigned long long todo = __ballot(il < upto && !(len == cap && worst_now > sc_l));
                    while (todo) {
                    const int bit = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int ii = i0 + bit;
                    const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)id_l, bit);
                    const long long sc = (long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)sc_l >> 32), bit) << 32) |
                                                     (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(unsigned long long)sc_l, bit));
                    // Re-offering an entry (the pre-buffer quirk) changes nothing while no equal scores have been met in this
                    // iteration's sequence: it is either still there (the search lands on it: same id, skipped), or it was rejected /
                    // pushed out by strictly better entries and is rejected again.  Only once two different ids have tied on a score
                    // can a re-offer land next to its copy and duplicate it, so from then on every offer is replayed in full.
                    if (ii < fresh_from && !ties) continue;
                    if (len == cap && nb_sc[len - 1] > sc) continue;
                    int loc = 0;
                    // Position by counting with all 64 lanes: the list is sorted, so with no equal score the insertion point is
                    // the number of larger entries, and with exactly one equal score binary_search_by can only land on it.
                    // Only when several entries tie with the new score does the landing slot depend on the probe sequence,
                    // and then the reference's loop is replayed below.
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
                    if (n_eq >= 2 || (n_eq == 1 && nb_id[eq_pos] != id)) ties = true;
                    if (n_eq == 0) {
                        loc = n_gt;
                    } else if (n_eq == 1) {
                        loc = eq_pos;
                    } else if (len > 0) {   // binary_search_by over the descending scores (lib.rs:122-125)
                        int size = len, base = 0;
                        while (size > 1) {
                            const int half = size / 2, mid = base + half;
                            base = (sc > nb_sc[mid]) ? base : mid;
                            size -= half;
                        }
                        const long long c = nb_sc[base];
                        loc = (sc == c) ? base : base + (sc < c ? 1 : 0);
                    }
                    if (loc < len && nb_id[loc] == id) continue;
                    const int newlen = len < cap ? len + 1 : cap;
                    for (int top = newlen - 1; top > loc; top -= 64) {   // shift [loc, newlen-1) up by one, 64 entries at a time
                        const int idx = top - lane;
                        const bool act = idx > loc;
                        uint32_t mi = 0, mv = 0;
                        long long ms = 0;
                        if (act) { mi = nb_id[idx - 1]; ms = nb_sc[idx - 1]; mv = nb_vis[idx - 1]; }
                        if (act) { nb_id[idx] = mi; nb_sc[idx] = ms; nb_vis[idx] = (uint8_t)mv; }
                    }
                    if (lane == 0) { nb_id[loc] = id; nb_sc[loc] = sc; nb_vis[loc] = 0; }
                    len = newlen;
                    if (nu < 0 || loc < nu) nu = loc;
                    }   // offers of this group of 64 that a full list does not reject outright
                }
            }
            if (lane == 0) { s_len = len; s_next = nu; }
        }
        __syncthreads();
        PHASE_STAMP(7);
    }

    const int len = s_len;
    for (int e = tid; e < len; e += BS_THREADS) {
        a.out_ids[qi * a.L + e] = nb_id[e];
        a.out_scores[qi * a.L + e] = nb_sc[e];
    }
    if (a.fill_vis) {
        if (tid == 0) s_npre = (int)(n_vis < a.vis_cap ? n_vis : a.vis_cap);
        __syncthreads();
        for (size_t e = (size_t)s_npre + tid; e < a.vis_cap; e += BS_THREADS) {
            a.vis_ids[qi * a.vis_cap + e] = 0xffffffffu;
            a.vis_scores[qi * a.vis_cap + e] = (long long)INT64_MIN;
        }
    }
    if (tid == 0) {
        a.out_len[qi] = (uint32_t)len;
        a.n_visited[qi] = n_vis;
        a.cmps[qi] = cmps;
        a.pq_cmps[qi] = pq_cmps;
        if (a.totals) {   // measurement only: what this search gathered (n_adj - 1 = neighbours that entered a pre-buffer)
            // rows gathered for an exact score: every neighbour that entered a pre-buffer (exact scoring) -- a fetched node's record then
            // takes its score from the list, only the entry point is gathered on its own --, or every fetched node (ADC scoring)
            atomicAdd(&a.totals[0], EXACT ? (unsigned long long)n_adj : (unsigned long long)cmps);
            atomicAdd(&a.totals[1], (unsigned long long)cmps);
            atomicAdd(&a.totals[2], EXACT ? 0ull : (unsigned long long)(n_adj - 1));
            atomicAdd(&a.totals[3], (unsigned long long)n_iter);
            atomicAdd(&a.totals[4], (unsigned long long)n_replayed);
#ifdef MSE_BEAM_PHASES
            for (int k_ = 1; k_ < 8; k_++) atomicAdd(&a.totals[4 + k_], ph_acc[k_]);
#endif
        }
    }
}

// fused request path: copies of the entry records' vectors; start node of a query = node id of its best entry row
__global__ void gather_entry_rows_kernel(const uint16_t* __restrict__ base, int d, const uint32_t* __restrict__ ids, uint16_t* __restrict__ out) {
    const uint4* src = reinterpret_cast<const uint4*>(base + (size_t)ids[blockIdx.x] * d);
    uint4* dst = reinterpret_cast<uint4*>(out + (size_t)blockIdx.x * d);
    for (int e = threadIdx.x; e < d / 8; e += blockDim.x) dst[e] = src[e];
}
__global__ void entry_starts_kernel(const uint32_t* __restrict__ best_row, const uint32_t* __restrict__ entry_ids, size_t n_entries, size_t nq,
                                    uint32_t* __restrict__ starts) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t r = best_row[q];
    starts[q] = entry_ids[r < n_entries ? r : 0];
}

// Entry step of a row table for a SMALL batch (round 5): exact top-1 of every query over the entry rows in two launches.  The
// brute-force searcher's matrix-core path (scan, tournament, re-score, certificate: a dozen launches and a host synchronisation for
// the margins) is the right tool for thousands of queries; for the few dozen queries of a coalesced submission its fixed cost was
// most of the call.  A workgroup takes 8 queries (their f16 copies in LDS) and a range of rows; a lane quad owns a row at a time,
// loads it ONCE and runs the reference's fast_dot chain (exact_dot.h: accumulator `part`, t ascending, the fixed reduction tree)
// against all 8 queries; per-quad best (score, row) -> per-workgroup best -> partial[chunk][query]; the search kernel's first wave picks
// the best chunk (BeamArgs::entry_psc: a launch of its own for that cost ~35 us of a ~280 us pass).  Same answer as the searcher's exact top-1: i64 scores in the reference's order, ties by the lower row.
constexpr int ET_Q = 8, ET_THREADS = 256;
__global__ __launch_bounds__(ET_THREADS) void entry_top1_rows_kernel(const uint16_t* __restrict__ rows, int n_rows, int d, const uint16_t* __restrict__ queries,
                                                                     int nq, int rows_per_wg, long long* __restrict__ part_sc, uint32_t* __restrict__ part_row) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* s_q = reinterpret_cast<uint16_t*>(smem);            // [ET_Q][d]
    __shared__ long long s_best[ET_THREADS / 4][ET_Q];
    __shared__ uint32_t s_brow[ET_THREADS / 4][ET_Q];
    const int tid = threadIdx.x, part = tid & 3, quad = tid >> 2;
    const int q0 = blockIdx.y * ET_Q, nqt = min(ET_Q, nq - q0);
    const int d8 = d / 8;
    for (int e = tid; e < ET_Q * d8; e += ET_THREADS) {
        const int j = e / d8, c = e - j * d8;
        reinterpret_cast<uint4*>(s_q)[e] = j < nqt ? reinterpret_cast<const uint4*>(queries + (size_t)(q0 + j) * d)[c] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    long long best[ET_Q];
    uint32_t brow[ET_Q];
#pragma unroll
    for (int j = 0; j < ET_Q; j++) { best[j] = (long long)INT64_MIN; brow[j] = 0xffffffffu; }
    const int r_begin = blockIdx.x * rows_per_wg, r_end = min(n_rows, r_begin + rows_per_wg);
    const int T = d / 32;
    for (int r0 = r_begin; r0 < r_end; r0 += ET_THREADS / 4) {
        const int r = r0 + quad;
        const int rr = r < r_end ? r : r_end - 1;                  // clamped: all four lanes of every quad stay in the exchanges
        const uint4* xp = reinterpret_cast<const uint4*>(rows + (size_t)rr * d) + part;
        float acc[ET_Q][8];
#pragma unroll
        for (int j = 0; j < ET_Q; j++)
#pragma unroll
            for (int l = 0; l < 8; l++) acc[j][l] = 0.0f;
        uint4 x = xp[0];
        for (int t = 0; t < T; t++) {
            const uint4 xn = xp[(t + 1 < T ? t + 1 : t) * 4];
#pragma unroll
            for (int j = 0; j < ET_Q; j++) quad_fma8(acc[j], x, reinterpret_cast<const uint4*>(s_q + (size_t)j * d)[t * 4 + part]);
            x = xn;
        }
#pragma unroll
        for (int j = 0; j < ET_Q; j++) {
            float v[8];
#pragma unroll
            for (int l = 0; l < 8; l++) v[l] = add_rn(acc[j][l], __shfl_xor(acc[j][l], 1));
            const float p0 = add_rn(v[0], v[1]), p1 = add_rn(v[2], v[3]);
            const float p2 = add_rn(v[4], v[5]), p3 = add_rn(v[6], v[7]);
            const float first = add_rn(p0, p2), second = add_rn(p1, p3);
            const float of = __shfl_xor(first, 2), os = __shfl_xor(second, 2);
            const bool low = (part & 2) == 0;
            const float s0 = low ? first : of, s1 = low ? second : os, s2 = low ? of : first, s3 = low ? os : second;
            const long long sc = scale_dot_result(add_rn(add_rn(add_rn(s0, s1), s2), s3));
            if (r < r_end && (sc > best[j] || (sc == best[j] && (uint32_t)r < brow[j]))) { best[j] = sc; brow[j] = (uint32_t)r; }
        }
    }
    if (part == 0) {
#pragma unroll
        for (int j = 0; j < ET_Q; j++) { s_best[quad][j] = best[j]; s_brow[quad][j] = brow[j]; }
    }
    __syncthreads();
    if (tid < ET_Q) {
        long long b = (long long)INT64_MIN;
        uint32_t br = 0xffffffffu;
        for (int qd = 0; qd < ET_THREADS / 4; qd++) {
            const long long sc = s_best[qd][tid];
            const uint32_t rw = s_brow[qd][tid];
            if (rw != 0xffffffffu && (br == 0xffffffffu || sc > b || (sc == b && rw < br))) { b = sc; br = rw; }
        }
        if (tid < nqt) {
            part_sc[(size_t)blockIdx.x * nq + q0 + tid] = b;
            part_row[(size_t)blockIdx.x * nq + q0 + tid] = br;
        }
    }
}
// The reference's own entry rule (src/query_disk_index.rs:254-256,447-450): the shard whose centroid has the largest
// scale_dot_result_f64(dot(centroid, query)) -- f32 operands, the sum carried in f64 in index order (the oracle's stated order for
// simsimd's f32 dot) -- `position_max_by_key` keeping the LAST maximum; the search starts at that shard's medioid.  One workgroup
// per query, a lane per shard (centroids transposed [d][E] so that the lanes of a wave read consecutive floats); the product of two
// f32 values is exact in f64, so multiply-then-add and a fused multiply-add round alike.
__global__ __launch_bounds__(256) void entry_by_centroid_kernel(const float* __restrict__ keys_t, int n_entries, int d, const float* __restrict__ q32,
                                                                const uint16_t* __restrict__ q16, const uint32_t* __restrict__ entry_ids,
                                                                uint32_t* __restrict__ starts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_q = reinterpret_cast<float*>(smem);
    __shared__ long long s_key[256];
    __shared__ int s_idx[256];
    const int tid = threadIdx.x;
    const size_t q = blockIdx.x;
    for (int e = tid; e < d; e += 256) s_q[e] = q32 ? q32[q * d + e] : __half2float(reinterpret_cast<const __half*>(q16)[q * d + e]);
    __syncthreads();
    long long best = (long long)INT64_MIN;
    int bi = -1;
    for (int e = tid; e < n_entries; e += 256) {
        double acc = 0.0;
        // the sum stays sequential in k (the stated order); 16 loads are in flight ahead of their 16 dependent adds (d % 32 == 0)
        for (int k0 = 0; k0 < d; k0 += 16) {
            float c[16];
#pragma unroll
            for (int u = 0; u < 16; u++) c[u] = keys_t[(size_t)(k0 + u) * n_entries + e];
#pragma unroll
            for (int u = 0; u < 16; u++) acc += (double)c[u] * (double)s_q[k0 + u];
        }
        const long long key = scale_dot_result_f64(acc);
        if (bi < 0 || key >= best) { best = key; bi = e; }   // e ascends within a lane: >= keeps the last maximum
    }
    s_key[tid] = best; s_idx[tid] = bi;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) {
            const long long ok = s_key[tid + w];
            const int oi = s_idx[tid + w];
            if (oi >= 0 && (s_idx[tid] < 0 || ok > s_key[tid] || (ok == s_key[tid] && oi > s_idx[tid]))) { s_key[tid] = ok; s_idx[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) starts[q] = entry_ids[s_idx[0] < 0 ? 0 : s_idx[0]];
}

// what the fused request path (mse_disk_query_topk) adds to a batched search: where the start nodes come from and what travels back
struct FusedQuery {
    const mse_graph* entries = nullptr;   // start node by the graph's entry table (NULL: `starts` from the host)
    mse_searcher* entry_s = nullptr;      // searcher over the entry rows borrowed from the graph's pool for this call (row tables only)
    size_t k = 0;                         // records selected per query on the device (the largest k of the batch)
    // query q's results go to dst[q] (its first dst[q].k records) when dst is given -- the coalesced calls of many threads --
    // otherwise to row q of the contiguous arrays below ([nq][k], [nq])
    const QueryDst* dst = nullptr;
    uint32_t* ids = nullptr;
    int64_t* scores = nullptr;
    uint32_t *n_visited = nullptr, *cmps = nullptr, *pq_cmps = nullptr;
    // a shard's hand-over (mse_disk_query_topk_block): the [nq][k] results stay on the device (dev_sc / dev_ids: the two halves of a packed
    // block) with id_offset added to the ids; only the counters travel to the host (ids / scores / dst are not used then)
    int64_t* dev_sc = nullptr;
    uint32_t* dev_ids = nullptr;
    uint64_t id_offset = 0;
};

// pinned host staging of a searcher (the fused path's ONE download per call; the coalescer's gathered inputs)
int ensure_pin(void** pin, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return 0;
    if (*pin) (void)hipHostFree(*pin);
    *pin = nullptr; *cap = 0;
    const size_t want = std::max<size_t>(2 * bytes, (size_t)1 << 16);
    MSE_HIP_TRY(hipHostMalloc(pin, want, hipHostMallocDefault));
    *cap = want;
    return 0;
}

}  // namespace

extern "C" {

mse_graph* mse_graph_from_host(const uint32_t* adj, const uint32_t* deg, size_t n, size_t max_deg, const uint8_t* has_url) {
    if (!adj || !deg || n == 0 || max_deg == 0) { fail("graph_from_host: bad argument"); return nullptr; }
    mse_graph* g = new (std::nothrow) mse_graph();
    if (!g) { fail("out of host memory"); return nullptr; }
    g->n = n; g->max_deg = max_deg;
    bool ok = hipMalloc((void**)&g->adj, n * max_deg * 4) == hipSuccess && hipMalloc((void**)&g->deg, n * 4) == hipSuccess;
    ok = ok && hipMemcpy(g->adj, adj, n * max_deg * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(g->deg, deg, n * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && has_url) ok = hipMalloc((void**)&g->has_url, n) == hipSuccess && hipMemcpy(g->has_url, has_url, n, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { mse_graph_free(g); fail("graph_from_host: device allocation/copy failed"); return nullptr; }
    return g;
}

void mse_graph_free(mse_graph* g) {
    if (!g) return;
    delete g->co;   // joins its workers; no search may be in flight
    g->co = nullptr;
    g->co_fast.store(nullptr);
    for (mse_graph::WorkerCtx& w : g->co_ctx) {
        if (w.s) mse_searcher_free(w.s);
        if (w.pin) (void)hipHostFree(w.pin);
    }
    g->co_ctx.clear();
    for (mse_searcher* es : g->entry_pool) mse_searcher_free(es);
    g->entry_pool.clear();
    if (g->entry_base) mse_base_free(g->entry_base);
    if (g->entry_rows) (void)hipFree(g->entry_rows);
    if (g->entry_ids) (void)hipFree(g->entry_ids);
    if (g->entry_keys_t) (void)hipFree(g->entry_keys_t);
    if (g->adj) (void)hipFree(g->adj);
    if (g->deg) (void)hipFree(g->deg);
    if (g->has_url) (void)hipFree(g->has_url);
    delete g;
}

static int disk_search_batch_impl(int visited_mode, mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts,
                                  const uint16_t* queries, const float* queries_f32, const float* luts, const float* scales, size_t nq,
                                  int disable_pq, size_t beamwidth, size_t search_list, uint32_t* buf_ids, int64_t* buf_scores,
                                  uint32_t* buf_len, uint32_t* visited_ids, int64_t* visited_scores, size_t visited_cap,
                                  uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps, const FusedQuery* fz = nullptr) {
    // with disable_pq and no descriptor bias neither the codec nor the codes are touched: both may be NULL then
    static const mse_codes no_codes{};
    if (!c && disable_pq && !scales) c = &no_codes;
    const bool codec_needed = !disable_pq;
    // fused request path (fz): the search list and the visited records stay on the device, only the k best visited records travel back
    if (!s || !s->base || (!pq && codec_needed) || !c || !g || (!starts && !(fz && fz->entries)) || (!queries && !queries_f32) ||
        (!luts && !queries_f32 && !disable_pq) || (!fz && (!buf_ids || !buf_scores || !buf_len || !n_visited || !cmps || !pq_cmps)))
        return fail("disk_search_batch: null argument");
    if (nq == 0) return 0;
    const mse_base* b = s->base;
    // visited sets: bit maps, or hash tables once the index is so large that the tables are the smaller ones (visited_set.h)
    const size_t words = (b->n + 31) / 32;
    // adjacency set: a search of list L fetches about L + a few nodes and meets <= max_deg new ids at each (measured: 2 400 inserts at
    // L = 32, R = 64); a search that outgrows its table is caught (half-full check, err bit 4) and the batch repeated with bit maps
    const int table_bits = visited_table_bits(std::min<size_t>(b->n, search_list * 128 + 2048));
    const char* vm = getenv("MSE_VISITED_MODE");   // test hook: "hash" / "bitmap"
    const bool use_hash = visited_mode >= 0 ? visited_mode == 1 : (vm ? !strcmp(vm, "hash") : words > ((size_t)1 << table_bits));
    const size_t set_words = use_hash ? (size_t)1 << table_bits : words;
    {   // two visited sets per query in flight: long batches go through in pieces of at most ~4 GiB of them
        const size_t per_query = set_words * 8, piece = std::max<size_t>(1, std::max(s->pool[4].cap, visited_budget_bytes()) / per_query);   // pool[4]: the bitmaps already held
        if (nq > piece) {
            for (size_t q0 = 0; q0 < nq; q0 += piece) {
                const size_t m = std::min(piece, nq - q0);
                FusedQuery fp;
                if (fz) {
                    fp = *fz;
                    if (fz->dev_sc) { fp.dev_sc = fz->dev_sc + q0 * fz->k; fp.dev_ids = fz->dev_ids + q0 * fz->k; }
                    if (fz->dst) fp.dst = fz->dst + q0;
                    else {
                        fp.ids = fz->ids + q0 * fz->k; fp.scores = fz->scores + q0 * fz->k;
                        fp.n_visited = fz->n_visited ? fz->n_visited + q0 : nullptr;
                        fp.cmps = fz->cmps ? fz->cmps + q0 : nullptr;
                        fp.pq_cmps = fz->pq_cmps ? fz->pq_cmps + q0 : nullptr;
                    }
                }
                const int prc = disk_search_batch_impl(visited_mode, s, pq, c, g, starts ? starts + q0 : nullptr, queries ? queries + q0 * b->d : nullptr,
                                                       queries_f32 ? queries_f32 + q0 * b->d : nullptr, luts ? luts + q0 * 16384 : nullptr,
                                                       scales ? scales + q0 * c->n_desc : nullptr, m, disable_pq, beamwidth, search_list,
                                                       buf_ids ? buf_ids + q0 * search_list : nullptr, buf_scores ? buf_scores + q0 * search_list : nullptr,
                                                       buf_len ? buf_len + q0 : nullptr, visited_ids ? visited_ids + q0 * visited_cap : nullptr,
                                                       visited_scores ? visited_scores + q0 * visited_cap : nullptr, visited_cap,
                                                       n_visited ? n_visited + q0 : nullptr, cmps ? cmps + q0 : nullptr,
                                                       pq_cmps ? pq_cmps + q0 : nullptr, fz ? &fp : nullptr);
                if (prc) return prc;
            }
            return 0;
        }
    }
    if ((c != &no_codes && c->n != b->n) || g->n != b->n) return fail("disk_search_batch: vectors, codes and graph differ in length");
    if (codec_needed && (pq->n_chunks != 64 || pq->n_centroids != 256 || c->code_size != 64)) return fail("disk_search_batch: needs the 64 x 256 codec");
    if (beamwidth == 0 || beamwidth > BS_BEAM_MAX) return fail("disk_search_batch: beamwidth must be 1..8");
    if (search_list == 0 || search_list > BS_LMAX) return fail("disk_search_batch: search_list must be 1..1024");
    if (g->max_deg > BS_DEG_MAX) return fail("disk_search_batch: at most 128 neighbours per node");
    if (c->n_desc > BS_DESC_MAX) return fail("disk_search_batch: at most 8 descriptors");
    if (b->d % 32 || b->d > 4096) return fail("disk_search_batch: vector width must be a multiple of 32");
    if (!fz && visited_cap && (!visited_ids || !visited_scores)) return fail("disk_search_batch: null visited arrays");
    if (fz && (fz->k == 0 || fz->k > visited_cap || fz->k > (size_t)TOPK_KMAX - 64 || (!fz->dst && !fz->dev_sc && (!fz->ids || !fz->scores))))
        return fail("disk_query_topk: bad k / outputs");
    if (fz && fz->entries) {
        const mse_graph* eg = fz->entries;
        const bool by_rows = eg->entry_base && fz->entry_s && eg->entry_base->d == b->d, by_keys = eg->entry_keys_t && eg->entry_keys_d == b->d;
        if (eg->n_entries == 0 || (!by_rows && !by_keys))
            return fail("disk_query_topk: the graph has no entry table for these vectors (mse_graph_set_entries / mse_graph_set_entry_centroids)");
    }
    for (size_t q = 0; starts && q < nq; q++)
        if (starts[q] >= b->n) return fail("disk_search_batch: start node out of range");
    hipStream_t st = s->stream;
    const size_t d = b->d;
    const bool bias = scales && c->n_desc && c->desc;
    DevBuf &dq = s->pool[0], &dl = s->pool[1], &dsc = s->pool[2], &dst = s->pool[3], &bm = s->pool[4], &oi = s->pool[5], &os = s->pool[6],
           &ol = s->pool[7], &vi = s->pool[8], &vs = s->pool[9], &cnt = s->pool[10], &qf = s->pool[11], &qt = s->pool[12], &fzb = s->pool[13];
    // fused path: entry top-1 [nq] (i64, u32) | the block that travels back in ONE copy: k best visited [nq][k] i64 scores, [nq][k] u32
    // ids, counters [3 nq + 1] (n_visited, cmps, pq_cmps, err)
    const size_t fz_block_off = (nq * 12 + 15) & ~(size_t)15;
    const size_t fz_block_bytes = fz ? nq * fz->k * 12 + (3 * nq + 1) * 4 : 0;
    if (fz && (fzb.ensure(fz_block_off + fz_block_bytes + 64) || ensure_pin(&s->pin, &s->pin_cap, fz_block_bytes))) return -1;
    if ((queries_f32 && (qf.ensure(nq * d * 4) || qt.ensure(nq * d * 4))) || dq.ensure(nq * d * 2) || dl.ensure(disable_pq ? 16 : nq * 65536) || dsc.ensure(nq * BS_DESC_MAX * 4 + 16) || dst.ensure(nq * 4) ||
        bm.ensure(nq * set_words * 8) || oi.ensure(nq * search_list * 4) || os.ensure(nq * search_list * 8) || ol.ensure(nq * 4) ||
        vi.ensure(nq * visited_cap * 4 + 16) || vs.ensure(nq * visited_cap * 8 + 16) || cnt.ensure(nq * 12 + 16))
        return -1;
    if (queries_f32) {
        // the caller's side of query_disk_index.rs:475-477 on the device: f16 copy of the query (RNE) for the exact scores,
        // preprocess_query (vector.rs:367-384) for the distance tables -- 64 KiB per query that never cross PCIe
        if (pq && pq->d != d) return fail("disk_search_batch: codec and vectors differ in width");
        MSE_HIP_TRY(hipMemcpyAsync(qf.p, queries_f32, nq * d * 4, hipMemcpyHostToDevice, st));
        if (launch_f32_to_f16(qf.as<float>(), nq * d, dq.as<uint16_t>(), st)) return -1;
        if (!disable_pq) {
            if (launch_pq_transform(pq->transform, (int)d, qf.as<float>(), nq, qt.as<float>(), st)) return -1;
            if (launch_pq_lut_batch(pq->centroids, (int)pq->n_centroids, (int)d, (int)pq->dpc, qt.as<float>(), nq, dl.as<float>(), st)) return -1;
        }
    } else {
        // hipMemcpyDefault: the f16 queries may sit in host memory or already on the device (the text tower's output): the runtime tells by the pointer
        MSE_HIP_TRY(hipMemcpyAsync(dq.p, queries, nq * d * 2, hipMemcpyDefault, st));
        if (!disable_pq) MSE_HIP_TRY(hipMemcpyAsync(dl.p, luts, nq * 65536, hipMemcpyHostToDevice, st));
    }
    if (bias) MSE_HIP_TRY(hipMemcpyAsync(dsc.p, scales, nq * c->n_desc * 4, hipMemcpyHostToDevice, st));
    const long long* entry_psc = nullptr;   // small-batch entry step: per-chunk bests, reduced by the search kernel itself
    const uint32_t* entry_prow = nullptr;
    int entry_chunks = 0;
    if (fz && fz->entries) {
        // the entry step of the request path (src/query_disk_index.rs:254-256,447-450: the medioid of the shard whose centroid is
        // closest to the query) on the device: exact top-1 of the f16 queries over the entry rows, on this search's stream
        const mse_graph* eg = fz->entries;
        if (eg->entry_keys_t) {
            // the reference's rule itself: centroids as keys, f32 query (an f16 query widened exactly), f64 sums, last maximum
            hipLaunchKernelGGL(entry_by_centroid_kernel, dim3((unsigned)nq), dim3(256), d * 4, st, eg->entry_keys_t, (int)eg->n_entries, (int)d,
                               queries_f32 ? qf.as<float>() : nullptr, dq.as<uint16_t>(), eg->entry_ids, dst.as<uint32_t>());
            MSE_HIP_TRY(hipGetLastError());
        } else if (nq * eg->n_entries <= ((size_t)1 << 22) && eg->n_entries <= ((size_t)1 << 20)) {
            // a small batch: exact top-1 over the entry rows in ONE launch (entry_top1_rows_kernel) + the search kernel's prologue
            const size_t E = eg->n_entries, n_qt = (nq + ET_Q - 1) / ET_Q;
            size_t rows_per_wg = (E + std::max<size_t>(1, 512 / n_qt) - 1) / std::max<size_t>(1, 512 / n_qt);
            rows_per_wg = std::max<size_t>(64, (rows_per_wg + 63) / 64 * 64);
            const size_t n_chunks = (E + rows_per_wg - 1) / rows_per_wg;
            DevBuf& ep = s->pool[14];
            if (ep.ensure(n_chunks * nq * 12 + 64)) return -1;
            long long* psc = ep.as<long long>();
            uint32_t* prow = reinterpret_cast<uint32_t*>(ep.as<char>() + n_chunks * nq * 8);
            MSE_DYN_LDS(entry_top1_rows_kernel, ET_Q * d * 2);
            hipLaunchKernelGGL(entry_top1_rows_kernel, dim3((unsigned)n_chunks, (unsigned)n_qt), dim3(ET_THREADS), ET_Q * d * 2, st, eg->entry_rows, (int)E,
                               (int)d, dq.as<uint16_t>(), (int)nq, (int)rows_per_wg, psc, prow);
            MSE_HIP_TRY(hipGetLastError());
            // (the best chunk per query is picked by the search kernel's first wave: no launch of its own)
            entry_psc = psc; entry_prow = prow; entry_chunks = (int)n_chunks;
        } else {
            int64_t* e_sc = fzb.as<int64_t>();
            uint32_t* e_row = reinterpret_cast<uint32_t*>(fzb.as<char>() + nq * 8);
            if (mse_searcher_set_stream(fz->entry_s, st)) return -1;
            if (mse_bruteforce_topk_f16_dev(fz->entry_s, dq.p, nq, 1, MSE_MODE_AUTO, 0, e_sc, e_row)) return -1;
            hipLaunchKernelGGL(entry_starts_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, e_row, eg->entry_ids, eg->n_entries, nq, dst.as<uint32_t>());
            MSE_HIP_TRY(hipGetLastError());
        }
    } else {
        MSE_HIP_TRY(hipMemcpyAsync(dst.p, starts, nq * 4, hipMemcpyHostToDevice, st));
    }
    MSE_HIP_TRY(hipMemsetAsync(bm.p, use_hash ? 0xff : 0, nq * set_words * 8, st));
    // counters: their own buffer, or (fused path) the tail of the block that travels back
    uint32_t* cnt_dev = fz ? reinterpret_cast<uint32_t*>(fzb.as<char>() + fz_block_off + nq * fz->k * 12) : cnt.as<uint32_t>();
    MSE_HIP_TRY(hipMemsetAsync(cnt_dev, 0, (3 * nq + 1) * 4, st));
    const size_t p_cap = beamwidth * ((g->max_deg + 63) / 64 * 64);
    BeamArgs a{};
    a.base = b->dev; a.n = b->n; a.d = (int)d;
    a.codes = c->codes; a.desc = bias ? c->desc : nullptr; a.n_desc = (int)c->n_desc;
    a.adj = g->adj; a.deg = g->deg; a.max_deg = (int)g->max_deg; a.has_url = g->has_url;
    a.queries = dq.as<uint16_t>(); a.luts = dl.as<float>(); a.scales = bias ? dsc.as<float>() : nullptr; a.starts = dst.as<uint32_t>();
    a.beam = (int)beamwidth; a.L = (int)search_list; a.disable_pq = disable_pq; a.p_cap = (int)p_cap;
    a.bm_adj = bm.as<uint32_t>(); a.bm_vis = bm.as<uint32_t>() + nq * set_words; a.bm_words = set_words; a.hash_bits = use_hash ? table_bits : 0;
    a.out_ids = oi.as<uint32_t>(); a.out_scores = os.as<long long>(); a.out_len = ol.as<uint32_t>();
    a.vis_ids = vi.as<uint32_t>(); a.vis_scores = vs.as<long long>(); a.vis_cap = visited_cap;
    a.n_visited = cnt_dev; a.cmps = cnt_dev + nq; a.pq_cmps = cnt_dev + 2 * nq;
    a.err = cnt_dev + 3 * nq;
    a.fill_vis = fz ? 1 : 0;
    a.entry_psc = entry_psc; a.entry_prow = entry_prow; a.entry_chunks = entry_chunks; a.entry_nq = (int)nq;
    a.entry_ids = entry_psc ? fz->entries->entry_ids : nullptr;
    size_t hash_slots = 64;
    while (hash_slots < 2 * p_cap) hash_slots *= 2;
    a.hash_slots = (int)hash_slots;
    // (beam_search_kernel's carving: table | query (exact scoring only) | list scores, pre-buffer scores, list ids, visited flags (one
    // byte each), pre-buffer ids, ranks | first-position table)
    const size_t lds = (disable_pq ? 0 : 65536) + (disable_pq ? ((d * 2 + 15) & ~(size_t)15) : 0) + search_list * 12 + ((search_list + 3) & ~(size_t)3) +
                       p_cap * 16 + hash_slots * 4;
    static const bool wide_only = MSE_DEV_KNOB("MSE_BEAM_FOUR_WAVES");   // developer library: the four-wave form for every search
    // one wave per query once the batch fills the chip on its own (16 queries per CU); a smaller batch is latency-bound, and four waves
    // finish a search sooner (round 5, scripts/beam_latency_probe.py, hard set: 64 queries at L = 12 0.40 ms against 0.98, at L = 200
    // 5.7 against 8.9; from 2048 queries on the two forms are level)
    const bool timed = s->beam_timing && s->bev0 && s->beam_tot.p;
    if (timed) {
        a.totals = s->beam_tot.as<unsigned long long>();
        MSE_HIP_TRY(hipEventRecord(s->bev0, st));
    }
    // A handful of queries (the request handler's one query per call): the search is a chain of dependent round trips and most of the chip
    // idles -- sixteen waves per query gather an iteration's ~200 neighbour rows in one round instead of three (round 6; same answers,
    // counters included: the tiling of the loops is all that changes).  MSE_BEAM_WAVES=4 / 8 / 16 forces a form (answer-preserving hook).
    static const int force_waves = [] { const char* e = getenv("MSE_BEAM_WAVES"); return e ? atoi(e) : 0; }();
    const int small_waves = force_waves ? force_waves : (nq <= BS_SMALL_NQ ? BS_SMALL_WAVES : 4);
    if (disable_pq && search_list <= 256 && p_cap <= 256 && !wide_only && nq > 1024 && !force_waves) {
        hipLaunchKernelGGL((beam_search_kernel<64, false>), dim3((unsigned)nq), dim3(64), lds, st, a);
    } else if (small_waves == 16) {
        if (disable_pq) {
            MSE_DYN_LDS((beam_search_kernel<1024, false>), 160 * 1024 - 1024);
            hipLaunchKernelGGL((beam_search_kernel<1024, false>), dim3((unsigned)nq), dim3(1024), lds, st, a);
        } else {
            MSE_DYN_LDS((beam_search_kernel<1024, true>), 160 * 1024 - 1024);
            hipLaunchKernelGGL((beam_search_kernel<1024, true>), dim3((unsigned)nq), dim3(1024), lds, st, a);
        }
    } else if (small_waves == 8) {
        if (disable_pq) {
            MSE_DYN_LDS((beam_search_kernel<512, false>), 160 * 1024 - 1024);
            hipLaunchKernelGGL((beam_search_kernel<512, false>), dim3((unsigned)nq), dim3(512), lds, st, a);
        } else {
            MSE_DYN_LDS((beam_search_kernel<512, true>), 160 * 1024 - 1024);
            hipLaunchKernelGGL((beam_search_kernel<512, true>), dim3((unsigned)nq), dim3(512), lds, st, a);
        }
    } else {
        if (disable_pq) {
            MSE_DYN_LDS((beam_search_kernel<BS_THREADS_MAX, false>), 160 * 1024 - 1024);
            hipLaunchKernelGGL((beam_search_kernel<BS_THREADS_MAX, false>), dim3((unsigned)nq), dim3(BS_THREADS_MAX), lds, st, a);
        } else {
            MSE_DYN_LDS((beam_search_kernel<BS_THREADS_MAX, true>), 160 * 1024 - 1024);
            hipLaunchKernelGGL((beam_search_kernel<BS_THREADS_MAX, true>), dim3((unsigned)nq), dim3(BS_THREADS_MAX), lds, st, a);
        }
    }
    MSE_HIP_TRY(hipGetLastError());
    if (timed) MSE_HIP_TRY(hipEventRecord(s->bev1, st));
    struct BeamTimed {   // read once the stream has been waited for (every path below does before it returns)
        mse_searcher* s; bool on; size_t nq;
        ~BeamTimed() {
            float ms = 0.0f;
            if (on && hipEventQuery(s->bev1) == hipSuccess && hipEventElapsedTime(&ms, s->bev0, s->bev1) == hipSuccess) {
                s->beam_ms_total += ms; s->beam_launches++; s->beam_queries += nq;
            } else if (on) {
                (void)hipGetLastError();
            }
        }
    } beam_timed{s, timed, nq};
    uint32_t err = 0;
    if (fz) {
        // the server's last step (src/query_disk_index.rs:529-540: the visited records ordered by exact score) cut to its first k, on
        // the device: the kernel has padded every visited list to visited_cap with (ID_NONE, INT64_MIN).  Scores, ids and the counters
        // come back in ONE copy into the searcher's pinned staging (round 5: five pageable copies before).
        int64_t* top_sc = reinterpret_cast<int64_t*>(fzb.as<char>() + fz_block_off);
        uint32_t* top_id = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(top_sc) + nq * fz->k * 8);
        if (g->dedup_threshold > 0.0f) {
            // the handler's runtime de-duplication (:482-527) before its sort: a visited record that resembles an already kept one (dot of
            // the f32-widened vectors above the threshold, visit order) leaves the list -- on the device, for every query of the batch
            DevBuf& db = s->pool[15];
            if (db.ensure(dedup_batch_scratch_bytes(nq, visited_cap))) return -1;
            if (launch_dedup_batch(b->dev, (int)d, vi.as<uint32_t>(), vs.as<long long>(), visited_cap, cnt_dev, nq, g->dedup_threshold, db.p, st)) return -1;
        }
        SelectArgs sa{};
        sa.kind = KEY_I64; sa.list_ids = vi.as<uint32_t>(); sa.list_keys = vs.p; sa.list_stride = visited_cap; sa.n_list = visited_cap;
        sa.k = (int)fz->k; sa.out_ids = top_id; sa.out_keys = top_sc; sa.out_stride = fz->k; sa.nq = (int)nq;
        if (launch_select(sa, st)) return -1;
        const char* blk = static_cast<const char*>(s->pin);
        if (fz->dev_sc) {   // results stay on the device (global ids); the counters come back alone, at their usual place in the staging
            if (launch_block_finish(top_sc, top_id, nq * fz->k, fz->id_offset, fz->dev_sc, fz->dev_ids, st)) return -1;
            MSE_HIP_TRY(hipMemcpyAsync(s->pin ? static_cast<char*>(s->pin) + nq * fz->k * 12 : nullptr, cnt_dev, (3 * nq + 1) * 4, hipMemcpyDeviceToHost, st));
        } else {
            MSE_HIP_TRY(hipMemcpyAsync(s->pin, top_sc, fz_block_bytes, hipMemcpyDeviceToHost, st));
        }
        MSE_HIP_TRY(hipStreamSynchronize(st));
        const int64_t* h_sc = reinterpret_cast<const int64_t*>(blk);
        const uint32_t* h_id = reinterpret_cast<const uint32_t*>(blk + nq * fz->k * 8);
        const uint32_t* h_cnt = reinterpret_cast<const uint32_t*>(blk + nq * fz->k * 12);
        err = h_cnt[3 * nq];
        if (err & 1u) return fail("disk_search_batch: a graph edge points outside the index");
        if (err & 4u)   // a search outgrew its table: the bit maps have room for everything
            return disk_search_batch_impl(0, s, pq, c, g, starts, queries, queries_f32, luts, scales, nq, disable_pq, beamwidth, search_list, buf_ids,
                                          buf_scores, buf_len, visited_ids, visited_scores, visited_cap, n_visited, cmps, pq_cmps, fz);
        // the reference keeps every visited record: a list that outgrew the device arrays means the caller repeats with larger ones
        for (size_t q = 0; q < nq; q++)
            if (h_cnt[q] > visited_cap) return -2;
        for (size_t q = 0; q < nq; q++) {
            QueryDst o = fz->dst ? fz->dst[q]
                                 : QueryDst{fz->ids + q * fz->k, fz->scores + q * fz->k, fz->n_visited ? fz->n_visited + q : nullptr,
                                            fz->cmps ? fz->cmps + q : nullptr, fz->pq_cmps ? fz->pq_cmps + q : nullptr, fz->k};
            // a caller's k records are the first k of the largest k's: the order (score descending, id ascending) is total
            if (!fz->dev_sc) {
                memcpy(o.ids, h_id + q * fz->k, o.k * 4);
                memcpy(o.scores, h_sc + q * fz->k, o.k * 8);
            }
            if (o.n_visited) *o.n_visited = h_cnt[q];
            if (o.cmps) *o.cmps = h_cnt[nq + q];
            if (o.pq_cmps) *o.pq_cmps = h_cnt[2 * nq + q];
        }
        return 0;
    }
    MSE_HIP_TRY(hipMemcpyAsync(buf_ids, oi.p, nq * search_list * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(buf_scores, os.p, nq * search_list * 8, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(buf_len, ol.p, nq * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(n_visited, a.n_visited, nq * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(cmps, a.cmps, nq * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(pq_cmps, a.pq_cmps, nq * 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipMemcpyAsync(&err, a.err, 4, hipMemcpyDeviceToHost, st));
    MSE_HIP_TRY(hipStreamSynchronize(st));
    if (err & 1u) return fail("disk_search_batch: a graph edge points outside the index");
    if (err & 4u)   // a search outgrew its table: the bit maps have room for everything
        return disk_search_batch_impl(0, s, pq, c, g, starts, queries, queries_f32, luts, scales, nq, disable_pq, beamwidth, search_list, buf_ids,
                                      buf_scores, buf_len, visited_ids, visited_scores, visited_cap, n_visited, cmps, pq_cmps, fz);
    if (visited_cap) {   // only the columns any query filled travel back (entries past n_visited[q] are unspecified)
        size_t widest = 0;
        for (size_t q = 0; q < nq; q++) widest = n_visited[q] > widest ? n_visited[q] : widest;
        if (widest > visited_cap) widest = visited_cap;
        if (widest) {
            MSE_HIP_TRY(hipMemcpy2DAsync(visited_ids, visited_cap * 4, vi.p, visited_cap * 4, widest * 4, nq, hipMemcpyDeviceToHost, st));
            MSE_HIP_TRY(hipMemcpy2DAsync(visited_scores, visited_cap * 8, vs.p, visited_cap * 8, widest * 8, nq, hipMemcpyDeviceToHost, st));
            MSE_HIP_TRY(hipStreamSynchronize(st));
        }
    }
    return 0;
}

}  // extern "C"

// ---- one query per call from many threads: the reference's request path --------------------------------------------------------
// query_disk_index serves every HTTP request with ONE greedy_search on its own task / thread (src/query_disk_index.rs:436-540,
// 711-736; the repo's only load test is 1000 one-query requests at concurrency 100, perf_test.py:6-29).  A one-query launch is one
// workgroup on a 256-CU part; T of them from T threads are T launches.  So small calls meet in the graph's coalescer (dispatch.h):
//   * mse_disk_search_batch(_f32) with nq = 1 (round 4): groups that can share a launch -- same vectors, codec, codes, graph, search
//     parameters and kind of inputs -- run as ONE batched search on the searcher of the group's first caller (blocked in its call);
//   * mse_disk_query_topk(_f32) with nq <= 16 from host memory (round 5): the WHOLE request -- entry node, search, top-k of the
//     visited records -- of every waiting caller in one entry step + one launch + one select on a searcher the WORKER owns (so
//     4096 request threads do not need 4096 streams and sets of scratch: a coalesced call only reads its searcher's `base`).
//     Queries are gathered straight into pinned memory, results come back in one copy and are scattered to the callers.
// Every caller gets exactly what its call returns when made alone (the batched kernels treat queries independently; a caller's k
// records are the first k of the largest k's, the order (score descending, id ascending) being total).  Two workers per graph: one
// pass's copies and host side overlap the other's kernels.
namespace {
enum : size_t { REQ_BEAM = 0, REQ_QUERY = 1 };
constexpr size_t FUSED_COALESCE_MAX = 16;   // calls with more queries go straight to the device on the caller's searcher

struct BeamCall {
    mse_searcher* s; mse_pq* pq; const mse_codes* c; const mse_graph* g; const uint32_t* starts;
    const uint16_t* queries; const float* queries_f32; const float* luts; const float* scales;
    int disable_pq; size_t beamwidth, search_list, visited_cap;
    uint32_t* buf_ids; int64_t* buf_scores; uint32_t* buf_len; uint32_t* visited_ids; int64_t* visited_scores;
    uint32_t *n_visited, *cmps, *pq_cmps;
    bool shares_with(const BeamCall& o) const {
        return s->base == o.s->base && pq == o.pq && c == o.c && g == o.g && disable_pq == o.disable_pq && beamwidth == o.beamwidth &&
               search_list == o.search_list && visited_cap == o.visited_cap && (queries != nullptr) == (o.queries != nullptr) &&
               (luts != nullptr) == (o.luts != nullptr) && (scales != nullptr) == (o.scales != nullptr) &&
               (visited_ids != nullptr) == (o.visited_ids != nullptr) && (visited_scores != nullptr) == (o.visited_scores != nullptr);
    }
};
int beam_call_alone(const BeamCall& k) {
    return disk_search_batch_impl(-1, k.s, k.pq, k.c, k.g, k.starts, k.queries, k.queries_f32, k.luts, k.scales, 1, k.disable_pq, k.beamwidth,
                                  k.search_list, k.buf_ids, k.buf_scores, k.buf_len, k.visited_ids, k.visited_scores, k.visited_cap, k.n_visited,
                                  k.cmps, k.pq_cmps);
}
void beam_run_group(std::vector<DispatchReq*>& grp) {
    const BeamCall& lead = *static_cast<const BeamCall*>(grp[0]->aux0);
    (void)hipSetDevice(lead.s->base->device);
    const size_t n = grp.size(), d = lead.s->base->d, L = lead.search_list, vc = lead.visited_cap;
    const size_t n_desc = (lead.scales && lead.c) ? lead.c->n_desc : 0;
    int rc = 0;
    if (n == 1) {
        rc = beam_call_alone(lead);
        grp[0]->rc = rc;
        if (rc) grp[0]->err = mse_last_error();
        return;
    }
    std::vector<uint32_t> starts(n), ids(n * L), len(n), nv(n), cm(n), pc(n), vids(lead.visited_ids ? n * vc : 0);
    std::vector<int64_t> sc(n * L), vsc(lead.visited_scores ? n * vc : 0);
    std::vector<uint16_t> q16(lead.queries ? n * d : 0);
    std::vector<float> q32(lead.queries_f32 ? n * d : 0), luts(lead.luts ? n * 16384 : 0), scl(n_desc ? n * n_desc : 0);
    for (size_t j = 0; j < n; j++) {
        const BeamCall& k = *static_cast<const BeamCall*>(grp[j]->aux0);
        starts[j] = k.starts[0];
        if (k.queries) memcpy(q16.data() + j * d, k.queries, d * 2);
        if (k.queries_f32) memcpy(q32.data() + j * d, k.queries_f32, d * 4);
        if (k.luts) memcpy(luts.data() + j * 16384, k.luts, 16384 * 4);
        if (n_desc) memcpy(scl.data() + j * n_desc, k.scales, n_desc * 4);
    }
    rc = disk_search_batch_impl(-1, lead.s, lead.pq, lead.c, lead.g, starts.data(), lead.queries ? q16.data() : nullptr,
                                lead.queries_f32 ? q32.data() : nullptr, lead.luts ? luts.data() : nullptr, n_desc ? scl.data() : lead.scales, n,
                                lead.disable_pq, lead.beamwidth, L, ids.data(), sc.data(), len.data(), lead.visited_ids ? vids.data() : nullptr,
                                lead.visited_scores ? vsc.data() : nullptr, vc, nv.data(), cm.data(), pc.data());
    for (size_t j = 0; j < n; j++) {
        const BeamCall& k = *static_cast<const BeamCall*>(grp[j]->aux0);
        if (rc) {     // the shared launch failed: each caller is repeated alone and sees only its own outcome
            grp[j]->rc = beam_call_alone(k);
            if (grp[j]->rc) grp[j]->err = mse_last_error();
            continue;
        }
        memcpy(k.buf_ids, ids.data() + j * L, L * 4);
        memcpy(k.buf_scores, sc.data() + j * L, L * 8);
        k.buf_len[0] = len[j]; k.n_visited[0] = nv[j]; k.cmps[0] = cm[j]; k.pq_cmps[0] = pc[j];
        if (k.visited_ids) memcpy(k.visited_ids, vids.data() + j * vc, vc * 4);
        if (k.visited_scores) memcpy(k.visited_scores, vsc.data() + j * vc, vc * 8);
        grp[j]->rc = 0;
    }
}

// the request path in one call: src/query_disk_index.rs:436-540
struct QueryCall {
    mse_searcher* s; mse_pq* pq; const mse_codes* c; const mse_graph* g; const uint32_t* starts;
    const uint16_t* queries; const float* queries_f32; const float* luts; const float* scales; size_t nq;
    int disable_pq; size_t beamwidth, search_list, k;
    uint32_t* ids; int64_t* scores; uint32_t *n_visited, *cmps, *pq_cmps;
    bool shares_with(const QueryCall& o) const {
        return s->base == o.s->base && pq == o.pq && c == o.c && g == o.g && disable_pq == o.disable_pq && beamwidth == o.beamwidth &&
               search_list == o.search_list && (starts != nullptr) == (o.starts != nullptr) && (queries != nullptr) == (o.queries != nullptr) &&
               (luts != nullptr) == (o.luts != nullptr) && (scales != nullptr) == (o.scales != nullptr);
    }
};

// Entry searcher (row tables), the shared hold on the entry table, and the grow-and-repeat loop around the batched search.  `fz`
// arrives with k and its destinations set.
int fused_run(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts, const uint16_t* q16, const float* q32,
              const float* luts, const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, FusedQuery fz) {
    // the table cannot be replaced under a call in flight (mse_graph_set_entries takes the lock exclusively)
    struct Shared {
        SharedExclusive* l = nullptr;
        ~Shared() { if (l) l->unlock_shared(); }
    } hold;
    // a searcher over the entry rows for the duration of this call (made on first use, returned to the graph's pool afterwards)
    struct Borrow {
        const mse_graph* g; mse_searcher* es = nullptr;
        ~Borrow() { if (es) { std::lock_guard<std::mutex> lk(g->entry_mu); g->entry_pool.push_back(es); } }
    } borrow{g};
    fz.entries = nullptr;
    if (!starts) {
        g->entry_lock.lock_shared();
        hold.l = &g->entry_lock;
        if (g->n_entries == 0 || (!g->entry_base && !g->entry_keys_t))
            return fail("disk_query_topk: the graph has no entry table (mse_graph_set_entries / mse_graph_set_entry_centroids)");
        fz.entries = g;
        if (!g->entry_keys_t) {
            {
                std::lock_guard<std::mutex> lk(g->entry_mu);
                if (!g->entry_pool.empty()) { borrow.es = g->entry_pool.back(); g->entry_pool.pop_back(); }
            }
            if (!borrow.es && !(borrow.es = mse_searcher_new(g->entry_base))) return -1;
            fz.entry_s = borrow.es;
        }
    }
    // visited records per query kept on the device: a search fetches about search_list + a few nodes; a list that outgrows the arrays
    // is never cut (the reference keeps every record) -- the call is repeated with four times the room
    size_t cap = (std::max(2 * search_list + 64, fz.k) + 63) / 64 * 64;
    for (;;) {
        const int rc = disk_search_batch_impl(-1, s, pq, c, g, starts, q16, q32, luts, scales, nq, disable_pq, beamwidth, search_list, nullptr,
                                              nullptr, nullptr, nullptr, nullptr, cap, nullptr, nullptr, nullptr, &fz);
        if (rc != -2) return rc;
        if (cap >= ((size_t)1 << 16)) return fail("disk_query_topk: a search visited more than 65536 records");
        // with the handler's de-duplication on, a query's similarity bits cover at most 4096 visited records: grow to exactly that
        // before giving up (the repeat used to jump from 2112 to 8448 and fail although 4096 would have held the list -- ADVICE r5)
        const size_t dedup_max = 4096;
        if (g->dedup_threshold > 0.0f && cap * 4 > dedup_max) {
            if (cap >= dedup_max) return fail("disk_query_topk: a search visited more than 4096 records (the limit with de-duplication on)");
            cap = dedup_max;
        } else {
            cap *= 4;
        }
    }
}

int query_call_on(mse_searcher* s, const QueryCall& k) {
    FusedQuery fz;
    fz.k = k.k; fz.ids = k.ids; fz.scores = k.scores; fz.n_visited = k.n_visited; fz.cmps = k.cmps; fz.pq_cmps = k.pq_cmps;
    return fused_run(s, k.pq, k.c, k.g, k.starts, k.queries, k.queries_f32, k.luts, k.scales, k.nq, k.disable_pq, k.beamwidth, k.search_list, fz);
}

// `n` waiting request-path calls that can share a submission = ONE entry step + ONE search launch + ONE select, on the worker's own
// searcher.  Inputs are gathered straight into pinned memory -- also for a single request: a pageable source of more than a few KB
// makes the runtime pin the caller's pages for the copy (a 16-query call from a numpy array: 4.8 ms against 0.5 ms).
int query_run_requests(mse_graph::WorkerCtx& ctx, DispatchReq* const* reqs, size_t n) {
    const QueryCall& lead = *static_cast<const QueryCall*>(reqs[0]->aux0);
    const mse_graph* g = lead.g;
    const size_t d = lead.s->base->d, n_desc = (lead.scales && lead.c) ? lead.c->n_desc : 0;
    size_t total = 0, kmax = 0;
    for (size_t i = 0; i < n; i++) {
        const QueryCall& k = *static_cast<const QueryCall*>(reqs[i]->aux0);
        total += k.nq;
        kmax = std::max(kmax, k.k);
    }
    const size_t q_bytes = total * d * (lead.queries ? 2 : 4), sc_bytes = total * n_desc * 4, st_bytes = lead.starts ? total * 4 : 0;
    const size_t lut_bytes = lead.luts ? total * 65536 : 0;
    const size_t off_sc = (q_bytes + 63) & ~(size_t)63, off_st = (off_sc + sc_bytes + 63) & ~(size_t)63, off_lut = (off_st + st_bytes + 63) & ~(size_t)63;
    if (ensure_pin(&ctx.pin, &ctx.pin_cap, off_lut + lut_bytes + 64)) return -1;
    char* p = static_cast<char*>(ctx.pin);
    ctx.dsts.resize(total);
    size_t row = 0;
    for (size_t i = 0; i < n; i++) {
        const QueryCall& k = *static_cast<const QueryCall*>(reqs[i]->aux0);
        if (k.queries) memcpy(p + row * d * 2, k.queries, k.nq * d * 2);
        else memcpy(p + row * d * 4, k.queries_f32, k.nq * d * 4);
        if (n_desc) memcpy(p + off_sc + row * n_desc * 4, k.scales, k.nq * n_desc * 4);
        if (k.starts) memcpy(p + off_st + row * 4, k.starts, k.nq * 4);
        if (k.luts) memcpy(p + off_lut + row * 65536, k.luts, k.nq * 65536);
        for (size_t q = 0; q < k.nq; q++, row++)
            ctx.dsts[row] = QueryDst{k.ids + q * k.k, k.scores + q * k.k, k.n_visited ? k.n_visited + q : nullptr, k.cmps ? k.cmps + q : nullptr,
                                     k.pq_cmps ? k.pq_cmps + q : nullptr, k.k};
    }
    FusedQuery fz;
    fz.k = kmax; fz.dst = ctx.dsts.data();
    return fused_run(ctx.s, lead.pq, lead.c, g, lead.starts ? reinterpret_cast<const uint32_t*>(p + off_st) : nullptr,
                     lead.queries ? reinterpret_cast<const uint16_t*>(p) : nullptr, lead.queries ? nullptr : reinterpret_cast<const float*>(p),
                     lead.luts ? reinterpret_cast<const float*>(p + off_lut) : nullptr, n_desc ? reinterpret_cast<const float*>(p + off_sc) : lead.scales,
                     total, lead.disable_pq, lead.beamwidth, lead.search_list, fz);
}

void query_run_group(std::vector<DispatchReq*>& grp) {
    const QueryCall& lead = *static_cast<const QueryCall*>(grp[0]->aux0);
    const mse_graph* g = lead.g;
    const mse_base* b = lead.s->base;
    (void)hipSetDevice(b->device);
    const int w = Coalescer::worker_index();
    mse_graph::WorkerCtx& ctx = g->co_ctx[(size_t)w < g->co_ctx.size() ? (size_t)w : 0];
    if (!ctx.s || ctx.s->base != b) {
        if (ctx.s) mse_searcher_free(ctx.s);
        ctx.s = mse_searcher_new(b);
        if (!ctx.s) {
            const std::string why = mse_last_error();
            for (DispatchReq* r : grp) { r->rc = -1; r->err = why; }
            return;
        }
    }
    if (query_run_requests(ctx, grp.data(), grp.size()) == 0) {
        for (DispatchReq* r : grp) r->rc = 0;
        return;
    }
    if (grp.size() == 1) {
        grp[0]->rc = -1;
        grp[0]->err = mse_last_error();
        return;
    }
    // the shared submission failed: each request on its own, so that a caller only ever sees its own outcome
    for (DispatchReq* r : grp) {
        r->rc = query_run_requests(ctx, &r, 1);
        if (r->rc) r->err = mse_last_error();
    }
}

void graph_run_batch(std::vector<DispatchReq*>& batch) {
    std::vector<char> taken(batch.size(), 0);
    std::vector<DispatchReq*> grp;
    for (size_t i = 0; i < batch.size(); i++) {
        if (taken[i]) continue;
        grp.clear();
        const size_t kind = batch[i]->aux_n;
        for (size_t j = i; j < batch.size(); j++) {
            if (taken[j] || batch[j]->aux_n != kind) continue;
            const bool same = kind == REQ_BEAM ? static_cast<const BeamCall*>(batch[i]->aux0)->shares_with(*static_cast<const BeamCall*>(batch[j]->aux0))
                                               : static_cast<const QueryCall*>(batch[i]->aux0)->shares_with(*static_cast<const QueryCall*>(batch[j]->aux0));
            if (same) { taken[j] = 1; grp.push_back(batch[j]); }
        }
        if (kind == REQ_BEAM) beam_run_group(grp); else query_run_group(grp);
    }
}

Coalescer* graph_coalescer(const mse_graph* g) {
    if (Coalescer* co = g->co_fast.load(std::memory_order_acquire)) return co;   // thousands of request threads pass here: no lock once it exists
    std::lock_guard<std::mutex> lk(g->co_mu);
    if (!g->co) {
        const int workers = g->co_workers > 0 ? g->co_workers : 3;
        // contexts beyond the new worker count own a searcher (device scratch, a stream) and pinned staging: freed, not dropped
        // (no worker is alive here: the previous coalescer was deleted, and with it its threads, before a new one is made)
        for (size_t w = (size_t)workers; w < g->co_ctx.size(); w++) {
            if (g->co_ctx[w].s) mse_searcher_free(g->co_ctx[w].s);
            if (g->co_ctx[w].pin) (void)hipHostFree(g->co_ctx[w].pin);
            g->co_ctx[w] = mse_graph::WorkerCtx{};
        }
        g->co_ctx.resize((size_t)workers);
        g->co = new (std::nothrow) Coalescer(g->co_max_queries ? g->co_max_queries : 1024, g->co_max_wait_us ? g->co_max_wait_us : 200,
                                             [](std::vector<DispatchReq*>& b) { graph_run_batch(b); }, nullptr, workers);
        if (!g->co) fail("out of host memory");
        g->co_fast.store(g->co, std::memory_order_release);
    }
    return g->co;
}

// nq == 1: through the graph's coalescer.  Argument errors that belong to one caller are found before it queues.
int beam_one_query(BeamCall& k) {
    if (!k.s || !k.s->base || !k.g || !k.starts || (!k.queries && !k.queries_f32) || !k.buf_ids || !k.buf_scores || !k.buf_len || !k.n_visited ||
        !k.cmps || !k.pq_cmps)
        return fail("disk_search_batch: null argument");
    if (k.beamwidth == 0 || k.beamwidth > BS_BEAM_MAX) return fail("disk_search_batch: beamwidth must be 1..8");
    if (k.search_list == 0 || k.search_list > BS_LMAX) return fail("disk_search_batch: search_list must be 1..1024");
    if (k.starts[0] >= k.g->n) return beam_call_alone(k);   // let the search report it in its own words
    Coalescer* co = graph_coalescer(k.g);
    if (!co) return -1;
    DispatchReq r;
    r.nq = 1;
    r.aux0 = &k;
    r.aux_n = REQ_BEAM;
    return co->submit(r);
}

// true when p points into device (or managed) memory; plain host memory is unknown to the runtime and reported as an error
bool is_device_pointer(const void* p) {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

int query_front(QueryCall& k) {
    if (!k.g || !k.ids || !k.scores || (!k.queries && !k.queries_f32)) return fail("disk_query_topk: null argument");
    if (k.nq == 0) return 0;
    if (!k.s || !k.s->base) return fail("disk_query_topk: null searcher");
    if (k.k == 0 || k.k > (size_t)TOPK_KMAX - 64) return fail("disk_query_topk: bad k / outputs");
    // f16 queries may be device-resident (embeddings that never left the GPU): such a call cannot be gathered by the host and goes straight
    // to the device.  f32 queries are host memory by contract (the handler's input, :436-477): no runtime call on the request thread.
    if (k.nq > FUSED_COALESCE_MAX || (k.queries && is_device_pointer(k.queries))) return query_call_on(k.s, k);
    // what belongs to this caller alone is found before it queues
    if (k.beamwidth == 0 || k.beamwidth > BS_BEAM_MAX) return fail("disk_search_batch: beamwidth must be 1..8");
    if (k.search_list == 0 || k.search_list > BS_LMAX) return fail("disk_search_batch: search_list must be 1..1024");
    if ((!k.disable_pq && (!k.pq || !k.c || (!k.luts && !k.queries_f32))) || (k.scales && !k.c)) return fail("disk_search_batch: null argument");
    for (size_t q = 0; k.starts && q < k.nq; q++)
        if (k.starts[q] >= k.g->n) return fail("disk_search_batch: start node out of range");
    Coalescer* co = graph_coalescer(k.g);
    if (!co) return -1;
    DispatchReq r;
    r.nq = k.nq;
    r.aux0 = &k;
    r.aux_n = REQ_QUERY;
    return co->submit(r);
}
}  // namespace

extern "C" {

int mse_disk_search_batch(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts,
                          const uint16_t* queries, const float* luts, const float* scales, size_t nq, int disable_pq,
                          size_t beamwidth, size_t search_list, uint32_t* buf_ids, int64_t* buf_scores, uint32_t* buf_len,
                          uint32_t* visited_ids, int64_t* visited_scores, size_t visited_cap, uint32_t* n_visited,
                          uint32_t* cmps, uint32_t* pq_cmps) {
    if (!queries) return fail("disk_search_batch: null argument");
    if (nq == 1) {
        BeamCall k{s, pq, c, g, starts, queries, nullptr, luts, scales, disable_pq, beamwidth, search_list, visited_cap, buf_ids, buf_scores, buf_len,
                   visited_ids, visited_scores, n_visited, cmps, pq_cmps};
        return beam_one_query(k);
    }
    return disk_search_batch_impl(-1, s, pq, c, g, starts, queries, nullptr, luts, scales, nq, disable_pq, beamwidth, search_list, buf_ids,
                                  buf_scores, buf_len, visited_ids, visited_scores, visited_cap, n_visited, cmps, pq_cmps);
}

// ---- the request path in one call (src/query_disk_index.rs:436-540 for a batch) ---------------------------------------------
static void clear_entries_locked(mse_graph* g) {   // entry_lock held exclusively
    {
        std::lock_guard<std::mutex> lk(g->entry_mu);
        for (mse_searcher* es : g->entry_pool) mse_searcher_free(es);
        g->entry_pool.clear();
    }
    if (g->entry_base) { mse_base_free(g->entry_base); g->entry_base = nullptr; }
    if (g->entry_rows) { (void)hipFree(g->entry_rows); g->entry_rows = nullptr; }
    if (g->entry_ids) { (void)hipFree(g->entry_ids); g->entry_ids = nullptr; }
    if (g->entry_keys_t) { (void)hipFree(g->entry_keys_t); g->entry_keys_t = nullptr; }
    g->entry_keys_d = 0;
    g->n_entries = 0;
}

static int set_entries_locked(mse_graph* g, const mse_base* b, const uint32_t* node_ids, size_t n_entries) {
    MSE_HIP_TRY(hipMalloc((void**)&g->entry_ids, n_entries * 4));
    MSE_HIP_TRY(hipMalloc((void**)&g->entry_rows, n_entries * b->d * 2));
    MSE_HIP_TRY(hipMemcpy(g->entry_ids, node_ids, n_entries * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(gather_entry_rows_kernel, dim3((unsigned)n_entries), dim3(64), 0, nullptr, b->dev, (int)b->d, g->entry_ids, g->entry_rows);
    MSE_HIP_TRY(hipGetLastError());
    MSE_HIP_TRY(hipDeviceSynchronize());
    g->entry_base = mse_base_wrap_device(g->entry_rows, n_entries, b->d);
    if (!g->entry_base) return -1;
    // one searcher now, used once: whatever the base prepares lazily for the matrix-core path (row norms) exists before callers on
    // several threads arrive
    mse_searcher* es = mse_searcher_new(g->entry_base);
    if (!es) return -1;
    int rc = 0;
    {
        DevBuf tmp;
        const size_t nqw = 16;
        hipError_t he = hipSuccess;
        if (tmp.ensure(nqw * b->d * 2 + nqw * 12)) rc = -1;
        else if ((he = hipMemset(tmp.p, 0, nqw * b->d * 2 + nqw * 12)) != hipSuccess) rc = fail(std::string("hipMemset: ") + hipGetErrorString(he));
        else {
            char* o = tmp.as<char>() + nqw * b->d * 2;
            rc = mse_bruteforce_topk_f16_dev(es, tmp.p, nqw, 1, MSE_MODE_MFMA, 0, o, o + nqw * 8);
            if ((he = hipStreamSynchronize(es->stream)) != hipSuccess && !rc) rc = fail(std::string("hipStreamSynchronize: ") + hipGetErrorString(he));
        }
    }
    if (rc) { mse_searcher_free(es); return -1; }
    g->n_entries = n_entries;
    std::lock_guard<std::mutex> lk(g->entry_mu);
    g->entry_pool.push_back(es);
    return 0;
}

int mse_graph_set_entries(mse_graph* g, const mse_base* b, const uint32_t* node_ids, size_t n_entries) {
    if (!g || !b || !b->dev || (!node_ids && n_entries)) return fail("graph_set_entries: null argument");
    if (b->n != g->n) return fail("graph_set_entries: vectors and graph differ in length");
    if (b->d % 64 || b->d == 0) return fail("graph_set_entries: vector width must be a multiple of 64");
    for (size_t i = 0; i < n_entries; i++)
        if (node_ids[i] >= g->n) return fail("graph_set_entries: entry id out of range");
    // exclusive: waits for the request-path calls in flight (they hold the lock shared for their duration) and keeps new ones out
    std::lock_guard<SharedExclusive> ex(g->entry_lock);
    clear_entries_locked(g);
    if (n_entries == 0) return 0;
    const int rc = set_entries_locked(g, b, node_ids, n_entries);
    if (rc) { const std::string why = mse_last_error(); clear_entries_locked(g); set_error(why); }   // nothing half-made stays behind
    return rc;
}

int mse_graph_set_entry_centroids(mse_graph* g, const float* centroids, size_t d, const uint32_t* node_ids, size_t n_entries) {
    if (!g || (n_entries && (!centroids || !node_ids))) return fail("graph_set_entry_centroids: null argument");
    if (n_entries && (d == 0 || d % 32 || d > 4096)) return fail("graph_set_entry_centroids: vector width must be a multiple of 32, at most 4096");
    for (size_t i = 0; i < n_entries; i++)
        if (node_ids[i] >= g->n) return fail("graph_set_entry_centroids: entry id out of range");
    std::lock_guard<SharedExclusive> ex(g->entry_lock);
    clear_entries_locked(g);
    if (n_entries == 0) return 0;
    std::vector<float> t(n_entries * d);   // transposed [d][n_entries]: a lane per shard reads consecutive floats
    for (size_t e = 0; e < n_entries; e++)
        for (size_t k = 0; k < d; k++) t[k * n_entries + e] = centroids[e * d + k];
    hipError_t he = hipMalloc((void**)&g->entry_ids, n_entries * 4);
    if (he == hipSuccess) he = hipMalloc((void**)&g->entry_keys_t, n_entries * d * 4);
    if (he == hipSuccess) he = hipMemcpy(g->entry_ids, node_ids, n_entries * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(g->entry_keys_t, t.data(), n_entries * d * 4, hipMemcpyHostToDevice);
    if (he != hipSuccess) { clear_entries_locked(g); return fail(std::string("graph_set_entry_centroids: ") + hipGetErrorString(he)); }
    g->entry_keys_d = d;
    g->n_entries = n_entries;
    return 0;
}

int mse_disk_query_topk(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts, const uint16_t* queries,
                        const float* luts, const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k,
                        uint32_t* ids, int64_t* scores, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps) {
    if (!queries) return fail("disk_query_topk: null argument");
    QueryCall q{s, pq, c, g, starts, queries, nullptr, luts, scales, nq, disable_pq, beamwidth, search_list, k, ids, scores, n_visited, cmps, pq_cmps};
    return query_front(q);
}

int mse_disk_query_topk_f32(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts, const float* queries_f32,
                            const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k, uint32_t* ids,
                            int64_t* scores, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps) {
    if (!queries_f32) return fail("disk_query_topk_f32: null argument");
    QueryCall q{s, pq, c, g, starts, nullptr, queries_f32, nullptr, scales, nq, disable_pq, beamwidth, search_list, k, ids, scores, n_visited, cmps, pq_cmps};
    return query_front(q);
}

// ---- the request path without a thread per request (round 5) ----------------------------------------------------------------
// A ticket owns everything a queued request needs after the submitting call returned: the request record, the call's arguments and
// a copy of the query (the caller's buffer is free again at once; the OUTPUT arrays stay the caller's and must outlive the ticket's
// completion).
struct mse_completion_queue {
    CompletionQueue q;
};

struct mse_ticket {
    DispatchReq r;
    QueryCall k;
    std::vector<float> q32;
    std::vector<float> scales;
    void* user = nullptr;
};

// Tickets are recycled per thread: a loop that submits and releases on one thread (the usual poller) allocates nothing in steady state
// (a ticket keeps the capacity of its query copy); tickets released on another thread fill that thread's cache up to its cap.
namespace {
constexpr size_t TICKET_CACHE_MAX = 8192;
struct TicketCache {
    std::vector<mse_ticket*> free_list;
    ~TicketCache() { for (mse_ticket* t : free_list) delete t; }
};
thread_local TicketCache tl_tickets;

mse_ticket* ticket_take() {
    if (!tl_tickets.free_list.empty()) {
        mse_ticket* t = tl_tickets.free_list.back();
        tl_tickets.free_list.pop_back();
        return t;
    }
    return new (std::nothrow) mse_ticket();
}

void ticket_give(mse_ticket* t) {
    if (tl_tickets.free_list.size() >= TICKET_CACHE_MAX) { delete t; return; }
    // back to the state of a new record (the vectors keep their capacity)
    t->r.rc = 0; t->r.err.clear(); t->r.flags = 0; t->r.next = nullptr; t->r.cnext = nullptr; t->r.cq = nullptr; t->user = nullptr;
    try {
        tl_tickets.free_list.push_back(t);
    } catch (const std::bad_alloc&) {
        delete t;
    }
}

int submit_f32(bool copy, mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const float* queries_f32, const float* scales,
               size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k, uint32_t* ids, int64_t* scores,
               uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps, void* user, mse_completion_queue* cq,
               mse_ticket** ticket_out) {
    if (!queries_f32 || !ticket_out || !g || !ids || !scores) return fail("disk_query_submit_f32: null argument");
    if (!s || !s->base) return fail("disk_query_submit_f32: null searcher");
    if (nq == 0 || nq > FUSED_COALESCE_MAX) return fail("disk_query_submit_f32: 1.." + std::to_string(FUSED_COALESCE_MAX) + " queries per request");
    if (k == 0 || k > (size_t)TOPK_KMAX - 64) return fail("disk_query_topk: bad k / outputs");
    if (beamwidth == 0 || beamwidth > BS_BEAM_MAX) return fail("disk_search_batch: beamwidth must be 1..8");
    if (search_list == 0 || search_list > BS_LMAX) return fail("disk_search_batch: search_list must be 1..1024");
    if ((!disable_pq && (!pq || !c)) || (scales && !c)) return fail("disk_search_batch: null argument");
    Coalescer* co = graph_coalescer(g);
    if (!co) return -1;
    mse_ticket* t = ticket_take();
    if (!t) return fail("out of host memory");
    const size_t d = s->base->d;
    const float *q_use = queries_f32, *sc_use = scales;
    if (copy) {
        try {
            t->q32.assign(queries_f32, queries_f32 + nq * d);
            if (scales) t->scales.assign(scales, scales + nq * c->n_desc);
        } catch (const std::bad_alloc&) {
            delete t;
            return fail("out of host memory");
        }
        q_use = t->q32.data();
        sc_use = scales ? t->scales.data() : nullptr;
    }
    t->user = user;
    t->k = QueryCall{s, pq, c, g, nullptr, nullptr, q_use, nullptr, sc_use, nq, disable_pq, beamwidth, search_list, k,
                     ids, scores, n_visited, cmps, pq_cmps};
    t->r.nq = nq;
    t->r.aux0 = &t->k;
    t->r.aux_n = REQ_QUERY;
    t->r.owner = t;
    t->r.cq = cq ? &cq->q : nullptr;
    *ticket_out = t;   // (before the record is queued: it may complete, and be handed to another thread, before submit_async returns)
    if (co->submit_async(t->r)) {
        *ticket_out = nullptr;
        delete t;
        return -1;
    }
    return 0;
}
}  // namespace

int mse_disk_query_submit_f32(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const float* queries_f32, const float* scales,
                              size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k, uint32_t* ids, int64_t* scores,
                              uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps, void* user, mse_completion_queue* cq,
                              mse_ticket** ticket_out) {
    return submit_f32(true, s, pq, c, g, queries_f32, scales, nq, disable_pq, beamwidth, search_list, k, ids, scores, n_visited, cmps, pq_cmps, user, cq,
                      ticket_out);
}

// the same without the copies: queries_f32 (and scales) must stay valid and unchanged until the ticket has come back
int mse_disk_query_submit_f32_nocopy(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const float* queries_f32,
                                     const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k, uint32_t* ids,
                                     int64_t* scores, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps, void* user, mse_completion_queue* cq,
                                     mse_ticket** ticket_out) {
    return submit_f32(false, s, pq, c, g, queries_f32, scales, nq, disable_pq, beamwidth, search_list, k, ids, scores, n_visited, cmps, pq_cmps, user, cq,
                      ticket_out);
}

long mse_graph_completions(const mse_graph* g, mse_ticket** out, size_t max, long timeout_us) {
    if (!g || !out) return fail("graph_completions: null argument");
    if (max == 0) return 0;
    Coalescer* co = graph_coalescer(g);
    if (!co) return -1;
    constexpr size_t CHUNK = 256;
    DispatchReq* got[CHUNK];
    const size_t n = co->completions(got, std::min(max, CHUNK), (int64_t)timeout_us);
    for (size_t i = 0; i < n; i++) out[i] = static_cast<mse_ticket*>(got[i]->owner);
    return (long)n;
}

// a completion queue of the caller's own (one per event loop): tickets submitted with it come back through it and nowhere else
mse_completion_queue* mse_completion_queue_new(void) {
    mse_completion_queue* q = new (std::nothrow) mse_completion_queue();
    if (!q) fail("out of host memory");
    return q;
}
void mse_completion_queue_free(mse_completion_queue* q) { delete q; }
int mse_completion_queue_fd(mse_completion_queue* q) { return q ? q->q.fd() : fail("completion_queue_fd: null argument"); }
long mse_completion_queue_wait(mse_completion_queue* q, mse_ticket** out, size_t max, long timeout_us) {
    if (!q || !out) return fail("completion_queue_wait: null argument");
    if (max == 0) return 0;
    constexpr size_t CHUNK = 256;
    DispatchReq* got[CHUNK];
    const size_t n = q->q.take(got, std::min(max, CHUNK), (int64_t)timeout_us, nullptr);
    for (size_t i = 0; i < n; i++) out[i] = static_cast<mse_ticket*>(got[i]->owner);
    return (long)n;
}

int mse_graph_completion_fd(const mse_graph* g) {
    if (!g) return fail("graph_completion_fd: null argument");
    Coalescer* co = graph_coalescer(g);
    if (!co) return -1;
    return co->completion_fd();
}

int mse_ticket_status(const mse_ticket* t) { return t ? t->r.rc : -1; }
const char* mse_ticket_error(const mse_ticket* t) { return t ? t->r.err.c_str() : "null ticket"; }
void* mse_ticket_user(const mse_ticket* t) { return t ? t->user : nullptr; }
void mse_ticket_free(mse_ticket* t) { if (t) ticket_give(t); }

// A shard's form of the request path: the [nq][k] results stay on the device as a packed block ([nq*k] i64 scores, [nq*k] u32 ids +
// id_offset; mse_topk_block_bytes) ready for the exchange; no coalescing (the shard's thread brings the whole batch).
int mse_disk_query_topk_block(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts, const uint16_t* queries,
                              const float* luts, const float* scales, size_t nq, int disable_pq, size_t beamwidth, size_t search_list, size_t k,
                              uint64_t id_offset, void* block_dev, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps) {
    if (!g || !queries || !block_dev) return fail("disk_query_topk_block: null argument");
    if (nq == 0) return 0;
    if (!s || !s->base) return fail("disk_query_topk_block: null searcher");
    if (k == 0 || k > (size_t)TOPK_KMAX - 64) return fail("disk_query_topk: bad k / outputs");
    FusedQuery fz;
    fz.k = k; fz.n_visited = n_visited; fz.cmps = cmps; fz.pq_cmps = pq_cmps;
    fz.dev_sc = reinterpret_cast<int64_t*>(block_dev);
    fz.dev_ids = reinterpret_cast<uint32_t*>(static_cast<char*>(block_dev) + nq * k * 8);
    fz.id_offset = id_offset;
    return fused_run(s, pq, c, g, starts, queries, nullptr, luts, scales, nq, disable_pq, beamwidth, search_list, fz);
}

int mse_graph_set_dedup(mse_graph* g, float threshold) {
    if (!g) return fail("null graph");
    if (!(threshold >= 0.0f)) return fail("graph_set_dedup: threshold must be >= 0 (0 = off)");
    g->dedup_threshold = threshold;   // no request-path call may be in flight
    return 0;
}

int mse_graph_set_coalescer(mse_graph* g, size_t max_queries_per_pass, uint32_t max_wait_us, int workers) {
    if (!g) return fail("null graph");
    if (workers < 0 || workers > 8) return fail("graph_set_coalescer: 1..8 workers (0 = default)");
    Coalescer* old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g->co_mu);
        old = g->co;
        g->co = nullptr;
        g->co_fast.store(nullptr, std::memory_order_release);
        g->co_max_queries = max_queries_per_pass; g->co_max_wait_us = max_wait_us; g->co_workers = workers;
    }
    delete old;   // joins its workers; no call may be in flight (as for mse_graph_free)
    return 0;
}

// Measurement hook for bench.py's gather roofline: HIP events around every beam_search_kernel launch of this searcher and device totals
// of what the searches gathered.  enable: 0 off, 1 on, 2 on + reset.  out (optional, 8 words): kernel ms x 1000 (integer microseconds),
// launches, queries, rows scored exactly (2304-byte gathers at d = 1152), nodes fetched (adjacency lists), ADC-scored neighbours,
// beam iterations, beam iterations whose inserts were replayed sequentially (equal scores in play).
int mse_searcher_beam_timing(mse_searcher* s, int enable, uint64_t out[8]) {
    if (!s) return fail("null searcher");
    if (s->base) (void)hipSetDevice(s->base->device);
    if (out) {
        unsigned long long tot[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (s->beam_tot.p) {
            MSE_HIP_TRY(hipStreamSynchronize(s->stream));
            MSE_HIP_TRY(hipMemcpy(tot, s->beam_tot.p, sizeof tot, hipMemcpyDeviceToHost));
        }
        out[0] = (uint64_t)(s->beam_ms_total * 1000.0 + 0.5); out[1] = s->beam_launches; out[2] = s->beam_queries;
        out[3] = tot[0]; out[4] = tot[1]; out[5] = tot[2]; out[6] = tot[3]; out[7] = tot[4];
#ifdef MSE_BEAM_PHASES   // probe build: the caller passes 16 words
        for (int k_ = 0; k_ < 7; k_++) out[8 + k_] = tot[5 + k_];
#endif
    }
    if (enable) {
        if (!s->bev0) {
            MSE_HIP_TRY(hipEventCreate(&s->bev0));
            MSE_HIP_TRY(hipEventCreate(&s->bev1));
        }
        if (!s->beam_tot.p) {
            if (s->beam_tot.ensure(128)) return -1;
            MSE_HIP_TRY(hipMemset(s->beam_tot.p, 0, 128));
        }
        if (enable == 2) {
            MSE_HIP_TRY(hipStreamSynchronize(s->stream));
            MSE_HIP_TRY(hipMemset(s->beam_tot.p, 0, 128));
            s->beam_ms_total = 0.0; s->beam_launches = 0; s->beam_queries = 0;
        }
    }
    s->beam_timing = enable != 0;
    return 0;
}

int mse_graph_coalescer_stats(const mse_graph* g, uint64_t out[6]) {
    if (!g || !out) return fail("null argument");
    DispatchStats st;
    {
        std::lock_guard<std::mutex> lk(g->co_mu);
        if (g->co) st = g->co->stats();
    }
    out[0] = st.queries; out[1] = st.requests; out[2] = st.passes; out[3] = st.max_pass_queries; out[4] = st.deadline_fires; out[5] = st.run_us;
    return 0;
}

int mse_disk_search_batch_f32(mse_searcher* s, mse_pq* pq, const mse_codes* c, const mse_graph* g, const uint32_t* starts,
                              const float* queries_f32, const float* scales, size_t nq, int disable_pq, size_t beamwidth,
                              size_t search_list, uint32_t* buf_ids, int64_t* buf_scores, uint32_t* buf_len, uint32_t* visited_ids,
                              int64_t* visited_scores, size_t visited_cap, uint32_t* n_visited, uint32_t* cmps, uint32_t* pq_cmps) {
    if (!queries_f32) return fail("disk_search_batch_f32: null argument");
    if (nq == 1) {
        BeamCall k{s, pq, c, g, starts, nullptr, queries_f32, nullptr, scales, disable_pq, beamwidth, search_list, visited_cap, buf_ids, buf_scores,
                   buf_len, visited_ids, visited_scores, n_visited, cmps, pq_cmps};
        return beam_one_query(k);
    }
    return disk_search_batch_impl(-1, s, pq, c, g, starts, nullptr, queries_f32, nullptr, scales, nq, disable_pq, beamwidth, search_list,
                                  buf_ids, buf_scores, buf_len, visited_ids, visited_scores, visited_cap, n_visited, cmps, pq_cmps);
}

// Orders this searcher's stream after everything `producer_stream` holds now: the way to hand device-resident inputs (queries that a
// tower wrote on ITS stream) to a call that copies them on the searcher's stream.
int mse_searcher_wait_stream(mse_searcher* s, void* producer_stream) {
    if (!s) return fail("null searcher");
    if (!s->ev_wait) MSE_HIP_TRY(hipEventCreateWithFlags(&s->ev_wait, hipEventDisableTiming));
    MSE_HIP_TRY(hipEventRecord(s->ev_wait, reinterpret_cast<hipStream_t>(producer_stream)));
    MSE_HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_wait, 0));
    return 0;
}

}  // extern "C"
