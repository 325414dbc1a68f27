// Host side of the graph searches: NeighbourBuffer and the in-RAM Vamana greedy search.
//
// NeighbourBuffer: diskann/src/lib.rs:74-155 -- capacity-bounded list sorted by score descending,
// with a visited flag per entry and the index of the best unvisited entry.
// greedy_search:   diskann/src/lib.rs:183-211 -- traversal stays on the host (pointer chasing), the
// per-hop neighbour scoring (lib.rs:201-207) is one gather-and-score launch on the device.
#include "../../include/mse.h"
#include "runtime.h"
#include <algorithm>
#include <vector>

struct mse_nb {
    std::vector<uint32_t> ids;
    std::vector<int64_t> scores;
    std::vector<uint8_t> visited;
    int64_t next_unvisited = -1;  // Option<u32>: -1 = None
    size_t cap = 0;
};

namespace {

// core::slice::binary_search_by with the comparator `score.partial_cmp(x)` over the descending
// list (lib.rs:122-125): returns the probe position on equality, else the insertion point.
size_t nb_locate(const mse_nb* b, int64_t score) {
    size_t size = b->scores.size();
    if (size == 0) return 0;
    size_t base = 0;
    while (size > 1) {
        const size_t half = size / 2, mid = base + half;
        base = (score > b->scores[mid]) ? base : mid;
        size -= half;
    }
    if (score == b->scores[base]) return base;
    return base + (score < b->scores[base] ? 1 : 0);
}

}  // namespace

extern "C" {

mse_nb* mse_nb_new(size_t cap) {
    mse_nb* b = new (std::nothrow) mse_nb();
    if (!b) { mse::fail("out of host memory"); return nullptr; }
    b->cap = cap;
    b->ids.reserve(cap + 1);
    b->scores.reserve(cap + 1);
    b->visited.reserve(cap + 1);
    return b;
}
void mse_nb_free(mse_nb* b) { delete b; }
void mse_nb_clear(mse_nb* b) {  // lib.rs:149-154
    b->ids.clear(); b->scores.clear(); b->visited.clear(); b->next_unvisited = -1;
}
size_t mse_nb_len(const mse_nb* b) { return b->ids.size(); }
size_t mse_nb_cap(const mse_nb* b) { return b->cap; }
const uint32_t* mse_nb_ids(const mse_nb* b) { return b->ids.data(); }
const int64_t* mse_nb_scores(const mse_nb* b) { return b->scores.data(); }

void mse_nb_insert(mse_nb* b, uint32_t id, int64_t score) {  // lib.rs:117-147
    const size_t len = b->ids.size();
    if (b->cap == 0) return;
    if (len == b->cap && b->scores[len - 1] > score) return;
    const size_t loc = nb_locate(b, score);
    if (loc < len && b->ids[loc] == id) return;
    b->ids.insert(b->ids.begin() + loc, id);
    b->scores.insert(b->scores.begin() + loc, score);
    b->visited.insert(b->visited.begin() + loc, 0);
    if (b->ids.size() > b->cap) { b->ids.resize(b->cap); b->scores.resize(b->cap); b->visited.resize(b->cap); }
    if (b->next_unvisited < 0 || (int64_t)loc < b->next_unvisited) b->next_unvisited = (int64_t)loc;
}

int mse_nb_next_unvisited(mse_nb* b, uint32_t* id) {  // lib.rs:93-107
    if (b->next_unvisited < 0) return 0;
    size_t cur = (size_t)b->next_unvisited;
    const size_t old = cur;
    b->visited[cur] = 1;
    while (cur < b->ids.size() && b->visited[cur]) cur++;
    b->next_unvisited = cur == b->ids.size() ? -1 : (int64_t)cur;
    *id = b->ids[old];
    return 1;
}

int mse_greedy_search(mse_searcher* s, const uint32_t* adj, const uint32_t* deg, size_t max_deg, uint32_t start,
                      const uint16_t* query, int base_vectors_only, uint32_t query_breakpoint, mse_nb* buf,
                      size_t* n_distances) {
    if (!s || !buf) return mse::fail("null searcher or buffer");
    const mse_base* b = s->base;
    if (start >= b->n) return mse::fail("start node out of range");
    const size_t d = b->d;
    hipStream_t st = s->stream;
    // device scratch: query, ids, scores
    if (s->q_stage.ensure(8 * d * 2) || s->cand_ids.ensure((max_deg + 1) * 4) || s->cand_scores.ensure((max_deg + 1) * 8))
        return -1;
    MSE_HIP_TRY(hipMemcpyAsync(s->q_stage.p, query, d * 2, hipMemcpyHostToDevice, st));
    std::vector<uint8_t> visited(b->n, 0);
    std::vector<uint32_t> pre;
    std::vector<int64_t> sc(max_deg + 1);
    pre.reserve(max_deg + 1);
    size_t distances = 0;
    mse_nb_clear(buf);

    auto score = [&](const uint32_t* ids, size_t n) -> int {
        MSE_HIP_TRY(hipMemcpyAsync(s->cand_ids.p, ids, n * 4, hipMemcpyHostToDevice, st));
        if (mse::launch_score_rows(b->dev, b->n, (int)d, s->q_stage.p, false, s->cand_ids.as<uint32_t>(), n, n,
                                   s->cand_scores.as<int64_t>(), nullptr, st)) return -1;
        MSE_HIP_TRY(hipMemcpyAsync(sc.data(), s->cand_scores.p, n * 8, hipMemcpyDeviceToHost, st));
        MSE_HIP_TRY(hipStreamSynchronize(st));
        return 0;
    };

    if (score(&start, 1)) return -1;  // lib.rs:188 (not counted in `distances`)
    mse_nb_insert(buf, start, sc[0]);
    visited[start] = 1;
    uint32_t pt;
    while (mse_nb_next_unvisited(buf, &pt)) {  // lib.rs:193
        pre.clear();
        for (uint32_t e = 0; e < deg[pt]; e++) {
            const uint32_t nb = adj[(size_t)pt * max_deg + e];
            if (nb >= b->n) return mse::fail("graph edge points outside the vector list");
            const bool fresh = !visited[nb];
            visited[nb] = 1;  // HashSet::insert happens before the base_vectors_only test (lib.rs:197)
            if (fresh && !(base_vectors_only && nb >= query_breakpoint)) pre.push_back(nb);
        }
        if (pre.empty()) continue;
        if (score(pre.data(), pre.size())) return -1;
        for (size_t i = 0; i < pre.size(); i++) {
            distances++;
            mse_nb_insert(buf, pre[i], sc[i]);
        }
    }
    if (n_distances) *n_distances = distances;
    return 0;
}

}  // extern "C"
