// The `visited` sets of the graph searches (HashSet<u32> in the reference: diskann/src/lib.rs:158, src/query_disk_index.rs:117-118),
// one per query in flight, in HBM.  Two layouts with the same membership semantics:
//   bits == 0   one bit per node of the index (n / 8 bytes per set; cleared with zeros)
//   bits  > 0   open-addressing table of 2^bits u32 slots (cleared with 0xFF bytes; empty = 0xFFFFFFFF), for indexes so large
//               that n / 8 bytes per query would leave room for only a few queries and make the clearing a cost of its own
//               (12.5 MB per set at 1e8 nodes against 256 KiB).  The caller keeps a table at most half full.
#pragma once
#include "common.h"
#include <cstdlib>

namespace mse {

// HashSet::insert: true if id was not in the set (and now is)
__device__ __forceinline__ bool visited_insert(uint32_t* set, int bits, uint32_t id) {
    if (bits == 0) {
        const uint32_t bit = 1u << (id & 31);
        return !(atomicOr(&set[id >> 5], bit) & bit);
    }
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t h = (id * 2654435761u) >> (32 - bits);
    // The callers keep a table at most half full between steps and a step adds <= 1024 ids to a table of >= 4096 slots, so a free
    // slot always exists; the probe count is bounded all the same so that a mis-sized table can never spin the GPU (a full table
    // reports "already present": the search stops growing instead of hanging, and the half-full check raises err bit 4).
    for (uint32_t probes = 0; probes <= mask; probes++) {
        const uint32_t old = atomicCAS(&set[h], 0xffffffffu, id);
        if (old == 0xffffffffu) return true;
        if (old == id) return false;
        h = (h + 1) & mask;
    }
    return false;
}

// slots (u32 words) of one set, and the table size for a search that inserts at most `max_inserts` ids
inline int visited_table_bits(size_t max_inserts) {
    if (const char* e = getenv("MSE_VISITED_TABLE_BITS")) {   // test hook; honoured only inside [12, 26] (one step inserts <= 1024 ids)
        const int v = atoi(e);
        if (v >= 12 && v <= 26) return v;
    }
    int bits = 14;
    while (((size_t)1 << bits) < 2 * max_inserts && bits < 26) bits++;
    return bits;
}

}  // namespace mse
