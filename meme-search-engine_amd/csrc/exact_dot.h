// fast_dot_noprefetch (diskann/src/vector.rs:255-306) for ONE base row by a quad of four adjacent lanes, bit-identical
// to the reference's AVX2 kernel: lane part a (= lane & 3) owns the reference's accumulator a and runs its chain of
// fused multiply-adds; the fixed reduction tree (:295-303) is replayed with two cross-lane exchanges.  The same
// arithmetic as scan_exact.hip, packaged for kernels that score a handful of rows in the middle of other work.
//
// The callers are latency bound (a graph search waits for these scores before it can take its next step), so the row
// is fetched six 16-byte pieces per lane at a time with the next six already in flight while the current ones are
// consumed: a 2304-byte row costs about three memory round trips instead of thirty-six.  The order of the
// multiply-adds is untouched (t ascending within each accumulator).
#pragma once
#include "common.h"

namespace mse {

__device__ __forceinline__ void quad_fma8(float (&acc)[8], const uint4& x, const uint4& q) {
    acc[0] = fma_h_lo(x.x, q.x, acc[0]);
    acc[1] = fma_h_hi(x.x, q.x, acc[1]);
    acc[2] = fma_h_lo(x.y, q.y, acc[2]);
    acc[3] = fma_h_hi(x.y, q.y, acc[3]);
    acc[4] = fma_h_lo(x.z, q.z, acc[4]);
    acc[5] = fma_h_hi(x.z, q.z, acc[5]);
    acc[6] = fma_h_lo(x.w, q.w, acc[6]);
    acc[7] = fma_h_hi(x.w, q.w, acc[7]);
}

// all four lanes of the quad must call this together (and be active); every lane returns the f32 sum.
// row, query: device / LDS pointers to d f16 values (16-byte aligned); d % 32 == 0.
template <int G = 6>
__device__ __forceinline__ float quad_fast_dot_f32(const uint16_t* row, const uint16_t* query, int d) {
    const int part = threadIdx.x & 3;
    const uint4* xp = reinterpret_cast<const uint4*>(row) + part;
    const uint4* qp = reinterpret_cast<const uint4*>(query) + part;
    float acc[8];
#pragma unroll
    for (int l = 0; l < 8; l++) acc[l] = 0.0f;
    const int T = d / 32, groups = T / G;
    if (groups > 0) {
        uint4 xa[G], xb[G];
#pragma unroll
        for (int u = 0; u < G; u++) xa[u] = xp[u * 4];
        // the group fetched ahead is clamped to the last one instead of being made conditional: every address formed
        // here lies inside the row whatever the compiler does with the loads
        for (int g = 0; g < groups; g += 2) {
            const int g1 = g + 1 < groups ? g + 1 : groups - 1;
#pragma unroll
            for (int u = 0; u < G; u++) xb[u] = xp[(g1 * G + u) * 4];
#pragma unroll
            for (int u = 0; u < G; u++) quad_fma8(acc, xa[u], qp[(g * G + u) * 4]);
            if (g + 1 < groups) {
                const int g2 = g + 2 < groups ? g + 2 : groups - 1;
#pragma unroll
                for (int u = 0; u < G; u++) xa[u] = xp[(g2 * G + u) * 4];
#pragma unroll
                for (int u = 0; u < G; u++) quad_fma8(acc, xb[u], qp[(g1 * G + u) * 4]);
            }
        }
    }
    for (int t = groups * G; t < T; t++) quad_fma8(acc, xp[t * 4], qp[t * 4]);
    float v[8];
#pragma unroll
    for (int l = 0; l < 8; l++) v[l] = add_rn(acc[l], __shfl_xor(acc[l], 1));  // acc1+acc2 | acc3+acc4
    const float p0 = add_rn(v[0], v[1]), p1 = add_rn(v[2], v[3]);               // hadd pairs
    const float p2 = add_rn(v[4], v[5]), p3 = add_rn(v[6], v[7]);
    const float first = add_rn(p0, p2);   // parts 0,1: s0   parts 2,3: s2
    const float second = add_rn(p1, p3);  // parts 0,1: s1   parts 2,3: s3
    const float of = __shfl_xor(first, 2), os = __shfl_xor(second, 2);
    const bool low = (part & 2) == 0;
    const float s0 = low ? first : of, s1 = low ? second : os;
    const float s2 = low ? of : first, s3 = low ? os : second;
    return add_rn(add_rn(add_rn(s0, s1), s2), s3);
}

}  // namespace mse
