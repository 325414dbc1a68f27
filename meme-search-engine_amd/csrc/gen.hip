// Deterministic synthetic rows on the device + small norm kernels.
//
// The generator is OURS, not the reference's: the reference's bench vectors come
// from Box-Muller over fastrand (diskann/src/vector.rs:24-42), whose log/cos cannot be reproduced
// bit-for-bit on a GPU.  Here: Philox-4x32-10 keyed by (seed, row); each component is a centred
// Irwin-Hall sum of four 16-bit uniforms; the row is scaled by 1/norm where norm comes from an
// exact int64 sum of squares and six Newton steps in f64 (only + * / : IEEE-exact on host and
// device alike), then f64 -> f32 -> f16 (RNE).  oracle/mse_oracle.c::orc_gen_rows_f16 performs the
// same steps; tests compare the two bit-for-bit.
#include "common.h"
#include "kernels.h"

namespace mse {
namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void pair_of(uint32_t seed, uint64_t row, uint32_t call, int& a, int& b) {
    uint32_t r[4];
    philox4x32_10(call, 0u, (uint32_t)row, (uint32_t)(row >> 32), seed, 0x5EEDu, r);
    a = (int)((r[0] & 0xffffu) + (r[0] >> 16) + (r[1] & 0xffffu) + (r[1] >> 16)) - 131070;
    b = (int)((r[2] & 0xffffu) + (r[2] >> 16) + (r[3] & 0xffffu) + (r[3] >> 16)) - 131070;
}

__device__ __forceinline__ uint16_t to_half_bits(float f) {
    _Float16 h = (_Float16)f;  // v_cvt_f16_f32, round-to-nearest-even
    return __builtin_bit_cast(uint16_t, h);
}

// one wave per row
__global__ __launch_bounds__(256) void generate_rows_kernel(uint16_t* __restrict__ out, uint32_t seed, uint64_t row0,
                                                            size_t n_rows, int d) {
    const int lane = threadIdx.x & 63;
    const int calls = d / 2;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n_rows;
         r += ((size_t)gridDim.x * blockDim.x) >> 6) {
        const uint64_t row = row0 + r;
        long long ss = 0;
        for (int c = lane; c < calls; c += 64) {
            int a, b;
            pair_of(seed, row, (uint32_t)c, a, b);
            ss += (long long)a * a + (long long)b * b;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        double norm = 1.0;
        if (ss != 0) {
            const double s = (double)ss;
            // Newton for sqrt(s) from a fixed start; no sqrt instruction, so host == device bitwise
            double y = 1284000.0;
#pragma unroll 1
            for (int i = 0; i < 12; i++) y = 0.5 * (y + s / y);
            norm = y;
        }
        uint32_t* dst = reinterpret_cast<uint32_t*>(out + r * (size_t)d);
        for (int c = lane; c < calls; c += 64) {
            int a, b;
            pair_of(seed, row, (uint32_t)c, a, b);
            const uint16_t ha = to_half_bits((float)((double)a / norm));
            const uint16_t hb = to_half_bits((float)((double)b / norm));
            dst[c] = (uint32_t)ha | ((uint32_t)hb << 16);
        }
    }
}

// 4 lanes per row like the scan; fp32 sum of squares, slight upward bias is applied by the caller.  out_bits[0] = the
// largest row norm (float bits, buffer zeroed first).  out_bits[1], [2] = the allowance for a matrix core that flushes f16
// subnormal inputs: the largest per-row sum of |x_i| over subnormal x_i and the largest |x_i| of the base -- flushing
// changes dot(x, y) by at most S_sub(x) * max|y| + S_sub(y) * max|x|.
__global__ __launch_bounds__(256) void row_norm_max_kernel(const uint16_t* __restrict__ base, size_t n_rows, int d,
                                                           uint32_t* __restrict__ out_bits) {
    const int lane = threadIdx.x & 63;
    const int part = lane & 3;
    float best = 0.0f, best_sub = 0.0f, best_abs = 0.0f;
    const size_t quad0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const size_t nquads = ((size_t)gridDim.x * blockDim.x) >> 2;
    const size_t n_round = (n_rows + 15) / 16 * 16;
    for (size_t row = quad0; row < n_round; row += nquads) {
        const size_t rr = row < n_rows ? row : n_rows - 1;
        const uint4* xp = reinterpret_cast<const uint4*>(base + rr * (size_t)d) + part;
        float s = 0.0f, sub = 0.0f;
        for (int t = 0; t < d / 32; t++) {
            const uint4 x = xp[t * 4];
            const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] & 0xffffu));
                const float hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w[i] >> 16));
                s = fmaf(lo, lo, s);
                s = fmaf(hi, hi, s);
                const float alo = fabsf(lo), ahi = fabsf(hi);
                if (alo < 6.103515625e-5f) sub += alo;          // below 2^-14: an f16 subnormal
                if (ahi < 6.103515625e-5f) sub += ahi;
                if (alo == alo) best_abs = fmaxf(best_abs, alo);
                if (ahi == ahi) best_abs = fmaxf(best_abs, ahi);
            }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        sub += __shfl_xor(sub, 1);
        sub += __shfl_xor(sub, 2);
        if (!(s == s)) s = __builtin_inff();  // NaN rows: force "no certificate"
        best = fmaxf(best, s);
        best_sub = fmaxf(best_sub, sub);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        best = fmaxf(best, __shfl_xor(best, o));
        best_sub = fmaxf(best_sub, __shfl_xor(best_sub, o));
        best_abs = fmaxf(best_abs, __shfl_xor(best_abs, o));
    }
    if (lane == 0) {
        atomicMax(out_bits, __float_as_uint(sqrtf(best) * 1.0001f));
        atomicMax(out_bits + 1, __float_as_uint(best_sub * 1.0001f));
        atomicMax(out_bits + 2, __float_as_uint(best_abs));
    }
}

__global__ void query_eps_kernel(const uint16_t* __restrict__ queries, int nq, int d,
                                 const uint32_t* __restrict__ max_norm_bits, float factor, float* __restrict__ eps) {
    const int q = blockIdx.x;
    const int lane = threadIdx.x;  // 64 threads
    float s = 0.0f, sub = 0.0f, mabs = 0.0f;
    for (int i = lane; i < d; i += 64) {
        const float v = (float)__builtin_bit_cast(_Float16, queries[(size_t)q * d + i]);
        s = fmaf(v, v, s);
        const float av = fabsf(v);
        if (av < 6.103515625e-5f) sub += av;
        if (av == av) mabs = fmaxf(mabs, av);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o);
        sub += __shfl_xor(sub, o);
        mabs = fmaxf(mabs, __shfl_xor(mabs, o));
    }
    if (lane == 0) {
        float e = factor * sqrtf(s) * 1.0001f * __uint_as_float(*max_norm_bits);
        // should the matrix core flush f16 subnormal inputs: the products it would drop (see row_norm_max_kernel)
        e += 1.0001f * (__uint_as_float(max_norm_bits[1]) * mabs + sub * __uint_as_float(max_norm_bits[2]));
        if (!(e == e)) e = __builtin_inff();
        eps[q] = e;
    }
}

// f32 queries whose f16 roundings q16 went through the matrix-core scan (flat index): the f16 bound above plus the rounding
// of the query, |x . (q - q16)| <= |x| |q - q16| with |q - q16| measured here (a q16 that overflowed to infinity gives eps = inf,
// i.e. no certificate, and the query repeats through the exact pass)
__global__ void query_eps_f32_kernel(const float* __restrict__ q32, const uint16_t* __restrict__ q16, int nq, int d,
                                     const uint32_t* __restrict__ max_norm_bits, float factor, float* __restrict__ eps) {
    const int q = blockIdx.x;
    const int lane = threadIdx.x;  // 64 threads
    float s = 0.0f, sub = 0.0f, mabs = 0.0f, dl = 0.0f;
    for (int i = lane; i < d; i += 64) {
        const float v = (float)__builtin_bit_cast(_Float16, q16[(size_t)q * d + i]);
        const float e = q32[(size_t)q * d + i] - v;
        s = fmaf(v, v, s);
        dl = fmaf(e, e, dl);
        const float av = fabsf(v);
        if (av < 6.103515625e-5f) sub += av;
        if (av == av) mabs = fmaxf(mabs, av);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        s += __shfl_xor(s, o);
        dl += __shfl_xor(dl, o);
        sub += __shfl_xor(sub, o);
        mabs = fmaxf(mabs, __shfl_xor(mabs, o));
    }
    if (lane == 0) {
        const float mn = __uint_as_float(*max_norm_bits);
        float e = factor * sqrtf(s) * 1.0001f * mn + sqrtf(dl) * 1.001f * mn;
        e += 1.0001f * (__uint_as_float(max_norm_bits[1]) * mabs + sub * __uint_as_float(max_norm_bits[2]));
        if (!(e == e)) e = __builtin_inff();
        eps[q] = e;
    }
}

}  // namespace

int launch_query_eps_f32(const float* q32, const uint16_t* q16, int nq, int d, const uint32_t* max_norm_bits, float factor, float* eps,
                         hipStream_t stream) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(query_eps_f32_kernel, dim3(nq), dim3(64), 0, stream, q32, q16, nq, d, max_norm_bits, factor, eps);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_generate_rows(uint16_t* out, uint32_t seed, uint64_t row0, size_t n_rows, int d, hipStream_t stream) {
    if (n_rows == 0) return 0;
    if (d % 64 != 0 || d <= 0 || d > D_MAX) return fail("vector width must be a positive multiple of 64");
    size_t blocks = (n_rows + 3) / 4;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(generate_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out, seed, row0, n_rows, d);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_row_norm_max(const uint16_t* base, size_t n_rows, int d, uint32_t* out_bits, hipStream_t stream) {
    if (n_rows == 0) return 0;
    size_t blocks = (n_rows + 63) / 64;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(row_norm_max_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, base, n_rows, d, out_bits);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_query_eps(const uint16_t* queries, int nq, int d, const uint32_t* max_norm_bits, float factor, float* eps,
                     hipStream_t stream) {
    if (nq == 0) return 0;
    hipLaunchKernelGGL(query_eps_kernel, dim3(nq), dim3(64), 0, stream, queries, nq, d, max_norm_bits, factor, eps);
    MSE_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mse
