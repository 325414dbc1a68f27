// Shared device/host helpers for libmse_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <stdlib.h>

namespace mse {

// ---- error plumbing: C ABI returns int status, message kept thread-local ----------------
void set_error(const std::string& msg);
int fail(const std::string& msg);  // sets the error, returns -1

#define MSE_HIP_TRY(expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            return ::mse::fail(std::string(#expr) + ": " + hipGetErrorString(_e));                \
        }                                                                                         \
    } while (0)

#define MSE_HIP_TRY_PTR(expr)                                                                     \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::mse::fail(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
            return nullptr;                                                                       \
        }                                                                                         \
    } while (0)

// Developer knobs (older kernels for A/B timing, ablations) exist only in the developer library (make dev, -DMSE_DEV_KERNELS):
// in the product build the environment variable is never read.
#ifdef MSE_DEV_KERNELS
#define MSE_DEV_KNOB(name) (getenv(name) != nullptr)
#else
#define MSE_DEV_KNOB(name) (false)
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: remembered per (kernel, device ordinal),
// raised when a larger request arrives.  (A flag per call site is wrong twice over: a second device never gets the attribute, and
// kernels of one signature share a function-template's statics.)
int ensure_dyn_lds(const void* kernel, int bytes);
#define MSE_DYN_LDS(kernel, bytes)                                                                \
    do {                                                                                          \
        if (::mse::ensure_dyn_lds(reinterpret_cast<const void*>(kernel), (int)(bytes))) return -1; \
    } while (0)

constexpr int D_MAX = 4096;  // largest embedding width the kernels accept (multiple of 64)

// ---- score conversions --------------------------------------------------------------------
// Rust `(x * SCALE) as i64` (diskann/src/vector.rs:408-411): truncate, saturate, NaN -> 0.
__host__ __device__ inline int64_t scale_dot_result(float x) {
    float v = x * 4294967296.0f;
    if (v != v) return 0;
    if (v >= 9223372036854775808.0f) return INT64_MAX;
    if (v <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t)v;
}
__host__ __device__ inline int64_t scale_dot_result_f64(double x) {
    double v = x * 4294967296.0;
    if (v != v) return 0;
    if (v >= 9223372036854775808.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}

// ---- order-preserving key maps ------------------------------------------------------------
__host__ __device__ inline uint64_t sortable_i64(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
__host__ __device__ inline int64_t unsortable_i64(uint64_t k) { return (int64_t)(k ^ 0x8000000000000000ull); }
__host__ __device__ inline uint32_t sortable_f32_bits(uint32_t b) { return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__host__ __device__ inline uint32_t unsortable_f32_bits(uint32_t k) { return (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; }

#ifdef __HIPCC__
// ---- exact mixed-precision FMA: f32 <- fma(f16, f16, f32) -----------------------------------
// v_fma_mix_f32 widens the selected f16 halves exactly and performs ONE fused multiply-add in
// f32: the same arithmetic as _mm256_cvtph_ps + _mm256_fmadd_ps (diskann/src/vector.rs:279-291).
__device__ __forceinline__ float fma_h_lo(uint32_t x, uint32_t q, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(x), "v"(q), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_h_hi(uint32_t x, uint32_t q, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(x), "v"(q), "v"(c));
    return d;
}
// f16 (packed pair x) times f32 query
__device__ __forceinline__ float fma_hf_lo(uint32_t x, float q, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(q), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_hf_hi(uint32_t x, float q, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(x), "v"(q), "v"(c));
    return d;
}
// plain (non-contracted) f32 add
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
#endif

}  // namespace mse
