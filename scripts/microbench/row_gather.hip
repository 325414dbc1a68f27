// Developer microbenchmark (needs a GPU): what random 2304-byte row gathers can reach on one MI355X -- the access pattern of the graph
// search's exact scoring (beam_search.hip: a lane quad owns a row, 36 dwordx4 loads per lane).  Rows are picked by a hash of the
// (wave, round) pair from a table of N rows; every quad xors what it loads and writes one word per row (so nothing is optimised away).
// Swept: loads in flight per lane (G, double-buffered like quad_fast_dot_f32) and waves per CU (occupancy, limited through LDS).
//   hipcc --offload-arch=gfx950 -O2 row_gather.hip -o row_gather && ./row_gather [rows=1e7]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}

template <int G>
__global__ __launch_bounds__(64) void gather_kernel(const uint4* __restrict__ base, uint32_t n_rows, int rounds, uint32_t* __restrict__ out) {
    extern __shared__ char pad[];   // occupancy limiter only
    const int lane = threadIdx.x, part = lane & 3, quad = lane >> 2;
    uint4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; r++) {
        const uint32_t row = mix((blockIdx.x * 16u + quad) * 2654435761u + (uint32_t)r * 40503u) % n_rows;
        const uint4* xp = base + (size_t)row * 144 + part;   // 2304 B = 144 x 16 B; lane `part` takes every fourth
        uint4 xa[G], xb[G];
        constexpr int groups = 36 / G;
#pragma unroll
        for (int u = 0; u < G; u++) xa[u] = xp[u * 4];
        for (int g = 0; g < groups; g += 2) {
            const int g1 = g + 1 < groups ? g + 1 : groups - 1;
#pragma unroll
            for (int u = 0; u < G; u++) xb[u] = xp[(g1 * G + u) * 4];
#pragma unroll
            for (int u = 0; u < G; u++) { acc.x ^= xa[u].x; acc.y ^= xa[u].y; acc.z ^= xa[u].z; acc.w ^= xa[u].w; }
            if (g + 1 < groups) {
                const int g2 = g + 2 < groups ? g + 2 : groups - 1;
#pragma unroll
                for (int u = 0; u < G; u++) xa[u] = xp[(g2 * G + u) * 4];
#pragma unroll
                for (int u = 0; u < G; u++) { acc.x ^= xb[u].x; acc.y ^= xb[u].y; acc.z ^= xb[u].z; acc.w ^= xb[u].w; }
            }
        }
    }
    if (pad[0] == 77) acc.x ^= 1;   // (keeps the LDS allocation alive)
    out[(size_t)blockIdx.x * 64 + lane] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// the same bytes as a stream: every wave reads whole consecutive rows (the copy-ceiling reference)
__global__ __launch_bounds__(64) void stream_kernel(const uint4* __restrict__ base, size_t n16, uint32_t* __restrict__ out) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 64) {
        const uint4 v = base[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int G> double run(const uint4* base, uint32_t n_rows, int waves_per_cu, uint32_t* out) {
    const int rounds = 64;
    const int lds = waves_per_cu >= 32 ? 0 : (160 * 1024 / waves_per_cu - 512) & ~255;
    const unsigned grid = 256u * 64u;   // 16384 waves x 16 rows x 64 rounds = 16.8 M rows = 38.7 GB
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gather_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather_kernel<G>, dim3(grid), dim3(64), lds, 0, base, n_rows, 4, out);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(gather_kernel<G>, dim3(grid), dim3(64), lds, 0, base, n_rows, rounds, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)grid * 16 * rounds * 2304.0 / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const uint32_t n_rows = argc > 1 ? (uint32_t)atof(argv[1]) : 10000000u;
    uint4* base; uint32_t* out;
    CHECK(hipMalloc(&base, (size_t)n_rows * 2304));
    CHECK(hipMemset(base, 0x5a, (size_t)n_rows * 2304));
    CHECK(hipMalloc(&out, (size_t)256 * 64 * 64 * 4 * 4));
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 32), dim3(64), 0, 0, base, (size_t)n_rows * 144, out);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(stream_kernel, dim3(256 * 32), dim3(64), 0, 0, base, (size_t)n_rows * 144, out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("# %u rows x 2304 B = %.1f GB; streaming read of all rows: %.0f GB/s\n", n_rows, n_rows * 2304.0 / 1e9, n_rows * 2304.0 / (ms * 1e-3) / 1e9);
    }
    printf("# random 2304-byte row gathers (a lane quad per row, dwordx4 loads), GB/s by loads in flight per lane (G, double-buffered) and waves per CU\n");
    printf("%10s %8s %8s %8s %8s\n", "waves/CU", "G=3", "G=6", "G=9", "G=18");
    for (int w : {4, 8, 13, 16, 24, 32}) {
        printf("%10d %8.0f %8.0f %8.0f %8.0f\n", w, run<3>(base, n_rows, w, out), run<6>(base, n_rows, w, out), run<9>(base, n_rows, w, out), run<18>(base, n_rows, w, out));
        fflush(stdout);
    }
    return 0;
}
