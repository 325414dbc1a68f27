// Developer microbenchmark: per-CU issue rate of 1 KiB LDS-DMA pieces (global_load_lds_dwordx4) and of plain
// global_load_dwordx4 from an L2-resident buffer, one workgroup per CU, W waves each.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>   // 0: LDS-DMA, 1: global_load_dwordx4 into VGPRs, 2: global_load + ds_write_b128
__global__ __launch_bounds__(1024) void rate_kernel(const char* __restrict__ src, size_t span, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* mine = smem + wave * 8192;
    const char* p = src + ((size_t)blockIdx.x * 65536 + wave * 8192) % span + lane * 16;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        const char* q = p + (size_t)(it & 7) * 8192 % span;
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(q + u * 1024),
                                                 (__attribute__((address_space(3))) void*)(mine + u * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const uint4*>(q + u * 1024);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (MODE == 2) *reinterpret_cast<uint4*>(mine + u * 1024 + lane * 16) = v[u];
                else acc ^= v[u].x ^ v[u].w;
            }
        }
    }
    if (MODE == 2) acc ^= *reinterpret_cast<unsigned*>(mine + lane * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t span = 2 << 20;   // 2 MiB: L2-resident per XCD
    char* src; unsigned* sink;
    hipMalloc(&src, span + (1 << 20)); hipMemset(src, 1, span + (1 << 20)); hipMalloc(&sink, 4);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++)
        for (int w : {1, 2, 4, 8, 16}) {
            auto launch = [&]() {
                const size_t lds = (size_t)w * 8192;
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(cus), dim3(w * 64), lds, 0, src, span, iters, sink);
                if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(cus), dim3(w * 64), lds, 0, src, span, iters, sink);
                if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(cus), dim3(w * 64), lds, 0, src, span, iters, sink);
            };
            hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes_per_cu = (double)iters * 8 * 1024 * w;
            printf("mode %d (%s) waves/CU %2d: %.3f ms, %.1f GB/s per CU, %.1f TB/s chip, %.1f ns per 1 KiB piece per CU\n", mode,
                   mode == 0 ? "lds-dma" : mode == 1 ? "global_load->vgpr" : "global_load->ds_write", w, ms,
                   bytes_per_cu / ms / 1e6, bytes_per_cu * cus / ms / 1e9, ms * 1e6 / (iters * 8.0 * w));
        }
    return 0;
}
