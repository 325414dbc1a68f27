// Developer probe (not part of the product): which workgroup shapes start next to a resident persistent kernel?
// A "scan-like" kernel (grid = CUs - spare, 1024 threads, `busy_lds` bytes of LDS, ~110 VGPRs) spins for a few milliseconds on one
// stream; while it runs, a probe kernel of a given shape (threads, LDS, grid) is launched on a second stream and the time from its
// launch to its completion is printed.  A probe that cannot be placed waits for the busy kernel to end (~ its remaining time).
//   hipcc --offload-arch=gfx950 -O2 coresidency_probe.hip -o coresidency_probe && ./coresidency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void busy_kernel(long long cycles, int* sink) {
    extern __shared__ char smem[];
    asm volatile("v_mov_b32 v108, 0" ::: "v108");   // ~110 VGPRs per lane, like pq_scan64x4_kernel<16, 8>
    smem[threadIdx.x] = (char)threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (smem[(threadIdx.x * 7) & 1023] == 77 && cycles == 1) sink[0] = 1;
}
template <int VG>
__global__ void probe_kernel(int* sink) {
    extern __shared__ char smem[];
    if (VG > 64) asm volatile("v_mov_b32 v100, 0" ::: "v100");
    if (threadIdx.x == 0 && sink[1] == 12345) sink[0] = smem[0];
}

int main() {
    int n_cu = 0;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    int* sink;
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(sink, 0, 64));
    CK(hipFuncSetAttribute((const void*)busy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)probe_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)probe_kernel<100>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const long long cycles = 300000;   // wall_clock64 ticks at 100 MHz: 3 ms
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Shape { int threads, lds, grid, vg; };
    const std::vector<Shape> shapes = {{64, 0, 8, 32}, {256, 0, 64, 32}, {1024, 0, 64, 32}, {1024, 0, 64, 100}, {256, 65536, 64, 32}, {1024, 65536, 64, 100},
                                       {1024, 65536, 8, 100}, {1024, 16384, 64, 100}, {512, 65536, 64, 100}, {1024, 32768, 64, 32}};
    printf("# %d CUs; busy kernel: 1024 threads, 3 ms; probe launched 0.5 ms into it; us from probe launch to probe end\n", n_cu);
    for (int busy_lds : {132 * 1024, 64 * 1024})
        for (int spare : {0, 8, 32}) {
            for (const Shape& s : shapes) {
                hipLaunchKernelGGL(busy_kernel, dim3(n_cu - spare), dim3(1024), busy_lds, sa, cycles, sink);
                // wait ~0.5 ms on the host so that the busy kernel is resident
                hipEvent_t ew; CK(hipEventCreate(&ew)); CK(hipEventRecord(ew, sa));
                for (volatile int spin = 0; spin < 3000000; spin++) {}
                CK(hipEventRecord(e0, sb));
                if (s.vg > 64) hipLaunchKernelGGL(probe_kernel<100>, dim3(s.grid), dim3(s.threads), s.lds, sb, sink);
                else hipLaunchKernelGGL(probe_kernel<32>, dim3(s.grid), dim3(s.threads), s.lds, sb, sink);
                CK(hipEventRecord(e1, sb));
                CK(hipStreamSynchronize(sb));
                CK(hipStreamSynchronize(sa));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("busy LDS %3d KiB, grid CUs-%-2d | probe %4d threads, %2d KiB LDS, %2d workgroups, %s VGPRs: %8.1f us\n", busy_lds / 1024, spare, s.threads,
                       s.lds / 1024, s.grid, s.vg > 64 ? ">64" : "few", ms * 1e3);
                CK(hipEventDestroy(ew));
            }
        }
    return 0;
}
