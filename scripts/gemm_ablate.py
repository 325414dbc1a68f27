import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSE_HIP_LIB", os.path.join(ROOT, "meme-search-engine_amd", "lib", "libmse_hip_dev.so"))   # developer library (make dev)
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
from mse import ffi
L = ffi.lib()
M, N, K = 186624, 4352, 1152
for abl, name in ((0, "old256 full"), (10, "pingpong"), (11, "pp no MFMA"), (12, "pp no DMA"), (13, "pp no LDS reads"), (14, "pp no epilogue"), (20, "pp no GELU"), (10, "pingpong"), (30, "persistent"), (31, "ps no stores"), (32, "ps no epilogue"), (33, "ps no GELU"), (34, "ps lax waits"), (30, "persistent")):
    ms = C.c_float()
    if L.mse_debug_gemm_ms(M, N, K, abl, 5, C.byref(ms)):
        print(f"{name:14s} not in this build (the first-generation kernels need -DMSE_DEV_KERNELS)")
        continue
    print(f"{name:14s} {ms.value:7.3f} ms  {2*M*N*K/ms.value/1e9:7.0f} TFLOP/s-equivalent", flush=True)
