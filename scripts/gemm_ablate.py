import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
from mse import ffi
L = ffi.lib()
M, N, K = 186624, 4352, 1152
for abl, name in ((0, "full"), (1, "no MFMA"), (2, "no DMA"), (3, "no LDS reads"), (0, "full")):
    ms = C.c_float()
    ffi.check(L.mse_debug_gemm_ms(M, N, K, abl, 5, C.byref(ms)))
    print(f"{name:14s} {ms.value:7.3f} ms  {2*M*N*K/ms.value/1e9:7.0f} TFLOP/s-equivalent")
