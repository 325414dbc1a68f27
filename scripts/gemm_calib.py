"""Calibration of the SigLIP GEMM shapes (M = 256 x 729 tokens): the library GEMM (torch.mm -> hipBLASLt) on random bf16
data next to this package's persistent ping-pong kernel on constant and on random data (developer hook mse_debug_gemm_ms).
Run on a GPU box:  python scripts/gemm_calib.py"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
from mse import ffi  # noqa: E402

M = 186624
SHAPES = (("qk", 2304, 1152), ("fc1", 4352, 1152), ("proj", 1280, 1152), ("fc2", 1280, 4352))


def lib_ms(n, k, iters=5):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand((M, k), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand((n, k), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    wt = w.t()
    torch.mm(x, wt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.mm(x, wt)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def mine_ms(n, k, random, abl=33):
    # developer library (make dev): mse_debug_gemm_ms and its MSE_GEMM_RANDOM switch exist only there
    env = dict(os.environ, MSE_HIP_LIB=os.path.join(ROOT, "meme-search-engine_amd", "lib", "libmse_hip_dev.so"))
    env.pop("MSE_GEMM_RANDOM", None)
    if random:
        env["MSE_GEMM_RANDOM"] = "1"
    code = ("import sys,ctypes as C;sys.path.insert(0,%r);from mse import ffi;L=ffi.lib();ms=C.c_float();"
            "ffi.check(L.mse_debug_gemm_ms(%d,%d,%d,%d,5,C.byref(ms)));print(ms.value)" % (os.path.join(ROOT, "meme-search-engine_amd"), M, n, k, abl))
    return float(subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()[-1])


if __name__ == "__main__":
    for name, n, k in SHAPES:
        fl = 2.0 * M * n * k
        a = lib_ms(n, k)
        b = mine_ms(n, k, False)
        c = mine_ms(n, k, True)
        print("%-5s N=%4d K=%4d  library %6.3f ms %5.0f TF/s | ping-pong const %6.3f ms %5.0f TF/s | random %6.3f ms %5.0f TF/s"
              % (name, n, k, a, fl / a / 1e9, b, fl / b / 1e9, c, fl / c / 1e9), flush=True)
