"""Small-batch GEMM kernels of the SigLIP towers, shape by shape: the large-batch kernels (0), the K-split skinny kernel (2), the
64 x 64 (3) and 128 x 128 (4) tiles, and what launch_gemm picks by size (1), at the row counts of 1..8 images (736 rows each) and
1..64 texts (64 rows each).  Weights stream from HBM (a ring of matrices larger than the last-level cache), as in a forward pass.
Prints microseconds per launch and whether the result is bit-equal to the large-batch kernels'.

    python scripts/gemm_small_probe.py            # one MI355X
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
from mse import ffi  # noqa: E402

L = ffi.lib()
SHAPES = [("qkv", 3456, 1152, 0), ("proj", 1152, 1152, 0), ("fc1", 4352, 1152, 1), ("fc2", 1152, 4352, 0)]
ROWS = [64, 128, 256, 512, 736, 1024, 1472, 2048, 2944, 4096, 5888]
print("rows  shape      big   auto  skinny   t64   t128   (us per launch; * = not bit-equal to the large-batch kernels, ! = more than two bf16 steps apart)")
for rows in ROWS:
    layer = [0.0] * 5
    for name, N, K, epi in SHAPES:
        cells = []
        for v, variant in enumerate((0, 1, 2, 3, 4)):
            if variant == 2 and rows > 1024:
                cells.append("     -")
                layer[v] = float("nan")
                continue
            ms, nd = C.c_float(), (C.c_uint64 * 2)()
            ffi.check(L.mse_debug_gemm_small(rows, N, K, epi, variant, 40, C.byref(ms), nd))
            cells.append("%6.1f%s" % (ms.value * 1e3, "!" if nd[1] else "*" if nd[0] else " "))
            layer[v] += ms.value * 1e3
        print("%5d %-5s %s" % (rows, name, " ".join(cells)), flush=True)
    print("%5d layer %s" % (rows, " ".join("%6.1f " % t for t in layer)), flush=True)
