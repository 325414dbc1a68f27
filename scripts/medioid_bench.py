"""medioid (diskann/src/lib.rs:54-68) over synthetic rows resident in HBM: python scripts/medioid_bench.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch  # noqa: F401
import mse
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
vl = mse.VectorList.generate(0x5EED0001, 0, n)
t0 = time.perf_counter(); m = mse.medioid(vl); dt = time.perf_counter() - t0
t0 = time.perf_counter(); m2 = mse.medioid(vl); dt2 = time.perf_counter() - t0
print(f"medioid of {n} x 1152 rows: id {m} ({m2}), {dt:.3f} s first call, {dt2:.3f} s second")
