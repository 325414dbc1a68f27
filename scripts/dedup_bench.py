"""mse_dedup_visited (src/query_disk_index.rs:482-527) timing: python scripts/dedup_bench.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch  # noqa: F401
import mse
vl = mse.VectorList.generate(0x5EED0001, 0, 200_000)
s = mse.Searcher(vl)
rng = np.random.default_rng(0)
for n in (200, 1000, 4000):
    ids = rng.choice(200_000, n, replace=False).astype(np.uint32)
    mse.dedup_visited(s, ids)
    t0 = time.perf_counter()
    for _ in range(5):
        k = mse.dedup_visited(s, ids)
    dt = (time.perf_counter() - t0) / 5
    print(f"dedup_visited n={n}: {dt*1e3:.2f} ms, kept {int(np.sum(k))}")
