#!/usr/bin/env python3
"""The entry step alone: exact top-1 of nq queries over a sampled entry table (small base, many queries) -- time and certificate counts.
python scripts/entry_step_probe.py [rows] [entries]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
import mse

D = 1152
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
ne = int(float(sys.argv[2])) if len(sys.argv) > 2 else max(4096, n // 1500)
clustered = bench.clustered_generator(n)
rows, queries = clustered(n, 1), clustered(8192, 2)
torch.cuda.synchronize()
e_idx = np.sort(np.random.default_rng(5).choice(n, ne, replace=False)).astype(np.int64)
e_rows = rows[torch.from_numpy(e_idx).cuda()].contiguous()
es = mse.Searcher(mse.VectorList.wrap_device(e_rows.data_ptr(), ne, D, keepalive=e_rows))
qh = queries.cpu().numpy().view(np.uint16)
for nq in (320, 1024, 2048, 4096, 8192):
    es.bruteforce_topk(qh[:nq], 1, mse.MODE_MFMA)
    t0 = time.perf_counter()
    for _ in range(3):
        _, top = es.bruteforce_topk(qh[:nq], 1, mse.MODE_MFMA)
    dt = (time.perf_counter() - t0) / 3
    st = es.last_stats() if hasattr(es, "last_stats") else None
    print(f"{nq:5d} queries x {ne} entries: {dt * 1e3:7.2f} ms per call (host arrays in and out); last stats {st}")
