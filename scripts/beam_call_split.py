#!/usr/bin/env python3
"""Where a batched beam-search call spends its wall time (host side): entry selection, the search call, the final top-k.
python scripts/beam_call_split.py [rows]"""
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
clustered = bench.clustered_generator(n)
rows, queries = clustered(n, 1), clustered(4096, 2)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, D, keepalive=rows)
s = mse.Searcher(vecs)
med = mse.medioid(vecs)
g = mse.BuildGraph(n, 64)
g.random_fill(1)
order = np.random.default_rng(3).permutation(n).astype(np.uint32)
g.build(s, order, med, mse.IndexBuildConfig(r=64, l=192, maxc=750), 4096)
qh = queries.cpu().numpy().view(np.uint16)
e_idx = np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.int64)
e_rows = rows[torch.from_numpy(e_idx).cuda()].contiguous()
es = mse.Searcher(mse.VectorList.wrap_device(e_rows.data_ptr(), len(e_idx), D, keepalive=e_rows))
for nq in (1024, 2048, 4096):
    q = qh[:nq]
    for rep in range(3):
        t0 = time.perf_counter()
        _, top = es.bruteforce_topk(q, 1, mse.MODE_MFMA)
        st = e_idx[top[:, 0]].astype(np.uint32)
        t1 = time.perf_counter()
        res = mse.disk_search_batch(s, None, None, g, st, q, None, None, True, 4, 32, 1024, as_arrays=True)
        t2 = time.perf_counter()
        top10 = mse.topk_of_visited(res, 10)
        t3 = time.perf_counter()
    print(f"{nq} queries per call: entry selection {1e3 * (t1 - t0):.2f} ms, search call {1e3 * (t2 - t1):.2f} ms, host top-k of visited {1e3 * (t3 - t2):.2f} ms; "
          f"widest visited list {int(res['n_visited'].max())}, node fetches per query {float(res['cmps'].mean()):.1f}")
