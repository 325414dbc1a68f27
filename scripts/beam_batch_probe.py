#!/usr/bin/env python3
"""Beam-search throughput (the request path in one call, mse_disk_query_topk) against queries per call (one-pass graph, sampled entries):
python scripts/beam_batch_probe.py [rows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mse  # noqa: E402

D = 1152
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
clustered = bench.clustered_generator(n)
rows, queries = clustered(n, 1), clustered(8192, 2)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, D, keepalive=rows)
s = mse.Searcher(vecs)
med = mse.medioid(vecs)
g = mse.BuildGraph(n, 64)
g.random_fill(1)
order = np.random.default_rng(3).permutation(n).astype(np.uint32)
t0 = time.perf_counter()
g.build(s, order, med, mse.IndexBuildConfig(r=64, l=192, maxc=750), 4096)
print("build s", time.perf_counter() - t0)
qh = queries.cpu().numpy().view(np.uint16)
_, truth = s.bruteforce_topk(qh, 10)
e_idx = np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32)
mse.set_entries(g, vecs, e_idx)
for nq in (256, 1024, 2048, 4096, 8192):
    for L in (32,):
        q = qh[:nq]
        def run():
            return mse.disk_query_topk(s, None, None, g, q, 10, None, None, None, True, 4, L)
        run()
        reps, ts = 4, []
        for _ in range(reps):
            t0 = time.perf_counter()
            top, _, stats = run()
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[reps // 2 - 1]          # lower median: a call that had to grow a device buffer does not set the figure
        rec = sum(len(set(top[i].tolist()) & set(truth[i].tolist())) for i in range(nq)) / (10 * nq)
        print(f"queries per call {nq:5d} L {L}: {nq / dt:9.0f} queries/s, recall@10 {rec:.4f}, {dt * 1e3:.2f} ms per call, {float(stats['cmps'].mean()):.1f} node fetches per query; calls ms " + " ".join(f"{t * 1e3:.2f}" for t in ts))
