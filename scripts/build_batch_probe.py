"""Developer probe (needs a GPU): build rate and graph quality against the batch size of the batched Vamana build.
python scripts/build_batch_probe.py [kind] [rows] [batch ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
import bench_ann as ba  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "easy"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
batches = [int(x) for x in sys.argv[3:]] or [4096, 65536, 262144]
K, R, nq = 10, 64, 2048
if kind == "easy":
    gen = ba.easy_generator(n)
    rows, queries = gen(n, 1), gen(nq, 2)
else:
    hs = ba.HardSet(n, **ba.HARD_PARAMS)
    rows, queries = hs.rows(n, 1), hs.rows(nq, 2)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
qh = queries.cpu().numpy().view(np.uint16)
_, truth = s.bruteforce_topk(qh, K)
med = mse.medioid(vecs)
perm = np.random.default_rng(3).permutation(n).astype(np.uint32)
e_idx = np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32)
for batch in batches:
    g = mse.BuildGraph(n, R)
    g.random_fill(1)
    t0 = time.perf_counter()
    g.build(s, perm, med, mse.IndexBuildConfig(r=R, l=192, maxc=750), batch)
    dt = time.perf_counter() - t0
    mse.set_entries(g, vecs, e_idx)
    rec = {}
    for L in ((12, 16, 32) if kind == "easy" else (64, 100, 200)):
        top, _, _ = mse.disk_query_topk(s, None, None, g, qh, K, None, None, None, True, 4, L)
        rec[L] = round(ba.recall_at(top, truth), 4)
    print(json.dumps({"kind": kind, "rows": n, "batch": batch, "build_s": round(dt, 2), "points_per_s": round(n / dt), "recall_by_L": rec}), flush=True)
    g.close()
