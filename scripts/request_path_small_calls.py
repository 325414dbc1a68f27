"""Developer probe (needs a GPU): latency of small request-path calls (mse_disk_query_topk_f32) by batch size and entry rule.
python scripts/request_path_small_calls.py [rows] [only_nq]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
import bench_ann as ba  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
build = only <= 0          # a negative batch size: that batch only, but on a built graph (for traces)
only = abs(only)
K, R, L = 10, 64, 12
gen = ba.easy_generator(n)
rows, queries = gen(n, 1), gen(4096, 2)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
g = mse.BuildGraph(n, R)
g.random_fill(1)
if build:
    med = mse.medioid(vecs)
    g.build(s, np.random.default_rng(3).permutation(n).astype(np.uint32), med, mse.IndexBuildConfig(r=R, l=192, maxc=750), 4096)
qf = queries.float().cpu().numpy()
e_idx = np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32)
cen = rows[torch.from_numpy(e_idx[:256].astype(np.int64)).cuda()].float().cpu().numpy()
for rule in ("rows", "centroids"):
    if rule == "rows":
        mse.set_entries(g, vecs, e_idx)
    else:
        mse.set_entry_centroids(g, cen, e_idx[:256])
    for nb in ([only] if only else [1, 8, 16, 17, 64, 256, 1024, 4096]):
        mse.disk_query_topk(s, None, None, g, qf[:nb], K, None, None, None, True, 4, L)
        t0 = time.perf_counter()
        reps = 5 if only else 30
        for _ in range(reps):
            mse.disk_query_topk(s, None, None, g, qf[:nb], K, None, None, None, True, 4, L)
        print(f"entry by {rule:9s}: call of {nb:5d} f32 queries: {(time.perf_counter() - t0) / reps * 1e3:7.3f} ms", flush=True)
