"""Developer probe (needs a GPU): beam width of the exactly scored request path on the HARD set -- queries/s and recall@10 of one
4096-query call for beam 1..8 at several search lists.  python scripts/beam_width_probe.py [rows]"""
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
hs = ba.HardSet(n, **ba.HARD_PARAMS)
rows, queries = hs.rows(n, 1), hs.rows(4096, 2)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
med = mse.medioid(vecs)
g = mse.BuildGraph(n, 64)
g.random_fill(1)
order = np.random.default_rng(3).permutation(n).astype(np.uint32)
t0 = time.perf_counter()
g.build(s, order, med, mse.IndexBuildConfig(r=64, l=192, maxc=750), 16384)
print("# hard set, %d rows, built in %.1f s; one call of 4096 f32 queries" % (n, time.perf_counter() - t0), flush=True)
mse.set_entries(g, vecs, np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32))
qf = queries.float().cpu().numpy()
_, truth = s.bruteforce_topk(queries.cpu().numpy().view(np.uint16), 10)
print("beam  L    queries/s  recall@10  node fetches/query", flush=True)
for beam in (1, 2, 4, 8):
    for L in (100, 150, 200, 300):
        mse.disk_query_topk(s, None, None, g, qf, 10, None, None, None, True, beam, L)
        t0 = time.perf_counter()
        top, _, st = mse.disk_query_topk(s, None, None, g, qf, 10, None, None, None, True, beam, L)
        dt = time.perf_counter() - t0
        print("%4d %4d %10.0f %9.4f %10.1f" % (beam, L, 4096 / dt, ba.recall_at(top, truth), float(st["cmps"].mean())), flush=True)
