"""Developer probe (needs a GPU; developer library for MSE_BEAM_FOUR_WAVES): latency of a request-path call by batch size and search
list, one wave per query against four.  MSE_HIP_LIB=.../libmse_hip_dev.so [MSE_BEAM_FOUR_WAVES=1] python scripts/beam_latency_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
import bench_ann as ba  # noqa: E402

n = 2_000_000
hs = ba.HardSet(n, **ba.HARD_PARAMS)
rows, queries = hs.rows(n, 1), hs.rows(4096, 2)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
g = mse.BuildGraph(n, 64)
g.random_fill(1)
g.build(s, np.random.default_rng(3).permutation(n).astype(np.uint32), mse.medioid(vecs), mse.IndexBuildConfig(r=64, l=192, maxc=750), 4096)
mse.set_entries(g, vecs, np.sort(np.random.default_rng(5).choice(n, 4096, replace=False)).astype(np.uint32))
qf = queries.float().cpu().numpy()
print("four waves" if os.environ.get("MSE_BEAM_FOUR_WAVES") else "one wave", flush=True)
for L in (12, 64, 200):
    for nb in (17, 64, 256, 512, 1024, 2048, 4096):
        mse.disk_query_topk(s, None, None, g, qf[:nb], 10, None, None, None, True, 4, L)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter()
            mse.disk_query_topk(s, None, None, g, qf[:nb], 10, None, None, None, True, 4, L)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"L {L:4d}  nq {nb:5d}  ms {sorted(ts)[4]:8.3f}", flush=True)
