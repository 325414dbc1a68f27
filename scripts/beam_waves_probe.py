"""Developer probe (needs a GPU): latency of small request-path calls on the HARD set by waves per query (MSE_BEAM_WAVES = 4 / 8 / 16,
an answer-preserving hook of the product library).  Run scripts/beam_hard_probe.py with the same row count first (it caches the graph in
/tmp).  python scripts/beam_waves_probe.py [rows]  ->  median ms per call by batch size; a checksum of the answers (must not change)."""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
import bench_ann as ba  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
hs = ba.HardSet(n, **ba.HARD_PARAMS)
rows, queries = hs.rows(n, 1), hs.rows(4096, 3)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
z = np.load(f"/tmp/beam_hard_graph_{n}.npz")
g = mse.DeviceGraph(mse.IndexGraph(z["adj"], z["deg"]))
mse.set_entries(g, vecs, np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32))
q16 = queries.cpu().numpy().view(np.uint16)
print("MSE_BEAM_WAVES =", os.environ.get("MSE_BEAM_WAVES", "(default)"), flush=True)
for L in (32, 200):
    for nb in (1, 4, 16, 64, 256, 1024):
        ids, sc, st = mse.disk_query_topk(s, None, None, g, q16[:nb], 10, None, None, None, True, 4, L)
        ts = []
        for i in range(12):
            qq = q16[(i * nb) % 2048:(i * nb) % 2048 + nb]
            t0 = time.perf_counter()
            mse.disk_query_topk(s, None, None, g, qq, 10, None, None, None, True, 4, L)
            ts.append((time.perf_counter() - t0) * 1e3)
        ck = zlib.crc32(ids.tobytes() + sc.tobytes() + st["cmps"].tobytes() + st["n_visited"].tobytes())
        print(f"L {L:4d}  nq {nb:5d}  ms {sorted(ts)[6]:8.3f}  checksum {ck:08x}", flush=True)
