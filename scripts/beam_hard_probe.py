"""Developer probe (needs a GPU): the exactly scored request path on the HARD set at its operating point (L = 200, beam 4, 4096 queries
per call) -- queries/s, recall@10, and the gather roofline from the searcher's own measurement hook (mse_searcher_beam_timing: HIP events
around beam_search_kernel + what the searches gathered).  The graph is cached in /tmp between runs of one gpurun call, so that the
rocprofv3 passes of scripts/prof_beam_r06.sh do not rebuild it.  python scripts/beam_hard_probe.py [rows] [L] [beam] [calls]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
import bench_ann as ba  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
beam = int(sys.argv[3]) if len(sys.argv) > 3 else 4
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 4
hs = ba.HardSet(n, **ba.HARD_PARAMS)
rows, queries = hs.rows(n, 1), hs.rows(4096, 3)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
cache = f"/tmp/beam_hard_graph_{n}.npz"
t0 = time.perf_counter()
if os.path.exists(cache):
    z = np.load(cache)
    g = mse.DeviceGraph(mse.IndexGraph(z["adj"], z["deg"]))
    how = "loaded from " + cache
else:
    med = mse.medioid(vecs)
    g = mse.BuildGraph(n, 64)
    g.random_fill(1)
    order = np.random.default_rng(3).permutation(n).astype(np.uint32)
    g.build(s, order, med, mse.IndexBuildConfig(r=64, l=192, maxc=750), 16384)
    h = g.to_host()
    np.savez(cache, adj=h.adj, deg=h.deg)
    how = "built"
print("# hard set, %d rows, graph %s in %.1f s; calls of 4096 held-out f16 queries, L = %d, beam %d" % (n, how, time.perf_counter() - t0, L, beam), flush=True)
mse.set_entries(g, vecs, np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32))
q16 = queries.cpu().numpy().view(np.uint16)
_, truth = s.bruteforce_topk(q16, 10)
mse.disk_query_topk(s, None, None, g, q16, 10, None, None, None, True, beam, L)
s.beam_timing(2)
t0 = time.perf_counter()
for _ in range(calls):
    top, _, st = mse.disk_query_topk(s, None, None, g, q16, 10, None, None, None, True, beam, L)
dt = (time.perf_counter() - t0) / calls
m = s.beam_timing(0)
rows_b, adj_b = m["rows_scored"] * ba.D * 2, m["nodes_fetched"] * (64 * 4 + 4)
gbps = (rows_b + adj_b) / (m["kernel_ms"] * 1e-3) / 1e9
print("queries/s %.0f (call %.2f ms), recall@10 %.4f, node fetches per query %.1f" % (4096 / dt, dt * 1e3, ba.recall_at(top, truth), float(st["cmps"].mean())), flush=True)
print("beam_search_kernel: %d launches, %.3f ms each = %.0f queries/s of the kernel alone; per query %.1f rows scored exactly (%.1f KB of %d-byte row gathers) + %.1f adjacency lists (%.1f KB)"
      % (m["launches"], m["kernel_ms"] / m["launches"], m["queries"] / (m["kernel_ms"] * 1e-3), m["rows_scored"] / m["queries"], rows_b / m["queries"] / 1e3, ba.D * 2,
         m["nodes_fetched"] / m["queries"], adj_b / m["queries"] / 1e3), flush=True)
print("beam iterations per query %.1f, of which replayed sequentially (equal scores in play) %.2f %%" % (m["iterations"] / m["queries"], 100.0 * m["iterations_replayed"] / max(1, m["iterations"])), flush=True)
print("gather roofline: %.1f MB algorithmic per 4096-query launch / %.3f ms = %.0f GB/s = %.3f of the 8 TB/s HBM peak" %
      ((rows_b + adj_b) / m["launches"] / 1e6, m["kernel_ms"] / m["launches"], gbps, gbps / 8000.0), flush=True)
