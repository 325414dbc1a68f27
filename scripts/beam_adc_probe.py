"""Developer probe (needs a GPU): the reference's default search (neighbours scored by ADC, fetched nodes exactly) on the HARD set with
the trained codec -- queries/s, recall@10 and what the kernel did (iterations, the share replayed sequentially, bytes gathered) at a few
search lists.  Uses the graph cached by scripts/beam_hard_probe.py.  python scripts/beam_adc_probe.py [rows] [L ...]"""
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
Ls = [int(x) for x in sys.argv[2:]] or [200, 400, 600]
beams = [int(x) for x in os.environ.get("BEAMS", "4").split(",")]
hs = ba.HardSet(n, **ba.HARD_PARAMS)
rows, queries = hs.rows(n, 1), hs.rows(4096, 3)
torch.cuda.synchronize()
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
z = np.load(f"/tmp/beam_hard_graph_{n}.npz")
g = mse.DeviceGraph(mse.IndexGraph(z["adj"], z["deg"]))
mse.set_entries(g, vecs, np.sort(np.random.default_rng(5).choice(n, max(4096, n // 1500), replace=False)).astype(np.uint32))
t0 = time.perf_counter()
sel = torch.from_numpy(np.sort(np.random.default_rng(4).choice(n, 100_000, replace=False))).cuda()
cents, T, info = ba.train_codec_aopq(rows[sel].float(), hs.rows(50_000, 7).float())
pq = mse.ProductQuantizer(cents, T, 18, ba.D)
codes = mse.Codes.quantize_base(pq, vecs)
print("# hard set, %d rows; codec trained + codes made in %.1f s; 4096 held-out f32 queries per call, beam 4" % (n, time.perf_counter() - t0), flush=True)
q32 = queries.float().cpu().numpy()
_, truth = s.bruteforce_topk(queries.cpu().numpy().view(np.uint16), 10)
for exact, beam in [(e, b) for e in (False, True) for b in beams]:
    for L in Ls:
        args = (s, None, None, g, q32, 10, None, None, None, True, beam, L) if exact else (s, pq, codes, g, q32, 10, None, None, None, False, beam, L)
        mse.disk_query_topk(*args)
        s.beam_timing(2)
        t0 = time.perf_counter()
        top, _, st = mse.disk_query_topk(*args)
        dt = time.perf_counter() - t0
        m = s.beam_timing(0)
        byt = m["rows_scored"] * ba.D * 2 + m["nodes_fetched"] * 260 + m["adc_scored"] * 68
        print("%s beam %d L %4d: %8.0f queries/s, recall@10 %.4f; kernel %.2f ms; per query %.1f iterations (%.1f %% replayed), %.0f rows + %.0f codes gathered = %.2f MB -> %.0f GB/s"
              % ("exact" if exact else "ADC  ", beam, L, 4096 / dt, ba.recall_at(top, truth), m["kernel_ms"], m["iterations"] / m["queries"],
                 100.0 * m["iterations_replayed"] / max(1, m["iterations"]), m["rows_scored"] / m["queries"], m["adc_scored"] / m["queries"],
                 byt / m["queries"] / 1e6, byt / (m["kernel_ms"] * 1e-3) / 1e9), flush=True)
