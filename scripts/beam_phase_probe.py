"""Developer probe (needs a GPU and the probe library: `make -C meme-search-engine_amd/csrc phases` -> lib/libmse_hip_phases.so, loaded through MSE_HIP_LIB): where a beam iteration's time goes,
phase by phase (100 MHz wall-clock stamps by thread 0 of every search, summed by the measurement hook).  Uses the graph cached by
scripts/beam_hard_probe.py.  MSE_HIP_LIB=.../libmse_hip_phases.so python scripts/beam_phase_probe.py [rows]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
from mse import ffi  # noqa: E402
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
sel = torch.from_numpy(np.sort(np.random.default_rng(4).choice(n, 100_000, replace=False))).cuda()
cents, T, _ = ba.train_codec_aopq(rows[sel].float(), hs.rows(50_000, 7).float(), rounds=1, iters=40)
pq = mse.ProductQuantizer(cents, T, 18, ba.D)
codes = mse.Codes.quantize_base(pq, vecs)
q32 = queries.float().cpu().numpy()
names = ["select next nodes", "adjacency + fetched rows' exact scores + visited", "first positions (LDS table)", "visited_adjacent inserts", "pre-buffer compaction + visited list",
         "neighbour scores (ADC / exact)", "merge / replay"]


def hook(enable):
    out = (C.c_uint64 * 16)()
    ffi.check(ffi.lib().mse_searcher_beam_timing(s._h, enable, C.cast(out, C.POINTER(C.c_uint64))), "beam_timing")
    return list(out)


for label, args, nq in (("ADC L 400, 4096 queries (four waves per query, two queries per CU)", (pq, codes, False, 400), 4096),
                        ("ADC L 400, 64 queries", (pq, codes, False, 400), 64),
                        ("exact L 200, 64 queries (four waves per query)", (None, None, True, 200), 64),
                        ("exact L 200, 4096 queries (one wave per query, 14 per CU)", (None, None, True, 200), 4096)):
    a = (s, args[0], args[1], g, q32[:nq], 10, None, None, None, args[2], 4, args[3])
    mse.disk_query_topk(*a)
    hook(2)
    t0 = time.perf_counter()
    mse.disk_query_topk(*a)
    dt = time.perf_counter() - t0
    o = hook(0)
    its = o[6]
    print(f"# {label}: call {dt * 1e3:.2f} ms, kernel {o[0] / 1e3:.2f} ms, {its / o[2]:.1f} iterations per query; per iteration (us, thread 0's view):", flush=True)
    tot = sum(o[8:15])
    for k, nm in enumerate(names):
        print(f"    {o[8 + k] / its / 100.0:8.2f}  {100.0 * o[8 + k] / max(1, tot):5.1f} %  {nm}", flush=True)
    print(f"    {tot / its / 100.0:8.2f}  total", flush=True)
