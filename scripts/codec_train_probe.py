"""Developer probe (needs a GPU): what the bench-side AOPQ trainer's knobs buy -- PQ-only recall@10 and the share of the exact top-10 inside the
ADC top-200 on the hard / ood sets at 1e6 rows, for a few (rounds, Adam steps, learning rate) settings.  python scripts/codec_train_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mse  # noqa: E402
import bench_ann as ba  # noqa: E402

n = 1_000_000
hs = ba.HardSet(n, **ba.HARD_PARAMS)
rows = hs.rows(n, 1)
vecs = mse.VectorList.wrap_device(rows.data_ptr(), n, ba.D, keepalive=rows)
s = mse.Searcher(vecs)
sel = torch.from_numpy(np.sort(np.random.default_rng(4).choice(n, 100_000, replace=False))).cuda()
samp = rows[sel].float()
for kind in ("hard", "ood"):
    if kind == "ood":
        train_q = hs.rows(50_000, 4, queries="ood", gap=ba.OOD_GAP, extra_noise=ba.OOD_EXTRA_NOISE).float()
        q = hs.rows(1024, 2, queries="ood", gap=ba.OOD_GAP, extra_noise=ba.OOD_EXTRA_NOISE)
    else:
        train_q = hs.rows(50_000, 7).float()
        q = hs.rows(1024, 2)
    q32 = q.float().cpu().numpy()
    _, truth = s.bruteforce_topk(q.cpu().numpy().view(np.uint16), 10)
    for rounds, iters, lr in ((0, 0, 0), (3, 120, 5e-4), (6, 120, 5e-4), (3, 300, 5e-4), (6, 200, 1e-3), (10, 100, 5e-4)):
        t0 = time.perf_counter()
        if rounds == 0:
            cents, T = ba.train_codec(samp[:20000].cpu().numpy())
            info = {}
        else:
            cents, T, info = ba.train_codec_aopq(samp, train_q, rounds=rounds, iters=iters, lr=lr)
        dt = time.perf_counter() - t0
        pq = mse.ProductQuantizer(cents, T, 18, ba.D)
        codes = mse.Codes.quantize_base(pq, vecs)
        top10 = np.concatenate([pq.scan_topk_batch(codes, q32[i:i + 64], 10, 10, None)[1] for i in range(0, 1024, 64)])
        top200 = np.concatenate([pq.scan_topk_batch(codes, q32[i:i + 64], 200, 200, None)[1] for i in range(0, 1024, 64)])
        inside = sum(len(set(top200[i].tolist()) & set(truth[i].tolist())) for i in range(1024)) / 10240
        last = (info.get("query_aware_loss_first_last_per_round") or [None])[-1]
        print(f"{kind}: rounds {rounds} x {iters} steps, lr {lr}: trained in {dt:5.1f} s, loss {last}, PQ-only recall@10 {ba.recall_at(top10, truth):.4f}, exact top-10 inside ADC top-200 {inside:.4f}", flush=True)
