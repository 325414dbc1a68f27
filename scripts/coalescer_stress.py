#!/usr/bin/env python3
"""Randomised stress of the cross-thread coalescer (not part of the test suite): `threads` Python threads issue random requests --
dispatcher searches of 1..40 queries with random k, AUTO-mode searches on per-thread searchers, index searches racing index adds,
one-query PQ scans -- for `seconds`, every answer checked against precomputed batched answers.
  python scripts/coalescer_stress.py [seconds] [threads]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402
import mse  # noqa: E402

D = 1152
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 96
rng = np.random.default_rng(0)
n = 60000
vl = mse.VectorList.generate(0x5EED0001, 0, n, D)
qd = mse.VectorList.generate(0x5EED0002, 0, 4096, D)
q = qd.rows(0, 4096)
ref = mse.Searcher(vl)
KMAX = 64
want_s, want_i = ref.bruteforce_topk(q, KMAX, mse.MODE_MFMA)
disp = mse.Dispatcher(vl)
searchers = [mse.Searcher(vl) for _ in range(T)]
# index: rows added in four stages; answers for every stage precomputed on a separate index
d2 = 256
x = (rng.standard_normal((8000, d2)) / np.sqrt(d2)).astype(np.float32)
qi = rng.standard_normal((512, d2)).astype(np.float32)
stages = [2000, 4000, 6000, 8000]
wants = []
for m in stages:
    tmp = mse.ScalarQuantizerIndex(d2)
    tmp.add(x[:m])
    r = tmp.search(qi, 5)
    wants.append((r.labels.copy(), r.distances.copy()))
    tmp.close()
idx = mse.ScalarQuantizerIndex(d2)
idx.add(x[:stages[0]])
stage = [0]
# pq
cents = (rng.standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
Tm = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
pq = mse.ProductQuantizer(cents, Tm, 18, D)
codes_h = rng.integers(0, 256, (n, 64), dtype=np.uint8)
codes = mse.Codes(codes_h, None)
qp = (rng.standard_normal((64, D)) / np.sqrt(D)).astype(np.float32)
want_pq = pq.scan_topk_batch(codes, qp, 100, 10)
errors, counts = [], [0] * T
stop = time.time() + secs


def worker(t):
    r = np.random.default_rng(100 + t)
    try:
        while time.time() < stop:
            op = r.integers(0, 10)
            if op < 4:
                m, k = int(r.integers(1, 41)), int(r.integers(1, KMAX + 1))
                lo = int(r.integers(0, 4096 - m))
                sc, ids = disp.search(q[lo:lo + m], k)
                assert np.array_equal(ids, want_i[lo:lo + m, :k]) and np.array_equal(sc, want_s[lo:lo + m, :k]), ("dispatcher", lo, m, k)
            elif op < 6:
                lo, k = int(r.integers(0, 4096)), int(r.integers(1, 20))
                sc, ids = searchers[t].bruteforce_topk(q[lo], k)
                assert np.array_equal(ids[0], want_i[lo, :k]) and np.array_equal(sc[0], want_s[lo, :k]), ("auto", lo, k)
            elif op < 8:
                lo, m = int(r.integers(0, 500)), int(r.integers(1, 12))
                before = stage[0]
                res = idx.search(qi[lo:lo + m], 5)
                after = stage[0]
                ok = any(np.array_equal(res.labels, wants[s_][0][lo:lo + m]) and np.array_equal(res.distances, wants[s_][1][lo:lo + m])
                         for s_ in range(before, min(after + 1, len(stages) - 1) + 1))
                assert ok, ("index", lo, m, before, after)
            elif op < 9:
                j = int(r.integers(0, 64))
                sc, ids = pq.scan_topk(codes, qp[j], 100, 10)
                assert np.array_equal(ids, want_pq[1][j]) and np.array_equal(sc, want_pq[0][j]), ("pq", j)
            else:
                if t == 0 and stage[0] < len(stages) - 1 and r.random() < 0.05:
                    s_ = stage[0]
                    idx.add(x[stages[s_]:stages[s_ + 1]])
                    stage[0] = s_ + 1
            counts[t] += 1
    except BaseException as e:  # noqa: BLE001
        errors.append(repr(e))


ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for th in ths:
    th.start()
for th in ths:
    th.join()
print("requests", sum(counts), "errors", len(errors), errors[:3], "dispatcher", disp.stats(), "index", idx.stats(), "stage", stage[0])
sys.exit(1 if errors else 0)
