"""ADC scan throughput (BASELINE config 5 shape): N x 64-byte PQ codes + 4 descriptor bytes, top-r by ADC, no re-score."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch  # noqa: F401  (HIP runtime order)
import mse

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
rng = np.random.default_rng(0)
D = 1152
cents = (rng.standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
T = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
pq = mse.ProductQuantizer(cents, T, 18, D)
codes = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
desc = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
gc = mse.Codes(codes, desc)
scales = np.array([0.5, 0, -0.25, 0], np.float32) / np.float32(512)
q = rng.standard_normal(D).astype(np.float32) / np.sqrt(D)
pq.scan_topk(gc, q, 200, 10, None, scales)
t0 = time.perf_counter()
it = 10
for _ in range(it):
    pq.scan_topk(gc, q, 200, 10, None, scales)
dt = (time.perf_counter() - t0) / it
print(f"n={n}: {dt*1e3:.2f} ms per query scan, {n*68/dt/1e9:.0f} GB/s of codes+descriptors, {1/dt:.1f} q/s")
if len(sys.argv) > 2:   # batched: queries go through in fours that share one pass over the codes
    nb = int(sys.argv[2])
    qs = (rng.standard_normal((nb, D)) / np.sqrt(D)).astype(np.float32)
    pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
    t0 = time.perf_counter()
    for _ in range(3):
        pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
    db = (time.perf_counter() - t0) / (3 * nb)
    print(f"n={n}: batched ({nb} per call, 4 per pass, {pq.last_uncertified} uncertified) {db*1e3:.3f} ms per query, {1/db:.1f} q/s, {n*68/(4*db)/1e9:.0f} GB/s of codes+descriptors per pass")
