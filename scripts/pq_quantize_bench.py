"""quantize_batch (diskann/src/vector.rs:331-364) throughput: 8192-row batches of 1152-d f32 through the C ABI (host arrays in, codes out)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch  # noqa: F401
import mse
D = 1152
rng = np.random.default_rng(0)
cents = (rng.standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
T = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
pq = mse.ProductQuantizer(cents, T, 18, D)
x = (rng.standard_normal((8192, D)) / np.sqrt(D)).astype(np.float32)
c0 = pq.quantize_batch(x)
t0 = time.perf_counter()
for _ in range(10):
    c = pq.quantize_batch(x)
dt = (time.perf_counter() - t0) / 10
print(f"quantize_batch 8192 x {D}: {dt*1e3:.2f} ms per batch = {8192/dt/1e6:.2f} M vectors/s; checksum {int(c.astype(np.uint64).sum())}")
