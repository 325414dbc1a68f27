#!/usr/bin/env python3
"""What profiles/r05_pq_scan_stats.txt is made from (run under rocprofv3 --kernel-trace): the PQ flat scan at 1e8 codes in the two
patterns bench.py's pq_scan.roofline reports.  python scripts/pq_trace_r05.py burst|sustained
  burst      40 calls of EIGHT queries: one scan per call on ONE stream, nothing before or beside it
  sustained  16 calls of 64 queries: eight scans per call back to back, alternating between two streams, tails beside them"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import mse  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "burst"
D, n = 1152, 100_000_000
rng = np.random.default_rng(0)
cents = (rng.standard_normal((256, D)) / np.sqrt(D)).astype(np.float32)
T = np.linalg.qr(rng.standard_normal((D, D)))[0].astype(np.float32)
pq = mse.ProductQuantizer(cents, T, 18, D)
blk = 1_000_000
block = rng.integers(0, 256, size=(blk, 64), dtype=np.uint8)
codes = np.empty((n, 64), np.uint8)
for c0 in range(0, n, blk):
    np.bitwise_xor(block, rng.integers(0, 256, size=64, dtype=np.uint8), out=codes[c0:c0 + blk])
desc = np.resize(rng.integers(0, 256, size=(blk, 4), dtype=np.uint8), (n, 4))
gc = mse.Codes(codes, desc)
del codes
scales = np.array([0.5, 0, -0.25, 0], np.float32) / np.float32(512)
qs = (rng.standard_normal((64, D)) / np.sqrt(D)).astype(np.float32)
pq.scan_timing(2)
if mode == "burst":
    for i in range(42):
        pq.scan_topk_batch(gc, qs[8 * (i % 8):8 * (i % 8) + 8], 200, 10, None, scales)
    ms, k = pq.scan_timing(0)
    print(f"HIP events: {k} scan launches, {ms / k:.4f} ms each")
else:
    for i in range(18):
        pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
    span, k = pq.scan_sustained()
    print(f"HIP events: {k} scans in calls of eight, {span / k:.4f} ms per scan from first start to last end")
