#!/usr/bin/env python3
"""Engine clock and socket power beside the batched PQ scan at 1e8 codes (rocm-smi every 0.5 s): 64-query calls back to back for a
few seconds, and the same number of 8-query calls (one scan, then its tail with nothing beside it).  python scripts/pq_power_probe.py [seconds]"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa: F401
import mse

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
D, n = 1152, 100_000_000


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        pw = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
        clk = {}
        for k, v in card.items():
            if k.startswith(("sclk", "mclk", "fclk")):
                clk[k.split()[0]] = int("".join(ch for ch in str(v).split("Mhz")[0].split("(")[-1] if ch.isdigit()))
        return clk, pw
    except Exception:  # noqa: BLE001
        return {}, None


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
print(f"# scripts/pq_power_probe.py {secs}: one MI355X, {n} codes of 64 B + 4 descriptor bytes, top-200 -> top-10; rocm-smi every 0.5 s")
for per_call in (64, 8):
    qs = (rng.standard_normal((per_call, D)) / np.sqrt(D)).astype(np.float32)
    for _ in range(3):
        pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
    stop, samples = threading.Event(), []

    def watch():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.5)

    th = threading.Thread(target=watch)
    th.start()
    pq.scan_timing(2)
    t0, calls = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
        calls += 1
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    ms, launches = pq.scan_timing(0)
    good = [x for x in samples[1:] if x[0]]
    keys = sorted({k for c, _ in good for k in c})
    clk = {k: sum(c.get(k, 0) for c, _ in good) / max(len(good), 1) for k in keys}
    pw = [p for _, p in good if p]
    print(f"{per_call:3d} queries per call: {per_call * calls / dt:7.1f} queries/s, {dt / calls * 1e3:7.3f} ms per call, scan kernel (HIP events) {ms / max(launches, 1):6.3f} ms "
          f"= {n * 68 / (ms / max(launches, 1)) / 1e6:5.0f} GB/s over {launches} launches; clocks MHz " + ", ".join(f"{k} {v:.0f}" for k, v in clk.items()) +
          f"; socket {sum(pw) / max(len(pw), 1):5.0f} W ({len(good)} samples)")
