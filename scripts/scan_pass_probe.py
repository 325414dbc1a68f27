#!/usr/bin/env python3
"""Queries per pass of the brute-force scan at the metric's size: 128 / 192 / 256 / 320 queries per pass over 1e8 x 1152 rows, each run
for a few seconds with the engine clock and socket power sampled beside it (rocm-smi).  One text block per point -> stdout.
  python scripts/scan_pass_probe.py [rows] [seconds per point]"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import mse  # noqa: E402

D = 1152
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        pw = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
        ck = next((v for k, v in card.items() if k.startswith("sclk")), None)
        mhz = int("".join(ch for ch in str(ck).split("Mhz")[0].split("(")[-1] if ch.isdigit())) if ck else None
        return mhz, pw
    except Exception:  # noqa: BLE001
        return None, None


vecs = mse.VectorList.generate(0x5EED0001, 0, rows, D)
s = mse.Searcher(vecs)
qs = mse.VectorList.generate(0x5EED0002, 0, 2048, D)
out_s = torch.empty((384, 10), dtype=torch.int64, device="cuda")
out_i = torch.empty((384, 10), dtype=torch.int32, device="cuda")
print(f"# scripts/scan_pass_probe.py {rows} {secs}: one MI355X, {rows} x {D} fp16 rows resident, top-10, matrix-core mode; sclk / socket power by rocm-smi every 0.5 s")
for nq in (128, 192, 256, 320):
    for i in range(3):
        s.bruteforce_topk_dev(qs.device_ptr, nq, 10, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []

    def watch():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.5)

    th = threading.Thread(target=watch)
    th.start()
    s.scan_timing(2)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        s.bruteforce_topk_dev(qs.device_ptr + (n % 4) * nq * D * 2, nq, 10, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    ms, launches = s.scan_timing(0)
    good = [x for x in samples[1:] if x[0]]
    clk = sum(x[0] for x in good) / len(good) if good else float("nan")
    pw = sum(x[1] for x in good if x[1]) / max(1, len([x for x in good if x[1]])) if good else float("nan")
    scan = ms / max(launches, 1)
    print(f"{nq:4d} queries/pass: {nq * n / dt:8.1f} queries/s, step {dt / n * 1e3:7.3f} ms, scan kernel {scan:7.3f} ms = {rows * D * 2 / scan / 1e6:6.0f} GB/s "
          f"({rows * D * 2 / scan / 1e6 / 8000:.3f} of 8 TB/s), {2 * rows * D * nq / scan / 1e9:6.0f} TFLOP/s; sclk {clk:5.0f} MHz, socket {pw:5.0f} W ({len(good)} samples)")
