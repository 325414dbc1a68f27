#!/usr/bin/env python3
"""Engine clock and socket power beside the SigLIP image tower (batch 256, depth 27, rocm-smi every 0.5 s), and beside its largest GEMM
alone with constant / random operands.  python scripts/siglip_power_probe.py [seconds]"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import torch
import mse  # noqa: F401
from mse import siglip

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0


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


def watched(fn, label, unit_per_call, unit):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []

    def watch():
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.5)

    th = threading.Thread(target=watch)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        fn()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    good = [x for x in samples[1:] if x[0]]
    clk = sum(x[0] for x in good) / max(len(good), 1)
    pws = [x[1] for x in good if x[1]]
    print(f"{label}: {unit_per_call * n / dt:8.1f} {unit}, {dt / n * 1e3:8.2f} ms per call; sclk {clk:5.0f} MHz, socket {sum(pws) / max(len(pws), 1):5.0f} W ({len(good)} samples)")


print(f"# scripts/siglip_power_probe.py {secs}: one MI355X; rocm-smi every 0.5 s")
batch, depth = 256, 27
cfg = dict(siglip.SO400M_384, depth=depth)
eng = siglip.SiglipImageEngine.from_state_dict(siglip.synthetic_state_dict(cfg), cfg, max_batch=batch)
img = torch.empty((batch, 3, 384, 384), dtype=torch.float16, device="cuda").uniform_(-1, 1)
watched(lambda: eng.encode_image_device(img.data_ptr(), batch), f"SigLIP SO400M/14-384 image tower, batch {batch}", batch, "img/s")
M, N, K = 256 * 729, 4352, 1152
for kind in ("random", "constant"):
    a = torch.empty((M, K), dtype=torch.bfloat16, device="cuda")
    w = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    if kind == "random":
        a.uniform_(-1, 1); w.uniform_(-1, 1)
    else:
        a.fill_(1.0); w.fill_(1.0)
    watched(lambda: torch.mm(a, w.t()), f"library GEMM {M} x {N} x {K} bf16, {kind} operands", 2.0 * M * N * K / 1e12, "TFLOP/s")
