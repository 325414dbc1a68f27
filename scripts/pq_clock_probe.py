#!/usr/bin/env python3
"""Developer probe (needs a GPU): why do back-to-back PQ scans take 8-10 % longer than scans with a pause in front?  Engine / memory clock
and socket power sampled every ~10 ms from sysfs (hwmon freq*_input, power1_average / power1_input) beside three patterns at 1e8 codes:
  burst    8-query calls (one scan each), a host round trip between scans
  sustained 64-query calls (eight scans back to back on two streams)
  paced    64-query calls' worth of scans issued as 8-query calls with a 2 ms sleep between them
python scripts/pq_clock_probe.py [seconds]"""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "meme-search-engine_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa: F401
import mse

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
D, n = 1152, 100_000_000


def find_sensors():
    out = {}
    pr = torch.cuda.get_device_properties(0)
    pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)      # OUR device, not the first card in /sys
    print("# device 0 =", pr.name, "at", pci, flush=True)
    for hw in glob.glob(f"/sys/bus/pci/devices/{pci}/hwmon/hwmon*"):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"):
            f = os.path.join(hw, name)
            if os.path.exists(f):
                lab = f.replace("_input", "_label").replace("_average", "_label")
                try:
                    label = open(lab).read().strip()
                except Exception:  # noqa: BLE001
                    label = name
                out.setdefault(name, (f, label))
    return out


sensors = find_sensors()
print("# sensors:", {k: v for k, v in sensors.items()}, flush=True)
for f in []:
    try:
        print("#", f, open(f).read().replace("\n", " | "))
    except Exception as e:  # noqa: BLE001
        print("#", f, repr(e))


def read_all():
    r = {}
    for k, (f, _) in sensors.items():
        try:
            r[k] = int(open(f).read())
        except Exception:  # noqa: BLE001
            pass
    return r


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


def pattern(name, body):
    for _ in range(3):
        body()
    stop, samples = threading.Event(), []

    def watch():
        while not stop.is_set():
            samples.append((time.perf_counter(), read_all()))
            time.sleep(0.01)

    th = threading.Thread(target=watch)
    th.start()
    pq.scan_timing(2)
    t0, scans = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        scans += body()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    ms, launches = pq.scan_timing(0)
    span, span_n = pq.scan_sustained()
    keys = sorted({k for _, r in samples for k in r})
    stat = {}
    for k in keys:
        v = np.array([r[k] for _, r in samples if k in r], np.float64)
        stat[k] = (v.mean(), v.min(), v.max())
    desc_s = "; ".join(f"{sensors[k][1]} mean {m / (1e6 if 'freq' in k or 'power' in k else 1e3):.0f} min {lo / (1e6 if 'freq' in k or 'power' in k else 1e3):.0f} max {hi / (1e6 if 'freq' in k or 'power' in k else 1e3):.0f}"
                       for k, (m, lo, hi) in stat.items())
    print(f"{name:10s}: {scans / dt:7.1f} scans/s; scan kernel by HIP events {ms / max(launches, 1):6.3f} ms over {launches} launches"
          + (f"; sustained span {span / max(span_n, 1):6.3f} ms per scan over {span_n}" if span_n else "") + f"; {len(samples)} samples: {desc_s}", flush=True)


def burst():
    pq.scan_topk_batch(gc, qs[:8], 200, 10, None, scales)
    return 1


def sustained():
    pq.scan_topk_batch(gc, qs, 200, 10, None, scales)
    return 8


def paced():
    for i in range(8):
        pq.scan_topk_batch(gc, qs[:8], 200, 10, None, scales)
        time.sleep(0.002)
    return 8


print(f"# scripts/pq_clock_probe.py {secs}: one MI355X, {n} codes; units MHz / W / C")
for name, body in (("burst", burst), ("sustained", sustained), ("paced", paced), ("sustained", sustained)):
    pattern(name, body)
