"""Developer timing of the MFMA scan kernel alone (HIP events around the kernel): python scripts/scan_ablate.py [rows] [queries]
MSE_SCAN_ABL=<bits> selects an ablated variant of the 256-query kernel (scan_mfma.hip); results are then meaningless."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSE_HIP_LIB", os.path.join(ROOT, "meme-search-engine_amd", "lib", "libmse_hip_dev.so"))   # developer library (make dev)
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch  # noqa: F401,E402
import mse  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if os.environ.get("ZERO_X"):   # power / DVFS probe: same traffic, operands that do not toggle the multipliers
    zt = torch.zeros(n * 1152, dtype=torch.float16, device="cuda")
    if os.environ["ZERO_X"] == "2":
        zt.fill_(1.0)
    torch.cuda.synchronize()
    vl = mse.VectorList.wrap_device(zt.data_ptr(), n, 1152, keepalive=zt)
else:
    vl = mse.VectorList.generate(0x5EED0001, 0, n)
qs = mse.VectorList.generate(0x5EED0002, 0, nq)
s = mse.Searcher(vl)
out_s = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
out_i = torch.empty((nq, 10), dtype=torch.int32, device="cuda")
for _ in range(2):
    s.bruteforce_topk_dev(qs.device_ptr, nq, 10, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)
s.scan_timing(2)
t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
t0 = time.perf_counter()
for _ in range(5):
    s.bruteforce_topk_dev(qs.device_ptr, nq, 10, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5
ms, k = s.scan_timing(0)
print(f"2d={os.environ.get('MSE_SCAN_2D', '-')} abl={os.environ.get('MSE_SCAN_ABL', '0')} S={os.environ.get('MSE_SCAN_S', '3')} rows={n} nq={nq}: scan {ms / k:.3f} ms "
      f"({n * 2304 / (ms / k) / 1e6:.0f} GB/s), step {wall * 1e3:.3f} ms")
if os.environ.get("CHECK") and os.environ.get("MSE_SCAN_2D") != "162":   # the variant's answers against the exact-order kernel (first 8 queries)
    chk_s = torch.empty((8, 10), dtype=torch.int64, device="cuda")
    chk_i = torch.empty((8, 10), dtype=torch.int32, device="cuda")
    s.bruteforce_topk_dev(qs.device_ptr, 8, 10, chk_s.data_ptr(), chk_i.data_ptr(), mse.MODE_EXACT)
    torch.cuda.synchronize()
    print("answers equal the exact-order kernel:", bool(torch.equal(chk_s, out_s[:8]) and torch.equal(chk_i, out_i[:8])), s.last_stats())
if os.environ.get("MSE_SCAN_2D") == "161":   # developer library: where a wave's cycles go (scan_mfma.hip, PROF)
    import ctypes
    from mse import ffi
    out = (ctypes.c_ulonglong * 4)()
    ffi.lib().mse_dev_scan_prof(out)          # clear what the warm-up and the timed loop left
    s.bruteforce_topk_dev(qs.device_ptr, nq, 10, out_s.data_ptr(), out_i.data_ptr(), mse.MODE_MFMA)
    torch.cuda.synchronize()
    ffi.lib().mse_dev_scan_prof(out)
    issue, wait, bar, nkb = [int(x) for x in out]
    tot = issue + wait + bar
    print(f"profiled 2-D scan: per K block and wave {tot / nkb:.0f} cycles = issue {issue / nkb:.0f} + vmcnt wait {wait / nkb:.0f} + barrier {bar / nkb:.0f}"
          f" ({nkb} wave-K-blocks)")
