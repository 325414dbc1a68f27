"""SigLIP image tower throughput on one GPU: batch of synthetic images resident in HBM, seeded synthetic
weights (no checkpoint or dataset is available offline).  Prints img/s and MFMA utilisation."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
import torch
import mse
from mse import siglip, ffi

def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 27
    cfg = dict(siglip.SO400M_384, depth=depth)
    eng = siglip.SiglipImageEngine.from_state_dict(siglip.synthetic_state_dict(cfg), cfg, max_batch=batch)
    img = torch.empty((batch, 3, 384, 384), dtype=torch.float16, device="cuda").uniform_(-1, 1)
    torch.cuda.synchronize()
    eng.encode_image_device(img.data_ptr(), batch)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.encode_image_device(img.data_ptr(), batch)
    dt = (time.perf_counter() - t0) / steps
    gflop_img = 0.988 + depth * 24.647 + 3.9
    print(f"batch {batch} depth {depth}: {dt*1e3:.1f} ms/batch  {batch/dt:.0f} img/s  {batch/dt*gflop_img/1e3:.0f} TFLOP/s "
          f"({batch/dt*gflop_img*1e9/2.5e15*100:.1f}% of 2.5 PF)")

main()

if os.environ.get("MSE_ATT64_ABL") == "7":   # developer library: where an attention workgroup's cycles go
    import ctypes
    out = (ctypes.c_ulonglong * 4)()
    ffi.lib().mse_dev_att_prof(out)
    pro, loop, epi, n = [int(x) for x in out]
    if n:
        print(f"attention workgroups (wave 0, shader cycles per workgroup over {n} workgroups): prologue {pro / n:.0f}, main loop {loop / n:.0f}, "
              f"epilogue {epi / n:.0f}  -> {100 * pro / (pro + loop + epi):.1f} % / {100 * loop / (pro + loop + epi):.1f} % / {100 * epi / (pro + loop + epi):.1f} %")
