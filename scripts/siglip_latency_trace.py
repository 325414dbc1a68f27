"""Developer probe (needs a GPU): one small-batch forward of each SigLIP tower, for rocprofv3 --kernel-trace.
python scripts/siglip_latency_trace.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mse import siglip  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tcfg = dict(siglip.SO400M_TEXT)
teng = siglip.SiglipTextEngine.from_state_dict(siglip.synthetic_text_state_dict(tcfg), tcfg, max_batch=32)
tok = np.random.default_rng(7).integers(2, tcfg["vocab_size"], size=(32, tcfg["context_length"]), dtype=np.int64)
for _ in range(3):
    teng.encode_text(tok[:b])
t0 = time.perf_counter()
for _ in range(5):
    teng.encode_text(tok[:b])
print("text batch", b, "ms", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
cfg = dict(siglip.SO400M_384)
eng = siglip.SiglipImageEngine.from_state_dict(siglip.synthetic_state_dict(cfg), cfg, max_batch=32)
img = torch.empty((32, 3, cfg["img_size"], cfg["img_size"]), dtype=torch.float16, device="cuda").uniform_(-1, 1)
torch.cuda.synchronize()
for _ in range(3):
    eng.encode_image_device(img.data_ptr(), b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.encode_image_device(img.data_ptr(), b)
torch.cuda.synchronize()
print("image batch", b, "ms", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
