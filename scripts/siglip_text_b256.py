"""Developer probe (needs a GPU): the text tower at batch 256 alone, for rocprofv3 --kernel-trace --stats.  python scripts/siglip_text_b256.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from mse import siglip  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tcfg = dict(siglip.SO400M_TEXT)
teng = siglip.SiglipTextEngine.from_state_dict(siglip.synthetic_text_state_dict(tcfg), tcfg, max_batch=b)
tok = np.random.default_rng(7).integers(2, tcfg["vocab_size"], size=(b, tcfg["context_length"]), dtype=np.int64)
teng.encode_text(tok)
t0 = time.perf_counter()
for _ in range(5):
    teng.encode_text(tok)
dt = (time.perf_counter() - t0) / 5
print(f"text batch {b}: {dt * 1e3:.3f} ms, {b / dt:.0f} texts/s, {b / dt * 27 * 1.968e9 / 2.5e15:.3f} of the dense bf16 peak", flush=True)
