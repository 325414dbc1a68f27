"""Golden vectors for the SigLIP image tower.

The reference's model code (open_clip/timm via clip_server.py) cannot be imported in this image
(open_clip, timm, torchvision absent; no weights), so the independent implementation available here --
HuggingFace `transformers.SiglipVisionModel` -- is run on seeded synthetic weights and images, and its
outputs are committed.  tests/test_siglip_oracle.py checks oracle/siglip_ref.py (the restatement of the
reference's own aitemplate/model.py graph) against these vectors; the HIP engine is then checked against
the oracle.  HF uses tanh-GELU and eps 1e-6, so the fixture is generated with those settings.

Run from the repo root:  python tests/golden/make_siglip_golden.py      (needs `transformers`)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import siglip_ref as ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    from transformers import SiglipVisionConfig, SiglipVisionModel
    torch.set_grad_enabled(False)
    cfg = dict(ref.CONFIG, depth=2)        # full widths, 2 blocks: the weights regenerate from the seed
    sd = ref.synthetic_weights(cfg, seed=0x5EED0005)
    hf_cfg = SiglipVisionConfig(hidden_size=cfg["emb_dim"], intermediate_size=cfg["mlp_dim"],
                                num_hidden_layers=cfg["depth"], num_attention_heads=cfg["num_heads"],
                                image_size=cfg["img_size"], patch_size=cfg["patch_size"], attn_implementation="eager")
    model = SiglipVisionModel(hf_cfg).eval()
    target = model.vision_model if hasattr(model, "vision_model") else model   # transformers 4.x nests it, 5.x does not
    missing = target.load_state_dict(ref.to_hf_state_dict(sd, cfg), strict=True)
    images = ref.synthetic_images(2, cfg, seed=0x5EED0004)
    out = model(pixel_values=images)
    pooled = out.pooler_output.numpy()
    last = out.last_hidden_state.numpy()
    np.savez_compressed(os.path.join(OUT, "siglip_hf_depth2.npz"), pooled=pooled.astype(np.float32),
                        last_hidden_head=last[:, :4, :64].astype(np.float32), depth=2, seed_weights=0x5EED0005,
                        seed_images=0x5EED0004, gelu="tanh", eps=1e-6)
    print("pooled", pooled.shape, float(np.abs(pooled).mean()), missing)


def text():
    """Same for the text tower: HF SiglipTextModel on seeded weights and token ids (depth 2)."""
    from transformers import SiglipTextConfig, SiglipTextModel
    torch.set_grad_enabled(False)
    cfg = dict(ref.TEXT_CONFIG, layers=2)
    sd = ref.synthetic_text_weights(cfg, seed=0x5EED0006)
    hf_cfg = SiglipTextConfig(hidden_size=cfg["width"], intermediate_size=cfg["mlp_dim"], num_hidden_layers=cfg["layers"],
                              num_attention_heads=cfg["heads"], vocab_size=cfg["vocab_size"],
                              max_position_embeddings=cfg["context_length"], attn_implementation="eager",
                              bos_token_id=None, eos_token_id=None, pad_token_id=1)
    model = SiglipTextModel(hf_cfg).eval()
    target = model.text_model if hasattr(model, "text_model") else model
    target.load_state_dict(ref.text_to_hf_state_dict(sd, cfg), strict=True)
    tokens = ref.synthetic_tokens(3, cfg, seed=0x5EED0007)
    pooled = model(input_ids=tokens).pooler_output.numpy()
    np.savez_compressed(os.path.join(OUT, "siglip_text_hf_depth2.npz"), pooled=pooled.astype(np.float32), layers=2,
                        seed_weights=0x5EED0006, seed_tokens=0x5EED0007, gelu="tanh", eps=1e-6)
    print("text pooled", pooled.shape, float(np.abs(pooled).mean()))


if __name__ == "__main__":
    main()
    text()
