"""CPU fp32 restatement of the SigLIP ViT-SO400M/14-384 image tower.

TEST INFRASTRUCTURE ONLY (see oracle/mse_oracle.c header): imported by tests/, smoke() and the
golden-fixture script, never by the product.

The reference serves this model through open_clip/timm (third-party, absent here) and restates the
graph itself for its AITemplate engine; this file follows THAT restatement:
    aitemplate/model.py:13-24   MLPBlock        fc1 (+GELU) -> fc2 (+residual)
    aitemplate/model.py:26-44   Encoder1DBlock  x = x + MHA(LN1 x);  x = x + MLP(LN2 x)
    aitemplate/model.py:46-55   Encoder         27 blocks then a final LayerNorm
    aitemplate/model.py:57-63   PositionalEmbeddings   x + pos_emb[1,729,1152]
    aitemplate/model.py:65-80   PatchEmbedder   conv k=s=14, bias, flatten to [B,729,1152]
    aitemplate/model.py:82-111  MAPHead         probe->q; x->kv; 16-head SDPA 1x729; proj; x + MLP(LN x)
    aitemplate/model.py:113-123 VisionTransformer   pool(encoder(pos_emb(patch_embed(image))))
    aitemplate/run.py:47-55     hyper-parameters (384 / 1152 / 27 / 16 / 4304 / 14 / 3)
    clip_server.py:40-57        state-dict key names (timm `visual.trunk.*`)
    clip_server.py:77           output = ys[0][:, 0, :]
    clip_server.py:99,115       features /= features.norm(dim=-1, keepdim=True)

Unverifiable third-party details are parameters: `gelu` ("erf": timm nn.GELU and
AIT specialization="gelu"; "tanh": HF/big_vision) and `eps` (1e-6 in timm/open_clip/HF).
PARITY: unpinned by the reference (its only check, aitemplate/run.py:150-159, is commented out); this
oracle is cross-checked against HuggingFace `SiglipVisionModel` with remapped seeded weights
(tests/golden/make_siglip_golden.py), an independent implementation of the same architecture.
"""
import math

import numpy as np
import torch

CONFIG = dict(img_size=384, emb_dim=1152, depth=27, num_heads=16, mlp_dim=4304, patch_size=14, in_chans=3)


def param_shapes(cfg):
    d, m, p, c = cfg["emb_dim"], cfg["mlp_dim"], cfg["patch_size"], cfg["in_chans"]
    n = (cfg["img_size"] // p) ** 2
    s = {"trunk.patch_embed.proj.weight": (d, c, p, p), "trunk.patch_embed.proj.bias": (d,),
         "trunk.pos_embed": (1, n, d), "trunk.norm.weight": (d,), "trunk.norm.bias": (d,),
         "trunk.attn_pool.latent": (1, 1, d),
         "trunk.attn_pool.q.weight": (d, d), "trunk.attn_pool.q.bias": (d,),
         "trunk.attn_pool.kv.weight": (2 * d, d), "trunk.attn_pool.kv.bias": (2 * d,),
         "trunk.attn_pool.proj.weight": (d, d), "trunk.attn_pool.proj.bias": (d,),
         "trunk.attn_pool.norm.weight": (d,), "trunk.attn_pool.norm.bias": (d,),
         "trunk.attn_pool.mlp.fc1.weight": (m, d), "trunk.attn_pool.mlp.fc1.bias": (m,),
         "trunk.attn_pool.mlp.fc2.weight": (d, m), "trunk.attn_pool.mlp.fc2.bias": (d,)}
    for i in range(cfg["depth"]):
        b = f"trunk.blocks.{i}."
        s.update({b + "norm1.weight": (d,), b + "norm1.bias": (d,), b + "attn.qkv.weight": (3 * d, d),
                  b + "attn.qkv.bias": (3 * d,), b + "attn.proj.weight": (d, d), b + "attn.proj.bias": (d,),
                  b + "norm2.weight": (d,), b + "norm2.bias": (d,), b + "mlp.fc1.weight": (m, d),
                  b + "mlp.fc1.bias": (m,), b + "mlp.fc2.weight": (d, m), b + "mlp.fc2.bias": (d,)})
    return s


def synthetic_weights(cfg, seed=0x5EED0005):
    """Deterministic weights (never committed: ~856 MB at full size): linears N(0, 1/fan_in), LayerNorm
    gamma ~ 1 +- 0.1, small biases, pos_embed / latent N(0, 0.02..1).  One Philox stream per tensor name
    so any subset can be regenerated independently."""
    out = {}
    for idx, (name, shape) in enumerate(sorted(param_shapes(cfg).items())):
        g = np.random.Generator(np.random.Philox(key=seed + idx))
        if name.endswith("norm.weight") or ".norm1.weight" in name or ".norm2.weight" in name:
            w = 1.0 + 0.1 * g.standard_normal(shape)
        elif name.endswith(".bias"):
            w = 0.02 * g.standard_normal(shape)
        elif name.endswith("pos_embed"):
            w = 0.02 * g.standard_normal(shape)
        elif name.endswith("latent"):
            w = g.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            w = g.standard_normal(shape) / math.sqrt(fan_in)
        out[name] = torch.from_numpy(w.astype(np.float32))
    return out


def synthetic_images(batch, cfg, seed=0x5EED0004):
    """uint8 RGB noise mapped to x/127.5 - 1 (open_clip Normalize(mean=std=0.5)), NCHW fp32."""
    g = np.random.Generator(np.random.Philox(key=seed))
    u8 = g.integers(0, 256, size=(batch, cfg["in_chans"], cfg["img_size"], cfg["img_size"]), dtype=np.uint8)
    return torch.from_numpy(u8.astype(np.float32) / np.float32(127.5) - np.float32(1.0))


def _gelu(x, kind):
    if kind == "erf":
        return torch.nn.functional.gelu(x)
    if kind == "tanh":
        return torch.nn.functional.gelu(x, approximate="tanh")
    raise ValueError(kind)


def _ln(x, w, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def _mlp(x, sd, prefix, gelu):   # model.py:13-24 (residual added by the caller)
    h = _gelu(x @ sd[prefix + "fc1.weight"].T + sd[prefix + "fc1.bias"], gelu)
    return h @ sd[prefix + "fc2.weight"].T + sd[prefix + "fc2.bias"]


def encode_image(images, sd, cfg=CONFIG, gelu="erf", eps=1e-6, normalize=True, taps=None):
    """images: [B,3,H,W] float32 (already normalised).  Returns [B, emb_dim] float32."""
    d, heads, p = cfg["emb_dim"], cfg["num_heads"], cfg["patch_size"]
    dh = d // heads
    x = images.float()
    B = x.shape[0]
    # PatchEmbedder (model.py:65-80): conv k=s=p == per-patch GEMM over (c, ky, kx)
    x = torch.nn.functional.conv2d(x, sd["trunk.patch_embed.proj.weight"], sd["trunk.patch_embed.proj.bias"], stride=p)
    x = x.flatten(2).transpose(1, 2)                       # [B, 729, d]
    x = x + sd["trunk.pos_embed"]                          # model.py:57-63,122
    n = x.shape[1]
    if taps is not None:
        taps["embed"] = x.clone()
    for i in range(cfg["depth"]):                          # model.py:26-44
        b = f"trunk.blocks.{i}."
        h = _ln(x, sd[b + "norm1.weight"], sd[b + "norm1.bias"], eps)
        qkv = h @ sd[b + "attn.qkv.weight"].T + sd[b + "attn.qkv.bias"]          # [B, n, 3d]
        qkv = qkv.reshape(B, n, 3, heads, dh).permute(2, 0, 3, 1, 4)              # [3, B, heads, n, dh]
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        o = (att @ v).transpose(1, 2).reshape(B, n, d)
        x = x + (o @ sd[b + "attn.proj.weight"].T + sd[b + "attn.proj.bias"])
        x = x + _mlp(_ln(x, sd[b + "norm2.weight"], sd[b + "norm2.bias"], eps), sd, b + "mlp.", gelu)
        if taps is not None:
            taps[f"block{i}"] = x.clone()
    x = _ln(x, sd["trunk.norm.weight"], sd["trunk.norm.bias"], eps)              # model.py:50,55
    # MAPHead (model.py:82-111)
    ap = "trunk.attn_pool."
    ql = sd[ap + "latent"].expand(B, -1, -1)
    q = (ql @ sd[ap + "q.weight"].T + sd[ap + "q.bias"]).reshape(B, 1, heads, dh).transpose(1, 2)    # [B,h,1,dh]
    kv = (x @ sd[ap + "kv.weight"].T + sd[ap + "kv.bias"]).reshape(B, n, 2, heads, dh).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, 1, d)
    o = o @ sd[ap + "proj.weight"].T + sd[ap + "proj.bias"]
    o = o + _mlp(_ln(o, sd[ap + "norm.weight"], sd[ap + "norm.bias"], eps), sd, ap + "mlp.", gelu)
    feat = o[:, 0, :]                                                              # clip_server.py:77
    if taps is not None:
        taps["pooled"] = feat.clone()
    if normalize:
        feat = feat / feat.norm(dim=-1, keepdim=True)                              # clip_server.py:115
    return feat


def to_hf_state_dict(sd, cfg):
    """Remap timm-style keys to HuggingFace SiglipVisionModel keys (independent implementation)."""
    d = cfg["emb_dim"]
    out = {"embeddings.patch_embedding.weight": sd["trunk.patch_embed.proj.weight"],
           "embeddings.patch_embedding.bias": sd["trunk.patch_embed.proj.bias"],
           "embeddings.position_embedding.weight": sd["trunk.pos_embed"][0],
           "post_layernorm.weight": sd["trunk.norm.weight"], "post_layernorm.bias": sd["trunk.norm.bias"],
           "head.probe": sd["trunk.attn_pool.latent"],
           "head.attention.in_proj_weight": torch.cat([sd["trunk.attn_pool.q.weight"], sd["trunk.attn_pool.kv.weight"]]),
           "head.attention.in_proj_bias": torch.cat([sd["trunk.attn_pool.q.bias"], sd["trunk.attn_pool.kv.bias"]]),
           "head.attention.out_proj.weight": sd["trunk.attn_pool.proj.weight"],
           "head.attention.out_proj.bias": sd["trunk.attn_pool.proj.bias"],
           "head.layernorm.weight": sd["trunk.attn_pool.norm.weight"], "head.layernorm.bias": sd["trunk.attn_pool.norm.bias"],
           "head.mlp.fc1.weight": sd["trunk.attn_pool.mlp.fc1.weight"], "head.mlp.fc1.bias": sd["trunk.attn_pool.mlp.fc1.bias"],
           "head.mlp.fc2.weight": sd["trunk.attn_pool.mlp.fc2.weight"], "head.mlp.fc2.bias": sd["trunk.attn_pool.mlp.fc2.bias"]}
    for i in range(cfg["depth"]):
        b, h = f"trunk.blocks.{i}.", f"encoder.layers.{i}."
        qkv_w, qkv_b = sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            out[h + f"self_attn.{nm}.weight"] = qkv_w[j * d:(j + 1) * d]
            out[h + f"self_attn.{nm}.bias"] = qkv_b[j * d:(j + 1) * d]
        out[h + "self_attn.out_proj.weight"], out[h + "self_attn.out_proj.bias"] = sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"]
        out[h + "layer_norm1.weight"], out[h + "layer_norm1.bias"] = sd[b + "norm1.weight"], sd[b + "norm1.bias"]
        out[h + "layer_norm2.weight"], out[h + "layer_norm2.bias"] = sd[b + "norm2.weight"], sd[b + "norm2.bias"]
        for nm in ("fc1", "fc2"):
            out[h + f"mlp.{nm}.weight"], out[h + f"mlp.{nm}.bias"] = sd[b + f"mlp.{nm}.weight"], sd[b + f"mlp.{nm}.bias"]
    return out


# ---------------------------------------------------------------------------------------------------------
# Text tower (`model.encode_text`, clip_server.py:98).  The reference holds NO restatement of it: it comes from
# open_clip (third-party, absent).  Known from the repo: output width = model.text.text_projection.out_features
# (clip_server.py:107,182); width 1152, 27 layers, ctx 64, vocab 32000, sentencepiece c4_en, pad id 1
# (misc/clip_accursed.py:31-55).  Published architecture restated: token + positional
# embedding, the same pre-LN blocks WITHOUT a causal mask, final LayerNorm, the LAST position pooled,
# Linear projection with bias.  Key names follow open_clip's TextTransformer; cross-checked against
# HuggingFace SiglipTextModel (independent implementation) by tests/golden/make_siglip_golden.py.
# ---------------------------------------------------------------------------------------------------------
TEXT_CONFIG = dict(vocab_size=32000, context_length=64, width=1152, layers=27, heads=16, mlp_dim=4304)


def text_param_shapes(cfg):
    d, m = cfg["width"], cfg["mlp_dim"]
    s = {"text.token_embedding.weight": (cfg["vocab_size"], d), "text.positional_embedding": (cfg["context_length"], d),
         "text.ln_final.weight": (d,), "text.ln_final.bias": (d,),
         "text.text_projection.weight": (d, d), "text.text_projection.bias": (d,)}
    for i in range(cfg["layers"]):
        b = f"text.transformer.resblocks.{i}."
        s.update({b + "ln_1.weight": (d,), b + "ln_1.bias": (d,), b + "attn.in_proj_weight": (3 * d, d),
                  b + "attn.in_proj_bias": (3 * d,), b + "attn.out_proj.weight": (d, d), b + "attn.out_proj.bias": (d,),
                  b + "ln_2.weight": (d,), b + "ln_2.bias": (d,), b + "mlp.c_fc.weight": (m, d), b + "mlp.c_fc.bias": (m,),
                  b + "mlp.c_proj.weight": (d, m), b + "mlp.c_proj.bias": (d,)})
    return s


def synthetic_text_weights(cfg, seed=0x5EED0006):
    out = {}
    for idx, (name, shape) in enumerate(sorted(text_param_shapes(cfg).items())):
        g = np.random.Generator(np.random.Philox(key=seed + idx))
        if ".ln_" in name and name.endswith("weight"):
            w = 1.0 + 0.1 * g.standard_normal(shape)
        elif name.endswith("bias"):
            w = 0.02 * g.standard_normal(shape)
        elif name.endswith("token_embedding.weight"):
            w = 0.5 * g.standard_normal(shape)
        elif name.endswith("positional_embedding"):
            w = 0.1 * g.standard_normal(shape)
        else:
            w = g.standard_normal(shape) / math.sqrt(shape[1])
        out[name] = torch.from_numpy(w.astype(np.float32))
    return out


def synthetic_tokens(batch, cfg, seed=0x5EED0007):
    g = np.random.Generator(np.random.Philox(key=seed))
    t = g.integers(2, cfg["vocab_size"], size=(batch, cfg["context_length"]), dtype=np.int64)
    lens = g.integers(3, cfg["context_length"], size=batch)
    for i, n in enumerate(lens):
        t[i, n:] = 1                                        # pad id 1 (misc/clip_accursed.py:55)
    return torch.from_numpy(t)


def encode_text(tokens, sd, cfg=TEXT_CONFIG, gelu="erf", eps=1e-6, normalize=True):
    d, heads = cfg["width"], cfg["heads"]
    dh = d // heads
    B, n = tokens.shape
    x = sd["text.token_embedding.weight"][tokens] + sd["text.positional_embedding"][:n]
    for i in range(cfg["layers"]):
        b = f"text.transformer.resblocks.{i}."
        h = _ln(x, sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], eps)
        qkv = (h @ sd[b + "attn.in_proj_weight"].T + sd[b + "attn.in_proj_bias"]).reshape(B, n, 3, heads, dh).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] @ qkv[1].transpose(-1, -2)) / math.sqrt(dh), dim=-1)   # no causal mask
        o = (att @ qkv[2]).transpose(1, 2).reshape(B, n, d)
        x = x + (o @ sd[b + "attn.out_proj.weight"].T + sd[b + "attn.out_proj.bias"])
        h = _ln(x, sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], eps)
        h = _gelu(h @ sd[b + "mlp.c_fc.weight"].T + sd[b + "mlp.c_fc.bias"], gelu)
        x = x + (h @ sd[b + "mlp.c_proj.weight"].T + sd[b + "mlp.c_proj.bias"])
    x = _ln(x, sd["text.ln_final.weight"], sd["text.ln_final.bias"], eps)
    feat = x[:, -1, :] @ sd["text.text_projection.weight"].T + sd["text.text_projection.bias"]   # last position
    if normalize:
        feat = feat / feat.norm(dim=-1, keepdim=True)                                           # clip_server.py:99
    return feat


def text_to_hf_state_dict(sd, cfg):
    d = cfg["width"]
    out = {"embeddings.token_embedding.weight": sd["text.token_embedding.weight"],
           "embeddings.position_embedding.weight": sd["text.positional_embedding"],
           "final_layer_norm.weight": sd["text.ln_final.weight"], "final_layer_norm.bias": sd["text.ln_final.bias"],
           "head.weight": sd["text.text_projection.weight"], "head.bias": sd["text.text_projection.bias"]}
    for i in range(cfg["layers"]):
        b, h = f"text.transformer.resblocks.{i}.", f"encoder.layers.{i}."
        w, bb = sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"]
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            out[h + f"self_attn.{nm}.weight"], out[h + f"self_attn.{nm}.bias"] = w[j * d:(j + 1) * d], bb[j * d:(j + 1) * d]
        out[h + "self_attn.out_proj.weight"], out[h + "self_attn.out_proj.bias"] = sd[b + "attn.out_proj.weight"], sd[b + "attn.out_proj.bias"]
        out[h + "layer_norm1.weight"], out[h + "layer_norm1.bias"] = sd[b + "ln_1.weight"], sd[b + "ln_1.bias"]
        out[h + "layer_norm2.weight"], out[h + "layer_norm2.bias"] = sd[b + "ln_2.weight"], sd[b + "ln_2.bias"]
        out[h + "mlp.fc1.weight"], out[h + "mlp.fc1.bias"] = sd[b + "mlp.c_fc.weight"], sd[b + "mlp.c_fc.bias"]
        out[h + "mlp.fc2.weight"], out[h + "mlp.fc2.bias"] = sd[b + "mlp.c_proj.weight"], sd[b + "mlp.c_proj.bias"]
    return out
