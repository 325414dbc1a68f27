"""SigLIP image tower: (CPU) the oracle restatement of aitemplate/model.py against the committed outputs of
an independent implementation (HF transformers, tests/golden/make_siglip_golden.py); (GPU) the HIP engine
against the oracle.  Tolerance from BASELINE.json north_star: cosine within 1e-3 of the fp32 reference."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

torch.set_grad_enabled(False)


def torch_from(a):
    import torch
    return torch.from_numpy(np.asarray(a, np.float32))


def cosine(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)


@pytest.fixture(scope="module")
def ref():
    from oracle import siglip_ref
    return siglip_ref


def test_oracle_matches_independent_implementation(ref):
    g = np.load(os.path.join(GOLDEN, "siglip_hf_depth2.npz"))
    cfg = dict(ref.CONFIG, depth=int(g["depth"]))
    sd = ref.synthetic_weights(cfg, seed=int(g["seed_weights"]))
    img = ref.synthetic_images(2, cfg, seed=int(g["seed_images"]))
    out = ref.encode_image(img, sd, cfg, gelu=str(g["gelu"]), eps=float(g["eps"]), normalize=False).numpy()
    assert np.abs(out - g["pooled"]).max() < 2e-5 * np.abs(g["pooled"]).max() + 1e-5
    assert np.all(cosine(out, g["pooled"]) > 1 - 1e-6)


def test_oracle_structure(ref):
    # shapes, normalisation and the two GELU flavours of SURVEY Appendix C
    cfg = dict(ref.CONFIG, depth=1)
    sd = ref.synthetic_weights(cfg)
    assert set(sd) == set(ref.param_shapes(cfg))
    img = ref.synthetic_images(1, cfg)
    assert img.shape == (1, 3, 384, 384) and float(img.min()) >= -1.0 and float(img.max()) <= 1.0
    taps = {}
    f = ref.encode_image(img, sd, cfg, taps=taps)
    assert f.shape == (1, 1152) and abs(float(f.norm()) - 1.0) < 1e-5
    assert taps["embed"].shape == (1, 729, 1152)
    f2 = ref.encode_image(img, sd, cfg, gelu="tanh")
    assert 1 - 1e-3 < float(cosine(f.numpy(), f2.numpy())[0]) < 1.0     # close but not identical


def test_text_oracle_matches_independent_implementation(ref):
    g = np.load(os.path.join(GOLDEN, "siglip_text_hf_depth2.npz"))
    cfg = dict(ref.TEXT_CONFIG, layers=int(g["layers"]))
    sd = ref.synthetic_text_weights(cfg, seed=int(g["seed_weights"]))
    tok = ref.synthetic_tokens(3, cfg, seed=int(g["seed_tokens"]))
    assert tok.shape == (3, 64) and int(tok.min()) >= 0 and int(tok.max()) < 32000
    assert (tok[:, -1] == 1).any()                      # some rows end in padding (pad id 1), as real captions do
    out = ref.encode_text(tok, sd, cfg, gelu=str(g["gelu"]), eps=float(g["eps"]), normalize=False).numpy()
    assert np.abs(out - g["pooled"]).max() < 2e-5 * np.abs(g["pooled"]).max() + 1e-5
    assert np.all(cosine(out, g["pooled"]) > 1 - 1e-6)


def test_text_host_side(mse):
    from mse import siglip
    from oracle import siglip_ref
    assert {k: tuple(v) for k, v in siglip.text_weight_shapes(siglip.SO400M_TEXT).items()} == \
        {k: tuple(v) for k, v in siglip_ref.text_param_shapes(siglip_ref.TEXT_CONFIG).items()}
    t = siglip.pad_tokens([[5, 6, 7], list(range(100, 200))])
    assert t.shape == (2, 64) and t.dtype == np.int64
    assert t[0].tolist() == [5, 6, 7] + [1] * 61 and t[1].tolist() == list(range(100, 164))


def test_engine_rejects_bad_use(mse):
    from mse import ffi, siglip
    if ffi.lib().mse_device_count() > 0:
        pytest.skip("needs no device")
    with pytest.raises(mse.MseError):
        siglip.SiglipImageEngine(dict(siglip.SO400M_384, depth=1), max_batch=1)


@pytest.mark.gpu
@pytest.mark.parametrize("depth,gelu,batch", [(2, "tanh", 2), (2, "erf", 3), (27, "erf", 2), (2, "erf", 5), (27, "tanh", 6)])
def test_engine_matches_oracle(gpu, mse, ref, depth, gelu, batch):
    """Batches of up to 4 images run the small-batch kernels (LayerNorm as its own pass), larger ones the batch kernels with
    LayerNorm folded into the GEMMs: both against the oracle."""
    from mse import siglip
    max_batch = 4 if batch <= 4 else 8
    cfg = dict(ref.CONFIG, depth=depth)
    sd = ref.synthetic_weights(cfg)
    img = ref.synthetic_images(batch, cfg)
    taps = {}
    want = ref.encode_image(img, sd, cfg, gelu=gelu, normalize=True, taps=taps).numpy()
    eng = siglip.SiglipImageEngine.from_state_dict({"visual." + k: v for k, v in sd.items()},
                                                   dict(siglip.SO400M_384, depth=depth), max_batch=max_batch, gelu=gelu)
    got = eng.encode_image(img.numpy())
    resid = eng.debug_residual(batch)
    cos_resid = cosine(resid.reshape(batch, -1), taps[f"block{depth - 1}"].numpy().reshape(batch, -1))
    assert np.all(cos_resid > 1 - 1e-3), cos_resid
    # headroom of the fp16 residual stream: the largest magnitude after the last block against fp16's 65504
    peak = float(np.abs(resid).max())
    assert np.isfinite(resid).all() and peak < 65504 / 16, peak
    cos = cosine(got, want)
    assert np.all(cos > 1 - 1e-3), cos                                    # north_star tolerance
    assert np.all(np.abs(np.linalg.norm(got, axis=1) - 1) < 1e-3)
    # fp16 serialisation path (clip_server.py:166) and fp16 inputs (the server feeds .half() images, :140)
    got16 = eng.encode_image(img.numpy().astype(np.float16), out="f16").view(np.float16).astype(np.float32)
    assert np.all(cosine(got16, want) > 1 - 1e-3)
    # smaller batch after a larger one (stale padded rows must not leak)
    got1 = eng.encode_image(img.numpy()[:1])
    assert np.all(cosine(got1, want[:1]) > 1 - 1e-3)
    with pytest.raises(mse.MseError):
        eng.encode_image(np.zeros((max_batch + 1, 3, 384, 384), np.float32))   # > max_batch (clip_server.py:139)


@pytest.mark.gpu
def test_fused_layernorm_equals_separate_layernorm(gpu, mse, ref, monkeypatch):
    """The image tower folds LN1 / LN2 and the residual adds into the GEMMs; MSE_SIGLIP_NOFUSE=1 runs them as
    passes of their own.  Same weights, same images: the two engines must agree far inside the oracle tolerance, the residual
    streams too, and planted LayerNorm gains / offsets (the seeded weights have gamma = 1, beta = 0) must be honoured."""
    from mse import siglip
    cfg = dict(ref.CONFIG, depth=3)
    sd = ref.synthetic_weights(cfg)
    g = torch.Generator().manual_seed(5)
    for i in range(3):
        for nm in ("norm1", "norm2"):
            sd[f"trunk.blocks.{i}.{nm}.weight"] = 0.5 + torch.rand(1152, generator=g)
            sd[f"trunk.blocks.{i}.{nm}.bias"] = 0.2 * torch.randn(1152, generator=g)
    img = ref.synthetic_images(5, cfg)                                      # (more than 4 images: the batch kernels)
    want = ref.encode_image(img, sd, cfg, normalize=True).numpy()
    named = {"visual." + k: v for k, v in sd.items()}
    fused = siglip.SiglipImageEngine.from_state_dict(named, dict(siglip.SO400M_384, depth=3), max_batch=8)
    got_f = fused.encode_image(img.numpy())
    res_f = fused.debug_residual(5)
    monkeypatch.setenv("MSE_SIGLIP_NOFUSE", "1")
    plain = siglip.SiglipImageEngine.from_state_dict(named, dict(siglip.SO400M_384, depth=3), max_batch=8)
    monkeypatch.delenv("MSE_SIGLIP_NOFUSE")
    got_p = plain.encode_image(img.numpy())
    res_p = plain.debug_residual(5)
    assert np.all(cosine(got_f, want) > 1 - 1e-3) and np.all(cosine(got_p, want) > 1 - 1e-3)
    assert np.all(cosine(got_f, got_p) > 1 - 1e-4), cosine(got_f, got_p)
    assert np.all(cosine(res_f.reshape(5, -1), res_p.reshape(5, -1)) > 1 - 1e-4)
    assert not np.array_equal(got_f, got_p)     # two different code paths really ran


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [32, 41])
def test_sub_batches_on_two_streams_change_nothing(gpu, mse, ref, monkeypatch, batch):
    """A forward pass of >= 32 images runs as two sub-batches on two streams; every row's arithmetic is the same as on
    one stream, so the features must be bit-equal (41 images split 24 + 17: the second sub-batch starts on a 256-row block boundary
    and ends inside one)."""
    from mse import siglip
    cfg = dict(ref.CONFIG, depth=2)
    sd = {"visual." + k: v for k, v in ref.synthetic_weights(cfg).items()}
    img = ref.synthetic_images(batch, cfg).numpy().astype(np.float16)
    two = siglip.SiglipImageEngine.from_state_dict(sd, dict(siglip.SO400M_384, depth=2), max_batch=48)
    got2 = two.encode_image(img)
    monkeypatch.setenv("MSE_SIGLIP_STREAMS", "1")
    one = siglip.SiglipImageEngine.from_state_dict(sd, dict(siglip.SO400M_384, depth=2), max_batch=48)
    monkeypatch.delenv("MSE_SIGLIP_STREAMS")
    got1 = one.encode_image(img)
    assert np.array_equal(got1, got2)
    assert np.isfinite(got2).all() and np.all(np.abs(np.linalg.norm(got2, axis=1) - 1) < 1e-3)
    # and a small batch (one stream either way) after the large one still equals its rows of the large batch
    assert np.array_equal(two.encode_image(img[:5]), got2[:5])


@pytest.mark.gpu
def test_batch_256_rows_equal_small_batch_rows(gpu, mse, ref):
    """BASELINE configs[1]'s batch (256 images, two sub-batches of 128 on two streams), depth 2: a row's arithmetic does not depend
    on what else is in the batch, so rows of the batch-256 forward are BIT-equal to the same images encoded five at a time --
    at the start, across the sub-batch seam and at the end -- and within the north star's cosine of the fp32 oracle.  (Calls of up to
    four images run other kernels -- test_small_batches_agree_with_the_batch_kernels.)"""
    from mse import siglip
    cfg = dict(ref.CONFIG, depth=2)
    sd = ref.synthetic_weights(cfg)
    img = ref.synthetic_images(256, cfg)
    x16 = img.numpy().astype(np.float16)
    eng = siglip.SiglipImageEngine.from_state_dict({"visual." + k: v for k, v in sd.items()}, dict(siglip.SO400M_384, depth=2),
                                                   max_batch=256)
    big = eng.encode_image(x16)
    assert big.shape == (256, 1152) and np.isfinite(big).all()
    assert np.all(np.abs(np.linalg.norm(big, axis=1) - 1) < 1e-3)
    for lo in (0, 125, 128, 251):
        assert np.array_equal(eng.encode_image(x16[lo:lo + 5]), big[lo:lo + 5]), lo
    rows = [0, 127, 128, 255]
    want = ref.encode_image(torch_from(x16[rows]), sd, cfg).numpy()
    assert np.all(cosine(big[rows], want) > 1 - 1e-3)
    assert np.array_equal(eng.encode_image(x16), big)                      # idempotent at full batch


@pytest.mark.gpu
def test_small_batches_agree_with_the_batch_kernels(gpu, mse, ref, monkeypatch):
    """Calls of 1..4 images (<= 3072 token rows) run the small-batch GEMM kernels and LayerNorm as a pass of its own (one image:
    7.6 -> 3.3 ms).  Same weights, same images: both within the oracle's tolerance, each
    other far inside it, every small call idempotent and independent of what ran before; and the GEMM kernels themselves, shape by
    shape at ragged row counts, within two bf16 steps of the batch kernels' output (summation order is all that differs)."""
    import ctypes as C
    from mse import ffi, siglip
    cfg = dict(ref.CONFIG, depth=3)
    sd = ref.synthetic_weights(cfg)
    g = torch.Generator().manual_seed(9)
    for i in range(3):
        for nm in ("norm1", "norm2"):
            sd[f"trunk.blocks.{i}.{nm}.weight"] = 0.5 + torch.rand(1152, generator=g)
            sd[f"trunk.blocks.{i}.{nm}.bias"] = 0.2 * torch.randn(1152, generator=g)
    img = ref.synthetic_images(6, cfg)
    x16 = img.numpy().astype(np.float16)
    want = ref.encode_image(torch_from(x16), sd, cfg, normalize=True).numpy()
    named = {"visual." + k: v for k, v in sd.items()}
    eng = siglip.SiglipImageEngine.from_state_dict(named, dict(siglip.SO400M_384, depth=3), max_batch=8)
    big = eng.encode_image(x16)                                             # six images: the batch kernels
    assert np.all(cosine(big, want) > 1 - 1e-3)
    for b in (1, 2, 3, 4):
        got = eng.encode_image(x16[:b])
        assert np.all(cosine(got, want[:b]) > 1 - 1e-3), (b, cosine(got, want[:b]))
        assert np.all(cosine(got, big[:b]) > 1 - 1e-4), (b, cosine(got, big[:b]))
        assert not np.array_equal(got, big[:b])                             # other kernels really ran
        assert np.array_equal(eng.encode_image(x16[:b]), got)
    one = eng.encode_image(x16[5:6])
    eng.encode_image(x16)                                                   # a larger batch in between leaves nothing behind
    assert np.array_equal(eng.encode_image(x16[5:6]), one)
    # a row of a small call does not depend on its neighbours (two to four images run the same kernels with the same K order; ONE
    # image also splits fc2's K range across workgroups, which changes the summation order)
    four = eng.encode_image(x16[2:6])
    assert np.array_equal(four[2:], eng.encode_image(x16[4:6])) and np.array_equal(four[:3], eng.encode_image(x16[2:5]))
    assert cosine(four[3:4], one)[0] > 1 - 1e-4
    # MSE_SIGLIP_NOSMALL=1 (read when an engine is created): small calls run the batch kernels too, bit-equal to a larger batch's rows
    monkeypatch.setenv("MSE_SIGLIP_NOSMALL", "1")
    plain = siglip.SiglipImageEngine.from_state_dict(named, dict(siglip.SO400M_384, depth=3), max_batch=8)
    monkeypatch.delenv("MSE_SIGLIP_NOSMALL")
    assert np.array_equal(plain.encode_image(x16), big) and np.array_equal(plain.encode_image(x16[:3]), big[:3])
    L = ffi.lib()
    for rows in (1, 37, 100, 736, 1000, 2944):
        for (N, K, epi) in ((3456, 1152, 0), (1152, 1152, 0), (4352, 1152, 1), (1152, 4352, 0)):
            for variant in (1, 3, 4) + ((2,) if rows <= 512 else ()):
                ms, nd = C.c_float(), (C.c_uint64 * 2)()
                ffi.check(L.mse_debug_gemm_small(rows, N, K, epi, variant, 1, C.byref(ms), nd))
                assert nd[1] == 0, (rows, N, K, epi, variant, nd[0], nd[1])


@pytest.mark.gpu
def test_config1_full_shape_depth_27_batch_256_rows_match_the_oracle(gpu, mse, ref):
    """BASELINE configs[1] at its FULL shape -- SO400M/14-384, all 27 blocks, batch 256 -- checked directly: four rows of the
    batch-256 forward (first, both sides of the sub-batch seam, last) against the fp32 oracle on the same four images, within the
    north star's 1e-3 cosine; and the same rows bit-equal to a forward of just those images (row independence at depth 27)."""
    from mse import siglip
    cfg = dict(ref.CONFIG)                                                  # depth 27
    sd = ref.synthetic_weights(cfg)
    img = ref.synthetic_images(256, cfg)
    x16 = img.numpy().astype(np.float16)
    eng = siglip.SiglipImageEngine.from_state_dict({"visual." + k: v for k, v in sd.items()}, dict(siglip.SO400M_384), max_batch=256)
    big = eng.encode_image(x16)
    assert big.shape == (256, 1152) and np.isfinite(big).all()
    rows = [0, 127, 128, 255]
    want = ref.encode_image(torch_from(x16[rows]), sd, cfg).numpy()
    cos = cosine(big[rows], want)
    assert np.all(cos > 1 - 1e-3), cos
    five = rows + [64]                                                      # (five images: still the batch kernels)
    assert np.array_equal(eng.encode_image(x16[five]), big[five])
    # the same four images as ONE call each and as a call of four: the small-batch kernels at depth 27, against the oracle and
    # against the batch kernels' rows
    alone = np.concatenate([eng.encode_image(x16[r:r + 1]) for r in rows])
    four = eng.encode_image(x16[rows])
    for got in (alone, four):
        assert np.all(cosine(got, want) > 1 - 1e-3), cosine(got, want)
        assert np.all(cosine(got, big[rows]) > 1 - 2e-4), cosine(got, big[rows])


@pytest.mark.gpu
def test_engine_with_massive_activation_channels(gpu, mse, ref):
    """Trained ViTs carry a few residual channels hundreds of times larger than the rest ("massive activations"); the seeded
    Gaussian weights of the other tests never do.  Plant them -- biases of +3000 / -800 / +12000 on three channels of the
    first block's proj / fc2 outputs -- and require the same 1e-3 cosine against the fp32 restatement: the fp16 residual
    stream (11-bit mantissa, range 65504) must carry them, and the fp32 LayerNorm statistics must see through them."""
    from mse import siglip
    cfg = dict(ref.CONFIG, depth=3)
    sd = ref.synthetic_weights(cfg)
    sd["trunk.blocks.0.attn.proj.bias"][100] = -800.0
    sd["trunk.blocks.0.mlp.fc2.bias"][7] = 3000.0
    sd["trunk.blocks.1.mlp.fc2.bias"][640] = 12000.0
    img = ref.synthetic_images(2, cfg)
    taps = {}
    want = ref.encode_image(img, sd, cfg, normalize=True, taps=taps).numpy()
    eng = siglip.SiglipImageEngine.from_state_dict({"visual." + k: v for k, v in sd.items()}, dict(siglip.SO400M_384, depth=3), max_batch=2)
    got = eng.encode_image(img.numpy())
    resid = eng.debug_residual(2)
    want_resid = taps["block2"].numpy()
    assert 11000 < float(np.abs(resid).max()) < 65504 and np.isfinite(resid).all()
    assert abs(float(np.abs(resid).max()) - float(np.abs(want_resid).max())) < 64        # fp16 spacing at 12000 is 8; two more blocks add to it
    assert np.all(cosine(got, want) > 1 - 1e-3), cosine(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("layers,gelu,batch", [(2, "tanh", 3), (27, "erf", 5), (3, "erf", 8), (2, "erf", 2)])
def test_text_engine_matches_oracle(gpu, mse, ref, layers, gelu, batch):
    from mse import siglip
    cfg = dict(ref.TEXT_CONFIG, layers=layers)
    sd = ref.synthetic_text_weights(cfg)
    tok = ref.synthetic_tokens(batch, cfg)
    want = ref.encode_text(tok, sd, cfg, gelu=gelu, normalize=True).numpy()
    eng = siglip.SiglipTextEngine.from_state_dict(sd, dict(siglip.SO400M_TEXT, layers=layers), max_batch=8, gelu=gelu)
    got = eng.encode_text(tok.numpy())
    cos = cosine(got, want)
    assert np.all(cos > 1 - 1e-3), cos                                    # north_star tolerance
    assert np.all(np.abs(np.linalg.norm(got, axis=1) - 1) < 1e-3)
    got16 = eng.encode_text(tok.numpy(), out="f16").view(np.float16).astype(np.float32)
    assert np.all(cosine(got16, want) > 1 - 1e-3)
    raw = eng.encode_text(tok.numpy(), normalize=False)
    want_raw = ref.encode_text(tok, sd, cfg, gelu=gelu, normalize=False).numpy()
    assert np.all(np.abs(np.linalg.norm(raw, axis=1) / np.linalg.norm(want_raw, axis=1) - 1) < 2e-2)
    got1 = eng.encode_text(tok.numpy()[1:2])                              # smaller batch after a larger one
    assert np.all(cosine(got1, want[1:2]) > 1 - 1e-3)
    with pytest.raises(mse.MseError):
        eng.encode_text(np.ones((9, 64), np.int64))                        # > max_batch
    with pytest.raises(mse.MseError):
        eng.encode_text(np.ones((1, 32), np.int64))                        # wrong context length
    with pytest.raises(mse.MseError):
        eng.encode_text(np.full((1, 64), 32000, np.int64))                 # outside the vocabulary


@pytest.mark.gpu
@pytest.mark.parametrize("layers,gelu,batch", [(3, "erf", 112), (2, "tanh", 256)])
def test_text_large_batch_runs_the_fused_layernorm_kernels_and_matches_the_oracle(gpu, mse, ref, layers, gelu, batch):
    """Round 6: parts of more than 3072 token rows (batches of ~100 texts and more) run the image tower's LayerNorm-fused GEMMs -- QKV /
    fc1 read the fp16 residual stream against gamma-folded weights, proj / fc2 add into it and emit the row statistics.  Every row
    against the fp32 oracle (cosine within 1e-3), against the same texts encoded eight at a time by the unfused small-batch kernels
    (bf16 rounding apart), and MSE_SIGLIP_NOFUSE=1 (LayerNorm as passes of its own at every size) as a third opinion."""
    import os
    from mse import siglip
    cfg = dict(ref.TEXT_CONFIG, layers=layers)
    sd = ref.synthetic_text_weights(cfg)
    tok = ref.synthetic_tokens(batch, cfg)
    want = ref.encode_text(tok, sd, cfg, gelu=gelu, normalize=True).numpy()
    eng = siglip.SiglipTextEngine.from_state_dict(sd, dict(siglip.SO400M_TEXT, layers=layers), max_batch=256, gelu=gelu)
    got = eng.encode_text(tok.numpy())
    assert np.all(cosine(got, want) > 1 - 1e-3), cosine(got, want).min()
    assert np.all(np.abs(np.linalg.norm(got, axis=1) - 1) < 1e-3)
    small = np.concatenate([eng.encode_text(tok.numpy()[i:i + 8]) for i in range(0, batch, 8)])
    assert np.all(cosine(got, small) > 1 - 2e-4), cosine(got, small).min()
    again = eng.encode_text(tok.numpy())
    assert np.array_equal(got, again)                                    # a call is deterministic
    eng.close()
    os.environ["MSE_SIGLIP_NOFUSE"] = "1"
    try:
        plain = siglip.SiglipTextEngine.from_state_dict(sd, dict(siglip.SO400M_TEXT, layers=layers), max_batch=256, gelu=gelu)
    finally:
        del os.environ["MSE_SIGLIP_NOFUSE"]
    unfused = plain.encode_text(tok.numpy())
    plain.close()
    assert np.all(cosine(unfused, want) > 1 - 1e-3)
    assert np.all(cosine(got, unfused) > 1 - 2e-4) and not np.array_equal(got, unfused)       # two arithmetic paths, one answer


@pytest.mark.gpu
def test_device_preprocessing_equals_host_preprocessing(gpu, mse, ref):
    """encode_rgb8 (ToTensor / Normalize / .half() on the device) == encode_image(host-preprocessed fp16 NCHW), bit for bit."""
    from mse import siglip
    cfg = dict(siglip.SO400M_384, depth=1)
    eng = siglip.SiglipImageEngine.from_state_dict(siglip.synthetic_state_dict(cfg), cfg, max_batch=3)
    u8 = np.random.default_rng(5).integers(0, 256, size=(3, 384, 384, 3), dtype=np.uint8)
    u8[0, :2, :2] = [[[0, 255, 51], [1, 2, 3]], [[127, 128, 129], [254, 253, 252]]]
    host = (u8.astype(np.float32) / np.float32(127.5) - np.float32(1.0)).transpose(0, 3, 1, 2).astype(np.float16)
    a = eng.encode_rgb8(u8, out="f16")
    b = eng.encode_image(np.ascontiguousarray(host), out="f16")
    assert np.array_equal(a, b)
    with pytest.raises(mse.MseError):
        eng.encode_rgb8(np.zeros((1, 100, 100, 3), np.uint8))


@pytest.mark.gpu
def test_engine_requires_all_weights(gpu, mse, ref):
    from mse import siglip
    cfg = dict(ref.CONFIG, depth=1)
    sd = ref.synthetic_weights(cfg)
    sd.pop("trunk.blocks.0.mlp.fc2.bias")
    with pytest.raises(mse.MseError):
        siglip.SiglipImageEngine.from_state_dict(sd, dict(siglip.SO400M_384, depth=1), max_batch=1)
    eng = siglip.SiglipImageEngine(dict(siglip.SO400M_384, depth=1), max_batch=1)
    with pytest.raises(mse.MseError):
        eng.set_weight("trunk.norm.weight", np.zeros(7, np.float32))       # wrong size
    with pytest.raises(mse.MseError):
        eng.encode_image(np.zeros((1, 3, 384, 384), np.float32))           # not finalised
