"""SigLIP image engine: host-side loader + the `fast_image_fns` seam of the reference's clip_server.py.

`SiglipImageEngine.from_state_dict` plays the role of `load_pretrained` + `generate_wrapper`
(clip_server.py:40-82): it takes an open_clip/timm state dict (keys `visual.trunk.*`), hands every tensor to
the HIP engine by name, and returns a callable `engine(images[b,3,384,384]) -> features[b,1152]`.
PyTorch is used only to read weights (any mapping name -> array-like works); all arithmetic runs in
libmse_hip.so.
"""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import MseError, check, check_ptr

SO400M_384 = dict(img_size=384, patch_size=14, in_chans=3, emb_dim=1152, depth=27, num_heads=16, mlp_dim=4304)  # aitemplate/run.py:47-55


def weight_shapes(cfg):
    """open_clip/timm tensor names (without `visual.`) and shapes of the image tower (clip_server.py:40-57)."""
    d, m, p, c = cfg["emb_dim"], cfg["mlp_dim"], cfg["patch_size"], cfg["in_chans"]
    n = (cfg["img_size"] // p) ** 2
    s = {"trunk.patch_embed.proj.weight": (d, c, p, p), "trunk.patch_embed.proj.bias": (d,), "trunk.pos_embed": (1, n, d),
         "trunk.norm.weight": (d,), "trunk.norm.bias": (d,), "trunk.attn_pool.latent": (1, 1, d),
         "trunk.attn_pool.q.weight": (d, d), "trunk.attn_pool.q.bias": (d,), "trunk.attn_pool.kv.weight": (2 * d, d),
         "trunk.attn_pool.kv.bias": (2 * d,), "trunk.attn_pool.proj.weight": (d, d), "trunk.attn_pool.proj.bias": (d,),
         "trunk.attn_pool.norm.weight": (d,), "trunk.attn_pool.norm.bias": (d,), "trunk.attn_pool.mlp.fc1.weight": (m, d),
         "trunk.attn_pool.mlp.fc1.bias": (m,), "trunk.attn_pool.mlp.fc2.weight": (d, m), "trunk.attn_pool.mlp.fc2.bias": (d,)}
    for i in range(cfg["depth"]):
        b = f"trunk.blocks.{i}."
        s.update({b + "norm1.weight": (d,), b + "norm1.bias": (d,), b + "attn.qkv.weight": (3 * d, d),
                  b + "attn.qkv.bias": (3 * d,), b + "attn.proj.weight": (d, d), b + "attn.proj.bias": (d,),
                  b + "norm2.weight": (d,), b + "norm2.bias": (d,), b + "mlp.fc1.weight": (m, d), b + "mlp.fc1.bias": (m,),
                  b + "mlp.fc2.weight": (d, m), b + "mlp.fc2.bias": (d,)})
    return s


def synthetic_state_dict(cfg, seed=0x5EED0005):
    """Random-init weights of the named architecture (no checkpoint can be downloaded offline): linears
    N(0, 1/fan_in), LayerNorm gain ~ 1, small biases.  Used by bench.py and by the server's
    `synthetic_weights` mode; results are architecture-faithful timings, not meaningful embeddings."""
    out = {}
    for idx, (name, shape) in enumerate(sorted(weight_shapes(cfg).items())):
        g = np.random.Generator(np.random.Philox(key=seed + idx))
        if name.endswith("norm.weight") or ".norm1.weight" in name or ".norm2.weight" in name:
            w = 1.0 + 0.1 * g.standard_normal(shape, dtype=np.float32)
        elif name.endswith(".bias") or name.endswith("pos_embed"):
            w = 0.02 * g.standard_normal(shape, dtype=np.float32)
        elif name.endswith("latent"):
            w = g.standard_normal(shape, dtype=np.float32)
        else:
            w = g.standard_normal(shape, dtype=np.float32) / np.float32(np.sqrt(np.prod(shape[1:])))
        out[name] = w.astype(np.float32)
    return out


def _to_numpy_f32(t):
    if hasattr(t, "detach"):
        t = t.detach().float().cpu().numpy()
    return np.ascontiguousarray(t, np.float32)


def is_plain_bmp(data, size):
    """True if `data` is a 24-bit uncompressed BMP of (width, height) = size: the file the device path takes as it is."""
    w, h = C.c_uint32(), C.c_uint32()
    if ffi.lib().mse_bmp24_info(bytes(data), len(data), C.byref(w), C.byref(h), None, None) != 0:
        return False
    return (w.value, h.value) == tuple(size)


class SiglipImageEngine:
    def __init__(self, config=None, max_batch=32, eps=1e-6, gelu="erf"):
        cfg = dict(SO400M_384 if config is None else config)
        self.cfg = cfg
        self.max_batch = max_batch
        c = ffi.SiglipConfig(cfg["img_size"], cfg["patch_size"], cfg["in_chans"], cfg["emb_dim"], cfg["depth"],
                             cfg["num_heads"], cfg["mlp_dim"], eps, 1 if gelu == "tanh" else 0, max_batch)
        self._h = check_ptr(ffi.lib().mse_siglip_create(C.byref(c)), "mse_siglip_create")
        self.embedding_size = cfg["emb_dim"]

    def weight_names(self):
        L = ffi.lib()
        return [L.mse_siglip_weight_name(self._h, i).decode() for i in range(L.mse_siglip_n_weights(self._h))]

    def set_weight(self, name, tensor):
        a = _to_numpy_f32(tensor)
        shape = (C.c_size_t * a.ndim)(*a.shape)
        check(ffi.lib().mse_siglip_set_weight(self._h, name.encode(), a.ctypes.data_as(ffi.f32p), shape, a.ndim),
              f"set_weight({name})")

    @classmethod
    def from_state_dict(cls, state, config=None, max_batch=32, eps=1e-6, gelu="erf"):
        """state: mapping of open_clip names (`visual.trunk.blocks.0.attn.qkv.weight`, ...) or the same without
        the `visual.` prefix.  Missing tensors are an error (the engine refuses to run half-loaded)."""
        eng = cls(config, max_batch, eps, gelu)
        for name in eng.weight_names():
            key = name if name in state else "visual." + name
            if key not in state:
                raise MseError(f"state dict lacks '{name}'")
            eng.set_weight(name, state[key])
        check(ffi.lib().mse_siglip_finalize(eng._h), "siglip_finalize")
        return eng

    def encode_image(self, images, normalize=True, out="f32"):
        """images: [b,3,H,W] float32 or float16 numpy array, already normalised (x/127.5-1).
        Returns float32 [b, emb] (out="f32") or the IEEE fp16 bit patterns the server serialises (out="f16")."""
        a = np.asarray(images)
        if a.dtype == np.float16:
            dtype = 1
        else:
            a = a.astype(np.float32, copy=False)
            dtype = 0
        a = np.ascontiguousarray(a)
        b = a.shape[0]
        if b > self.max_batch:
            raise MseError(f"max batch size is {self.max_batch}")   # assert at clip_server.py:139
        of32 = np.empty((b, self.embedding_size), np.float32) if out == "f32" else None
        of16 = np.empty((b, self.embedding_size), np.uint16) if out == "f16" else None
        check(ffi.lib().mse_siglip_encode_image(self._h, a.ctypes.data_as(C.c_void_p), dtype, 0, b, int(normalize),
                                                of32.ctypes.data_as(ffi.f32p) if of32 is not None else None,
                                                of16.ctypes.data_as(ffi.u16p) if of16 is not None else None),
              "siglip_encode_image")
        return of32 if out == "f32" else of16

    def encode_rgb8(self, images_u8, normalize=True, out="f32"):
        """images_u8: decoded RGB bytes [b,H,W,3] uint8.  ToTensor / Normalize(0.5) / .half() (clip_server.py:140-141) run
        on the device; the result equals encode_image(preprocessed fp16 NCHW) bit for bit."""
        a = np.ascontiguousarray(images_u8, np.uint8)
        s = self.cfg["img_size"]
        if a.ndim != 4 or a.shape[1:] != (s, s, self.cfg["in_chans"]):
            raise MseError(f"images must be [batch, {s}, {s}, {self.cfg['in_chans']}] uint8")
        b = a.shape[0]
        if b > self.max_batch:
            raise MseError(f"max batch size is {self.max_batch}")
        of32 = np.empty((b, self.embedding_size), np.float32) if out == "f32" else None
        of16 = np.empty((b, self.embedding_size), np.uint16) if out == "f16" else None
        check(ffi.lib().mse_siglip_encode_rgb8(self._h, a.ctypes.data_as(ffi.u8p), b, int(normalize),
                                               of32.ctypes.data_as(ffi.f32p) if of32 is not None else None,
                                               of16.ctypes.data_as(ffi.u16p) if of16 is not None else None),
              "siglip_encode_rgb8")
        return of32 if out == "f32" else of16

    def encode_bmp(self, files, normalize=True, out="f32"):
        """files: the request's image bytes, every one a 24-bit uncompressed BMP of the model's size (what the reference's clients
        send, src/common.rs:31-54).  Header check on the host, everything else on the device; equal to decoding with PIL and
        encode_rgb8 bit for bit.  Raises MseError for any other file: use `is_plain_bmp` first (or catch and decode on the host)."""
        files = [bytes(f) for f in files]
        b = len(files)
        if b > self.max_batch:
            raise MseError(f"max batch size is {self.max_batch}")
        ptrs = (C.c_char_p * b)(*files)
        sizes = (ffi.sz * b)(*[len(f) for f in files])
        of32 = np.empty((b, self.embedding_size), np.float32) if out == "f32" else None
        of16 = np.empty((b, self.embedding_size), np.uint16) if out == "f16" else None
        check(ffi.lib().mse_siglip_encode_bmp(self._h, ptrs, sizes, b, int(normalize),
                                              of32.ctypes.data_as(ffi.f32p) if of32 is not None else None,
                                              of16.ctypes.data_as(ffi.u16p) if of16 is not None else None), "siglip_encode_bmp")
        return of32 if out == "f32" else of16

    def encode_image_device(self, dev_ptr, batch, dtype_f16=True, normalize=True):
        """Images already resident in HBM (the `fast_image_fns` call shape); result stays on the device:
        returns (device pointer to [batch, emb] f32, device pointer to the fp16 copy)."""
        check(ffi.lib().mse_siglip_encode_image(self._h, dev_ptr, 1 if dtype_f16 else 0, 1, batch, int(normalize), None, None),
              "siglip_encode_image")
        L = ffi.lib()
        return L.mse_siglip_output_device(self._h, 0), L.mse_siglip_output_device(self._h, 1)

    def __call__(self, images):
        return self.encode_image(images)

    def debug_residual(self, batch):
        n = (self.cfg["img_size"] // self.cfg["patch_size"]) ** 2
        out = np.empty((batch * n, self.embedding_size), np.float32)
        check(ffi.lib().mse_siglip_debug_residual(self._h, out.ctypes.data_as(ffi.f32p)), "debug_residual")
        return out.reshape(batch, n, self.embedding_size)

    def close(self):
        if getattr(self, "_h", None):
            ffi.lib().mse_siglip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- text tower ------------------------------------------------------------------------------------------------
SO400M_TEXT = dict(width=1152, layers=27, heads=16, mlp_dim=4304, context_length=64, vocab_size=32000)  # misc/clip_accursed.py:31-55
PAD_ID = 1                                                                                                # clip_server.py:129 (sentencepiece pad)


def text_weight_shapes(cfg):
    """open_clip tensor names and shapes of the text tower (`model.text`, clip_server.py:98,107)."""
    d, m = cfg["width"], cfg["mlp_dim"]
    s = {"text.token_embedding.weight": (cfg["vocab_size"], d), "text.positional_embedding": (cfg["context_length"], d),
         "text.ln_final.weight": (d,), "text.ln_final.bias": (d,), "text.text_projection.weight": (d, d),
         "text.text_projection.bias": (d,)}
    for i in range(cfg["layers"]):
        b = f"text.transformer.resblocks.{i}."
        s.update({b + "ln_1.weight": (d,), b + "ln_1.bias": (d,), b + "attn.in_proj_weight": (3 * d, d),
                  b + "attn.in_proj_bias": (3 * d,), b + "attn.out_proj.weight": (d, d), b + "attn.out_proj.bias": (d,),
                  b + "ln_2.weight": (d,), b + "ln_2.bias": (d,), b + "mlp.c_fc.weight": (m, d), b + "mlp.c_fc.bias": (m,),
                  b + "mlp.c_proj.weight": (d, m), b + "mlp.c_proj.bias": (d,)})
    return s


def synthetic_text_state_dict(cfg, seed=0x5EED0006):
    """Random-init text tower (see synthetic_state_dict)."""
    out = {}
    for idx, (name, shape) in enumerate(sorted(text_weight_shapes(cfg).items())):
        g = np.random.Generator(np.random.Philox(key=seed + idx))
        if ".ln_1.weight" in name or ".ln_2.weight" in name or name.endswith("ln_final.weight"):
            w = 1.0 + 0.1 * g.standard_normal(shape, dtype=np.float32)
        elif name.endswith("bias") or name.endswith("positional_embedding"):
            w = 0.02 * g.standard_normal(shape, dtype=np.float32)
        elif name.endswith("token_embedding.weight"):
            w = g.standard_normal(shape, dtype=np.float32)
        else:
            w = g.standard_normal(shape, dtype=np.float32) / np.float32(np.sqrt(shape[1]))
        out[name] = w.astype(np.float32)
    return out


def pad_tokens(token_lists, context_length=64, pad_id=PAD_ID):
    """What open_clip's SigLIP tokenizer wrapper does after sentencepiece: truncate to the context length and
    pad with the pad id (clip_server.py:129 `tokenizer(texts)`); returns int64 [n, context_length]."""
    out = np.full((len(token_lists), context_length), pad_id, np.int64)
    for i, t in enumerate(token_lists):
        t = list(t)[:context_length]
        out[i, :len(t)] = t
    return out


class SiglipTextEngine:
    def __init__(self, config=None, max_batch=32, eps=1e-6, gelu="erf"):
        cfg = dict(SO400M_TEXT if config is None else config)
        self.cfg = cfg
        self.max_batch = max_batch
        c = ffi.SiglipTextConfig(cfg["width"], cfg["layers"], cfg["heads"], cfg["mlp_dim"], cfg["context_length"],
                                 cfg["vocab_size"], eps, 1 if gelu == "tanh" else 0, max_batch)
        self._h = check_ptr(ffi.lib().mse_siglip_text_create(C.byref(c)), "mse_siglip_text_create")
        self.embedding_size = cfg["width"]
        self.context_length = cfg["context_length"]

    def weight_names(self):
        L = ffi.lib()
        return [L.mse_siglip_text_weight_name(self._h, i).decode() for i in range(L.mse_siglip_text_n_weights(self._h))]

    def set_weight(self, name, tensor):
        a = _to_numpy_f32(tensor)
        shape = (C.c_size_t * a.ndim)(*a.shape)
        check(ffi.lib().mse_siglip_text_set_weight(self._h, name.encode(), a.ctypes.data_as(ffi.f32p), shape, a.ndim),
              f"text set_weight({name})")

    @classmethod
    def from_state_dict(cls, state, config=None, max_batch=32, eps=1e-6, gelu="erf"):
        """state: open_clip state dict (keys `text.*`); missing tensors are an error."""
        eng = cls(config, max_batch, eps, gelu)
        for name in eng.weight_names():
            if name not in state:
                raise MseError(f"state dict lacks '{name}'")
            eng.set_weight(name, state[name])
        check(ffi.lib().mse_siglip_text_finalize(eng._h), "siglip_text_finalize")
        return eng

    def encode_text(self, tokens, normalize=True, out="f32"):
        """tokens: int [b, context_length] (already tokenised and padded).  Returns float32 [b, width] or fp16 bits."""
        t = np.ascontiguousarray(np.asarray(tokens), np.int64)
        if t.ndim != 2 or t.shape[1] != self.context_length:
            raise MseError(f"tokens must be [batch, {self.context_length}]")
        b = t.shape[0]
        if b > self.max_batch:
            raise MseError(f"max batch size is {self.max_batch}")
        if t.min() < 0 or t.max() >= self.cfg["vocab_size"]:
            raise MseError("token id outside the vocabulary")
        of32 = np.empty((b, self.embedding_size), np.float32) if out == "f32" else None
        of16 = np.empty((b, self.embedding_size), np.uint16) if out == "f16" else None
        check(ffi.lib().mse_siglip_text_encode(self._h, t.ctypes.data_as(ffi.i64p), b, int(normalize),
                                               of32.ctypes.data_as(ffi.f32p) if of32 is not None else None,
                                               of16.ctypes.data_as(ffi.u16p) if of16 is not None else None),
              "siglip_text_encode")
        return of32 if out == "f32" else of16

    def encode_text_device(self, tokens, normalize=True):
        """The same forward with the features left on the device: returns (f32 device pointer, f16 device pointer, hipStream_t of
        the engine) without waiting.  The query path hands the f16 rows to the search with no host round trip:
        `searcher.wait_stream(stream); mse.disk_query_topk(searcher, ..., (f16_ptr, b), ...)`.  The token array is kept alive on the
        engine until the next call."""
        t = np.ascontiguousarray(np.asarray(tokens), np.int64)
        if t.ndim != 2 or t.shape[1] != self.context_length:
            raise MseError(f"tokens must be [batch, {self.context_length}]")
        if t.shape[0] > self.max_batch:
            raise MseError(f"max batch size is {self.max_batch}")
        L = ffi.lib()
        check(L.mse_siglip_text_encode_dev(self._h, t.ctypes.data_as(ffi.i64p), t.shape[0], int(normalize)), "siglip_text_encode_dev")
        self._tokens_in_flight = t
        return L.mse_siglip_text_output_device(self._h, 0), L.mse_siglip_text_output_device(self._h, 1), L.mse_siglip_text_stream(self._h)

    def __call__(self, tokens):
        return self.encode_text(tokens)

    def close(self):
        if getattr(self, "_h", None):
            ffi.lib().mse_siglip_text_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
