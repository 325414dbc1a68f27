"""Host mirror of the index-packing pieces of dump_processor: ScoreModel (src/score_model.rs) and the CDF inversion that
turns score channels into descriptor bytes (src/dump_processor.rs:483-491).  quantize_batch lives on ProductQuantizer."""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import MseError, check, check_ptr
from .vector import _p


class ScoreModel:
    """src/score_model.rs:4-49.  Tensors as stored in model.safetensors: up_proj [d_hidden, d_emb], bias [d_hidden],
    down_proj [output_channels, d_hidden]."""

    def __init__(self, up_proj, bias, down_proj):
        up = np.ascontiguousarray(up_proj, np.float32)
        b = np.ascontiguousarray(bias, np.float32).reshape(-1)
        down = np.ascontiguousarray(down_proj, np.float32)
        if up.ndim != 2 or down.ndim != 2 or up.shape[0] != b.size or down.shape[1] != b.size:
            raise MseError("score model: inconsistent tensor shapes")
        self.d_emb, self.d_hidden, self.output_channels = up.shape[1], up.shape[0], down.shape[0]
        self._h = check_ptr(ffi.lib().mse_score_model_load(_p(up, C.c_float), _p(b, C.c_float), _p(down, C.c_float), self.d_emb,
                                                           self.d_hidden, self.output_channels), "mse_score_model_load")

    @classmethod
    def load(cls, path):
        from safetensors.numpy import load_file
        t = load_file(path)
        return cls(t["up_proj"], t["bias"], t["down_proj"])

    def score_batch(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.d_emb)
        out = np.empty((x.shape[0], self.output_channels), np.float32)
        check(ffi.lib().mse_score_model_score_batch(self._h, _p(x, C.c_float), x.shape[0], _p(out, C.c_float)), "score_batch")
        return out

    def close(self):
        if getattr(self, "_h", None):
            ffi.lib().mse_score_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def descriptor_buckets(cdfs, scores):
    """dump_processor.rs:483-491: scores [n, n_desc] -> bytes [n, n_desc] through the ascending CDF of each channel."""
    c = np.ascontiguousarray(cdfs, np.float32)
    s = np.ascontiguousarray(scores, np.float32)
    if c.ndim != 2 or s.ndim != 2 or s.shape[1] != c.shape[0]:
        raise MseError("cdfs must be [n_desc, cdf_len] and scores [n, n_desc]")
    out = np.empty(s.shape, np.uint8)
    check(ffi.lib().mse_descriptor_buckets(_p(c, C.c_float), c.shape[0], c.shape[1], _p(s, C.c_float), s.shape[0], _p(out, C.c_uint8)),
          "descriptor_buckets")
    return out
