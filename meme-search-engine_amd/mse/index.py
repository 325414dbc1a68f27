"""Flat in-memory index with the semantics the small-scale server relies on:
FAISS `ScalarQuantizerIndexImpl::new(d, QT_fp16, InnerProduct)` (reference: src/main.rs:822),
`add` (:858,:892), `search` (:900), `ntotal` (:1015,:1053)."""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import check, check_ptr
from .vector import _p


class SearchResult:
    def __init__(self, distances, labels):
        self.distances, self.labels = distances, labels


class ScalarQuantizerIndex:
    def __init__(self, d):
        self.d = d
        self._h = check_ptr(ffi.lib().mse_index_new(d), "mse_index_new")

    def add(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.d)
        check(ffi.lib().mse_index_add(self._h, _p(x, C.c_float), x.shape[0]), "index.add")

    def ntotal(self):
        return int(ffi.lib().mse_index_ntotal(self._h))

    def search(self, query, k):
        """labels == -1 marks an empty slot (main.rs:908: `id.get()?`)."""
        q = np.ascontiguousarray(query, np.float32).reshape(-1, self.d)
        nq = q.shape[0]
        dist = np.empty((nq, k), np.float32)
        lab = np.empty((nq, k), np.int64)
        check(ffi.lib().mse_index_search(self._h, _p(q, C.c_float), nq, k, _p(dist, C.c_float), _p(lab, C.c_int64)),
              "index.search")
        return SearchResult(dist, lab)

    def stats(self):
        """Coalescer counters of this index (mse_index_stats)."""
        out = (C.c_uint64 * 6)()
        check(ffi.lib().mse_index_stats(self._h, out), "index.stats")
        return dict(zip(("queries", "requests", "passes", "max_pass_queries", "deadline_fires", "retried_alone"), map(int, out)))

    def close(self):
        if self._h:
            ffi.lib().mse_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
