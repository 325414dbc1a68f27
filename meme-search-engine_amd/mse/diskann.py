"""Host mirror of the `diskann` crate's search pieces (reference: diskann/src/lib.rs)."""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import MseError, check, check_ptr
from .vector import Searcher, _bits, _p


class NeighbourBuffer:
    """lib.rs:74-155: candidate list sorted by score descending, bounded by `size`."""

    def __init__(self, size):
        self._h = check_ptr(ffi.lib().mse_nb_new(size), "mse_nb_new")

    def insert(self, idx, score):
        ffi.lib().mse_nb_insert(self._h, int(idx), int(score))

    def next_unvisited(self):
        out = C.c_uint32()
        return int(out.value) if ffi.lib().mse_nb_next_unvisited(self._h, C.byref(out)) else None

    def clear(self):
        ffi.lib().mse_nb_clear(self._h)

    def __len__(self):
        return int(ffi.lib().mse_nb_len(self._h))

    def len(self):
        return len(self)

    def cap(self):
        return int(ffi.lib().mse_nb_cap(self._h))

    @property
    def ids(self):
        n = len(self)
        return np.ctypeslib.as_array(ffi.lib().mse_nb_ids(self._h), (n,)).copy() if n else np.empty(0, np.uint32)

    @property
    def scores(self):
        n = len(self)
        return np.ctypeslib.as_array(ffi.lib().mse_nb_scores(self._h), (n,)).copy() if n else np.empty(0, np.int64)

    def close(self):
        if self._h:
            ffi.lib().mse_nb_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class IndexGraph:
    """lib.rs:16-39 as a fixed-degree table: adj[n][max_deg], deg[n]."""

    def __init__(self, adj, deg):
        self.adj = np.ascontiguousarray(adj, np.uint32)
        self.deg = np.ascontiguousarray(deg, np.uint32)


def greedy_search(searcher: Searcher, start, base_vectors_only, query, graph: IndexGraph, l,
                  query_breakpoint=0xFFFFFFFF):
    """lib.rs:183-211.  Returns (NeighbourBuffer, distances); results are buffer.ids, best first."""
    q = _bits(query).reshape(-1)
    buf = NeighbourBuffer(l)
    nd = C.c_size_t()
    check(ffi.lib().mse_greedy_search(searcher._h, _p(graph.adj, C.c_uint32), _p(graph.deg, C.c_uint32),
                                      graph.adj.shape[1], int(start), _p(q, C.c_uint16), int(bool(base_vectors_only)),
                                      int(query_breakpoint), buf._h, C.byref(nd)), "greedy_search")
    return buf, int(nd.value)
