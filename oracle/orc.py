"""ctypes loader for the CPU oracle (oracle/mse_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package (meme-search-engine_amd/) never
imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "mse_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _SO


class BuildConfig(C.Structure):
    """IndexBuildConfig (diskann/src/lib.rs:42-52); defaults of generate_index_shard.rs:22-33,85-94."""
    _fields_ = [("r", C.c_uint64), ("l", C.c_uint64), ("maxc", C.c_uint64), ("alpha", C.c_int64), ("query_alpha", C.c_int64),
                ("saturate_graph", C.c_uint32), ("query_breakpoint", C.c_uint32), ("max_add_per_stitch_iter", C.c_uint64)]

    @classmethod
    def make(cls, r=64, l=192, maxc=750, alpha=65536, query_alpha=65536, saturate_graph=False, query_breakpoint=0xFFFFFFFF,
             max_add_per_stitch_iter=16):
        return cls(r, l, maxc, alpha, query_alpha, int(saturate_graph), query_breakpoint, max_add_per_stitch_iter)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u16p, u32p, u8p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        f32p, i64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        sz = C.c_size_t
        sig = {
            "orc_have_avx2": (C.c_int, []),
            "orc_scale_dot_result": (C.c_int64, [C.c_float]),
            "orc_scale_dot_result_f64": (C.c_int64, [C.c_double]),
            "orc_f16_to_f32": (C.c_float, [C.c_uint16]),
            "orc_f32_to_f16": (C.c_uint16, [C.c_float]),
            "orc_f16_to_f32_n": (None, [u16p, sz, f32p]),
            "orc_f32_to_f16_n": (None, [f32p, sz, u16p]),
            "orc_fast_dot_f32": (C.c_float, [u16p, u16p, sz]),
            "orc_fast_dot": (C.c_int64, [u16p, u16p, sz]),
            "orc_fast_dot_scalar": (C.c_int64, [u16p, u16p, sz]),
            "orc_dot_f64": (C.c_int64, [u16p, u16p, sz]),
            "orc_score_all": (None, [u16p, sz, sz, u16p, i64p]),
            "orc_score_rows": (None, [u16p, sz, u32p, sz, u16p, i64p]),
            "orc_topk_from_scores": (sz, [i64p, sz, sz, i64p, u32p]),
            "orc_bruteforce_topk": (None, [u16p, sz, sz, u16p, sz, sz, i64p, u32p]),
            "orc_ranks_from_scores": (None, [i64p, sz, u32p]),
            "orc_recall_at_k": (C.c_double, [u32p, u32p, sz, sz]),
            "orc_descriptor_product": (C.c_int64, [f32p, sz, u8p, C.c_uint32]),
            "orc_pq_apply_transform": (None, [f32p, sz, f32p, sz, f32p]),
            "orc_pq_preprocess_query": (None, [f32p, sz, f32p, sz, sz, f32p, f32p]),
            "orc_pq_lut_from_transformed": (None, [f32p, sz, sz, sz, f32p, f32p]),
            "orc_pq_quantize_batch": (C.c_int, [f32p, sz, f32p, sz, sz, f32p, sz, C.c_int, u8p]),
            "orc_pq_adc": (None, [f32p, sz, sz, u8p, sz, i64p]),
            "orc_pq_adc_desc": (None, [f32p, sz, sz, u8p, u8p, sz, f32p, sz, i64p]),
            "orc_select_shard": (sz, [f32p, sz, sz, f32p]),
            "orc_nb_new": (C.c_void_p, [sz]),
            "orc_nb_free": (None, [C.c_void_p]),
            "orc_nb_clear": (None, [C.c_void_p]),
            "orc_nb_len": (sz, [C.c_void_p]),
            "orc_nb_cap": (sz, [C.c_void_p]),
            "orc_nb_ids": (u32p, [C.c_void_p]),
            "orc_nb_scores": (i64p, [C.c_void_p]),
            "orc_nb_visited": (u8p, [C.c_void_p]),
            "orc_nb_insert": (None, [C.c_void_p, C.c_uint32, C.c_int64]),
            "orc_nb_next_unvisited": (C.c_int, [C.c_void_p, u32p]),
            "orc_greedy_search": (sz, [u16p, sz, sz, u32p, u32p, sz, C.c_uint32, u16p, C.c_int, C.c_uint32, C.c_void_p]),
            "orc_disk_greedy_search": (sz, [u16p, sz, sz, u32p, u32p, sz, u8p, u8p, sz, sz, u8p, sz, C.c_uint32, u16p, f32p,
                                            f32p, C.c_int, sz, C.c_void_p, u32p, i64p, sz, C.POINTER(sz), C.POINTER(sz)]),
            "orc_dedup_keep": (None, [u16p, sz, sz, C.c_float, u8p]),
            "orc_score_batch": (None, [f32p, f32p, f32p, sz, sz, sz, f32p, sz, f32p]),
            "orc_descriptor_bucket": (C.c_uint8, [f32p, sz, C.c_float]),
            "orc_centroid_f16": (None, [u16p, sz, sz, u16p]),
            "orc_medioid": (C.c_uint32, [u16p, sz, sz]),
            "orc_index_ip": (C.c_float, [u16p, f32p, sz, C.c_int]),
            "orc_index_search": (None, [u16p, sz, sz, f32p, sz, sz, C.c_int, f32p, i64p]),
            "orc_total_embedding": (None, [u16p, f32p, sz, sz, f32p]),
            "orc_gen_row_ints": (None, [C.c_uint32, C.c_uint64, sz, i32p]),
            "orc_gen_rows_f16": (None, [C.c_uint32, C.c_uint64, sz, sz, u16p]),
            "orc_greedy_search_visited": (sz, [u16p, sz, sz, u32p, u32p, sz, C.c_uint32, u16p, C.c_int, C.c_uint32, C.c_void_p,
                                               u32p, i64p, sz]),
            "orc_robust_prune": (sz, [u16p, sz, u32p, i64p, sz, C.c_uint32, C.POINTER(BuildConfig), u32p]),
            "orc_build_graph": (None, [u16p, sz, sz, u32p, u32p, u32p, sz, sz, C.c_uint32, C.POINTER(BuildConfig)]),
            "orc_robust_stitch": (None, [u16p, sz, sz, u32p, u32p, u32p, C.POINTER(BuildConfig)]),
            "orc_random_fill_graph": (None, [C.c_uint32, sz, sz, u32p, u32p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---- thin numpy helpers -------------------------------------------------------------

def f16_bits(a):
    """float array -> uint16 IEEE half bits (RNE) using the oracle's own converter."""
    a = _c(a, np.float32)
    out = np.empty(a.shape, np.uint16)
    lib().orc_f32_to_f16_n(_p(a, C.c_float), a.size, _p(out, C.c_uint16))
    return out


def f16_to_f32(h):
    h = _c(h, np.uint16)
    out = np.empty(h.shape, np.float32)
    lib().orc_f16_to_f32_n(_p(h, C.c_uint16), h.size, _p(out, C.c_float))
    return out


def fast_dot(x, y):
    x, y = _c(x, np.uint16), _c(y, np.uint16)
    return int(lib().orc_fast_dot(_p(x, C.c_uint16), _p(y, C.c_uint16), x.size))


def fast_dot_scalar(x, y):
    x, y = _c(x, np.uint16), _c(y, np.uint16)
    return int(lib().orc_fast_dot_scalar(_p(x, C.c_uint16), _p(y, C.c_uint16), x.size))


def fast_dot_f32(x, y):
    x, y = _c(x, np.uint16), _c(y, np.uint16)
    return float(lib().orc_fast_dot_f32(_p(x, C.c_uint16), _p(y, C.c_uint16), x.size))


def dot_f64(x, y):
    x, y = _c(x, np.uint16), _c(y, np.uint16)
    return int(lib().orc_dot_f64(_p(x, C.c_uint16), _p(y, C.c_uint16), x.size))


def score_all(base, q):
    base, q = _c(base, np.uint16), _c(q, np.uint16)
    n, d = base.shape
    out = np.empty(n, np.int64)
    lib().orc_score_all(_p(base, C.c_uint16), n, d, _p(q, C.c_uint16), _p(out, C.c_int64))
    return out


def score_rows(base, ids, q):
    base, q, ids = _c(base, np.uint16), _c(q, np.uint16), _c(ids, np.uint32)
    out = np.empty(ids.size, np.int64)
    lib().orc_score_rows(_p(base, C.c_uint16), base.shape[1], _p(ids, C.c_uint32), ids.size, _p(q, C.c_uint16),
                         _p(out, C.c_int64))
    return out


def topk_from_scores(scores, k):
    scores = _c(scores, np.int64)
    os_, oi = np.empty(k, np.int64), np.empty(k, np.uint32)
    n = lib().orc_topk_from_scores(_p(scores, C.c_int64), scores.size, k, _p(os_, C.c_int64), _p(oi, C.c_uint32))
    return os_[:n], oi[:n]


def bruteforce_topk(base, q, k):
    base, q = _c(base, np.uint16), _c(q, np.uint16)
    q = q.reshape(-1, base.shape[1])
    n, d = base.shape
    nq = q.shape[0]
    os_, oi = np.empty((nq, k), np.int64), np.empty((nq, k), np.uint32)
    lib().orc_bruteforce_topk(_p(base, C.c_uint16), n, d, _p(q, C.c_uint16), nq, k, _p(os_, C.c_int64),
                              _p(oi, C.c_uint32))
    return os_, oi


def ranks_from_scores(scores):
    scores = _c(scores, np.int64)
    out = np.empty(scores.size, np.uint32)
    lib().orc_ranks_from_scores(_p(scores, C.c_int64), scores.size, _p(out, C.c_uint32))
    return out


def recall_at_k(rank_of_id, found_ids, k):
    r, f = _c(rank_of_id, np.uint32), _c(found_ids, np.uint32)
    return float(lib().orc_recall_at_k(_p(r, C.c_uint32), _p(f, C.c_uint32), f.size, k))


def descriptor_product(scales, descriptors, idx):
    s, d = _c(scales, np.float32), _c(descriptors, np.uint8)
    return int(lib().orc_descriptor_product(_p(s, C.c_float), s.size, _p(d, C.c_uint8), idx))


class PQ:
    """Mirror of diskann::vector::ProductQuantizer for the oracle."""

    def __init__(self, centroids, transform, n_dims_per_code, n_dims):
        self.centroids = _c(centroids, np.float32).reshape(-1, n_dims)
        self.transform = _c(transform, np.float32).reshape(n_dims, n_dims)
        self.dpc, self.d = n_dims_per_code, n_dims
        self.n_centroids = self.centroids.shape[0]
        self.n_chunks = n_dims // n_dims_per_code

    def apply_transform(self, x):
        x = _c(x, np.float32).reshape(-1, self.d)
        out = np.empty_like(x)
        lib().orc_pq_apply_transform(_p(self.transform, C.c_float), self.d, _p(x, C.c_float), x.shape[0],
                                     _p(out, C.c_float))
        return out

    def preprocess_query(self, q):
        q = _c(q, np.float32).reshape(self.d)
        lut = np.empty((self.n_chunks, self.n_centroids), np.float32)
        lib().orc_pq_preprocess_query(_p(self.centroids, C.c_float), self.n_centroids, _p(self.transform, C.c_float),
                                      self.d, self.dpc, _p(q, C.c_float), _p(lut, C.c_float))
        return lut

    def lut_from_transformed(self, t):
        t = _c(t, np.float32).reshape(self.d)
        lut = np.empty((self.n_chunks, self.n_centroids), np.float32)
        lib().orc_pq_lut_from_transformed(_p(self.centroids, C.c_float), self.n_centroids, self.d, self.dpc,
                                          _p(t, C.c_float), _p(lut, C.c_float))
        return lut

    def quantize_batch(self, x, pre_transformed=False):
        x = _c(x, np.float32).reshape(-1, self.d)
        codes = np.empty((x.shape[0], self.n_chunks), np.uint8)
        rc = lib().orc_pq_quantize_batch(_p(self.centroids, C.c_float), self.n_centroids, _p(self.transform, C.c_float),
                                         self.d, self.dpc, _p(x, C.c_float), x.shape[0], int(pre_transformed),
                                         _p(codes, C.c_uint8))
        if rc != 0:
            raise ValueError("more than 256 centroids")
        return codes

    def asymmetric_dot_product(self, lut, codes):
        lut, codes = _c(lut, np.float32), _c(codes, np.uint8).reshape(-1, self.n_chunks)
        out = np.empty(codes.shape[0], np.int64)
        lib().orc_pq_adc(_p(lut, C.c_float), self.n_chunks, self.n_centroids, _p(codes, C.c_uint8), codes.shape[0],
                         _p(out, C.c_int64))
        return out

    def adc_desc(self, lut, codes, descriptors, scales):
        lut, codes = _c(lut, np.float32), _c(codes, np.uint8).reshape(-1, self.n_chunks)
        desc, scales = _c(descriptors, np.uint8), _c(scales, np.float32)
        out = np.empty(codes.shape[0], np.int64)
        lib().orc_pq_adc_desc(_p(lut, C.c_float), self.n_chunks, self.n_centroids, _p(codes, C.c_uint8),
                              _p(desc, C.c_uint8), scales.size, _p(scales, C.c_float), codes.shape[0],
                              _p(out, C.c_int64))
        return out


def select_shard(centroids, q):
    c, q = _c(centroids, np.float32), _c(q, np.float32)
    return int(lib().orc_select_shard(_p(c, C.c_float), c.shape[0], c.shape[1], _p(q, C.c_float)))


class NeighbourBuffer:
    def __init__(self, cap):
        self._h = lib().orc_nb_new(cap)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_nb_free(self._h)
            self._h = None

    def insert(self, idx, score):
        lib().orc_nb_insert(self._h, int(idx), int(score))

    def next_unvisited(self):
        out = C.c_uint32()
        return int(out.value) if lib().orc_nb_next_unvisited(self._h, C.byref(out)) else None

    def clear(self):
        lib().orc_nb_clear(self._h)

    def __len__(self):
        return int(lib().orc_nb_len(self._h))

    @property
    def cap(self):
        return int(lib().orc_nb_cap(self._h))

    @property
    def ids(self):
        n = len(self)
        return np.ctypeslib.as_array(lib().orc_nb_ids(self._h), (n,)).copy() if n else np.empty(0, np.uint32)

    @property
    def scores(self):
        n = len(self)
        return np.ctypeslib.as_array(lib().orc_nb_scores(self._h), (n,)).copy() if n else np.empty(0, np.int64)

    @property
    def visited(self):
        n = len(self)
        return np.ctypeslib.as_array(lib().orc_nb_visited(self._h), (n,)).copy() if n else np.empty(0, np.uint8)


def greedy_search(vecs, adj, deg, start, query, cap, base_vectors_only=False, query_breakpoint=0xFFFFFFFF):
    vecs, adj, deg, query = _c(vecs, np.uint16), _c(adj, np.uint32), _c(deg, np.uint32), _c(query, np.uint16)
    nb = NeighbourBuffer(cap)
    n, d = vecs.shape
    dist = lib().orc_greedy_search(_p(vecs, C.c_uint16), n, d, _p(adj, C.c_uint32), _p(deg, C.c_uint32), adj.shape[1],
                                   start, _p(query, C.c_uint16), int(base_vectors_only), query_breakpoint, nb._h)
    return nb, int(dist)


def disk_greedy_search(vecs, adj, deg, codes, descriptors, start, query, lut, scales=None, disable_pq=False, beamwidth=1,
                       cap=1000, has_url=None, n_centroids=256):
    """src/query_disk_index.rs:144-212; returns (buffer, visited_ids, visited_scores, cmps, pq_cmps)."""
    vecs, adj, deg, query = _c(vecs, np.uint16), _c(adj, np.uint32), _c(deg, np.uint32), _c(query, np.uint16)
    codes, lut = _c(codes, np.uint8), _c(lut, np.float32)
    n, d = vecs.shape
    nd = 0
    if descriptors is not None:
        descriptors = _c(descriptors, np.uint8).reshape(n, -1)
        nd = descriptors.shape[1]
    sc = None if scales is None else _c(scales, np.float32)
    hu = None if has_url is None else _c(has_url, np.uint8)
    nb = NeighbourBuffer(cap)
    vids, vsc = np.empty(n, np.uint32), np.empty(n, np.int64)
    cm, pc = C.c_size_t(), C.c_size_t()
    nv = lib().orc_disk_greedy_search(
        _p(vecs, C.c_uint16), n, d, _p(adj, C.c_uint32), _p(deg, C.c_uint32), adj.shape[1],
        _p(hu, C.c_uint8) if hu is not None else None, _p(codes, C.c_uint8), codes.shape[1], n_centroids,
        _p(descriptors, C.c_uint8) if descriptors is not None else None, nd, start, _p(query, C.c_uint16),
        _p(lut, C.c_float), _p(sc, C.c_float) if sc is not None else None, int(disable_pq), beamwidth, nb._h,
        _p(vids, C.c_uint32), _p(vsc, C.c_int64), n, C.byref(cm), C.byref(pc))
    return nb, vids[:nv].copy(), vsc[:nv].copy(), int(cm.value), int(pc.value)


def dedup_keep(vecs, threshold=0.95):
    """src/query_disk_index.rs:482-527: keep mask over the visited vectors (visit order)."""
    vecs = _c(vecs, np.uint16)
    keep = np.zeros(vecs.shape[0], np.uint8)
    lib().orc_dedup_keep(_p(vecs, C.c_uint16), vecs.shape[0], vecs.shape[1], threshold, _p(keep, C.c_uint8))
    return keep


def score_batch(up, bias, down, x):
    up, bias, down = _c(up, np.float32), _c(bias, np.float32), _c(down, np.float32)
    x = _c(x, np.float32).reshape(-1, up.shape[1])
    out = np.empty((x.shape[0], down.shape[0]), np.float32)
    lib().orc_score_batch(_p(up, C.c_float), _p(bias, C.c_float), _p(down, C.c_float), up.shape[1], up.shape[0], down.shape[0],
                          _p(x, C.c_float), x.shape[0], _p(out, C.c_float))
    return out


def descriptor_buckets(cdfs, scores):
    cdfs, scores = _c(cdfs, np.float32), _c(scores, np.float32)
    out = np.empty(scores.shape, np.uint8)
    for i in range(scores.shape[0]):
        for j in range(scores.shape[1]):
            out[i, j] = lib().orc_descriptor_bucket(_p(cdfs[j], C.c_float), cdfs.shape[1], float(scores[i, j]))
    return out


def centroid_f16(vecs):
    vecs = _c(vecs, np.uint16)
    out = np.empty(vecs.shape[1], np.uint16)
    lib().orc_centroid_f16(_p(vecs, C.c_uint16), vecs.shape[0], vecs.shape[1], _p(out, C.c_uint16))
    return out


def medioid(vecs):
    vecs = _c(vecs, np.uint16)
    return int(lib().orc_medioid(_p(vecs, C.c_uint16), vecs.shape[0], vecs.shape[1]))


def index_search(codes, q, k, order=0):
    codes, q = _c(codes, np.uint16), _c(q, np.float32).reshape(-1, codes.shape[1])
    nq = q.shape[0]
    dist, lab = np.empty((nq, k), np.float32), np.empty((nq, k), np.int64)
    lib().orc_index_search(_p(codes, C.c_uint16), codes.shape[0], codes.shape[1], _p(q, C.c_float), nq, k, order,
                           _p(dist, C.c_float), _p(lab, C.c_int64))
    return dist, lab


def total_embedding(embs_f16, weights):
    e, w = _c(embs_f16, np.uint16), _c(weights, np.float32)
    out = np.empty(e.shape[1], np.float32)
    lib().orc_total_embedding(_p(e, C.c_uint16), _p(w, C.c_float), e.shape[0], e.shape[1], _p(out, C.c_float))
    return out


def gen_rows_f16(seed, row0, n_rows, d=1152):
    out = np.empty((n_rows, d), np.uint16)
    lib().orc_gen_rows_f16(seed, row0, n_rows, d, _p(out, C.c_uint16))
    return out


def gen_row_ints(seed, row, d=1152):
    out = np.empty(d, np.int32)
    lib().orc_gen_row_ints(seed, row, d, _p(out, C.c_int32))
    return out


# ---- Vamana graph build (diskann/src/lib.rs:213-389) -----------------------------------

def greedy_search_visited(vecs, adj, deg, start, query, cap, base_vectors_only=False, query_breakpoint=0xFFFFFFFF):
    """lib.rs:183-211 keeping scratch.visited_list: (buffer, visited ids, visited scores)."""
    vecs, adj, deg, query = _c(vecs, np.uint16), _c(adj, np.uint32), _c(deg, np.uint32), _c(query, np.uint16)
    nb = NeighbourBuffer(cap)
    n, d = vecs.shape
    vi, vs = np.empty(n, np.uint32), np.empty(n, np.int64)
    nv = lib().orc_greedy_search_visited(_p(vecs, C.c_uint16), n, d, _p(adj, C.c_uint32), _p(deg, C.c_uint32), adj.shape[1],
                                         start, _p(query, C.c_uint16), int(base_vectors_only), query_breakpoint, nb._h,
                                         _p(vi, C.c_uint32), _p(vs, C.c_int64), n)
    return nb, vi[:nv].copy(), vs[:nv].copy()


def robust_prune(vecs, cand_ids, cand_scores, p, cfg):
    vecs, ci, cs = _c(vecs, np.uint16), _c(cand_ids, np.uint32), _c(cand_scores, np.int64)
    out = np.empty(int(cfg.r), np.uint32)
    nn = lib().orc_robust_prune(_p(vecs, C.c_uint16), vecs.shape[1], _p(ci, C.c_uint32), _p(cs, C.c_int64), ci.size, p,
                                C.byref(cfg), _p(out, C.c_uint32))
    return out[:nn].copy()


def build_graph(vecs, adj, deg, order, medioid, cfg, batch=1):
    """In place on adj [n][r] / deg [n]."""
    vecs, order = _c(vecs, np.uint16), _c(order, np.uint32)
    assert adj.dtype == np.uint32 and deg.dtype == np.uint32 and adj.flags.c_contiguous and adj.shape[1] == cfg.r
    n, d = vecs.shape
    lib().orc_build_graph(_p(vecs, C.c_uint16), n, d, _p(adj, C.c_uint32), _p(deg, C.c_uint32), _p(order, C.c_uint32),
                          order.size, batch, medioid, C.byref(cfg))


def robust_stitch(vecs, adj, deg, queries_order, cfg):
    vecs, qo = _c(vecs, np.uint16), _c(queries_order, np.uint32)
    assert adj.dtype == np.uint32 and deg.dtype == np.uint32 and adj.flags.c_contiguous and adj.shape[1] == cfg.r
    n, d = vecs.shape
    assert qo.size == n - cfg.query_breakpoint
    lib().orc_robust_stitch(_p(vecs, C.c_uint16), n, d, _p(adj, C.c_uint32), _p(deg, C.c_uint32), _p(qo, C.c_uint32), C.byref(cfg))


def random_fill_graph(seed, n, r, adj=None, deg=None):
    if adj is None:
        adj, deg = np.zeros((n, r), np.uint32), np.zeros(n, np.uint32)
    lib().orc_random_fill_graph(seed, n, r, _p(adj, C.c_uint32), _p(deg, C.c_uint32))
    return adj, deg
