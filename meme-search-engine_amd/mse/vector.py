"""Host mirror of `diskann::vector` (reference: diskann/src/vector.rs) over the C ABI.

Names, argument meaning and error behaviour follow the Rust module so that tests read like
tests of the reference.  f16 vectors are numpy uint16 arrays of IEEE binary16 bit patterns
(np.float16 arrays are accepted and viewed as bits).  All arithmetic happens on the device.
"""
import ctypes as C

import numpy as np

from . import ffi
from .ffi import MseError, check, check_ptr

SCALE = 4294967296.0  # vector.rs:46
ID_NONE = 0xFFFFFFFF

MODE_AUTO, MODE_EXACT, MODE_MFMA = 0, 1, 2


def _bits(a):
    a = np.asarray(a)
    if a.dtype == np.float16:
        a = a.view(np.uint16)
    if a.dtype != np.uint16:
        raise TypeError("f16 vectors must be np.float16 or np.uint16 bit patterns")
    return np.ascontiguousarray(a)


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def scale_dot_result(x):
    """vector.rs:408-411"""
    return int(ffi.lib().mse_scale_dot_f32(float(x)))


def scale_dot_result_f64(x):
    """vector.rs:413-416"""
    return int(ffi.lib().mse_scale_dot_f64(float(x)))


def fast_dot_noprefetch(x, y):
    """vector.rs:255-306.  len % 64 == 0 is required (debug_assert at :259)."""
    x, y = _bits(x).reshape(-1), _bits(y).reshape(-1)
    if x.size != y.size:
        raise MseError("fast_dot: length mismatch")
    out = C.c_int64()
    check(ffi.lib().mse_fast_dot_f16(_p(x, C.c_uint16), _p(y, C.c_uint16), x.size, C.byref(out)), "fast_dot")
    return int(out.value)


def fast_dot(x, y, prefetch=None):
    """vector.rs:192-252: same arithmetic; the third vector is only prefetched."""
    return fast_dot_noprefetch(x, y)


class VectorList:
    """vector.rs:118-186: row-major contiguous f16 rows, here resident in HBM."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keep = keepalive

    @classmethod
    def from_f16s(cls, f16s, d):
        a = _bits(f16s).reshape(-1)
        if a.size % d != 0:
            raise MseError("from_f16s: data length is not a multiple of d")  # assert at vector.rs:174
        n = a.size // d
        return cls(check_ptr(ffi.lib().mse_base_from_host(_p(a, C.c_uint16), n, d), "mse_base_from_host"))

    @classmethod
    def generate(cls, seed, first_row, n_rows, d=1152):
        """Synthetic unit-norm rows made on the device (bit-identical to oracle.gen_rows_f16)."""
        return cls(check_ptr(ffi.lib().mse_base_generate(seed, first_row, n_rows, d), "mse_base_generate"))

    @classmethod
    def wrap_device(cls, dev_ptr, n_rows, d, keepalive=None):
        return cls(check_ptr(ffi.lib().mse_base_wrap_device(dev_ptr, n_rows, d), "mse_base_wrap_device"), keepalive)

    def __len__(self):
        return int(ffi.lib().mse_base_len(self._h))

    @property
    def d_emb(self):
        return int(ffi.lib().mse_base_dim(self._h))

    @property
    def device_ptr(self):
        return ffi.lib().mse_base_device_ptr(self._h)

    def rows(self, first, n):
        out = np.empty((n, self.d_emb), np.uint16)
        check(ffi.lib().mse_base_read_rows(self._h, first, n, _p(out, C.c_uint16)), "mse_base_read_rows")
        return out

    def __getitem__(self, i):
        return self.rows(i, 1)[0]

    def close(self):
        if self._h:
            ffi.lib().mse_base_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Searcher:
    """Per-thread scratch + stream (reference `Scratch`: lib.rs:157-175, query_disk_index.rs:116-123)."""

    def __init__(self, vecs: VectorList):
        self.vecs = vecs
        self._h = check_ptr(ffi.lib().mse_searcher_new(vecs._h), "mse_searcher_new")

    def set_stream(self, hip_stream):
        check(ffi.lib().mse_searcher_set_stream(self._h, hip_stream), "set_stream")

    def wait_stream(self, producer_stream):
        """Order this searcher's stream after what `producer_stream` (a hipStream_t value) holds now: call it before handing the
        searcher device-resident inputs that another stream wrote."""
        check(ffi.lib().mse_searcher_wait_stream(self._h, producer_stream), "wait_stream")

    def beam_timing(self, enable):
        """Measurement hook of the graph search (mse_searcher_beam_timing): returns what has accumulated so far -- kernel_ms, launches,
        queries, rows_scored (2304-byte row gathers at d = 1152), nodes_fetched (adjacency lists), adc_scored (64-byte code gathers) --
        then sets the switch (0 off, 1 on, 2 on and reset)."""
        out = (C.c_uint64 * 8)()
        check(ffi.lib().mse_searcher_beam_timing(self._h, int(enable), out), "beam_timing")
        return {"kernel_ms": out[0] / 1e3, "launches": int(out[1]), "queries": int(out[2]), "rows_scored": int(out[3]),
                "nodes_fetched": int(out[4]), "adc_scored": int(out[5]), "iterations": int(out[6]), "iterations_replayed": int(out[7])}

    def bruteforce_topk(self, queries, k, mode=MODE_AUTO):
        """Brute-force scan + ranking of `evaluate` (query_disk_index.rs:262-273) for a query batch.
        Returns (scores int64 [nq,k], ids uint32 [nq,k])."""
        d = self.vecs.d_emb
        q = _bits(queries).reshape(-1, d)
        nq = q.shape[0]
        scores = np.empty((nq, k), np.int64)
        ids = np.empty((nq, k), np.uint32)
        check(ffi.lib().mse_bruteforce_topk_f16(self._h, _p(q, C.c_uint16), nq, k, mode, _p(scores, C.c_int64),
                                                _p(ids, C.c_uint32)), "bruteforce_topk")
        return scores, ids

    def bruteforce_topk_dev(self, queries_dev, nq, k, scores_dev, ids_dev, mode=MODE_AUTO, id_offset=0):
        check(ffi.lib().mse_bruteforce_topk_f16_dev(self._h, queries_dev, nq, k, mode, id_offset, scores_dev, ids_dev),
              "bruteforce_topk_dev")

    def merge_topk_dev(self, gathered_scores_dev, gathered_ids_dev, n_shards, nq, k, out_scores_dev, out_ids_dev):
        """k-way merge of all-gathered [n_shards][nq][k] shard results (device pointers)."""
        check(ffi.lib().mse_merge_topk_dev(self._h, gathered_scores_dev, gathered_ids_dev, n_shards, nq, k,
                                           out_scores_dev, out_ids_dev), "merge_topk_dev")

    def scores(self, query):
        q = _bits(query).reshape(-1)
        out = np.empty(len(self.vecs), np.int64)
        check(ffi.lib().mse_bruteforce_scores_f16(self._h, _p(q, C.c_uint16), _p(out, C.c_int64)), "bruteforce_scores")
        return out

    def ranks(self, query, ids):
        q = _bits(query).reshape(-1)
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.empty(ids.size, np.uint32)
        check(ffi.lib().mse_bruteforce_ranks_f16(self._h, _p(q, C.c_uint16), _p(ids, C.c_uint32), ids.size,
                                                 _p(out, C.c_uint32)), "bruteforce_ranks")
        return out

    def score_rows(self, ids, query):
        """out[i] = fast_dot(query, vecs[ids[i]]) (lib.rs:201-207, query_disk_index.rs:168-169)."""
        q = _bits(query).reshape(-1)
        ids = np.ascontiguousarray(ids, np.uint32)
        out = np.empty(ids.size, np.int64)
        check(ffi.lib().mse_score_rows_f16(self._h, _p(ids, C.c_uint32), ids.size, _p(q, C.c_uint16), _p(out, C.c_int64)),
              "score_rows")
        return out

    def scan_timing(self, enable):
        """HIP-event totals of the scan kernel so far -> (total_ms, launches); then set mode
        (0 off, 1 on, 2 on + reset)."""
        ms, n = C.c_double(), C.c_uint64()
        check(ffi.lib().mse_searcher_scan_timing(self._h, enable, C.byref(ms), C.byref(n)), "scan_timing")
        return float(ms.value), int(n.value)

    def last_stats(self):
        a, b = C.c_uint32(), C.c_uint32()
        check(ffi.lib().mse_searcher_last_stats(self._h, C.byref(a), C.byref(b)), "last_stats")
        return {"widened_queries": int(a.value), "max_groups": int(b.value)}

    def close(self):
        if self._h:
            ffi.lib().mse_searcher_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dispatcher:
    """Cross-thread query coalescer over a VectorList (include/mse.h `mse_dispatcher`): the meeting point of the reference's
    one-query-per-request threads (src/main.rs:896-934,1043-1049; src/query_disk_index.rs:711-736).  `search` may be called
    from any number of threads; callers waiting at the same time share one pass over the rows."""

    def __init__(self, vecs: VectorList, max_queries_per_pass=0, max_wait_us=0):
        self.vecs = vecs
        self._h = check_ptr(ffi.lib().mse_dispatcher_new(vecs._h, max_queries_per_pass, max_wait_us), "mse_dispatcher_new")

    def search(self, queries, k):
        d = self.vecs.d_emb
        q = _bits(queries).reshape(-1, d)
        nq = q.shape[0]
        scores = np.empty((nq, k), np.int64)
        ids = np.empty((nq, k), np.uint32)
        check(ffi.lib().mse_dispatcher_topk_f16(self._h, _p(q, C.c_uint16), nq, k, _p(scores, C.c_int64), _p(ids, C.c_uint32)),
              "dispatcher.search")
        return scores, ids

    def stats(self):
        out = (C.c_uint64 * 6)()
        check(ffi.lib().mse_dispatcher_stats(self._h, out), "dispatcher.stats")
        return dict(zip(("queries", "requests", "passes", "max_pass_queries", "deadline_fires", "retried_alone"), map(int, out)))

    def searcher_handle(self):
        return ffi.lib().mse_dispatcher_searcher(self._h)

    def close(self):
        if self._h:
            ffi.lib().mse_dispatcher_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QueryLUT:
    """vector.rs:316-317: chunk-major table [n_chunks][n_centroids] of f32."""

    def __init__(self, table):
        self.table = np.ascontiguousarray(table, np.float32)


class ProductQuantizer:
    """vector.rs:308-406.  Fields as serialised in opq.msgpack (diskann/aopq_train.py:87-93)."""

    def __init__(self, centroids, transform, n_dims_per_code, n_dims):
        centroids = np.ascontiguousarray(centroids, np.float32).reshape(-1)
        transform = np.ascontiguousarray(transform, np.float32).reshape(-1)
        if transform.size != n_dims * n_dims:
            raise MseError("transform must be n_dims x n_dims")  # assert_eq at vector.rs:334
        if centroids.size % n_dims != 0:
            raise MseError("centroids must be rows of n_dims")
        self.n_dims, self.n_dims_per_code = n_dims, n_dims_per_code
        self.n_centroids = centroids.size // n_dims
        self.n_chunks = n_dims // n_dims_per_code
        self._h = check_ptr(ffi.lib().mse_pq_load(_p(centroids, C.c_float), self.n_centroids, _p(transform, C.c_float),
                                                  n_dims, n_dims_per_code), "mse_pq_load")

    @classmethod
    def from_msgpack(cls, blob):
        import msgpack
        m = msgpack.unpackb(blob, raw=False)
        return cls(m["centroids"], m["transform"], m["n_dims_per_code"], m["n_dims"])

    def apply_transform(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.n_dims)
        out = np.empty_like(x)
        check(ffi.lib().mse_pq_apply_transform(self._h, _p(x, C.c_float), x.shape[0], _p(out, C.c_float)),
              "apply_transform")
        return out

    def quantize_batch(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.n_dims)
        codes = np.empty((x.shape[0], self.n_chunks), np.uint8)
        check(ffi.lib().mse_pq_quantize_batch(self._h, _p(x, C.c_float), x.shape[0], _p(codes, C.c_uint8)),
              "quantize_batch")
        return codes

    def preprocess_query(self, query):
        q = np.ascontiguousarray(query, np.float32).reshape(self.n_dims)
        lut = np.empty((self.n_chunks, self.n_centroids), np.float32)
        check(ffi.lib().mse_pq_preprocess_query(self._h, _p(q, C.c_float), _p(lut, C.c_float)), "preprocess_query")
        return QueryLUT(lut)

    def asymmetric_dot_product(self, lut, pq_vectors):
        table = lut.table if isinstance(lut, QueryLUT) else np.ascontiguousarray(lut, np.float32)
        codes = np.ascontiguousarray(pq_vectors, np.uint8).reshape(-1, self.n_chunks)
        out = np.empty(codes.shape[0], np.int64)
        check(ffi.lib().mse_pq_adc(self._h, _p(table, C.c_float), _p(codes, C.c_uint8), codes.shape[0],
                                   _p(out, C.c_int64)), "asymmetric_dot_product")
        return out

    def adc_gather(self, codes, lut, ids, scales=None):
        table = lut.table if isinstance(lut, QueryLUT) else np.ascontiguousarray(lut, np.float32)
        ids = np.ascontiguousarray(ids, np.uint32)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float32)
        out = np.empty(ids.size, np.int64)
        check(ffi.lib().mse_pq_adc_gather(self._h, codes._h, _p(table, C.c_float),
                                          _p(sc, C.c_float) if sc is not None else None, _p(ids, C.c_uint32), ids.size,
                                          _p(out, C.c_int64)), "adc_gather")
        return out

    def scan_topk(self, codes, query_f32, r, k, searcher=None, scales=None):
        q = np.ascontiguousarray(query_f32, np.float32).reshape(self.n_dims)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float32)
        scores = np.empty(k, np.int64)
        ids = np.empty(k, np.uint32)
        check(ffi.lib().mse_pq_scan_topk(self._h, codes._h, searcher._h if searcher is not None else None,
                                         _p(q, C.c_float), _p(sc, C.c_float) if sc is not None else None, r, k,
                                         _p(scores, C.c_int64), _p(ids, C.c_uint32)), "pq_scan_topk")
        return scores, ids

    def scan_topk_batch(self, codes, queries_f32, r, k, searcher=None, scales=None):
        """scan_topk for [nq, n_dims] queries in one call (one upload, the scans back to back, one download) -> ([nq,k], [nq,k])."""
        q = np.ascontiguousarray(queries_f32, np.float32).reshape(-1, self.n_dims)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float32)
        scores = np.empty((q.shape[0], k), np.int64)
        ids = np.empty((q.shape[0], k), np.uint32)
        check(ffi.lib().mse_pq_scan_topk_batch(self._h, codes._h, searcher._h if searcher is not None else None,
                                               _p(q, C.c_float), q.shape[0], _p(sc, C.c_float) if sc is not None else None, r, k,
                                               _p(scores, C.c_int64), _p(ids, C.c_uint32)), "pq_scan_topk_batch")
        return scores, ids

    def scan_timing(self, enable):
        """HIP-event totals of the four-query scan kernel so far -> (total_ms, launches); then set mode (0 off, 1 on, 2 on + reset)."""
        ms, n = C.c_double(), C.c_uint64()
        check(ffi.lib().mse_pq_scan_timing(self._h, enable, C.byref(ms), C.byref(n)), "pq_scan_timing")
        return float(ms.value), int(n.value)

    def scan_sustained(self):
        """(span_ms, scans) of the batch calls with at least four scans made while timing was on: first scan's start to last scan's
        end -- the back-to-back cost of a pass (reset by scan_timing(2))."""
        ms, n = C.c_double(), C.c_uint64()
        check(ffi.lib().mse_pq_scan_sustained(self._h, C.byref(ms), C.byref(n)), "pq_scan_sustained")
        return float(ms.value), int(n.value)

    @property
    def last_uncertified(self):
        """Queries of the last scan_topk_batch call that the four-query scan could not certify and repeated through the exact scan."""
        return int(ffi.lib().mse_pq_last_uncertified(self._h))

    def debug_group_max(self, codes, lut0, lut1=None, scales=None):
        """Test hook: the flat scan's group maxima (one i64 per 64 vectors) for one table, or for a pair through the
        two-queries-per-pass kernel."""
        ng = (len(codes) + 63) // 64
        l0 = np.ascontiguousarray(lut0, np.float32)
        l1 = None if lut1 is None else np.ascontiguousarray(lut1, np.float32)
        sc = None if scales is None else np.ascontiguousarray(scales, np.float32)
        o0 = np.empty(ng, np.int64)
        o1 = np.empty(ng, np.int64) if l1 is not None else None
        check(ffi.lib().mse_debug_pq_group_max(self._h, codes._h, _p(l0, C.c_float), _p(l1, C.c_float) if l1 is not None else None,
                                               _p(sc, C.c_float) if sc is not None else None, _p(o0, C.c_int64),
                                               _p(o1, C.c_int64) if o1 is not None else None), "debug_pq_group_max")
        return (o0, o1) if l1 is not None else o0

    def close(self):
        if self._h:
            ffi.lib().mse_pq_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Codes:
    """PQ codes (+ descriptor bytes) in HBM: index.pq-codes.bin / index.descriptor-codes.bin
    (query_disk_index.rs:686-709)."""

    def __init__(self, codes, descriptors=None):
        codes = np.ascontiguousarray(codes, np.uint8)
        n, cs = codes.shape
        if descriptors is not None:
            descriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(n, -1)
            nd = descriptors.shape[1]
            dp = _p(descriptors, C.c_uint8)
        else:
            nd, dp = 0, None
        self._h = check_ptr(ffi.lib().mse_codes_from_host(_p(codes, C.c_uint8), n, cs, dp, nd), "mse_codes_from_host")

    @classmethod
    def quantize_base(cls, pq, vecs, descriptors=None):
        """Codes of rows already resident in HBM (mse_codes_quantize_base): quantize_batch over the f32 widenings of the f16 rows,
        on the device -- the encode step of src/dump_processor.rs:468-481."""
        self = cls.__new__(cls)
        if descriptors is not None:
            descriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(len(vecs), -1)
            nd, dp = descriptors.shape[1], _p(descriptors, C.c_uint8)
        else:
            nd, dp = 0, None
        self._h = check_ptr(ffi.lib().mse_codes_quantize_base(pq._h, vecs._h, dp, nd), "mse_codes_quantize_base")
        return self

    def __len__(self):
        return int(ffi.lib().mse_codes_len(self._h))

    def close(self):
        if self._h:
            ffi.lib().mse_codes_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def descriptor_product(scales, descriptors, idx):
    """query_disk_index.rs:135-142"""
    s = np.ascontiguousarray(scales, np.float32)
    d = np.ascontiguousarray(descriptors, np.uint8)
    return int(ffi.lib().mse_descriptor_product(_p(s, C.c_float), s.size, _p(d, C.c_uint8), int(idx)))
