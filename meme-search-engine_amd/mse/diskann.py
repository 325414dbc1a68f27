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


def medioid(vecs):
    """lib.rs:52-68: id of the row closest to the (f16-rounded, running-mean) centroid."""
    out = C.c_uint32()
    check(ffi.lib().mse_medioid(vecs._h, C.byref(out)), "medioid")
    return int(out.value)


def select_shard(centroids, query):
    """query_disk_index.rs:254-256,447-450: index of the shard whose centroid has the largest dot with the query
    (last maximum on ties)."""
    c = np.ascontiguousarray(centroids, np.float32)
    q = np.ascontiguousarray(query, np.float32).reshape(-1)
    if c.ndim != 2 or c.shape[1] != q.size:
        raise MseError("centroids must be [n_shards, d]")
    out = C.c_size_t()
    check(ffi.lib().mse_select_shard(_p(c, C.c_float), c.shape[0], c.shape[1], _p(q, C.c_float), C.byref(out)), "select_shard")
    return int(out.value)


DUPLICATES_THRESHOLD = 0.95   # query_disk_index.rs:99


def dedup_visited(searcher: Searcher, visited_ids, threshold=DUPLICATES_THRESHOLD):
    """query_disk_index.rs:482-527: boolean keep mask over the visited list (visit order)."""
    ids = np.ascontiguousarray(visited_ids, np.uint32)
    keep = np.zeros(ids.size, np.uint8)
    check(ffi.lib().mse_dedup_visited(searcher._h, _p(ids, C.c_uint32), ids.size, float(threshold), _p(keep, C.c_uint8)),
          "dedup_visited")
    return keep.astype(bool)


class DiskSearchResult:
    """What query_disk_index::greedy_search leaves behind: the Scratch's neighbour buffer and visited list
    (ids + exact scores, fetch order) and the returned (cmps, pq_cmps)."""

    def __init__(self, buffer, visited_ids, visited_scores, cmps, pq_cmps):
        self.neighbour_buffer = buffer
        self.visited_ids = visited_ids
        self.visited_scores = visited_scores
        self.cmps = cmps
        self.pq_cmps = pq_cmps


def disk_greedy_search(searcher: Searcher, quantizer, codes, graph: IndexGraph, start, query, query_preprocessed,
                       descriptor_scales=None, disable_pq=False, beamwidth=1, search_list=1000, has_url=None):
    """query_disk_index.rs:144-212 with the index HBM-resident.  `searcher` wraps the record vectors, `codes` the PQ
    codes + descriptor bytes, `graph` the adjacency, `query_preprocessed` a QueryLUT, `search_list` = capacity of
    the NeighbourBuffer (config.search_list / 1000 in evaluate, :288)."""
    q = _bits(query).reshape(-1)
    table = getattr(query_preprocessed, "table", query_preprocessed)
    table = np.ascontiguousarray(table, np.float32)
    sc = None if descriptor_scales is None else np.ascontiguousarray(descriptor_scales, np.float32)
    hu = None if has_url is None else np.ascontiguousarray(has_url, np.uint8)
    n = len(codes)
    buf = NeighbourBuffer(search_list)
    vids = np.empty(n, np.uint32)
    vsc = np.empty(n, np.int64)
    nv, cm, pc = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(ffi.lib().mse_disk_greedy_search(
        searcher._h, quantizer._h, codes._h, _p(graph.adj, C.c_uint32), _p(graph.deg, C.c_uint32), graph.adj.shape[1],
        _p(hu, C.c_uint8) if hu is not None else None, int(start), _p(q, C.c_uint16), _p(table, C.c_float),
        _p(sc, C.c_float) if sc is not None else None, int(bool(disable_pq)), int(beamwidth), buf._h,
        _p(vids, C.c_uint32), _p(vsc, C.c_int64), n, C.byref(nv), C.byref(cm), C.byref(pc)), "disk_greedy_search")
    k = int(nv.value)
    return DiskSearchResult(buf, vids[:k].copy(), vsc[:k].copy(), int(cm.value), int(pc.value))


class DeviceGraph:
    """Adjacency (and the has-url flags of the records) resident in HBM for disk_search_batch."""

    def __init__(self, graph: IndexGraph, has_url=None):
        hu = None if has_url is None else np.ascontiguousarray(has_url, np.uint8)
        self._h = check_ptr(ffi.lib().mse_graph_from_host(_p(graph.adj, C.c_uint32), _p(graph.deg, C.c_uint32), graph.adj.shape[0],
                                                          graph.adj.shape[1], _p(hu, C.c_uint8) if hu is not None else None),
                            "mse_graph_from_host")

    def close(self):
        if getattr(self, "_h", None):
            ffi.lib().mse_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def disk_search_batch(searcher: Searcher, quantizer, codes, dgraph: DeviceGraph, starts, queries, luts=None, descriptor_scales=None,
                      disable_pq=False, beamwidth=1, search_list=1000, visited_cap=4096, as_arrays=False):
    """query_disk_index::greedy_search for a batch of queries, entirely on the device (one workgroup per query).
    as_arrays=True returns the padded output arrays themselves (dict: buf_ids, buf_scores, buf_len, visited_ids,
    visited_scores, n_visited, cmps, pq_cmps) instead of one tuple per query.
    queries: f16 rows with `luts` (QueryLUTs / [nq][64*256] f32; not needed with disable_pq), or f32 rows with luts=None --
    then the f16 copies and the tables are made on the device (query_disk_index.rs:475-477).
    Returns a list of (buffer ids, buffer scores, visited ids, visited scores, cmps, pq_cmps)."""
    qa = np.asarray(queries)
    from_f32 = qa.dtype == np.float32 and luts is None
    if from_f32:
        q = np.ascontiguousarray(qa.reshape(-1, qa.shape[-1]), np.float32)
    else:
        q = _bits(queries)
        q = q.reshape(-1, q.shape[-1])
    nq = q.shape[0]
    tables = None
    if not from_f32 and luts is not None:
        tables = np.ascontiguousarray(np.stack([getattr(t, "table", t) for t in luts]) if not isinstance(luts, np.ndarray) else luts,
                                      np.float32).reshape(nq, -1)
    st = np.ascontiguousarray(starts, np.uint32).reshape(nq)
    sc = None
    if descriptor_scales is not None:
        sc = np.ascontiguousarray(descriptor_scales, np.float32)
        if sc.ndim == 1:
            sc = np.ascontiguousarray(np.broadcast_to(sc, (nq, sc.size)))
    bi, bs, bl = np.empty((nq, search_list), np.uint32), np.empty((nq, search_list), np.int64), np.empty(nq, np.uint32)
    vi, vs = np.empty((nq, visited_cap), np.uint32), np.empty((nq, visited_cap), np.int64)
    nv, cm, pc = np.empty(nq, np.uint32), np.empty(nq, np.uint32), np.empty(nq, np.uint32)
    tail = (nq, int(bool(disable_pq)), int(beamwidth), int(search_list), _p(bi, C.c_uint32), _p(bs, C.c_int64), _p(bl, C.c_uint32),
            _p(vi, C.c_uint32), _p(vs, C.c_int64), visited_cap, _p(nv, C.c_uint32), _p(cm, C.c_uint32), _p(pc, C.c_uint32))
    scp = _p(sc, C.c_float) if sc is not None else None
    if from_f32:
        check(ffi.lib().mse_disk_search_batch_f32(searcher._h, quantizer._h if quantizer is not None else None,
                                                  codes._h if codes is not None else None, dgraph._h, _p(st, C.c_uint32), _p(q, C.c_float),
                                                  scp, *tail), "disk_search_batch_f32")
    else:
        check(ffi.lib().mse_disk_search_batch(searcher._h, quantizer._h if quantizer is not None else None,
                                              codes._h if codes is not None else None, dgraph._h, _p(st, C.c_uint32), _p(q, C.c_uint16),
                                              _p(tables, C.c_float) if tables is not None else None, scp, *tail), "disk_search_batch")
    if as_arrays:
        return {"buf_ids": bi, "buf_scores": bs, "buf_len": bl, "visited_ids": vi, "visited_scores": vs, "n_visited": nv, "cmps": cm,
                "pq_cmps": pc}
    if int(nv.max(initial=0)) > visited_cap:
        # the reference keeps every visited record (query_disk_index.rs:168-186); a silently shortened list would change results
        raise MseError(f"a search visited {int(nv.max())} records but visited_cap is {visited_cap}: raise visited_cap "
                       "(as_arrays=True returns the truncated arrays together with n_visited instead)")
    out = []
    for i in range(nq):
        k = min(int(nv[i]), visited_cap)
        out.append((bi[i, :bl[i]].copy(), bs[i, :bl[i]].copy(), vi[i, :k].copy(), vs[i, :k].copy(), int(cm[i]), int(pc[i])))
    return out


def set_entries(dgraph, vecs, node_ids):
    """The entry table of disk_query_topk: `node_ids` name the entry records (the reference: the shard medioids,
    query_disk_index.rs:254-256,447-450; for a one-piece index a sample of its rows); copies of their vectors stay with the graph."""
    ids = np.ascontiguousarray(node_ids, np.uint32).reshape(-1)
    check(ffi.lib().mse_graph_set_entries(dgraph._h, vecs._h, _p(ids, C.c_uint32), ids.size), "graph_set_entries")


def set_entry_centroids(dgraph, centroids, medioid_ids):
    """The reference's own entry rule (query_disk_index.rs:254-256,447-450): `centroids` [n_shards, d] f32 are the shard centroids of
    the index header, `medioid_ids` the shards' start nodes; a search starts at the medioid of the shard whose centroid has the largest
    scale_dot_result_f64(dot(centroid, query)), last maximum on ties."""
    c = np.ascontiguousarray(centroids, np.float32)
    ids = np.ascontiguousarray(medioid_ids, np.uint32).reshape(-1)
    if c.ndim != 2 or c.shape[0] != ids.size:
        raise MseError("centroids must be [n_shards, d] with one medioid id per shard")
    check(ffi.lib().mse_graph_set_entry_centroids(dgraph._h, _p(c, C.c_float), c.shape[1], _p(ids, C.c_uint32), ids.size),
          "graph_set_entry_centroids")


def set_dedup(dgraph, threshold=DUPLICATES_THRESHOLD):
    """The handler's runtime de-duplication (query_disk_index.rs:482-527) inside disk_query_topk: visited records that resemble an
    already kept one (dot above `threshold`, visit order) leave the list before it is ordered.  0 switches it off (the default)."""
    check(ffi.lib().mse_graph_set_dedup(dgraph._h, float(threshold)), "graph_set_dedup")


def set_coalescer(dgraph, max_queries_per_pass=0, max_wait_us=0, workers=0):
    """How the graph's coalescer serves small request-path calls from many threads (0 = the defaults: 1024, 200 us, 3 workers)."""
    check(ffi.lib().mse_graph_set_coalescer(dgraph._h, int(max_queries_per_pass), int(max_wait_us), int(workers)), "graph_set_coalescer")


def coalescer_stats(dgraph):
    out = (C.c_uint64 * 6)()
    check(ffi.lib().mse_graph_coalescer_stats(dgraph._h, out), "graph_coalescer_stats")
    return {"queries": int(out[0]), "requests": int(out[1]), "passes": int(out[2]), "max_pass_queries": int(out[3]),
            "deadline_fires": int(out[4]), "run_us": int(out[5])}


def disk_query_topk(searcher: Searcher, quantizer, codes, dgraph, queries, k, starts=None, luts=None, descriptor_scales=None,
                    disable_pq=False, beamwidth=1, search_list=1000):
    """The request path of query_disk_index (:436-540) for a batch in one device submission: entry node (by the graph's entry table
    when `starts` is None), greedy_search, the visited records ordered by exact score and cut to the first k.  f16 query rows in
    (a host array, or `(device_pointer, nq)` for rows already on the device),
    (ids [nq, k] uint32, scores [nq, k] int64, stats dict) out; ids / scores equal topk_of_visited(disk_search_batch(...)) for the
    same start nodes.  Rows with fewer than k visited records are padded with ID_NONE / INT64_MIN."""
    from_f32 = False
    if isinstance(queries, tuple):      # (device pointer, nq): f16 rows already resident on the searcher's device, contiguous
        q_ptr, nq = C.cast(C.c_void_p(int(queries[0])), C.POINTER(C.c_uint16)), int(queries[1])
    elif np.asarray(queries).dtype == np.float32 and luts is None:
        # f32 rows in, as the reference's handler gets them (:436-477): f16 copies and distance tables are made on the device
        from_f32 = True
        q = np.ascontiguousarray(np.asarray(queries).reshape(-1, np.asarray(queries).shape[-1]), np.float32)
        nq = q.shape[0]
        q_ptr = _p(q, C.c_float)
    else:
        q = _bits(queries)
        q = q.reshape(-1, q.shape[-1])
        nq = q.shape[0]
        q_ptr = _p(q, C.c_uint16)
    tables = None
    if luts is not None:
        tables = np.ascontiguousarray(np.stack([getattr(t, "table", t) for t in luts]) if not isinstance(luts, np.ndarray) else luts,
                                      np.float32).reshape(nq, -1)
    st = None if starts is None else np.ascontiguousarray(starts, np.uint32).reshape(nq)
    sc = None
    if descriptor_scales is not None:
        sc = np.ascontiguousarray(descriptor_scales, np.float32)
        if sc.ndim == 1:
            sc = np.ascontiguousarray(np.broadcast_to(sc, (nq, sc.size)))
    ids, scores = np.empty((nq, k), np.uint32), np.empty((nq, k), np.int64)
    nv, cm, pc = np.empty(nq, np.uint32), np.empty(nq, np.uint32), np.empty(nq, np.uint32)
    tail = (nq, int(bool(disable_pq)), int(beamwidth), int(search_list), int(k), _p(ids, C.c_uint32), _p(scores, C.c_int64), _p(nv, C.c_uint32),
            _p(cm, C.c_uint32), _p(pc, C.c_uint32))
    head = (searcher._h, quantizer._h if quantizer is not None else None, codes._h if codes is not None else None, dgraph._h,
            _p(st, C.c_uint32) if st is not None else None, q_ptr)
    scp = _p(sc, C.c_float) if sc is not None else None
    if from_f32:
        check(ffi.lib().mse_disk_query_topk_f32(*head, scp, *tail), "disk_query_topk_f32")
    else:
        check(ffi.lib().mse_disk_query_topk(*head, _p(tables, C.c_float) if tables is not None else None, scp, *tail), "disk_query_topk")
    return ids, scores, {"n_visited": nv, "cmps": cm, "pq_cmps": pc}


class QueryTickets:
    """The request path without a thread per request (include/mse.h "WITHOUT A THREAD PER REQUEST"): one host thread keeps many
    one-query requests of a graph in flight, the way the reference's monoio tasks would (src/query_disk_index.rs:640-655,716-732).
    submit() queues a request and returns its key; collect() hands back requests executed since, as (key, ids [nq, k], scores
    [nq, k]) -- the answers of disk_query_topk for the same queries, bit for bit.  By default a graph has ONE completion list:
    collect() of any such QueryTickets object of that graph returns whatever has completed, whichever object (search settings)
    submitted it (the output arrays of requests in flight are kept alive on the graph object); with own_queue=True the object has a
    completion queue of its own."""

    def __init__(self, searcher: Searcher, quantizer, codes, dgraph, k, disable_pq=False, beamwidth=1, search_list=1000, own_queue=False):
        """own_queue=True: a completion queue of this object's own (mse_completion_queue_*): its requests come back through its
        collect() / fileno() and nowhere else -- one per event loop of a host that runs several."""
        import threading
        self._s, self._pq, self._codes, self._g = searcher, quantizer, codes, dgraph
        self.k, self.disable_pq, self.beamwidth, self.search_list = int(k), bool(disable_pq), int(beamwidth), int(search_list)
        self._q = None
        if own_queue:
            self._q = check_ptr(ffi.lib().mse_completion_queue_new(), "mse_completion_queue_new")
            self._reg = {"lock": threading.Lock(), "next": 1, "out": {}}
        else:
            if not hasattr(dgraph, "_tickets"):
                dgraph._tickets = {"lock": threading.Lock(), "next": 1, "out": {}}
            self._reg = dgraph._tickets

    def close(self):
        """Release the object's own completion queue (no request of it may be in flight)."""
        if getattr(self, "_q", None):
            if self._reg["out"]:
                raise MseError("completion queue closed with requests in flight")
            ffi.lib().mse_completion_queue_free(self._q)
            self._q = None

    def __del__(self):
        try:
            if getattr(self, "_q", None) and not self._reg["out"]:
                ffi.lib().mse_completion_queue_free(self._q)
                self._q = None
        except Exception:  # noqa: BLE001
            pass

    @property
    def in_flight(self):
        """Requests of the whole graph submitted and not yet collected."""
        return len(self._reg["out"])

    def submit(self, query_f32, descriptor_scales=None, key=None, copy=True):
        """copy=False: the library reads the query where it lies (mse_disk_query_submit_f32_nocopy); this object keeps the array alive
        until the request has been collected."""
        q = np.ascontiguousarray(np.asarray(query_f32, np.float32).reshape(-1, np.asarray(query_f32).shape[-1]))
        nq = q.shape[0]
        sc = None
        if descriptor_scales is not None:
            sc = np.asarray(descriptor_scales, np.float32)
            sc = np.ascontiguousarray(np.broadcast_to(sc, (nq, sc.shape[-1])))
        ids, scores = np.empty((nq, self.k), np.uint32), np.empty((nq, self.k), np.int64)
        with self._reg["lock"]:
            tag = self._reg["next"]
            self._reg["next"] += 1
            self._reg["out"][tag] = (key if key is not None else tag, ids, scores) + (() if copy else (q, sc))   # before the request can complete
        t = C.c_void_p()
        try:
            fn = ffi.lib().mse_disk_query_submit_f32 if copy else ffi.lib().mse_disk_query_submit_f32_nocopy
            check(fn(self._s._h, self._pq._h if self._pq is not None else None,
                                                      self._codes._h if self._codes is not None else None, self._g._h, _p(q, C.c_float),
                                                      _p(sc, C.c_float) if sc is not None else None, nq, int(self.disable_pq), self.beamwidth,
                                                      self.search_list, self.k, _p(ids, C.c_uint32), _p(scores, C.c_int64), None, None, None,
                                                      C.c_void_p(tag), self._q, C.byref(t)), "disk_query_submit_f32")
        except MseError:
            with self._reg["lock"]:
                self._reg["out"].pop(tag, None)
            raise
        return key if key is not None else tag

    def fileno(self):
        """An eventfd that becomes readable when requests of the graph have completed (for select / epoll / asyncio's add_reader):
        read its 8-byte counter, then collect(timeout_us=0) until it returns []."""
        fd = ffi.lib().mse_completion_queue_fd(self._q) if self._q else ffi.lib().mse_graph_completion_fd(self._g._h)
        if fd < 0:
            check(-1, "completion fd")
        return fd

    def collect(self, max_tickets=256, timeout_us=-1):
        """Executed requests of the graph, in completion order: [(key, ids, scores)].  Sleeps up to timeout_us for the first (0: poll,
        < 0: until one is there); [] if none came in time.  A request that failed when executed raises MseError with its own
        message once the successful ones of the same call have been returned (they are handed out first, the failure on the next
        call)."""
        pending = self._reg.setdefault("failed", [])
        if pending:
            raise pending.pop(0)
        buf = (C.c_void_p * int(max_tickets))()
        if self._q:
            n = ffi.lib().mse_completion_queue_wait(self._q, buf, int(max_tickets), int(timeout_us))
        else:
            n = ffi.lib().mse_graph_completions(self._g._h, buf, int(max_tickets), int(timeout_us))
        if n < 0:
            check(-1, "completions")
        done = []
        for i in range(n):
            t = buf[i]
            tag = int(ffi.lib().mse_ticket_user(t) or 0)
            rc = ffi.lib().mse_ticket_status(t)
            msg = ffi.lib().mse_ticket_error(t).decode() if rc else ""
            ffi.lib().mse_ticket_free(t)
            with self._reg["lock"]:
                key, ids, scores = self._reg["out"].pop(tag)[:3]
            if rc:
                pending.append(MseError(f"request {key!r}: {msg or 'search failed'}"))
            else:
                done.append((key, ids, scores))
        if not done and pending:
            raise pending.pop(0)
        return done


def topk_of_visited(res, k, keep=None):
    """Batch form of the server's last step (query_disk_index.rs:529-540): the reference sorts the WHOLE visited list by exact
    score after the dedup filter (:482-527) and returns all of it; this helper cuts that sorted list to its first k ids per query
    (stable order), from disk_search_batch(..., as_arrays=True).  `keep` ([nq, visited_cap] bool, e.g. from dedup_visited per
    query) drops filtered records first; without it no dedup is applied.  Always returns [nq, k]: rows with fewer than k
    records are padded with ID_NONE.  Raises if a search visited more records than visited_cap held (the reference keeps all)."""
    vi, vs, nv = res["visited_ids"], res["visited_scores"], res["n_visited"]
    cap = vi.shape[1]
    if int(nv.max(initial=0)) > cap:
        raise ValueError(f"a search visited {int(nv.max())} records but visited_cap is {cap}: raise visited_cap (the reference keeps every visited record)")
    nq = vi.shape[0]
    w = int(min(cap, max(int(nv.max(initial=0)), 1)))
    lowest = np.iinfo(np.int64).min
    sc = vs[:, :w].copy()
    dead = np.arange(w)[None, :] >= nv[:, None]
    if keep is not None:
        dead |= ~np.asarray(keep, bool)[:, :w]
    sc[dead] = lowest
    # two stable passes: order by score descending without the precision loss of a float cast
    order = np.argsort(-(sc >> 1), axis=1, kind="stable")          # coarse (halved scores cannot overflow when negated)
    fine = np.take_along_axis(sc, order, axis=1)
    fix = np.lexsort((np.arange(w)[None, :].repeat(nq, 0), -(fine & 1), -(fine >> 1)), axis=1)
    order = np.take_along_axis(order, fix, axis=1)
    ids = np.full((nq, k), 0xFFFFFFFF, np.int64)
    m = min(k, w)
    top = order[:, :m]
    ids[:, :m] = np.take_along_axis(vi[:, :w], top, axis=1)
    ids[:, :m][np.take_along_axis(dead, top, axis=1)] = 0xFFFFFFFF
    return ids


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


# ---- Vamana graph build on the device (diskann/src/lib.rs:213-389; src/generate_index_shard.rs:85-133) -------------

def IndexBuildConfig(r=64, l=192, maxc=750, alpha=65536, query_alpha=65536, saturate_graph=False, query_breakpoint=0xFFFFFFFF,
                     max_add_per_stitch_iter=16):
    """IndexBuildConfig (lib.rs:42-52) with generate_index_shard's defaults (:22-33,85-94)."""
    return ffi.BuildConfig(r, l, maxc, alpha, query_alpha, int(bool(saturate_graph)), query_breakpoint, max_add_per_stitch_iter)


class BuildGraph:
    """The graph under construction, resident in HBM (IndexGraph::empty + the build passes)."""

    def __init__(self, n, r, graph: IndexGraph = None):
        if graph is not None:
            assert graph.adj.shape == (n, r)
            self._h = check_ptr(ffi.lib().mse_graph_from_host(_p(graph.adj, C.c_uint32), _p(graph.deg, C.c_uint32), n, r, None),
                                "mse_graph_from_host")
        else:
            self._h = check_ptr(ffi.lib().mse_graph_new(n, r), "mse_graph_new")
        self.n, self.r = n, r

    def random_fill(self, seed, r=None):
        """random_fill_graph (lib.rs:376-389) from a counter-based generator (a seed names one graph)."""
        check(ffi.lib().mse_graph_random_fill(self._h, int(seed), self.r if r is None else r), "graph_random_fill")

    def build(self, searcher: Searcher, order, medioid, config, batch=1024):
        """build_graph (lib.rs:287-324); `order` = the shuffled point ids (sigmas); batch = 1 is the sequential form."""
        o = np.ascontiguousarray(order, np.uint32)
        check(ffi.lib().mse_build_graph(searcher._h, self._h, _p(o, C.c_uint32), o.size, int(batch), int(medioid), C.byref(config)),
              "build_graph")

    def robust_stitch(self, searcher: Searcher, queries_order, config):
        """robust_stitch (lib.rs:326-374)."""
        o = np.ascontiguousarray(queries_order, np.uint32)
        check(ffi.lib().mse_robust_stitch(searcher._h, self._h, _p(o, C.c_uint32), C.byref(config)), "robust_stitch")

    def to_host(self) -> IndexGraph:
        adj, deg = np.empty((self.n, self.r), np.uint32), np.empty(self.n, np.uint32)
        check(ffi.lib().mse_graph_to_host(self._h, _p(adj, C.c_uint32), _p(deg, C.c_uint32)), "graph_to_host")
        return IndexGraph(adj, deg)

    def search_batch(self, searcher: Searcher, starts, queries, l, base_vectors_only=False, query_breakpoint=0xFFFFFFFF, as_arrays=False):
        """diskann::greedy_search (lib.rs:183-211) for a batch of queries on the device: list of (ids, scores, distances), or with
        as_arrays=True the padded arrays (ids [nq][l], scores [nq][l], len [nq], distances [nq])."""
        q = _bits(queries)
        q = q.reshape(-1, q.shape[-1])
        nq = q.shape[0]
        st = np.ascontiguousarray(np.broadcast_to(np.asarray(starts, np.uint32), (nq,)))
        bi, bs = np.empty((nq, l), np.uint32), np.empty((nq, l), np.int64)
        bl, nd = np.empty(nq, np.uint32), np.empty(nq, np.uint32)
        check(ffi.lib().mse_graph_search_batch(searcher._h, self._h, _p(st, C.c_uint32), _p(q, C.c_uint16), nq, int(l),
                                               int(bool(base_vectors_only)), int(query_breakpoint), _p(bi, C.c_uint32),
                                               _p(bs, C.c_int64), _p(bl, C.c_uint32), _p(nd, C.c_uint32)), "graph_search_batch")
        if as_arrays:
            return bi, bs, bl, nd
        return [(bi[i, :bl[i]].copy(), bs[i, :bl[i]].copy(), int(nd[i])) for i in range(nq)]

    def close(self):
        if getattr(self, "_h", None):
            ffi.lib().mse_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def robust_prune(searcher: Searcher, cand_ids, cand_scores, p, config):
    """robust_prune (lib.rs:227-285) on a candidate list; returns the new neighbour list."""
    ci, cs = np.ascontiguousarray(cand_ids, np.uint32), np.ascontiguousarray(cand_scores, np.int64)
    out = np.empty(int(config.r), np.uint32)
    nn = C.c_size_t()
    check(ffi.lib().mse_robust_prune(searcher._h, _p(ci, C.c_uint32), _p(cs, C.c_int64), ci.size, int(p), C.byref(config),
                                     _p(out, C.c_uint32), C.byref(nn)), "robust_prune")
    return out[:nn.value].copy()
