"""File formats either side of the graph build (src/generate_index_shard.rs): the shard input stream written by
dump-processor and the shard output pair read back by it.  Byte-level cases are written out by hand from rmp-serde's
rules (structs as positional arrays, serde_bytes as bin, f32 as float32, integers in their shortest form)."""
import os
import struct

import numpy as np
import pytest

D = 1152


def test_shard_input_bytes_by_hand(tmp_path):
    from mse import generate_index_shard as gis
    v0, v1 = (np.arange(D) % 7).astype("<u2"), (np.arange(D) % 5 + 1).astype("<u2")
    blob = bytes([0x92, 0x03, 0x92, 0xCA, 0x3F, 0x80, 0x00, 0x00, 0xCA, 0xC0, 0x20, 0x00, 0x00])       # [3, [1.0f32, -2.5f32]]
    blob += bytes([0x92, 0x07, 0xC5, 0x09, 0x00]) + v0.tobytes()                                     # [7, bin16(2304 bytes)]
    blob += bytes([0x92, 0xCE, 0x00, 0x01, 0x11, 0x70, 0xC5, 0x09, 0x00]) + v1.tobytes()             # [70000 as uint32, bin16]
    p = tmp_path / "3.shard.msgpack"
    p.write_bytes(blob)
    header, ids, vecs = gis.read_shard_input(str(p))
    assert header == {"id": 3, "centroid": [1.0, -2.5]}
    assert ids.tolist() == [7, 70000] and np.array_equal(vecs[0], v0) and np.array_equal(vecs[1], v1)
    # and our writer produces the same bytes
    q = tmp_path / "w.msgpack"
    gis.write_shard_input(str(q), 3, [1.0, -2.5], [7, 70000], np.stack([v0, v1]))
    assert q.read_bytes() == blob


def test_shard_input_rejects_short_vectors(tmp_path):
    from mse import generate_index_shard as gis
    p = tmp_path / "bad.msgpack"
    p.write_bytes(bytes([0x92, 0x00, 0x90, 0x92, 0x01, 0xC4, 0x02, 0x00, 0x00]))
    with pytest.raises(ValueError):
        gis.read_shard_input(str(p))


def test_shard_output_layout(tmp_path):
    from mse import generate_index_shard as gis
    from mse.diskann import IndexGraph
    adj = np.array([[5, 6, 0], [7, 0, 0], [1, 2, 3], [9, 9, 9]], np.uint32)
    deg = np.array([2, 1, 3, 3], np.uint32)                      # node 3 is a query node (behind the breakpoint)
    ids = np.array([100, 300, 200], np.uint32)
    gis.write_shard_output(str(tmp_path), {"id": 4, "centroid": [0.5]}, 2, ids, IndexGraph(adj, deg), 3)
    raw = (tmp_path / "4.shard.bin").read_bytes()
    assert np.frombuffer(raw, "<u4").tolist() == [5, 6, 7, 1, 2, 3]                 # base nodes only, lists back to back
    hdr = (tmp_path / "4.shard-header.msgpack").read_bytes()
    # ShardHeader {id, max, centroid, medioid, offsets, mapping} as a positional array; offsets in bytes, one extra at the end
    assert hdr == bytes([0x96, 0x04, 0xCD, 0x01, 0x2C, 0x91, 0xCA, 0x3F, 0x00, 0x00, 0x00, 0x02,
                         0x94, 0x00, 0x08, 0x0C, 0x18, 0x93, 0x64, 0xCD, 0x01, 0x2C, 0xCC, 0xC8])
    h, lists = gis.read_shard_output(str(tmp_path), 4)
    assert h["medioid"] == 2 and h["max"] == 300 and h["mapping"] == [100, 300, 200]
    assert [l.tolist() for l in lists] == [[5, 6], [7], [1, 2, 3]]


@pytest.mark.gpu
def test_generate_index_shard_end_to_end(gpu, mse, orc, tmp_path):
    """The whole tool on a small shard with query vectors; the written lists equal the oracle's build run with the same
    shuffles and initial graph."""
    from mse import generate_index_shard as gis
    from test_gpu_graph_build import rows
    from conftest import SEED_QUERY
    n, nq, r = 1500, 120, 16
    vecs = rows(orc, n, seed=12)
    ids = (np.arange(n) * 3 + 11).astype(np.uint32)
    inp = tmp_path / "5.shard.msgpack"
    gis.write_shard_input(str(inp), 5, np.linspace(-1, 1, D), ids, vecs)
    queries = orc.gen_rows_f16(SEED_QUERY, 0, nq)
    qb = tmp_path / "queries.bin"
    queries.astype("<u2").tofile(str(qb))
    out = tmp_path / "out"
    out.mkdir()
    info, host = gis.generate_index_shard(str(inp), str(out), str(qb), l=40, r=r, maxc=100, alpha=65536, alpha_2=78643,
                                          second_pass=True, seed=99, batch=64, log=lambda *_: None)
    header, lists = gis.read_shard_output(str(out), 5)
    assert header["mapping"] == ids.tolist() and header["max"] == int(ids.max()) and len(lists) == n
    assert header["offsets"][-1] == os.path.getsize(out / "5.shard.bin")
    # the oracle, fed the same random choices
    allv = np.concatenate([vecs, queries])
    rng = np.random.default_rng(99)
    adj, deg = orc.random_fill_graph(int(rng.integers(0, 2 ** 32)), n + nq, r)
    med = int(orc.medioid(allv))
    assert header["medioid"] == med
    kw = dict(r=r, l=40, maxc=100, query_breakpoint=n, max_add_per_stitch_iter=16)
    orc.build_graph(allv, adj, deg, rng.permutation(n + nq).astype(np.uint32), med, orc.BuildConfig.make(**kw), 64)
    cfg2 = orc.BuildConfig.make(alpha=78643, **kw)
    orc.build_graph(allv, adj, deg, rng.permutation(n + nq).astype(np.uint32), med, cfg2, 64)
    orc.robust_stitch(allv, adj, deg, (n + rng.permutation(nq)).astype(np.uint32), cfg2)
    for i in range(n):
        assert np.array_equal(lists[i], adj[i, :deg[i]]), i


# ---- the index directory of query-disk-index (src/query_disk_index.rs:658-709) ------------------------------------

def test_index_header_named_map_by_hand(tmp_path):
    """to_vec_named => a MAP keyed by field name; tuples are arrays; the reader must not depend on key order."""
    import msgpack
    from mse import disk_index as di
    d, dpc = 4, 2
    blob = bytes([0x86])                                                   # map of 6
    blob += bytes([0xA6]) + b"shards" + bytes([0x91, 0x92, 0x94]) + b"".join(bytes([0xCA]) + struct.pack(">f", v) for v in (1, 0, 0, 0)) + bytes([0x05])
    blob += bytes([0xA5]) + b"count" + bytes([0x03])
    blob += bytes([0xAA]) + b"dead_count" + bytes([0x01])
    blob += bytes([0xAF]) + b"record_pad_size" + bytes([0xCD, 0x10, 0x00])
    quant = {"centroids": [0.5] * (2 * d), "transform": [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0], "n_dims_per_code": dpc, "n_dims": d}
    blob += bytes([0xA9]) + b"quantizer" + msgpack.packb(quant, use_single_float=True)
    blob += bytes([0xAF]) + b"descriptor_cdfs" + bytes([0x92, 0x92, 0xCA, 0x00, 0x00, 0x00, 0x00, 0xCA, 0x3F, 0x80, 0x00, 0x00, 0x90])
    (tmp_path / "index.msgpack").write_bytes(blob)
    h = di.read_index_header(str(tmp_path / "index.msgpack"))
    assert (h.count, h.dead_count, h.record_pad_size, h.pq_code_size, h.n_descriptors) == (3, 1, 4096, 2, 2)
    assert h.shards[0][1] == 5 and h.shards[0][0].tolist() == [1, 0, 0, 0]
    assert h.descriptor_cdfs[0].tolist() == [0.0, 1.0] and h.descriptor_cdfs[1].size == 0
    # writer -> reader round trip, and the writer keeps the struct's field order (shards first, descriptor_cdfs last)
    di.write_index_header(str(tmp_path / "w.msgpack"), h)
    raw = (tmp_path / "w.msgpack").read_bytes()
    assert raw[:8] == bytes([0x86, 0xA6]) + b"shards"
    h2 = di.read_index_header(str(tmp_path / "w.msgpack"))
    assert h2.count == 3 and np.array_equal(h2.quantizer["transform"], h.quantizer["transform"])
    with pytest.raises(ValueError):
        (tmp_path / "bad.msgpack").write_bytes(msgpack.packb({"count": 1}))
        di.read_index_header(str(tmp_path / "bad.msgpack"))


def test_index_directory_records_and_code_files(tmp_path):
    from mse import disk_index as di
    d, dpc, count = 4, 2, 3
    quant = {"centroids": np.zeros(2 * d, np.float32), "transform": np.eye(d, dtype=np.float32).reshape(-1), "n_dims_per_code": dpc, "n_dims": d}
    hdr = di.IndexHeader([(np.ones(d, np.float32), 0)], count, 0, 64, quant, [np.array([0, 1], np.float32)])
    di.write_index_header(str(tmp_path / "index.msgpack"), hdr)
    payloads = [b"abc", b"", bytes(range(62))]
    di.write_records(str(tmp_path / "index.bin"), payloads, 64)
    raw = (tmp_path / "index.bin").read_bytes()
    assert len(raw) == 3 * 64 and raw[:5] == b"\x03\x00abc" and raw[5:64] == bytes(59) and raw[128:130] == b"\x3e\x00"
    np.arange(count * 2, dtype=np.uint8).tofile(str(tmp_path / "index.pq-codes.bin"))
    np.array([9, 8, 7], np.uint8).tofile(str(tmp_path / "index.descriptor-codes.bin"))
    idx = di.DiskIndex(str(tmp_path), decode_entry=lambda b: b.decode("latin1"))
    assert [idx.record_payload(i) for i in range(3)] == payloads and idx.read_node(0) == "abc"
    assert idx.pq_codes.tolist() == [[0, 1], [2, 3], [4, 5]] and idx.descriptors.tolist() == [[9], [8], [7]]
    with pytest.raises(NotImplementedError):
        di.DiskIndex(str(tmp_path)).read_node(0)            # no silent default: the bitcode restatement is unpinned, opt-in only
    with pytest.raises(NotImplementedError):
        list(di.DiskIndex(str(tmp_path)).entries())
    with pytest.raises(NotImplementedError):
        di.write_index(str(tmp_path), hdr, [], np.zeros((0, 2), np.uint8), None)
    with pytest.raises(ValueError):                         # b"abc" is not a PackedIndexEntry: the opted-in decoder says so
        di.DiskIndex(str(tmp_path), decode_entry=di.UNPINNED_BITCODE06_DECODE).read_node(0)
    with pytest.raises(ValueError):
        di.write_records(str(tmp_path / "x.bin"), [bytes(63)], 64)
    np.arange(5, dtype=np.uint8).tofile(str(tmp_path / "index.pq-codes.bin"))
    with pytest.raises(ValueError):
        di.DiskIndex(str(tmp_path))


def test_merge_shards_by_hand(tmp_path):
    """Two shards sharing records 10 and 30 (spill 2): lists are mapped back to original ids and united in shard order,
    first occurrence kept (src/dump_processor.rs:264-293)."""
    from mse import generate_index_shard as gis
    from mse.diskann import IndexGraph
    # shard 0 holds originals [10, 20, 30]; shard 1 holds [30, 40, 10]
    g0 = IndexGraph(np.array([[1, 2], [0, 0], [0, 1]], np.uint32), np.array([2, 1, 2], np.uint32))
    g1 = IndexGraph(np.array([[1, 2], [0, 0], [1, 0]], np.uint32), np.array([2, 1, 2], np.uint32))
    gis.write_shard_output(str(tmp_path), {"id": 0, "centroid": [1.0, 0.0]}, 1, np.array([10, 20, 30], np.uint32), g0, 3)
    gis.write_shard_output(str(tmp_path), {"id": 1, "centroid": [0.0, 1.0]}, 2, np.array([30, 40, 10], np.uint32), g1, 3)
    adj, deg, shards_of, specs = gis.merge_shards(str(tmp_path))
    lists = {i: adj[i, :deg[i]].tolist() for i in (10, 20, 30, 40)}
    assert lists == {10: [20, 30, 40], 20: [10], 30: [10, 20, 40], 40: [30]}        # 10: [20, 30] from shard 0, then 40 (30 is a repeat)
    assert shards_of[10].tolist() == [0, 1] and shards_of[20].tolist() == [0, -1] and shards_of[5].tolist() == [-1, -1]
    assert [m for _, m in specs] == [20, 10] and specs[1][0].tolist() == [0.0, 1.0]   # medioids as original ids
    gis.write_shard_output(str(tmp_path), {"id": 2, "centroid": [0.0, 0.0]}, 0, np.array([10], np.uint32), IndexGraph(np.zeros((1, 2), np.uint32), np.zeros(1, np.uint32)), 1)
    with pytest.raises(ValueError):
        gis.merge_shards(str(tmp_path))                                               # record 10 in three shards


@pytest.mark.gpu
def test_sharded_pipeline_end_to_end(gpu, mse, orc, tmp_path):
    """The reference's large-index recipe in small: records spilled to their two closest shards (dump_processor.rs:438-455), one
    graph per shard (generate-index-shard), lists merged (:264-293), a query enters at the medioid of the shard whose centroid it is
    closest to (query_disk_index.rs:447-450) and is searched over the merged graph on the device."""
    from mse import generate_index_shard as gis
    from test_gpu_graph_build import rows
    n, S, R, K = 6000, 3, 24, 10
    vecs = rows(orc, n, seed=31)
    x = orc.f16_to_f32(vecs)
    cents = x[[10, 2000, 4000]].copy()
    for _ in range(3):                                                   # a few k-means steps for the shard centroids (kmeans.py)
        a = np.argmax(x @ cents.T, axis=1)
        cents = np.stack([x[a == s].mean(axis=0) for s in range(S)])
    two = np.argsort(-(x @ cents.T), axis=1)[:, :gis.SHARD_SPILL]
    sh_in, sh_out = tmp_path / "in", tmp_path / "out"
    sh_in.mkdir(); sh_out.mkdir()
    for s in range(S):
        ids = np.flatnonzero((two == s).any(axis=1)).astype(np.uint32)
        gis.write_shard_input(str(sh_in / f"{s}.shard.msgpack"), s, cents[s], ids, vecs[ids])
        gis.generate_index_shard(str(sh_in / f"{s}.shard.msgpack"), str(sh_out), l=64, r=R, maxc=200, seed=s, batch=256, log=lambda *_: None)
    adj, deg, shards_of, specs = gis.merge_shards(str(sh_out))
    assert (shards_of >= 0).all() and deg.max() <= 2 * R and deg.min() >= 1
    searcher = mse.Searcher(mse.VectorList.from_f16s(vecs, D))
    dgraph = mse.DeviceGraph(mse.IndexGraph(adj, deg))
    nq = 64
    q = rows(orc, nq, seed=32)
    qf = orc.f16_to_f32(q)
    centroids = np.stack([c for c, _ in specs])
    starts = np.array([specs[mse.select_shard(centroids, qf[i])][1] for i in range(nq)], np.uint32)
    # neighbours scored exactly: neither codec nor codes are needed (and asking for ADC without them is an error)
    res = mse.disk_search_batch(searcher, None, None, dgraph, starts, q, None, None, True, 4, search_list=64, visited_cap=1024, as_arrays=True)
    with pytest.raises(mse.MseError):
        mse.disk_search_batch(searcher, None, None, dgraph, starts, qf, None, None, False, 4, search_list=64, visited_cap=1024)
    top = mse.topk_of_visited(res, K)
    _, truth = searcher.bruteforce_topk(q, K)
    recall = np.mean([len(set(top[i].tolist()) & set(truth[i].tolist())) / K for i in range(nq)])
    assert recall > 0.9


def test_merge_shards_random_against_per_record_loop(tmp_path):
    """The vectorised merge against read_out_vertices written out record by record (dump_processor.rs:264-293): random shards with
    spill 2, ragged lists, repeated ids inside a list and across the two shards of a record."""
    from mse import generate_index_shard as gis
    from mse.diskann import IndexGraph
    rng = np.random.default_rng(41)
    n, S, R = 400, 5, 7
    member = np.stack([rng.permutation(S)[:2] for _ in range(n)])            # the two shards of every record
    member[rng.random(n) < 0.2, 1] = -1                                       # some records sit in one shard only
    shard_data = {}
    for s in range(S):
        ids = np.flatnonzero((member == s).any(axis=1)).astype(np.uint32)
        m = len(ids)
        adj = rng.integers(0, m, size=(m, R)).astype(np.uint32)               # repeats inside a list happen
        deg = rng.integers(0, R + 1, size=m).astype(np.uint32)
        gis.write_shard_output(str(tmp_path), {"id": s, "centroid": [float(s)]}, int(rng.integers(0, m)), ids, IndexGraph(adj, deg), m)
        shard_data[s] = (ids, adj, deg)
    adj, deg, shards_of, specs = gis.merge_shards(str(tmp_path))
    for gid in range(n):
        want, want_sh = [], []
        for s in range(S):                                                    # shard slots in ascending shard id
            ids, a, d = shard_data[s]
            pos = np.flatnonzero(ids == gid)
            if len(pos) == 0:
                continue
            want_sh.append(s)
            for w in a[pos[0], :d[pos[0]]]:
                g = int(ids[w])
                if g not in want:
                    want.append(g)
        assert adj[gid, :deg[gid]].tolist() == want, gid
        assert [x for x in shards_of[gid].tolist() if x >= 0] == want_sh
    assert len(specs) == S


def test_merge_shards_repeated_id_takes_next_slot(tmp_path):
    """dump_processor.rs:245-259: each occurrence of an id in a shard's mapping takes the record's next empty slot, also when the
    SAME shard lists the id twice; a third occurrence (spill = 2) is the "shard processing inconsistency" error."""
    from mse import generate_index_shard as gis
    from mse.diskann import IndexGraph
    ids = np.array([0, 1, 1, 2], np.uint32)                      # id 1 twice in shard 0
    adj = np.array([[1, 0], [0, 3], [3, 0], [1, 2]], np.uint32)  # within-shard ids
    deg = np.array([1, 2, 1, 2], np.uint32)
    gis.write_shard_output(str(tmp_path), {"id": 0, "centroid": [0.0]}, 0, ids, IndexGraph(adj, deg), 4)
    m_adj, m_deg, shards_of, _ = gis.merge_shards(str(tmp_path))
    # record 1: slot 0 = list of occurrence 0 -> [ids[0], ids[3]] = [0, 2]; slot 1 = occurrence 1 -> [ids[3]] = [2] (already there)
    assert m_adj[1, :m_deg[1]].tolist() == [0, 2] and shards_of[1].tolist() == [0, 0]
    assert m_adj[0, :m_deg[0]].tolist() == [1] and shards_of[0].tolist() == [0, -1]
    ids3 = np.array([1, 1, 1], np.uint32)
    (tmp_path / "x").mkdir()
    gis.write_shard_output(str(tmp_path / "x"), {"id": 0, "centroid": [0.0]}, 0, ids3,
                           IndexGraph(np.zeros((3, 1), np.uint32), np.ones(3, np.uint32)), 3)
    with pytest.raises(ValueError, match="inconsistency"):
        gis.merge_shards(str(tmp_path / "x"))
