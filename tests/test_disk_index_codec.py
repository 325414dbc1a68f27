"""index.bin payload codec (mse/bitcode06.py: the bitcode 0.6 layout of PackedIndexEntry, src/common.rs:154-164) and the index
directory writer / reader around it.  The crate itself is unavailable here (parity unpinned, see the module header): these tests
pin the restated layout rules byte by byte against hand-assembled strings, and the encoder against the decoder."""
import numpy as np
import pytest


def bc():
    from mse import bitcode06
    return bitcode06


def test_pack_ints_by_hand():
    b = bc()
    assert b.pack_ints([7], 4) == bytes([4, 7])                                   # u32 -> u8: two sizes down, header 2*2
    assert b.pack_ints([300], 4) == bytes([2]) + (300).to_bytes(2, "little")      # u32 -> u16: one size down
    assert b.pack_ints([70000], 4) == bytes([0]) + (70000).to_bytes(4, "little")  # not narrowed
    assert b.pack_ints([1700000000], 8) == bytes([2]) + (1700000000).to_bytes(4, "little")   # u64 -> u32
    assert b.pack_ints([], 8) == bytes([6])                                       # empty: narrowest form, no data
    assert b.pack_ints([1152], 8) == bytes([4]) + (1152).to_bytes(2, "little")
    # offset form: more than five values whose spread fits a narrower size than their maximum
    v = [1000, 1001, 1002, 1003, 1004, 1005]
    assert b.pack_ints(v, 2) == bytes([1]) + (1000).to_bytes(2, "little") + bytes([0, 1, 2, 3, 4, 5])
    assert b.pack_ints(v[:5], 2) == bytes([0]) + np.array(v[:5], "<u2").tobytes()  # five values: never offset
    for vals, nb in (([7], 4), (v, 2), ([], 8), ([2 ** 40, 5], 8), (list(range(300, 310)), 4)):
        got, pos = b.unpack_ints(b.pack_ints(vals, nb), 0, len(vals), nb)
        assert got.tolist() == vals and pos == len(b.pack_ints(vals, nb))
    with pytest.raises(b.BitcodeError):
        b.unpack_ints(bytes([1, 0, 0]), 0, 1, 2)        # "not narrowed, with offset" is unrepresentable... header 1 = one size down + offset: truncated here
    with pytest.raises(b.BitcodeError):
        b.unpack_ints(bytes([9]), 0, 1, 4)              # narrower than a byte


def test_pack_bytes_and_lengths_by_hand():
    b = bc()
    assert b.pack_bytes([255]) == bytes([0, 255])                 # raw
    assert b.pack_bytes([64]) == bytes([0, 64])
    assert b.pack_bytes([3]) == bytes([6, 3])                     # four-valued digits, header 2*3
    assert b.pack_bytes([0]) == bytes([10, 0])                    # two-valued
    assert b.pack_bytes([1, 0, 1, 1, 0, 0, 0, 1, 1]) == bytes([10, 0b10001101, 1])   # eight per byte, little end first
    assert b.pack_bytes([15, 1, 2]) == bytes([2, 15 + 16 * 1, 2])                    # sixteen-valued, two per byte
    assert b.pack_bytes([5, 4, 3]) == bytes([4, 5 + 6 * 4 + 36 * 3])                 # six-valued, three per byte
    assert b.pack_bytes([200, 201, 200, 201, 200, 201, 200]) == bytes([9, 200, 0b0101010])   # offset + two-valued: header 2*5 - 1
    for vals in ([255], [3], [0] * 9, [5, 4, 3, 0], list(range(100, 116)) * 2, [17, 200, 3]):
        got, pos = b.unpack_bytes(b.pack_bytes(vals), 0, len(vals))
        assert got == vals and pos == len(b.pack_bytes(vals))
    # LengthEncoder: small byte (255 = look in the large list) then the large lengths as u64 ints
    assert b.pack_lengths([3]) == bytes([6, 3]) + bytes([6])
    assert b.pack_lengths([1152]) == bytes([0, 255]) + bytes([4]) + (1152).to_bytes(2, "little")
    assert b.unpack_lengths(b.pack_lengths([1152, 3, 255, 0]), 0, 4)[0] == [1152, 3, 255, 0]


def test_f32_split_by_hand():
    b = bc()
    # 1.0 = 0x3F800000: rotated left by one = 0x7F000000 -> mantissa+sign bytes 00 00 00, exponent byte 0x7F
    # -1.5 = 0xBFC00000: rotated = 0x7F800001 -> 01 00 80, exponent 0x7F
    assert b.pack_f32([1.0, -1.5]) == bytes([0, 0, 0, 1, 0, 0x80, 0x7F, 0x7F])
    x = np.array([0.0, -0.0, 1e-40, 3.14159, -2.5e10, np.inf], np.float32)
    y, pos = b.unpack_f32(b.pack_f32(x), 0, len(x))
    assert y.view(np.uint32).tolist() == x.view(np.uint32).tolist() and pos == 4 * len(x)


def entry(i, d=1152, deg=64, url="https://example.org/%d.jpg"):
    rng = np.random.default_rng(i)
    return {"vector": (rng.standard_normal(d) / np.sqrt(d)).astype(np.float16).view(np.uint16), "vertices": rng.integers(0, 1000, deg).astype(np.uint32),
            "id": i, "timestamp": 1_600_000_000 + i, "dimensions": (640 + i, 480), "scores": rng.random(3).astype(np.float32),
            "url": url % i if "%" in url else url, "shards": np.array([i % 7, (i + 1) % 7], np.uint32)}


def test_entry_layout_and_round_trip():
    b = bc()
    e = entry(5)
    p = b.encode_packed_index_entry(e)
    # field by field, in the struct's declaration order (src/common.rs:155-164)
    want = (bytes([0, 255, 4]) + (1152).to_bytes(2, "little") + bytes([0]) + e["vector"].astype("<u2").tobytes()      # Vec<u16>: fp16 bits span the 16 bits
            + bytes([0, 64, 6]) + bytes([2]) + e["vertices"].astype("<u2").tobytes()                                # Vec<u32> below 65536 -> u16
            + bytes([4, 5])                                                                                          # id 5
            + bytes([2]) + (1_600_000_005).to_bytes(4, "little")                                                     # timestamp u64 -> u32
            + bytes([2]) + (645).to_bytes(2, "little") + bytes([2]) + (480).to_bytes(2, "little")                    # (u32, u32)
            + bytes([6, 3, 6]) + b.pack_f32(e["scores"])
            + bytes([0, len(e["url"]), 6]) + e["url"].encode()
            + bytes([8, 2, 6]) + bytes([4, 5, 6]))                                                                   # length 2: three-valued digit; shards [5, 6] -> u8
    assert p == want
    d = b.decode_packed_index_entry(p)
    assert all(np.array_equal(d[k], e[k]) for k in ("vector", "vertices", "shards")) and d["scores"].tolist() == e["scores"].tolist()
    assert (d["id"], d["timestamp"], d["dimensions"], d["url"]) == (5, 1_600_000_005, (645, 480), e["url"])
    assert len(p) < 4094                                       # a full record (1152-d vector, 64 neighbours) fits the 4 KiB sector
    for bad in (p[:-1], p + b"\\0", p[:10]):
        with pytest.raises(ValueError):
            b.decode_packed_index_entry(bad)
    # edge cases: empty lists and strings, non-ASCII URL, a long URL, big ids
    e2 = dict(entry(1), vertices=np.zeros(0, np.uint32), scores=np.zeros(0, np.float32), url="", shards=np.zeros(0, np.uint32), id=2 ** 32 - 1,
              timestamp=2 ** 63)
    d2 = b.decode_packed_index_entry(b.encode_packed_index_entry(e2))
    assert d2["url"] == "" and len(d2["vertices"]) == 0 and d2["id"] == 2 ** 32 - 1 and d2["timestamp"] == 2 ** 63
    e3 = dict(entry(2), url="https://пример.example/" + "x" * 400)
    assert b.decode_packed_index_entry(b.encode_packed_index_entry(e3))["url"] == e3["url"]


def test_write_index_then_open(tmp_path):
    """dump-processor's output files written by write_index, opened by DiskIndex: framing, dead records, code files, header counts."""
    from mse import disk_index as di
    d, count = 64, 6
    quant = {"centroids": np.zeros(4 * d, np.float32), "transform": np.eye(d, dtype=np.float32).reshape(-1), "n_dims_per_code": 16, "n_dims": d}
    hdr = di.IndexHeader([(np.ones(d, np.float32), 2)], 0, 0, 512, quant, [np.array([0, 1], np.float32)] * 2)
    ents = [entry(i, d=d, deg=5) for i in range(count)]
    ents[3]["url"] = "https://example.org/" + "y" * 600          # does not fit a 512-byte record: survives as a graph-only node
    for e in ents:
        e["vertices"] = e["vertices"] % count
    codes = np.arange(count * 4, dtype=np.uint8).reshape(count, 4)
    desc = np.arange(count * 2, dtype=np.uint8).reshape(count, 2)
    out = di.write_index(str(tmp_path), hdr, ents, codes, desc, encode_entry=di.UNPINNED_BITCODE06_ENCODE)
    assert (out.count, out.dead_count) == (count, 1)
    assert (tmp_path / "index.bin").stat().st_size == count * 512
    idx = di.DiskIndex(str(tmp_path), decode_entry=di.UNPINNED_BITCODE06_DECODE)
    assert idx.header.count == count and idx.header.dead_count == 1
    got = list(idx.entries())
    assert [g["url"] for g in got] == [e["url"] if i != 3 else "" for i, e in enumerate(ents)]
    assert all(np.array_equal(g["vector"], e["vector"]) and np.array_equal(g["vertices"], e["vertices"]) for g, e in zip(got, ents))
    assert idx.read_node(4)["timestamp"] == ents[4]["timestamp"]
    assert idx.pq_codes.tolist() == codes.tolist() and idx.descriptors.tolist() == desc.tolist()
