"""Payload codec of `index.bin` records: `bitcode::encode(&PackedIndexEntry)` / `bitcode::decode` (src/common.rs:154-164,
src/dump_processor.rs:510, src/query_disk_index.rs:80) -- a restatement of the bitcode 0.6 wire format for exactly the types
that struct uses.

STATUS: parity UNPINNED.  bitcode is a third-party crate (Cargo.toml: `bitcode = "0.6"`, Cargo.lock 0.6.7); its source is not
in the reference tree, the reference ships no sample index, and no Rust toolchain exists here to produce one.  What follows is
the crate's published design (derive encoders are columnar: a struct writes its fields one after another in declaration
order, every field through the encoder of its type) restated from knowledge of the 0.6 sources, with encoder and decoder
written together and pinned against each other and against hand-assembled byte strings (tests/test_disk_index_codec.py).
A real index.bin must be checked against this before it is trusted -- `DiskIndex(decode_entry=...)` still accepts another decoder.

Layout rules restated (all integers little endian):
  struct / tuple      fields in declaration order, nothing in between
  u16 / u32 / u64     `pack_ints`: one header byte h, then the values narrowed to the smallest of 8/16/32/64 bits that holds the
                      largest one.  h = 2*steps - offset, steps = how many sizes below the native one (0 = not narrowed), offset = 1
                      when the minimum was subtracted first (then the minimum follows the header at native width); the offset form
                      is only considered for more than OFFSET_MIN_LEN values and only when it narrows further.
  Vec<T>              LengthEncoder, then T's encoder over all elements
  LengthEncoder       one u8 per length (255 = "see the large list") through `pack_bytes`, then the large lengths as u64 `pack_ints`
  pack_bytes          header h = 2*p - offset with p = 0 raw | 1 sixteen-valued (2 per byte) | 2 six-valued (3 per byte) |
                      3 four-valued (4 per byte) | 4 three-valued (5 per byte) | 5 two-valued (8 per byte); digits little end first
  f32                 bits rotated left by one (sign to bit 0): all 3-byte mantissa+sign parts, then all exponent bytes
  String              LengthEncoder over the byte length, then the UTF-8 bytes unpacked
"""
import struct

import numpy as np

OFFSET_MIN_LEN = 5          # offset packing is considered for MORE than this many values
_INT_SIZES = (8, 4, 2, 1)   # bytes, descending: u64, u32, u16, u8
_BYTE_FACTORS = (256, 16, 6, 4, 3, 2)
_PER_BYTE = {16: 2, 6: 3, 4: 4, 3: 5, 2: 8}


class BitcodeError(ValueError):
    pass


# ---- pack_ints ---------------------------------------------------------------------------------------------------------------
def _width_for(max_value):
    for w in (1, 2, 4, 8):
        if max_value < (1 << (8 * w)):
            return w
    raise BitcodeError("integer does not fit 64 bits")


def pack_ints(values, native_bytes):
    v = [int(x) for x in values]
    hi, lo = (max(v), min(v)) if v else (0, 0)
    if v and (lo < 0 or hi >= 1 << (8 * native_bytes)):
        raise BitcodeError("value out of range for its type")
    w = min(_width_for(hi), native_bytes)
    offset = False
    if len(v) > OFFSET_MIN_LEN:
        wo = min(_width_for(hi - lo), native_bytes)
        if wo < w:
            w, offset = wo, True
    steps = _INT_SIZES.index(w) - _INT_SIZES.index(native_bytes)
    out = bytearray([2 * steps - int(offset)])
    if offset:
        out += lo.to_bytes(native_bytes, "little")
        v = [x - lo for x in v]
    out += np.asarray(v, dtype=np.dtype("<u%d" % w) if v else np.uint8).astype("<u%d" % w).tobytes()
    return bytes(out)


def unpack_ints(buf, pos, count, native_bytes):
    if pos >= len(buf):
        raise BitcodeError("truncated: integer header")
    h = buf[pos]
    pos += 1
    steps, offset = (h + 1) // 2, h & 1
    idx = _INT_SIZES.index(native_bytes) + steps
    if idx >= len(_INT_SIZES) or (steps == 0 and offset):
        raise BitcodeError("bad integer packing header %d" % h)
    w = _INT_SIZES[idx]
    base = 0
    if offset:
        base = int.from_bytes(buf[pos:pos + native_bytes], "little")
        pos += native_bytes
    end = pos + w * count
    if end > len(buf):
        raise BitcodeError("truncated: integer data")
    vals = np.frombuffer(buf, dtype="<u%d" % w, count=count, offset=pos).astype(np.uint64) + np.uint64(base)
    if count and int(vals.max()) >= 1 << (8 * native_bytes):
        raise BitcodeError("offset integer overflows its type")
    return vals, end


# ---- pack_bytes --------------------------------------------------------------------------------------------------------------
def _byte_packing(max_value):
    for p in range(len(_BYTE_FACTORS) - 1, -1, -1):      # the smallest factor that still holds max_value
        if max_value < _BYTE_FACTORS[p]:
            return p
    return 0


def pack_bytes(values):
    v = [int(x) for x in values]
    hi, lo = (max(v), min(v)) if v else (0, 0)
    p, offset = _byte_packing(hi), False
    if len(v) > OFFSET_MIN_LEN:
        po = _byte_packing(hi - lo)
        if po > p:
            p, offset = po, True
    out = bytearray([2 * p - int(offset)])
    if offset:
        out.append(lo)
        v = [x - lo for x in v]
    f = _BYTE_FACTORS[p]
    if f == 256:
        out += bytes(v)
    else:
        per = _PER_BYTE[f]
        for i in range(0, len(v), per):
            b, m = 0, 1
            for x in v[i:i + per]:
                b += x * m
                m *= f
            out.append(b)
    return bytes(out)


def unpack_bytes(buf, pos, count):
    if pos >= len(buf):
        raise BitcodeError("truncated: byte header")
    h = buf[pos]
    pos += 1
    p, offset = (h + 1) // 2, h & 1
    if p >= len(_BYTE_FACTORS) or (p == 0 and offset):
        raise BitcodeError("bad byte packing header %d" % h)
    base = 0
    if offset:
        base = buf[pos]
        pos += 1
    f = _BYTE_FACTORS[p]
    out = []
    if f == 256:
        out = list(buf[pos:pos + count])
        pos += count
    else:
        per = _PER_BYTE[f]
        nb = (count + per - 1) // per
        for b in buf[pos:pos + nb]:
            for _ in range(per):
                out.append(b % f)
                b //= f
        pos += nb
        out = out[:count]
    if len(out) != count:
        raise BitcodeError("truncated: byte data")
    return [x + base for x in out], pos


# ---- lengths, f32, str ------------------------------------------------------------------------------------------------------
def pack_lengths(lengths):
    small = [n if n < 255 else 255 for n in lengths]
    large = [n for n in lengths if n >= 255]
    return pack_bytes(small) + pack_ints(large, 8)


def unpack_lengths(buf, pos, count):
    small, pos = unpack_bytes(buf, pos, count)
    n_large = sum(1 for x in small if x == 255)
    large, pos = unpack_ints(buf, pos, n_large, 8)
    it = iter(int(x) for x in large)
    return [x if x != 255 else next(it) for x in small], pos


def pack_f32(values):
    bits = np.asarray(values, np.float32).view(np.uint32)
    rot = ((bits << np.uint32(1)) | (bits >> np.uint32(31))).astype(np.uint32)      # sign to bit 0
    mant = np.stack([rot & 0xFF, (rot >> 8) & 0xFF, (rot >> 16) & 0xFF], axis=1).astype(np.uint8)
    return mant.tobytes() + (rot >> 24).astype(np.uint8).tobytes()


def unpack_f32(buf, pos, count):
    end = pos + 4 * count
    if end > len(buf):
        raise BitcodeError("truncated: f32 data")
    mant = np.frombuffer(buf, np.uint8, 3 * count, pos).reshape(count, 3).astype(np.uint32)
    exp = np.frombuffer(buf, np.uint8, count, pos + 3 * count).astype(np.uint32)
    rot = mant[:, 0] | (mant[:, 1] << 8) | (mant[:, 2] << 16) | (exp << 24)
    bits = ((rot >> np.uint32(1)) | (rot << np.uint32(31))).astype(np.uint32)
    return bits.view(np.float32).copy(), end


# ---- PackedIndexEntry ---------------------------------------------------------------------------------------------------------
FIELDS = ("vector", "vertices", "id", "timestamp", "dimensions", "scores", "url", "shards")   # src/common.rs:155-164


def encode_packed_index_entry(e) -> bytes:
    """e: mapping with FIELDS; vector = u16 f16 bit patterns, vertices / shards = u32 lists, dimensions = (w, h), url = str."""
    url = e["url"].encode("utf-8") if isinstance(e["url"], str) else bytes(e["url"])
    out = bytearray()
    out += pack_lengths([len(e["vector"])]) + pack_ints(e["vector"], 2)
    out += pack_lengths([len(e["vertices"])]) + pack_ints(e["vertices"], 4)
    out += pack_ints([e["id"]], 4)
    out += pack_ints([e["timestamp"]], 8)
    out += pack_ints([e["dimensions"][0]], 4) + pack_ints([e["dimensions"][1]], 4)
    out += pack_lengths([len(e["scores"])]) + pack_f32(e["scores"])
    out += pack_lengths([len(url)]) + url
    out += pack_lengths([len(e["shards"])]) + pack_ints(e["shards"], 4)
    return bytes(out)


def decode_packed_index_entry(payload) -> dict:
    buf = bytes(payload)
    pos = 0
    (n,), pos = unpack_lengths(buf, pos, 1)
    vector, pos = unpack_ints(buf, pos, n, 2)
    (n,), pos = unpack_lengths(buf, pos, 1)
    vertices, pos = unpack_ints(buf, pos, n, 4)
    (id_,), pos = unpack_ints(buf, pos, 1, 4)
    (ts,), pos = unpack_ints(buf, pos, 1, 8)
    (w,), pos = unpack_ints(buf, pos, 1, 4)
    (h,), pos = unpack_ints(buf, pos, 1, 4)
    (n,), pos = unpack_lengths(buf, pos, 1)
    scores, pos = unpack_f32(buf, pos, n)
    (n,), pos = unpack_lengths(buf, pos, 1)
    if pos + n > len(buf):
        raise BitcodeError("truncated: url")
    url = buf[pos:pos + n].decode("utf-8")
    pos += n
    (n,), pos = unpack_lengths(buf, pos, 1)
    shards, pos = unpack_ints(buf, pos, n, 4)
    if pos != len(buf):
        raise BitcodeError("%d trailing bytes after the entry" % (len(buf) - pos))   # bitcode::decode rejects them too
    return {"vector": vector.astype(np.uint16), "vertices": vertices.astype(np.uint32), "id": int(id_), "timestamp": int(ts),
            "dimensions": (int(w), int(h)), "scores": scores, "url": url, "shards": shards.astype(np.uint32)}


def _selftest():   # pragma: no cover
    e = {"vector": list(range(1000, 2152)), "vertices": [5, 70000, 3], "id": 7, "timestamp": 1700000000, "dimensions": (640, 480),
         "scores": [0.5, -1.25, 3.0], "url": "https://example.org/a.png", "shards": [1, 2]}
    d = decode_packed_index_entry(encode_packed_index_entry(e))
    assert list(d["vector"]) == e["vector"] and d["url"] == e["url"] and struct.pack("<3f", *d["scores"]) == struct.pack("<3f", *e["scores"])
