"""Developer probe run on the GPU box: quick parity + timing of the brute-force kernels.
Usage: python scripts/dev_probe.py [n_rows_for_timing]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "meme-search-engine_amd"))
from oracle import orc  # noqa: E402  (developer probe: compares against the checker, never shipped)

L = C.CDLL(os.path.join(ROOT, "meme-search-engine_amd", "lib", "libmse_hip.so"))
vp, sz = C.c_void_p, C.c_size_t
u16p, u32p, i64p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_int64)
L.mse_last_error.restype = C.c_char_p
L.mse_base_generate.restype = vp
L.mse_base_generate.argtypes = [C.c_uint32, C.c_uint64, sz, sz]
L.mse_base_from_host.restype = vp
L.mse_base_from_host.argtypes = [u16p, sz, sz]
L.mse_base_free.argtypes = [vp]
L.mse_base_read_rows.argtypes = [vp, sz, sz, u16p]
L.mse_searcher_new.restype = vp
L.mse_searcher_new.argtypes = [vp]
L.mse_searcher_free.argtypes = [vp]
L.mse_bruteforce_topk_f16.argtypes = [vp, u16p, sz, sz, C.c_int, i64p, u32p]
L.mse_bruteforce_scores_f16.argtypes = [vp, u16p, i64p]
L.mse_score_rows_f16.argtypes = [vp, u32p, sz, u16p, i64p]
L.mse_fast_dot_f16.argtypes = [u16p, u16p, sz, i64p]
L.mse_searcher_last_stats.argtypes = [vp, u32p, u32p]
L.mse_device_mem_info.argtypes = [C.POINTER(sz), C.POINTER(sz)]


def P(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def ck(rc):
    if rc != 0:
        raise RuntimeError(L.mse_last_error().decode())


def topk(s, q, k, mode):
    q = np.ascontiguousarray(q)
    nq = q.shape[0]
    sc = np.empty((nq, k), np.int64)
    ids = np.empty((nq, k), np.uint32)
    ck(L.mse_bruteforce_topk_f16(s, P(q, C.c_uint16), nq, k, mode, P(sc, C.c_int64), P(ids, C.c_uint32)))
    return sc, ids


def main():
    f, t = sz(), sz()
    ck(L.mse_device_mem_info(C.byref(f), C.byref(t)))
    print("mem free/total GB", f.value / 1e9, t.value / 1e9)
    d = 1152
    # 1. generator parity
    n = 4096
    b = L.mse_base_generate(0x5EED0001, 0, n, d)
    assert b, L.mse_last_error()
    dev_rows = np.empty((n, d), np.uint16)
    ck(L.mse_base_read_rows(b, 0, n, P(dev_rows, C.c_uint16)))
    host_rows = orc.gen_rows_f16(0x5EED0001, 0, n, d)
    print("generator mismatching halves:", int((dev_rows != host_rows).sum()))
    s = L.mse_searcher_new(b)
    q = orc.gen_rows_f16(0x5EED0002, 0, 130, d)
    # 2. exact scores parity
    sc = np.empty(n, np.int64)
    ck(L.mse_bruteforce_scores_f16(s, P(q[0], C.c_uint16), P(sc, C.c_int64)))
    ref = orc.score_all(host_rows, q[0])
    print("exact score mismatches:", int((sc != ref).sum()), "of", n)
    # fast_dot single
    out = C.c_int64()
    ck(L.mse_fast_dot_f16(P(q[0], C.c_uint16), P(host_rows[5], C.c_uint16), d, C.byref(out)))
    print("fast_dot single ok:", out.value == orc.fast_dot(q[0], host_rows[5]))
    # 3. top-k parity small
    for mode, name, nq in ((1, "exact", 11), (2, "mfma", 130)):
        rs, ri = orc.bruteforce_topk(host_rows, q[:nq], 10)
        gs, gi = topk(s, q[:nq], 10, mode)
        print(f"top10 {name} n={n}: id mismatches {int((gi != ri).sum())} score mismatches {int((gs != rs).sum())}")
    L.mse_searcher_free(s)
    L.mse_base_free(b)

    # 4. larger parity (multi-level descent) against the oracle on a subset of queries
    for n in (300_000, 5_000_000):
        b = L.mse_base_generate(0x5EED0001, 0, n, d)
        assert b, L.mse_last_error()
        s = L.mse_searcher_new(b)
        t0 = time.time()
        ge, gie = topk(s, q[:8], 10, 1)
        t1 = time.time()
        gm, gim = topk(s, q[:128], 10, 2)
        t2 = time.time()
        print(f"n={n}: exact(8q) {t1 - t0:.3f}s mfma(128q) {t2 - t1:.3f}s; exact==mfma ids {bool((gie == gim[:8]).all())} "
              f"scores {bool((ge == gm[:8]).all())}")
        a, bb = C.c_uint32(), C.c_uint32()
        L.mse_searcher_last_stats(s, C.byref(a), C.byref(bb))
        print("   mfma stats widened", a.value, "max_groups", bb.value)
        if n <= 300_000:
            host = orc.gen_rows_f16(0x5EED0001, 0, n, d)
            rs, ri = orc.bruteforce_topk(host, q[:4], 10)
            print("   oracle check ids", bool((ri == gie[:4]).all()), "scores", bool((rs == ge[:4]).all()))
        else:
            # spot-check returned scores with the oracle on regenerated rows
            ok = True
            for qi in range(2):
                for j in range(10):
                    row = orc.gen_rows_f16(0x5EED0001, int(gie[qi, j]), 1, d)[0]
                    ok &= orc.fast_dot(q[qi], row) == int(ge[qi, j])
            print("   spot check of returned scores vs oracle:", ok)
        # timing
        for mode, name, nq in ((1, "exact", 1), (1, "exact", 8), (2, "mfma", 128)):
            topk(s, q[:nq], 10, mode)
            t0 = time.time()
            reps = 3
            for _ in range(reps):
                topk(s, q[:nq], 10, mode)
            dt = (time.time() - t0) / reps
            print(f"   {name} nq={nq}: {dt * 1e3:.2f} ms/scan  {n * d * 2 / dt / 1e9:.0f} GB/s  {nq / dt:.0f} QPS")
        L.mse_searcher_free(s)
        L.mse_base_free(b)

    if len(sys.argv) > 1:
        n = int(float(sys.argv[1]))
        b = L.mse_base_generate(0x5EED0001, 0, n, d)
        assert b, L.mse_last_error()
        s = L.mse_searcher_new(b)
        for mode, name, nq in ((1, "exact", 1), (1, "exact", 4), (1, "exact", 8), (2, "mfma", 128)):
            topk(s, q[:nq], 10, mode)
            t0 = time.time()
            reps = 5
            for _ in range(reps):
                topk(s, q[:nq], 10, mode)
            dt = (time.time() - t0) / reps
            print(f"n={n} {name} nq={nq}: {dt * 1e3:.2f} ms/scan  {n * d * 2 / dt / 1e9:.0f} GB/s  {nq / dt:.0f} QPS")
        L.mse_searcher_free(s)
        L.mse_base_free(b)


if __name__ == "__main__":
    main()
