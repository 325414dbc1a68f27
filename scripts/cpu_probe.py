import os, sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from oracle import orc
orc.build()
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
block = orc.gen_rows_f16(0x5EED0001, 0, 100000)
D = 1152
rows = 1_000_000
base = np.empty((rows, D), np.uint16)
def par(fn, n):
    th = [threading.Thread(target=fn, args=(t, n)) for t in range(n)]
    t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; return time.perf_counter() - t0
def fill(t, n):
    chunk = (rows + n - 1) // n
    r, hi = t * chunk, min(rows, (t + 1) * chunk)
    while r < hi:
        off = r % 100000; m = min(hi - r, 100000 - off)
        np.copyto(base[r:r + m], block[off:off + m]); r += m
dt = par(fill, 64); print("fill 64 threads: %.2f s = %.1f GB/s" % (dt, rows * D * 2 / dt / 1e9))
q = orc.gen_rows_f16(0x5EED0002, 0, 300)
def readsum(t, n):
    chunk = (rows + n - 1) // n
    base[t * chunk:(t + 1) * chunk].sum(dtype=np.uint64)
for n in (16, 64, 128):
    dt = par(readsum, n); print("numpy sum over the sample with %d threads: %.1f GB/s" % (n, rows * D * 2 / dt / 1e9))
def work(t, n):
    orc.bruteforce_topk(base, q[t:t + 1], 10)
for n in (1, 8, 32, 64, 128, 256):
    dt = par(work, n); print("fair mode %3d threads: %.2f s, %.1f q/s on the sample, %.1f GB/s" % (n, dt, n / dt, n * rows * D * 2 / dt / 1e9))
