#!/usr/bin/env python3
"""A deliberate soak of the situation DESIGN.md section 5a is about, on the real GPU: T threads inside NyquistIO::Load at once
(each leasing a decoder), R rounds over the whole corpus, while one more thread runs batches over a device LIST ({0, 0}: two
device shards with their own feeders and pinned arenas) -- everything compared with the sequential single-file results.
Native stderr is not captured: whatever any runtime says stays in the log.  usage: concurrent_load_soak.py [threads] [rounds]"""
import ctypes as C
import glob
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process)
from test_host_decoder import load_host  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H = load_host()
G = os.path.join(ROOT, "tests", "golden")
paths = [p for p in sorted(glob.glob(os.path.join(G, "corpus", "*.opus"))) if "unsupported" not in p] + [os.path.join(G, "short.opus")]
raws = [open(p, "rb").read() for p in paths]
want = []
info = np.zeros(8, np.int64)
for r in raws:
    n = H.nyqh_nyquistio_load_buffer(r, len(r), None, 0, info)
    a = np.zeros(n, np.float32)
    assert H.nyqh_nyquistio_load_buffer(r, len(r), a.ctypes.data_as(C.c_void_p), n, info) == n
    want.append(a)
errors, loads, batches = [], [0], [0]
lock = threading.Lock()
stop = threading.Event()


def loader(tid):
    inf = np.zeros(8, np.int64)
    for rep in range(R):
        for k in range(tid % len(raws), len(raws) + tid % len(raws)):
            k %= len(raws)
            out = np.zeros(want[k].size, np.float32)
            n = H.nyqh_nyquistio_load_buffer(raws[k], len(raws[k]), out.ctypes.data_as(C.c_void_p), out.size, inf)
            if n != want[k].size or not np.array_equal(out, want[k]):
                errors.append(("load", tid, rep, paths[k], n))
            with lock:
                loads[0] += 1


def batcher():
    cnt = len(raws)
    files = (C.c_char_p * cnt)(*raws)
    sizes = (C.c_long * cnt)(*[len(r) for r in raws])
    ns = (C.c_long * cnt)()
    two = (C.c_int * 2)(0, 0)
    cap = sum(a.size for a in want)
    while not stop.is_set():
        out = np.zeros(cap, np.float32)
        tot = H.nyqh_batch_load_devices(files, sizes, cnt, two, 2, ns, out.ctypes.data_as(C.c_void_p), cap)
        pos = 0
        for i, a in enumerate(want):
            if tot != cap or ns[i] != a.size or not np.array_equal(out[pos:pos + a.size], a):
                errors.append(("batch", i, tot))
                break
            pos += a.size
        batches[0] += 1


t0 = time.time()
th = [threading.Thread(target=loader, args=(t,)) for t in range(T)]
bt = threading.Thread(target=batcher)
bt.start()
for t in th:
    t.start()
for t in th:
    t.join()
stop.set()
bt.join()
counts = (C.c_long * 2)()
H.nyqh_decoder_pool_counts(counts)
print(f"{T} loading threads x {R} rounds x {len(raws)} files: {loads[0]} loads, {batches[0]} two-shard batches beside them, "
      f"{len(errors)} mismatches, decoders made {counts[0]}, torn down {counts[1]}, {time.time() - t0:.1f} s")
print(errors[:5])
sys.exit(1 if errors else 0)
