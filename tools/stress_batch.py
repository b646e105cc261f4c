#!/usr/bin/env python3
"""Stress of the overlapped batch decode path: many (count, threads) combinations over files of different
shapes; the first and the last stream of every batch must be bit-identical to the file decoded alone
(pieces, feeder threads, staging reuse and later-segment rounds must not depend on batch composition)."""
import ctypes as C
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from test_host_decoder import load_host  # noqa: E402

H = load_host()
files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "corpus", "*.opus"))) + [os.path.join(ROOT, "tests", "golden", "short.opus")]
files = [f for f in files if "unsupported" not in f and "surround" not in f]
stats = np.zeros(4, np.float64)
t0 = time.time()
runs = bad = 0
for f in files:
    raw = open(f, "rb").read()
    info = np.zeros(8, np.int64)
    n = H.nyqh_nyquistio_load_buffer(raw, len(raw), None, 0, info)
    alone = np.zeros(n, np.float32)
    assert H.nyqh_nyquistio_load_buffer(raw, len(raw), alone.ctypes.data_as(C.c_void_p), n, info) == n
    for count, threads in ((1, 1), (2, 16), (7, 3), (33, 16), (200, 5), (777, 16)):
        first, last = np.zeros(n, np.float32), np.zeros(n, np.float32)
        got = H.nyqh_batch_decode(raw, len(raw), count, threads, first.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), n, stats)
        runs += 1
        if got != n or not np.array_equal(first, alone) or not np.array_equal(last, alone):
            bad += 1
            print("MISMATCH", os.path.basename(f), count, threads, got, n)
print(f"{runs} batches over {len(files)} files in {time.time() - t0:.1f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
