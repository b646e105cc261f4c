#!/usr/bin/env python3
"""Latency of ONE file through the plugin surface (NyquistIO::Load on an in-memory buffer): first call
(contexts, staging memory) and steady state.   python tools/load_latency.py [file.opus]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process)
from test_host_decoder import load_host  # noqa: E402

H = load_host()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "short.opus")
raw = open(path, "rb").read()
info = np.zeros(8, np.int64)
n = H.nyqh_nyquistio_load_buffer(raw, len(raw), None, 0, info)
out = np.zeros(n, np.float32)
ts = []
for _ in range(21):
    t0 = time.perf_counter()
    assert H.nyqh_nyquistio_load_buffer(raw, len(raw), out.ctypes.data_as(C.c_void_p), n, info) == n
    ts.append(time.perf_counter() - t0)
print(json.dumps({"file": os.path.basename(path), "samples": int(n), "first_call_ms": ts[0] * 1e3,
                  "steady_ms_median": float(np.median(ts[1:])) * 1e3, "steady_ms_min": min(ts[1:]) * 1e3,
                  "audio_seconds": n / info[0] / 48000.0}))
