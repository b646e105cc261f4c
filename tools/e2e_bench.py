#!/usr/bin/env python3
"""End-to-end timing of the batched Opus decode (plugin surface path): `count` copies of
tests/golden/short.opus (220 stereo 20 ms frames + one 2.5 ms frame each; or sb-reverie.opus, 11184 frames)
decoded as one batch.
Reports the CPU entropy stage and the GPU stage (incl. PCIe copies) separately -- the Amdahl split
SURVEY.md section 7 asks to state openly.   python tools/e2e_bench.py [count] [threads] [file]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime per process)
from test_host_decoder import load_host  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = load_host()
fname = sys.argv[3] if len(sys.argv) > 3 else "short.opus"      # or sb-reverie.opus (224 s, BASELINE config 4's file)
raw = open(os.path.join(ROOT, "tests", "golden", fname), "rb").read()
n = {"short.opus": 421930, "sb-reverie.opus": 21472602, "sb-reverie-60ms-frames.opus": 21472602,
     "corpus/surround71_20ms_320k.opus": 384000}[fname]
first = np.zeros(n, np.float32)
stats = np.zeros(6, np.float64)
import time  # noqa: E402

H.nyqh_batch_decode_timed(raw, len(raw), count, threads, first.ctypes.data_as(C.c_void_p), None, n, stats)   # warm up: contexts, pinned staging
t0 = time.perf_counter()
got = H.nyqh_batch_decode_timed(raw, len(raw), count, threads, first.ctypes.data_as(C.c_void_p), None, n, stats)
wall = time.perf_counter() - t0
assert got == n
cpu_s, tail_s, frames, thr, wall_inside, ndev = stats
res = {"streams": count, "frames": int(frames), "threads": int(thr), "devices": int(ndev),
       "wall_s_of_the_call": wall, "wall_s_measured_inside_the_library": wall_inside,
       "frames_per_s": frames / wall, "realtime_factor": count * (n // 2) / 48000.0 / wall,
       "breakdown": {"cpu_entropy_s": cpu_s, "after_cpu_s": tail_s, "cpu_frames_per_s_per_thread": frames / cpu_s / thr,
                     "note": "GPU pieces (PCIe included) overlap the CPU stage; after_cpu_s = what was not hidden + trimming copy; "
                             "the results are consumed and released inside the timed call (sink form, pooled buffers)"},
       "file": fname, "audio_seconds": count * (n // 2) / 48000.0}
rp = os.path.join(ROOT, "oracle", "_ref", "libref_decode.so")
if os.path.exists(rp) and os.environ.get("E2E_NO_REF") is None:
    R = C.CDLL(rp)
    R.ref_decode_bench.restype = C.c_double
    R.ref_decode_bench.argtypes = [C.c_char_p, C.c_long, C.c_long, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_double)]
    ns_, ck = C.c_long(0), C.c_double(0)
    secs = R.ref_decode_bench(raw, len(raw), count, threads, C.byref(ns_), C.byref(ck))
    res["reference_decoder_wall_s"] = secs
    res["vs_reference_wall_over_wall"] = secs / wall
print(json.dumps(res))
