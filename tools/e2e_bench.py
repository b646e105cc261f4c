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
n = {"short.opus": 421930, "sb-reverie.opus": 21472602, "sb-reverie-60ms-frames.opus": 21472602}[fname]
first = np.zeros(n, np.float32)
stats = np.zeros(4, np.float64)
import time  # noqa: E402

H.nyqh_batch_decode(raw, len(raw), count, threads, first.ctypes.data_as(C.c_void_p), None, n, stats)   # warm up: contexts, pinned staging
t0 = time.perf_counter()
got = H.nyqh_batch_decode(raw, len(raw), count, threads, first.ctypes.data_as(C.c_void_p), None, n, stats)
wall = time.perf_counter() - t0
assert got == n
cpu_s, tail_s, frames, thr = stats
print(json.dumps({"streams": count, "frames": int(frames), "threads": int(thr),
                  "cpu_entropy_s": cpu_s, "cpu_frames_per_s": frames / cpu_s, "cpu_frames_per_s_per_thread": frames / cpu_s / thr,
                  "after_cpu_s": tail_s, "note": "GPU pieces (PCIe included) overlap the CPU stage; after_cpu_s = what was not hidden + trimming copy",
                  "wall_s_of_the_call": wall,
                  "file": fname, "audio_seconds": count * (n // 2) / 48000.0,
                  "realtime_factor": count * (n // 2) / 48000.0 / (cpu_s + tail_s)}))
