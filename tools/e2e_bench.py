#!/usr/bin/env python3
"""End-to-end timing of the batched Opus decode (plugin surface path): `count` copies of
tests/golden/short.opus (220 stereo 20 ms frames + one 2.5 ms frame each) decoded as one batch.
Reports the CPU entropy stage and the GPU stage (incl. PCIe copies) separately -- the Amdahl split
SURVEY.md section 7 asks to state openly.   python tools/e2e_bench.py [count] [threads]"""
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
raw = open(os.path.join(ROOT, "tests", "golden", "short.opus"), "rb").read()
n = 421930
first = np.zeros(n, np.float32)
stats = np.zeros(4, np.float64)
H.nyqh_batch_decode(raw, len(raw), 8, threads, first.ctypes.data_as(C.c_void_p), None, n, stats)   # warm up
got = H.nyqh_batch_decode(raw, len(raw), count, threads, first.ctypes.data_as(C.c_void_p), None, n, stats)
assert got == n
cpu_s, gpu_s, frames, thr = stats
print(json.dumps({"streams": count, "frames": int(frames), "threads": int(thr),
                  "cpu_entropy_s": cpu_s, "cpu_frames_per_s": frames / cpu_s, "cpu_frames_per_s_per_thread": frames / cpu_s / thr,
                  "gpu_stage_s_incl_pcie": gpu_s, "gpu_frames_per_s_incl_pcie": frames / gpu_s,
                  "audio_seconds": count * 210965 / 48000.0,
                  "realtime_factor": count * 210965 / 48000.0 / (cpu_s + gpu_s)}))
