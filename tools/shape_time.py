#!/usr/bin/env python3
"""tools/shape_time.py -- how long celt_shape_kernel takes on real symbols: tests/golden/sb-reverie.opus (11184 stereo 20 ms
frames) decoded to symbol records by the host entropy stage, replicated REP times on the device, nyq_celt_shape_dev timed
with the wall clock around a synchronised launch (best of 6).  Further arguments: variant libraries (tools/variant_ab.py
build name=-DNYQ_SHAPE_...), timed the same way after the product build."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import libnyquist_amd as nyq  # noqa: E402
from test_host_decoder import load_host  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 8
VARIANTS = sys.argv[2:]                                       # tools/variants/<name>.so built by tools/variant_ab.py build
H = load_host()
H.nyqh_symbol_bytes.argtypes = [C.c_int]
H.nyqh_symbol_bytes.restype = C.c_long
u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
H.nyqh_decode_to_symbols.argtypes = [C.c_char_p, C.c_long, C.c_long, u8] + list(H.nyqh_decode_to_freq.argtypes[4:])
raw = open(os.path.join(ROOT, "tests", "golden", "sb-reverie.opus"), "rb").read()
cap = 12000
rec = H.nyqh_symbol_bytes(2)
sym = np.zeros((cap, rec), np.uint8)
flags = np.zeros((cap, 4), np.int32)
gain = np.zeros(cap, np.float32)
rng = np.zeros(cap, np.uint32)
info = np.zeros(8, np.int64)
assert H.nyqh_decode_to_symbols(raw, len(raw), cap, sym, flags, gain, rng, info) == 0
nf = int(info[2])
heads = sym[:nf, :32].copy().view(np.uint16).reshape(nf, 16)
print(f"{nf} frames, {int(info[6])} host-built; per frame: leaves {heads[:, 2].mean():.1f} (max {heads[:, 2].max()}), "
      f"vectors {heads[:, 3].mean():.1f}, operations {heads[:, 4].mean():.1f}")
dev = torch.device("cuda", 0)
ctx = nyq.Context(0)
d_sym = torch.from_numpy(sym[:nf]).to(dev).repeat(REP, 1).contiguous()
d_freq = torch.empty((REP * nf, 2, 960), device=dev)
torch.cuda.synchronize(dev)
best = 1e9
for _ in range(6):
    t0 = time.perf_counter()
    ctx.celt_shape_dev(d_sym.data_ptr(), d_freq.data_ptr(), REP, nf, 2)
    ctx.synchronize()
    best = min(best, time.perf_counter() - t0)
n = REP * nf
print(f"celt_shape_kernel: {n} frames in {best * 1e3:.3f} ms = {n / best / 1e6:.2f} M frames/s "
      f"({n * (rec + 7680) / best / 1e9:.0f} GB/s of records in + freq out)")
for path in VARIANTS:
    c = nyq.Context.__new__(nyq.Context)
    c.lib = nyq.binding.load(path)
    h = C.c_void_p()
    assert c.lib.nyq_ctx_create(C.byref(h), 0) == 0
    c.h, c.device = h, 0
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        c.celt_shape_dev(d_sym.data_ptr(), d_freq.data_ptr(), REP, nf, 2)
        c.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"  variant {os.path.basename(path)}: {best * 1e3:.3f} ms = {n / best / 1e6:.2f} M frames/s")
