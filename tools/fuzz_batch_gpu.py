#!/usr/bin/env python3
"""Larger fuzz of the GPU batch path than the test tier runs: random byte damage / truncation of real files,
including a 224 s file (time slices already handed over when the damage is hit), single loads and mixed batches.
Nothing may crash, hang or return an impossible size.   python tools/fuzz_batch_gpu.py [iterations]"""
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
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
G = os.path.join(ROOT, "tests", "golden")
srcs = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(G, "corpus", "*.opus"))) + [os.path.join(G, "short.opus")]]
long_src = open(os.path.join(G, "sb-reverie.opus"), "rb").read()
rng = np.random.default_rng(77)


sys.path.insert(0, os.path.join(ROOT, "tools"))
import oggopus  # noqa: E402
import struct  # noqa: E402


def reseal(raw):
    raw = bytearray(raw)
    pos = 0
    while pos + 27 <= len(raw):
        if raw[pos:pos + 4] != b"OggS":
            pos += 1
            continue
        nseg = raw[pos + 26]
        if pos + 27 + nseg > len(raw):
            break
        ln = 27 + nseg + sum(raw[pos + 27:pos + 27 + nseg])
        if pos + ln > len(raw):
            break
        raw[pos + 22:pos + 26] = b"\0\0\0\0"
        raw[pos + 22:pos + 26] = struct.pack("<I", oggopus.ogg_crc(bytes(raw[pos:pos + ln])))
        pos += ln
    return raw


def damage(raw):
    raw = bytearray(raw)
    mode = int(rng.integers(0, 4))
    if mode == 0:
        for _ in range(int(rng.integers(1, 8))):
            raw[int(rng.integers(0, len(raw)))] = int(rng.integers(0, 256))
    elif mode == 1:
        raw = raw[: int(rng.integers(1, len(raw)))]
    elif mode == 2:
        lo = min(len(raw) - 1, 120)
        for _ in range(int(rng.integers(1, 16))):
            raw[int(rng.integers(lo, len(raw)))] ^= 1 << int(rng.integers(0, 8))
    if rng.integers(0, 4):                               # re-seal the pages so that the damage gets past the Ogg CRC
        raw = reseal(raw)
    return bytes(raw)                                    # mode 3: untouched


t0 = time.time()
ok = bad = 0
LOG = open(os.environ.get("FUZZ_LOG", "/dev/null"), "w")
DUMP = os.environ.get("FUZZ_DUMP")          # directory: the inputs of the current iteration are left there
info = np.zeros(8, np.int64)
for it in range(iters):
    if it % 50 == 49:                                    # a damaged long file now and then
        raw = damage(long_src)
        print(it, "long", len(raw), file=LOG, flush=True)
        if DUMP:
            open(os.path.join(DUMP, "cur_long.opus"), "wb").write(raw)
        n = H.nyqh_nyquistio_load_buffer(raw, len(raw), None, 0, info)
        assert n in (-1, -2) or 0 <= n <= 21472602 + 10, n
    else:
        cnt = int(rng.integers(1, 12))
        raws = [damage(srcs[int(rng.integers(0, len(srcs)))]) for _ in range(cnt)]
        files = (C.c_char_p * cnt)(*raws)
        sizes = (C.c_long * cnt)(*[len(r) for r in raws])
        ns = (C.c_long * cnt)()
        thr = int(rng.integers(1, 17))
        print(it, "batch", cnt, thr, [len(r) for r in raws], file=LOG, flush=True)
        if DUMP:
            for k, r in enumerate(raws):
                open(os.path.join(DUMP, f"cur_{k}.opus"), "wb").write(r)
            for k in range(cnt, 12):
                try:
                    os.unlink(os.path.join(DUMP, f"cur_{k}.opus"))
                except OSError:
                    pass
        total = H.nyqh_batch_decode_files(files, sizes, cnt, thr, ns, None, 0)
        assert total >= -1
        for k in range(cnt):
            assert ns[k] == -1 or 0 <= ns[k] <= 8 * 48000 * 12, ns[k]
            ok += ns[k] >= 0
            bad += ns[k] < 0
    if it % 100 == 99:
        print(f"{it + 1} iterations, {ok} decoded, {bad} refused, {time.time() - t0:.0f} s", flush=True)
print(f"done: {iters} iterations, {ok} decoded, {bad} refused")
