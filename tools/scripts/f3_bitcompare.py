#!/usr/bin/env python3
"""Row f3 gate: the files whose channel mapping / trim / gain now happen in the kernels' store phase must decode to the
SAME samples, bit for bit, as with the previous host library (per-sample mapping loop on the host), and a batch of them must
equal the single-file loads.  usage: f3_bitcompare.py <previous libnyquist_host.so>"""
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import libnyquist_amd as nyq  # noqa: E402  (loads torch's HIP runtime first)

nyq.load()
new = C.CDLL(os.path.join(ROOT, "libnyquist_amd", "libnyquist_host.so"))
old = C.CDLL(sys.argv[1])
for H in (new, old):
    H.nyqh_nyquistio_load_buffer.restype = C.c_long
    H.nyqh_nyquistio_load_buffer.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "corpus", "*.opus"))) + [os.path.join(ROOT, "tests", "golden", "short.opus")]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oggopus  # noqa: E402  (tools/oggopus.py: remuxes short.opus' packets into multistream variants)

# files made here: 4 channels in two coupled streams with a permuted mapping (tests/test_gpu_host.py), the same with a silent
# channel (mapping 255), one decoded channel named twice, and a header gain on a multistream file
_pk, _gr, _last = oggopus.read_packets(open(os.path.join(ROOT, "tests", "golden", "short.opus"), "rb").read())
_aud = _pk[2:]
_gran = [(i + 1) * 960 for i in range(220)] + [_last]
_two = [_aud, _aud[7:220] + _aud[:7] + _aud[220:]]


def _with_gain(raw, q8):
    # OpusHead is the first packet of the first page: patch output_gain (bytes 16..17 of the head) and the page checksum
    i = raw.index(b"OpusHead")
    out = bytearray(raw)
    out[i + 16:i + 18] = int(q8).to_bytes(2, "little", signed=True)
    p0 = raw.rindex(b"OggS", 0, i)
    nseg = out[p0 + 26]
    plen = 27 + nseg + sum(out[p0 + 27:p0 + 27 + nseg])
    out[p0 + 22:p0 + 26] = b"\0\0\0\0"
    out[p0 + 22:p0 + 26] = oggopus.ogg_crc(bytes(out[p0:p0 + plen])).to_bytes(4, "little")
    return bytes(out)


made = {"made: 4ch mapping [2,0,3,1]": oggopus.mux_family1(_two, 2, [2, 0, 3, 1], 312, _gran),
        "made: 4ch mapping [2,255,3,1] (silent channel)": oggopus.mux_family1(_two, 2, [2, 255, 3, 1], 312, _gran),
        "made: 4ch mapping [0,0,3,1] (one channel twice)": oggopus.mux_family1(_two, 2, [0, 0, 3, 1], 312, _gran)}
made["made: 4ch mapping [2,0,3,1], gain -3 dB"] = _with_gain(made["made: 4ch mapping [2,0,3,1]"], -768)


def load(H, raw):
    info = (C.c_long * 8)()
    n = H.nyqh_nyquistio_load_buffer(raw, len(raw), None, 0, info)
    if n < 0:
        return None, list(info)
    buf = np.zeros(n, np.float32)
    assert H.nyqh_nyquistio_load_buffer(raw, len(raw), buf.ctypes.data_as(C.c_void_p), n, info) == n
    return buf, list(info)


res = []
for f in files + list(made):
    raw = made[f] if f in made else open(f, "rb").read()
    a, ia = load(new, raw)
    b, ib = load(old, raw)
    t0 = time.perf_counter()
    for _ in range(3):
        load(new, raw)
    tn = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        load(old, raw)
    to = (time.perf_counter() - t0) / 3
    same = (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and np.array_equal(a, b))
    res.append({"file": os.path.basename(f), "channels": ia[0] if a is not None else None, "samples": None if a is None else int(a.size),
                "bit_identical_to_previous_host_library": bool(same), "load_ms_new": tn * 1e3, "load_ms_previous": to * 1e3})
    print(json.dumps(res[-1]), flush=True)
ok = all(r["bit_identical_to_previous_host_library"] for r in res)
print(json.dumps({"all_bit_identical": ok, "files": len(res)}))
sys.exit(0 if ok else 1)
