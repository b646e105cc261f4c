#!/usr/bin/env python3
"""Block digests of the reference decoder's PCM for the two files of the reference's own ctest
(CMakeLists.txt:199-217): sb-reverie.opus and sb-reverie-60ms-frames.opus -> tests/golden/sb_reverie_digest.npz.
TEST INFRASTRUCTURE: run here (needs oracle/_ref/libref_decode.so = the reference built from its own sources);
the GPU-tier test compares this build's NyquistIO::Load with these digests when the reference build is absent, so
the comparison never silently disappears.  Per file: block sums / sums of squares over 4800-sample blocks per channel
(float64), the first 9600 and last 2000 sample frames verbatim, and every 997th sample frame."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_decode.so"))
R.ref_decode_pcm.restype = C.c_long
R.ref_decode_pcm.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
out = {}
for key, fname in (("a", "sb-reverie.opus"), ("b", "sb-reverie-60ms-frames.opus")):
    raw = open(os.path.join(ROOT, "tests", "golden", fname), "rb").read()
    n = 21472602
    pcm = np.zeros(n, np.float32)
    assert R.ref_decode_pcm(raw, len(raw), pcm.ctypes.data_as(C.c_void_p), n, None) == n
    p2 = pcm.reshape(-1, 2).astype(np.float64)
    nb = p2.shape[0] // 4800
    blk = p2[: nb * 4800].reshape(nb, 4800, 2)
    out[key + "_file"] = np.array(fname)
    out[key + "_block_sum"] = blk.sum(axis=1)
    out[key + "_block_sq"] = (blk ** 2).sum(axis=1)
    out[key + "_head"] = pcm.reshape(-1, 2)[:9600].copy()
    out[key + "_tail"] = pcm.reshape(-1, 2)[-2000:].copy()
    out[key + "_every997"] = pcm.reshape(-1, 2)[::997].copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sb_reverie_digest.npz"), **out)
print({k: v.shape for k, v in out.items()})
