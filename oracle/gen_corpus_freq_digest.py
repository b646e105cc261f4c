#!/usr/bin/env python3
"""oracle/gen_corpus_freq_digest.py -- TEST INFRASTRUCTURE.  What the REFERENCE decoder's entropy stage hands to its
IMDCT for every file of tests/golden/corpus (one elementary stream each): oracle/_ref/ref_capture (NyquistIO::Load
built from the reference's own sources with the capture tap of oracle/tap/) digests the input of EVERY
clt_mdct_backward call -- a block of the frame's freq[] -- as (shift, stride, n2, sum, sum of squares, eight picked
coefficients).  tests/test_opus_corpus.py computes the same digests from the host entropy stage's freq[] on CPU.
Output: tests/golden/corpus_freq_digest.npz.  Run from the repo root in the build container (needs oracle/_ref)."""
import glob
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CALL = np.dtype([("shift", "<i4"), ("stride", "<i4"), ("n2", "<i4"), ("sum", "<f8"), ("ss", "<f8"), ("pick", "<f4", 8)])

out = {}
with tempfile.TemporaryDirectory() as tmp:
    for p in sorted(glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))):
        name = os.path.basename(p)[:-5]
        if name.startswith(("unsupported_", "surround", "twosize")):
            continue
        cap = os.path.join(tmp, "cap.bin")
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_capture"), p, cap, "1000000", "calls"], check=True,
                       stdout=subprocess.DEVNULL)
        raw = open(cap + ".calls", "rb").read()
        n = struct.unpack("<i", raw[:4])[0]
        a = np.frombuffer(raw[4:], CALL, n)
        out[name + "/shape"] = np.stack([a["shift"], a["stride"], a["n2"]], 1).astype(np.int16)
        out[name + "/sums"] = np.stack([a["sum"], a["ss"]], 1)
        out[name + "/pick"] = a["pick"].copy()
        print(name, n, "calls")
np.savez_compressed(os.path.join(GOLDEN, "corpus_freq_digest.npz"), **out)
