#!/usr/bin/env python3
"""Dump the 'real' post-filter parameter case (windows of sb-reverie.opus's own sequence, one per stream) as raw arrays for
tools/chainbench: usage dump_real_params.py out.bin [nstreams] [nframes]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 256
z = np.load(os.path.join(ROOT, "tests", "golden", "sb_reverie_pf_params.npz"))
tot = len(z["pf_pitch"])
idx = ((np.arange(ns) * 977) % (tot - nf))[:, None] + np.arange(nf)[None, :]
with open(out, "wb") as f:
    f.write(z["pf_pitch"][idx].astype(np.int32).tobytes())
    f.write((z["pf_gain_q"][idx].astype(np.float32) * np.float32(0.09375)).tobytes())
    f.write(z["pf_tapset"][idx].astype(np.int32).tobytes())
