#!/usr/bin/env python3
"""tests/golden/sb_reverie_pf_params.npz: the post-filter parameters (period, gain index, tapset) and transient flags of
the 11184 frames of sb-reverie.opus, as this repository's own CPU entropy stage decodes them (the stage is held to the
reference decoder sample for sample on this file by tests/test_gpu_host.py).  Used by tools/chain_time.py and
tools/sweep_ops.py as the 'real' parameter case; no reference code involved."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_host_decoder import entropy_decode, load_host  # noqa: E402

H = load_host()
raw = open(os.path.join(ROOT, "tests", "golden", "sb-reverie.opus"), "rb").read()
rc, freq, flags, gain, rng, info = entropy_decode(H, raw, max_frames=11300)
assert rc == 0 and int(info[2]) == 11184, (rc, info)
nf = 11184
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sb_reverie_pf_params.npz"), pf_pitch=flags[:nf, 1].astype(np.int16),
                    pf_gain_q=np.round(gain[:nf] / 0.09375).astype(np.uint8), pf_tapset=flags[:nf, 2].astype(np.uint8),
                    transient=flags[:nf, 0].astype(np.uint8))
