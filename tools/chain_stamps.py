#!/usr/bin/env python3
"""Per-role cycle accounting of the one-launch frames -> PCM kernel (celt_chain_kernel, diagnostic).

Needs the diagnostic library (s_memtime stamps compiled in):
  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -shared -fPIC -DNYQ_PIPE_STAMPS [-D variant flags] \
        -o tools/libnyq_imdct_diag.so libnyquist_amd/csrc/nyq_imdct.hip
usage: chain_stamps.py [nstreams] [nframes] [mix|real|short|long|off]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libnyquist_amd as nyq  # noqa: E402
from libnyquist_amd import binding  # noqa: E402

L = binding.load(os.environ.get("NYQ_DIAG_LIB", os.path.join(ROOT, "tools", "libnyq_imdct_diag.so")))
binding._lib = L
L.nyq_debug_chain_stamps.argtypes = [C.c_void_p, C.c_int]

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 256
case = sys.argv[3] if len(sys.argv) > 3 else "mix"
dev = torch.device("cuda", 0)
ctx = nyq.Context(0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(4)
ch, n = 2, 960
freq = torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30.0
trans = (torch.rand((ns, nf), generator=g, device=dev) < 0.028).to(torch.uint8)
if case == "real":
    z = np.load(os.path.join(ROOT, "tests", "golden", "sb_reverie_pf_params.npz"))
    tot = len(z["pf_pitch"])
    idx = ((np.arange(ns) * 977) % (tot - nf))[:, None] + np.arange(nf)[None, :]
    pitch = torch.from_numpy(z["pf_pitch"][idx].astype(np.int32)).to(dev)
    gain = torch.from_numpy(z["pf_gain_q"][idx].astype(np.float32) * np.float32(0.09375)).to(dev)
    tap = torch.from_numpy(z["pf_tapset"][idx].astype(np.int32)).to(dev)
    trans = torch.from_numpy(z["transient"][idx].astype(np.uint8)).to(dev)
else:
    lo, hi, on = {"mix": (15, 80, 0.7), "short": (15, 60, 1.0), "long": (300, 1000, 1.0), "off": (15, 80, 0.0)}[case]
    pitch = torch.randint(lo, hi, (ns, nf), generator=g, device=dev, dtype=torch.int32)
    gain = (torch.rand((ns, nf), generator=g, device=dev) < on).float() * (torch.randint(1, 9, (ns, nf), generator=g, device=dev) * 0.09375).float()
    tap = torch.randint(0, 3, (ns, nf), generator=g, device=dev, dtype=torch.int32)
out = torch.empty((ns, nf * n, ch), device=dev)


def run():
    ctx.celt_chain_dev(3, freq.data_ptr(), trans.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, 0,
                       out.data_ptr(), 0, 0, ns, nf, ch)


for _ in range(3):
    run()
torch.cuda.synchronize(dev)
z = (C.c_ulonglong * 32)()
L.nyq_debug_chain_stamps(None, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(stream)
run()
b.record(stream)
torch.cuda.synchronize(dev)
L.nyq_debug_chain_stamps(z, 0)
per = float(ns * nf)
v = [x / per for x in z]
res = {"case": f"{ns} x {nf} stereo frames, {case}", "ms_with_stamps": a.elapsed_time(b),
       "comb wave (chain 0), s_memtime ticks per frame": {"at barrier": v[0], "comb steps": v[1]},
       "I/O wave": {"at barrier": v[8], "carry-over + parameters": v[9], "de-emphasis + stores": v[10]},
       "transform wave": {"at barrier": v[16], "wait for coefficients": v[17], "S0": v[18], "S2": v[19], "S3": v[20], "S4": v[21],
                          "transient frames + load issue": v[22]}}
print(json.dumps(res, indent=1))
