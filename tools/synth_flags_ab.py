#!/usr/bin/env python3
"""nyq_celt_synth_dev per frame size with (a) 2.8 % transient frames, (b) a flag array of zeros, (c) no flag array:
what the transient-flag machinery of the long-frame kernel costs.  usage: synth_flags_ab.py [nstreams] [blocks per CU]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libnyquist_amd as nyq  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bpc = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # NYQ_OPT_BLOCKS_PER_CU (0 = built-in: 6 waves per CU)
dev = torch.device("cuda", 0)
ctx = nyq.Context(0)
ctx.set_option(nyq.binding.OPT_BLOCKS_PER_CU, bpc)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(4)
ch = 2
for lm in (3, 2, 1, 0):
    n = 120 << lm
    nf = 256 << (3 - lm)
    freq = torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30.0
    pcm = torch.empty((ns, ch, nf * n), device=dev)
    state = torch.zeros((ns * ch, 60), device=dev)
    work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    tr28 = (torch.rand((ns, nf), generator=g, device=dev) < 0.028).to(torch.uint8)
    tr0 = torch.zeros((ns, nf), device=dev, dtype=torch.uint8)
    cases = {"2.8 % transient": tr28.data_ptr(), "flags all zero": tr0.data_ptr(), "no flag array": 0}
    times = {k: [] for k in cases}
    for rnd in range(9):
        for k, ptr in cases.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(3):
                ctx.celt_synth_dev(lm, freq.data_ptr(), ptr, pcm.data_ptr(), state.data_ptr(), work.data_ptr(), ns, nf, ch)
            b.record(stream)
            torch.cuda.synchronize(dev)
            if rnd >= 2:
                times[k].append(a.elapsed_time(b) / 3)
    out = {"LM": lm, "shape": f"{ns} x {nf} x {ch}", "blocks_per_cu": bpc}
    for k in cases:
        ms = sorted(times[k])[len(times[k]) // 2]
        out[k] = {"ms": round(ms, 4), "GBps": round(ns * nf * ch * n * 8 / ms / 1e6, 1)}
    print(json.dumps(out))
    del freq, pcm, work
