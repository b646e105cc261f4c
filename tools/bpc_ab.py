#!/usr/bin/env python3
"""Waves per CU of the persistent row kernels (NYQ_OPT_BLOCKS_PER_CU), interleaved in ONE process: frame synthesis
(2.8 % transient frames) and the plain row kernel of every size.  usage: bpc_ab.py 6,7,8"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libnyquist_amd as nyq  # noqa: E402

bpcs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "6,7,8").split(",")]
dev = torch.device("cuda", 0)
ctx = nyq.Context(0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(4)
ns, ch = 1024, 2


def sweep(name, fn, unit_bytes):
    times = {b: [] for b in bpcs}
    for rnd in range(11):
        for b in bpcs:
            ctx.set_option(nyq.binding.OPT_BLOCKS_PER_CU, b)
            fn()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(3):
                fn()
            e.record(stream)
            torch.cuda.synchronize(dev)
            if rnd >= 2:
                times[b].append(a.elapsed_time(e) / 3)
    out = {"op": name}
    for b in bpcs:
        ms = sorted(times[b])[len(times[b]) // 2]
        out[f"bpc_{b}"] = {"ms": round(ms, 4), "GBps": round(unit_bytes / ms / 1e6, 1)}
    print(json.dumps(out), flush=True)


for lm in (3, 2, 1, 0):
    n = 120 << lm
    nf = 256 << (3 - lm)
    freq = torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30.0
    pcm = torch.empty((ns, ch, nf * n), device=dev)
    state = torch.zeros((ns * ch, 60), device=dev)
    work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    tr = (torch.rand((ns, nf), generator=g, device=dev) < 0.028).to(torch.uint8)
    sweep(f"celt_synth_dev LM {lm}, 2.8 % transient", lambda: ctx.celt_synth_dev(lm, freq.data_ptr(), tr.data_ptr(), pcm.data_ptr(), state.data_ptr(), work.data_ptr(), ns, nf, ch),
          ns * nf * ch * n * 8)
    rows = ns * nf * ch
    tail = torch.empty((rows, 60), device=dev)
    sweep(f"imdct_batch_dev nfft {60 << lm}", lambda: ctx.imdct_batch_dev(3 - lm, freq.data_ptr(), 0, pcm.data_ptr(), tail.data_ptr(), rows), rows * n * 8)
    sweep(f"ifft_batch_dev nfft {60 << lm}", lambda: ctx.ifft_batch_dev(60 << lm, freq.data_ptr(), pcm.data_ptr(), rows // 2), rows * n * 8)
    del freq, pcm, work, tail
