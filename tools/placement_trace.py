#!/usr/bin/env python3
"""Where and when the workgroups of celt_post_pipe_kernel run inside the frames -> PCM chain (diagnostic).

Needs the diagnostic library:
  hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -shared -fPIC -DNYQ_PIPE_STAMPS \
        -o tools/libnyq_imdct_diag.so libnyquist_amd/csrc/nyq_imdct.hip
usage: placement_trace.py [nstreams] [nframes]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libnyquist_amd as nyq  # noqa: E402
from libnyquist_amd import binding  # noqa: E402

L = binding.load(os.path.join(ROOT, "tools", "libnyq_imdct_diag.so"))
binding._lib = L
L.nyq_debug_pipe_wg.argtypes = [C.c_void_p, C.c_size_t]

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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
pitch = torch.randint(15, 80, (ns, nf), generator=g, device=dev, dtype=torch.int32)
gain = (torch.rand((ns, nf), generator=g, device=dev) < 0.7).float() * (torch.randint(1, 9, (ns, nf), generator=g, device=dev) * 0.09375).float()
tap = torch.randint(0, 3, (ns, nf), generator=g, device=dev, dtype=torch.int32)
out = torch.empty((ns, nf * n, ch), device=dev)
pcm = torch.empty((ns * ch, nf * n), device=dev)
work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
big = torch.empty(32 << 20, device=dev)


def synth():
    ctx.celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), 0, work.data_ptr(), ns, nf, ch)


def post():
    ctx.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch)


def report(what, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    fn()
    b.record(stream)
    torch.cuda.synchronize(dev)
    wg = np.zeros((4096, 4), np.uint64)
    assert L.nyq_debug_pipe_wg(wg.ctypes.data, wg.nbytes) == 0
    k = min(4096, ns * ch // 2)
    wg = wg[:k]
    hw, xcc = wg[:, 0].astype(np.int64), wg[:, 1].astype(np.int64) & 15
    cu = xcc * 1024 + ((hw >> 8) & 0xff)          # (cu_id, sh_id, se_id bits of HW_ID)
    simd = (hw >> 4) & 3
    t0 = int(wg[:, 2].min())
    st = (wg[:, 2].astype(np.int64) - t0) * 0.01
    life = (wg[:, 3].astype(np.int64) - wg[:, 2].astype(np.int64)) * 0.01
    _, cnt = np.unique(cu, return_counts=True)
    per = {int(c): int((cnt == c).sum()) for c in np.unique(cnt)}
    late = st > 20
    se = xcc * 8 + ((hw >> 13) & 7)
    _, secnt = np.unique(se, return_counts=True)
    seper = {int(c): int((secnt == c).sum()) for c in np.unique(secnt)}
    cucount = dict(zip(*np.unique(cu, return_counts=True)))
    latelist = sorted((int(se[i]), int(cu[i]) & 1023, int(cucount[cu[i]]), int(st[i])) for i in np.nonzero(late)[0])
    print(f"   workgroups per shader engine {seper} over {len(secnt)} engines; late ones (engine, cu bits, workgroups on that cu, start us): {latelist[:40]}")
    print(f"{what}: call {a.elapsed_time(b):.3f} ms; post kernel: {k} workgroups on {len(cnt)} CUs, workgroups per CU {per}; "
          f"started > 20 us after the first: {int(late.sum())} (their start: min {st[late].min() if late.any() else 0:.0f} max {st.max():.0f} us); "
          f"lifetime min/median/max {life.min():.0f}/{np.median(life):.0f}/{life.max():.0f} us; span {(int(wg[:, 3].max()) - t0) * 0.01:.0f} us; "
          f"first wave on SIMD {np.bincount(simd, minlength=4).tolist()}")


report("post alone", post)
report("synth + post", lambda: (synth(), post()))       # (the fill node that round 2's synthesis still had is gone: nothing to switch)
report("memset + post", lambda: (big.zero_(), post()))
report("post alone, again", post)
