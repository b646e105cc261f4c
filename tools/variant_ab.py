#!/usr/bin/env python3
"""Kernel-source variants of libnyq_imdct against each other in ONE process, interleaved: every variant is the product's
sources compiled with its own -D flags into tools/variants/<name>.so (build step, runs without a GPU), loaded side by side
(ctypes, RTLD_LOCAL), and the post-filter stage and the frames -> PCM chain are timed round-robin on the same buffers.
  build:   python tools/variant_ab.py build  base= nodeemph=-DNYQ_PIPE_NO_COMB_DEEMPH light64=-DNYQ_PIPE_LIGHT_PERIOD=64
  run:     python tools/variant_ab.py run [nstreams] [nframes] [mix|real|short|long|off ...]
Outputs of every variant are compared with the first one's (max |diff|)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "variants")


def build(specs):
    from libnyquist_amd import _build
    os.makedirs(VDIR, exist_ok=True)
    for f in os.listdir(VDIR):
        if f.endswith(".so"):
            os.unlink(os.path.join(VDIR, f))
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        cmd = [_build.hipcc(), "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
               "-Wl,-Bsymbolic-functions"] + flags.split() + ["-o", os.path.join(VDIR, name + ".so")] + _build.SOURCES
        procs.append((name, subprocess.Popen(cmd)))
    for name, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"variant {name} failed to build")
    json.dump([s.partition("=")[0] + ("  [" + s.partition("=")[2] + "]" if s.partition("=")[2] else "") for s in specs],
              open(os.path.join(VDIR, "order.json"), "w"))
    print("built", [n for n, _ in procs])


def run(ns, nf, cases):
    import numpy as np
    import torch
    import libnyquist_amd as nyq
    names = json.load(open(os.path.join(VDIR, "order.json")))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctxs = []
    for nm in names:
        c = nyq.Context.__new__(nyq.Context)
        c.lib = nyq.binding.load(os.path.join(VDIR, nm.split("  [")[0] + ".so"))
        import ctypes as C
        h = C.c_void_p()
        assert c.lib.nyq_ctx_create(C.byref(h), 0) == 0
        c.h, c.device = h, 0
        c.set_stream(stream.cuda_stream)
        ctxs.append(c)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    ch, n = 2, 960
    freq = torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30.0
    out = torch.empty((ns, nf * n, ch), device=dev)
    pcm = torch.empty((ns * ch, nf * n), device=dev)
    work = torch.empty(ctxs[0].celt_synth_work_floats(ns, nf, ch), device=dev)
    for case in cases:
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
            lo, hi, on = {"mix": (15, 80, 0.7), "short": (15, 60, 1.0), "long": (300, 1000, 1.0), "off": (15, 80, 0.0), "all": (15, 1000, 0.9)}[case]
            pitch = torch.randint(lo, hi, (ns, nf), generator=g, device=dev, dtype=torch.int32)
            gain = (torch.rand((ns, nf), generator=g, device=dev) < on).float() * (torch.randint(1, 9, (ns, nf), generator=g, device=dev) * 0.09375).float()
            tap = torch.randint(0, 3, (ns, nf), generator=g, device=dev, dtype=torch.int32)
        ctxs[0].celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), 0, work.data_ptr(), ns, nf, ch)
        torch.cuda.synchronize(dev)
        rows = ns * nf * ch
        tail_buf = torch.empty((rows, 60), device=dev)
        # the verdict's three "slow kernels" on the same bytes: nfft-60 rows (8 x as many rows of an eighth the length), LM 1
        # frame synthesis (4 x as many 240-sample frames, 2.8 % transient), post-filter of LM 1 frames
        rows60 = rows * 8
        tail60 = torch.empty((rows60, 60), device=dev) if "rows60" in (os.environ.get("VAB_OPS") or "") else tail_buf
        nf1 = nf * 4
        trans1 = (torch.rand((ns, nf1), generator=g, device=dev) < 0.028).to(torch.uint8)
        work1 = torch.empty(ctxs[0].celt_synth_work_floats(ns, nf1, ch), device=dev)
        pitch1, gain1, tap1 = (t.repeat_interleave(4, dim=1).contiguous() for t in (pitch, gain, tap))
        ops = {"imdct_rows": lambda c: c.imdct_batch_dev(0, freq.data_ptr(), 0, pcm.data_ptr(), tail_buf.data_ptr(), rows),
               "rows60": lambda c: c.imdct_batch_dev(3, freq.data_ptr(), 0, pcm.data_ptr(), tail60.data_ptr(), rows60),
               "synth1": lambda c: c.celt_synth_dev(1, freq.data_ptr(), trans1.data_ptr(), pcm.data_ptr(), 0, work1.data_ptr(), ns, nf1, ch),
               "post1": lambda c: c.celt_post_dev(1, pcm.data_ptr(), pitch1.data_ptr(), gain1.data_ptr(), tap1.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf1, ch),
               "synth": lambda c: c.celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), 0, work.data_ptr(), ns, nf, ch),
               "post": lambda c: c.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch),
               "chain": lambda c: c.celt_chain_dev(3, freq.data_ptr(), trans.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, 0,
                                                   out.data_ptr(), pcm.data_ptr(), work.data_ptr(), ns, nf, ch)}
        only = os.environ.get("VAB_OPS")           # e.g. VAB_OPS=chain,post
        for opname, op in ops.items():
            if only and opname not in only.split(","):
                continue
            times = [[] for _ in ctxs]
            outs = []
            for rnd in range(12):
                for i, c in enumerate(ctxs):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    for _ in range(3):
                        op(c)
                    b.record(stream)
                    torch.cuda.synchronize(dev)
                    if rnd >= 2:
                        times[i].append(a.elapsed_time(b) / 3)
                    if rnd == 0:
                        outs.append(out.clone())
            res = {"case": case, "op": opname, "shape": f"{ns} x {nf} x {ch}"}
            for i, nm in enumerate(names):
                ms = sorted(times[i])[len(times[i]) // 2]
                res[nm] = {"ms": round(ms, 4), "min_ms": round(min(times[i]), 4), "GBps": round(ns * nf * ch * 7680 / ms / 1e6, 1),
                           "max_abs_diff_vs_first": float((outs[i] - outs[0]).abs().max())}
            print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "build":
        build(sys.argv[2:] or ["base="])
    else:
        a = sys.argv[2:] if len(sys.argv) >= 2 and sys.argv[1] == "run" else sys.argv[1:]
        run(int(a[0]) if a else 1024, int(a[1]) if len(a) > 1 else 256, a[2:] or ["mix", "real"])
