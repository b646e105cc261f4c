#!/usr/bin/env python3
"""Time every device-resident operator of the C ABI on synthetic batches (HIP events on the operator's
stream) and report algorithmic GB/s.  Secondary to bench.py (which measures the headline nfft-480 rows)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libnyquist_amd as nyq  # noqa: E402

dev = torch.device("cuda", 0)
ctx = nyq.Context(0, ab=True)     # the tools' A/B build: the product's kernels + the round-1 post-filter forms
B = nyq.binding
BPC = int(os.environ.get('SWEEP_BPC', '0'))   # NYQ_OPT_BLOCKS_PER_CU of the persistent row kernels (0 = the per-size built-in)
ctx.set_option(B.OPT_BLOCKS_PER_CU, BPC)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(1)


def timeit(fn, reps=10):
    # steady clocks first (bench.py's pre-roll): the operator back to back for >= 40 ms, untimed
    spent = 0.0
    while spent < float(os.environ.get("SWEEP_PREROLL_MS", "40")):
        p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p0.record(stream)
        for _ in range(4):
            fn()
        p1.record(stream)
        torch.cuda.synchronize(dev)
        spent += p0.elapsed_time(p1)
    fn()
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps):
        fn()
    b.record(stream)
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / reps


res = []
ONLY = os.environ.get('SWEEP_ONLY', '')
for shift, rows in () if ONLY and ONLY != 'imdct' else ((0, 1 << 20), (1, 1 << 21), (2, 1 << 22), (3, 1 << 23)):
    n2 = 960 >> shift
    x = torch.rand((rows, n2), generator=g, device=dev) * 2 - 1
    fin = torch.empty_like(x)
    tail = torch.empty((rows, 60), device=dev)
    ms = timeit(lambda: ctx.imdct_batch_dev(shift, x.data_ptr(), 0, fin.data_ptr(), tail.data_ptr(), rows))
    res.append(dict(op=f"imdct_batch_dev nfft {480 >> shift}", rows=rows, ms=ms, alg_GBps=rows * n2 * 8 / ms / 1e6,
                    rows_per_s=rows / ms * 1e3))
    del x, fin, tail
for nfft, rows in () if ONLY and ONLY != "ifft" else ((60, 1 << 22), (120, 1 << 21), (240, 1 << 20), (480, 1 << 19)):
    x = torch.rand((rows, 2 * nfft), generator=g, device=dev)
    y = torch.empty_like(x)
    ms = timeit(lambda: ctx.ifft_batch_dev(nfft, x.data_ptr(), y.data_ptr(), rows))
    res.append(dict(op=f"ifft_batch_dev nfft {nfft}", rows=rows, ms=ms, alg_GBps=rows * nfft * 16 / ms / 1e6, rows_per_s=rows / ms * 1e3))
    del x, y
for n, rows in () if ONLY and ONLY != 'vorbis' else ((2048, 1 << 19), (256, 1 << 22), (4096, 1 << 18), (64, 1 << 23), (8192, 1 << 17)):
    x = torch.rand((rows, n // 2), generator=g, device=dev)
    y = torch.empty((rows, n), device=dev)
    ms = timeit(lambda: ctx.vorbis_imdct_batch_dev(n, x.data_ptr(), y.data_ptr(), rows))
    res.append(dict(op=f"vorbis_imdct_batch_dev n {n}", rows=rows, ms=ms, alg_GBps=rows * n * 6 / ms / 1e6, rows_per_s=rows / ms * 1e3))
    del x, y
# frame synthesis (nyq_celt_synth_dev) for every frame size: the same 2 GB of coefficients as 2.5 / 5 / 10 / 20 ms frames
for lm in () if ONLY and ONLY != 'synth' else (3, 2, 1, 0):
    n = 120 << lm
    sns, snf, sch = 1024, 256 << (3 - lm), 2
    sfreq = torch.randn((sns, snf, sch, n), generator=g, device=dev) * 30.0
    strans = (torch.rand((sns, snf), generator=g, device=dev) < 0.028).to(torch.uint8)
    spcm = torch.empty((sns, sch, snf * n), device=dev)
    sstate = torch.zeros((sns * sch, 60), device=dev)
    swork = torch.empty(ctx.celt_synth_work_floats(sns, snf, sch), device=dev)
    ms = timeit(lambda: ctx.celt_synth_dev(lm, sfreq.data_ptr(), strans.data_ptr(), spcm.data_ptr(), sstate.data_ptr(), swork.data_ptr(), sns, snf, sch), 5)
    res.append(dict(op=f"celt_synth_dev LM {lm} ({n}-sample frames), {sns} streams x {snf} frames x {sch}ch, 2.8 % transient", rows=sns * snf * sch, ms=ms,
                    alg_GBps=sns * snf * sch * n * 8 / ms / 1e6, rows_per_s=sns * snf * sch / ms * 1e3))
    del sfreq, spcm, swork, strans, sstate
ns, nf, ch = int(os.environ.get("SWEEP_NS", "1024")), int(os.environ.get("SWEEP_NF", "256")), 2
pcm = torch.randn((ns * ch, nf * 960), generator=g, device=dev) * 300
out = torch.empty((ns, nf * 960, ch), device=dev)
pt = torch.randint(0, 3, (ns, nf), generator=g, device=dev, dtype=torch.int32)
for label, lo, hi, gmax in () if ONLY and ONLY != 'post' else (("pitch 15..1000, gain 0..0.75", 15, 1000, 9), ("no post-filter (gain 0)", 15, 1000, 1),
                             ("pitch 300..1000", 300, 1000, 9), ("pitch 15..60", 15, 60, 9),
                             ("real-stream mix: 70 % filtered, pitch 15..80", 15, 80, -70)):
    pp = torch.randint(lo, hi, (ns, nf), generator=g, device=dev, dtype=torch.int32)
    if gmax < 0:      # filtered fraction in per cent, gains 0.09 .. 0.75
        on = (torch.rand((ns, nf), generator=g, device=dev) < (-gmax / 100.0)).float()
        pg = on * (torch.randint(1, 9, (ns, nf), generator=g, device=dev) * 0.09375).float()
    else:
        pg = (torch.randint(0, gmax, (ns, nf), generator=g, device=dev) * 0.09375).float()
    for mode in ("0", "1", "2"):      # one wave per channel vs one wave per stereo pair vs workgroup pipeline (default), same process, same box
        ctx.set_option(B.OPT_POST_FORM, {"0": B.POST_FORM_WAVE_PER_CHANNEL, "1": B.POST_FORM_WAVE_PER_PAIR, "2": B.POST_FORM_PIPELINE}[mode])
        ms = timeit(lambda: ctx.celt_post_dev(3, pcm.data_ptr(), pp.data_ptr(), pg.data_ptr(), pt.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch), 5)
        res.append(dict(op=f"celt_post_dev {ns} streams x {nf} frames x 2ch, {label}" + (" [stereo pairs]" if mode == "1" else " [pipeline]" if mode == "2" else ""),
                        rows=ns * nf * ch, ms=ms, alg_GBps=ns * nf * ch * 7680 / ms / 1e6, rows_per_s=ns * nf * ch / ms * 1e3))
    ctx.set_option(B.OPT_POST_FORM, B.POST_FORM_PIPELINE)
if not ONLY or ONLY == 'post':
    ns1, ch1 = 2048, 1
    out1 = torch.empty((ns1, nf * 960, ch1), device=dev)
    pp = torch.randint(15, 1000, (ns1, nf), generator=g, device=dev, dtype=torch.int32)
    pg = torch.zeros((ns1, nf), device=dev)
    pt1 = torch.zeros((ns1, nf), device=dev, dtype=torch.int32)
    ms = timeit(lambda: ctx.celt_post_dev(3, pcm.data_ptr(), pp.data_ptr(), pg.data_ptr(), pt1.data_ptr(), 0, 0, 0, 0, out1.data_ptr(), ns1, nf, ch1), 5)
    res.append(dict(op="celt_post_dev 2048 streams x 256 frames x 1ch, no post-filter", rows=ns1 * nf, ms=ms, alg_GBps=ns1 * nf * 7680 / ms / 1e6, rows_per_s=ns1 * nf / ms * 1e3))
for r in res:
    r["blocks_per_cu"] = BPC
    r["frac_8TBps"] = r["alg_GBps"] / 8000.0
    print(json.dumps(r))
