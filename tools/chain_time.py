#!/usr/bin/env python3
"""Time the frames -> PCM chain (nyq_celt_chain_dev) as ONE launch (nyq_chain_kernel.hpp) vs as two kernels, same process, interleaved; HIP events on
the operator's stream.  usage: chain_time.py [nstreams] [nframes] [mix|short|long|off|real]
CHAIN_SPLIT=1: the two stages alone next to the chain; CHAIN_WINDOWS="0,64,128": the two-kernel chain of the PRODUCT library
over time windows of that many frames (NYQ_OPT_CHAIN_WINDOW; 0 = one window), interleaved, medians.
The fused kernel and the round-1 post-filter form exist only in the tools' A/B build, which this tool loads."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libnyquist_amd as nyq  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 256
case = sys.argv[3] if len(sys.argv) > 3 else "mix"
dev = torch.device("cuda", 0)
ctx = nyq.Context(0, ab=os.environ.get('CHAIN_WINDOWS') is None or 'o' in os.environ['CHAIN_WINDOWS'])   # A/B build unless the product's windowed chain is what is timed
B = nyq.binding
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(4)
ch, n = 2, 960
freq = torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30.0
trans = (torch.rand((ns, nf), generator=g, device=dev) < 0.028).to(torch.uint8)
if case == "real":
    # every stream = a window of sb-reverie.opus's own parameter sequence (tests/golden/sb_reverie_pf_params.npz)
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "sb_reverie_pf_params.npz"))
    tot = len(z["pf_pitch"])
    off = (np.arange(ns) * 977) % (tot - nf)
    idx = off[:, None] + np.arange(nf)[None, :]
    pitch = torch.from_numpy(z["pf_pitch"][idx].astype(np.int32)).to(dev)
    gain = torch.from_numpy(z["pf_gain_q"][idx].astype(np.float32) * 0.09375).to(dev)
    tap = torch.from_numpy(z["pf_tapset"][idx].astype(np.int32)).to(dev)
    trans = torch.from_numpy(z["transient"][idx].astype(np.uint8)).to(dev)
else:
    lo, hi, on = {"mix": (15, 80, 0.7), "short": (15, 60, 1.0), "long": (300, 1000, 1.0), "off": (15, 80, 0.0)}[case]
    pitch = torch.randint(lo, hi, (ns, nf), generator=g, device=dev, dtype=torch.int32)
    gain = (torch.rand((ns, nf), generator=g, device=dev) < on).float() * (torch.randint(1, 9, (ns, nf), generator=g, device=dev) * 0.09375).float()
    tap = torch.randint(0, 3, (ns, nf), generator=g, device=dev, dtype=torch.int32)
out = torch.empty((ns, nf * n, ch), device=dev)
pcm = torch.empty((ns * ch, nf * n), device=dev)
work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)


state = torch.zeros((ns * ch, 60), device=dev)
WITH_STATE = os.environ.get("CHAIN_STATE") == "1"


def run():
    ctx.celt_chain_dev(3, freq.data_ptr(), trans.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0,
                       state.data_ptr() if WITH_STATE else 0, 0, 0, out.data_ptr(), pcm.data_ptr(), work.data_ptr(), ns, nf, ch)


def timed(fn, reps=10):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps):
        fn()
    b.record(stream)
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / reps


if os.environ.get("CHAIN_SPLIT") == "1":
    # the two stages timed alone (loops of 10 of the same launch) next to the chain timed as a unit
    t_s = timed(lambda: ctx.celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), state.data_ptr() if WITH_STATE else 0,
                                           work.data_ptr(), ns, nf, ch))
    t_p = timed(lambda: ctx.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch))
    t_c = timed(run)
    st2 = torch.zeros_like(state)

    def two_calls():
        ctx.celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), state.data_ptr() if WITH_STATE else 0, work.data_ptr(), ns, nf, ch)
        ctx.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch)
    t_2 = timed(two_calls)
    tiny = torch.zeros(64, device=dev)

    def with_tiny(where):
        def f():
            if where == "before_synth":
                tiny.add_(1.0)
            ctx.celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), state.data_ptr() if WITH_STATE else 0, work.data_ptr(), ns, nf, ch)
            if where == "between":
                tiny.add_(1.0)
            ctx.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch)
        return f
    extra = {w: timed(with_tiny(w)) for w in ("before_synth", "between")}
    ctx.set_option(B.OPT_POST_FORM, B.POST_FORM_WAVE_PER_CHANNEL)
    extra["two_calls_with_round1_post_kernel"] = timed(two_calls)
    extra["round1_post_kernel_alone"] = timed(lambda: ctx.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch))
    ctx.set_option(B.OPT_POST_FORM, B.POST_FORM_PIPELINE)
    print(json.dumps(extra))
    state.zero_()
    t_c0 = timed(run)
    print(json.dumps({"synth_alone_ms": t_s, "post_alone_ms": t_p, "sum_ms": t_s + t_p, "chain_as_a_unit_ms": t_c, "two_api_calls_as_a_unit_ms": t_2,
                      "chain_again_after_zeroing_state_ms": t_c0, "state_abs_max": float(state.abs().max()), "with_state": WITH_STATE}))
    sys.exit(0)


if os.environ.get("CHAIN_WINDOWS") is not None:
    # entries: W (sequential windows) or "Wo" (overlapped: post-filter of window k beside the synthesis of window k + 1)
    wins = os.environ["CHAIN_WINDOWS"].split(",")
    times = {w: [] for w in wins}
    outs = {}
    for rnd in range(int(os.environ.get("CHAIN_ROUNDS", "9"))):
        for w in wins:
            ctx.set_option(B.OPT_CHAIN_WINDOW, int(w.rstrip("o")))
            ctx.set_option(B.OPT_CHAIN_OVERLAP, 1 if w.endswith("o") else 0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            run()
            b.record(stream)
            torch.cuda.synchronize(dev)
            if rnd >= 2:
                times[w].append(a.elapsed_time(b))
            if rnd == 0:
                outs[w] = out.clone()
    res = {"case": f"{ns} streams x {nf} frames x 2 ch, LM 3, post-filter case {case}, product library"}
    for w in wins:
        ms = sorted(times[w])[len(times[w]) // 2]
        res[f"window_{w}"] = {"ms": ms, "min_ms": min(times[w]), "stereo_frames_per_sec": ns * nf / ms * 1e3,
                              "GBps_freq_in_plus_pcm_out": ns * nf * ch * 7680 / ms / 1e6,
                              "bit_identical_to_first": bool(torch.equal(outs[w], outs[wins[0]]))}
    print(json.dumps(res))
    sys.exit(0)

res = {}
outs = {}
times = {"0": [], "1": []}
for rnd in range(int(os.environ.get("CHAIN_ROUNDS", "12"))):
    for mode in ("0", "1"):
        ctx.set_option(B.OPT_CHAIN_FUSED, int(mode))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        run()
        b.record(stream)
        torch.cuda.synchronize(dev)
        if rnd >= 2:
            times[mode].append(a.elapsed_time(b))
        if rnd == 0:
            outs[mode] = out.clone()
for mode, name in (("0", "two kernels"), ("1", "one launch")):
    ms = sorted(times[mode])[len(times[mode]) // 2]
    res[name] = {"ms": ms, "stereo_frames_per_sec": ns * nf / ms * 1e3, "GBps_in_plus_out": ns * nf * ch * 7680 / ms / 1e6}
res["max_abs_diff_one_launch_vs_two_kernels"] = float((outs["0"] - outs["1"]).abs().max())
res["case"] = f"{ns} streams x {nf} frames x 2 ch, LM 3, 2.8 % transient, post-filter case {case}"
print(json.dumps(res))
