"""GPU tier: nyq_celt_chain_dev -- freq[] -> interleaved PCM as ONE launch (LM 3, stereo: nyq_chain_kernel.hpp) against the
oracle (compute_inv_mdcts + comb_filter + deemphasis restated in oracle/nyq_oracle.c) and against the two-kernel chain it
replaces, decoder state carried in and out; round 2's fused kernel (A/B build) against the same."""
import os

import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import libnyquist_amd as nyq
    c = nyq.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_ab():
    """a context of the tools' A/B build (tools/libnyq_imdct_ab.so, -DNYQ_AB_FORMS): the only build that has round 2's fused
    chain kernel; the product library answers NYQ_ERR_INVALID to NYQ_OPT_CHAIN_FUSED = NYQ_CHAIN_FUSED_R2"""
    import libnyquist_amd as nyq
    c = nyq.Context(0, ab=True)
    yield c
    c.close()


TWO_KERNELS, ONE_LAUNCH, FUSED_R2 = 0, 1, 2      # NYQ_OPT_CHAIN_FUSED values (include/nyq_imdct.h)


def _run_chain(ctx, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form, window=0, overlap=0):
    import libnyquist_amd as nyq
    import torch
    dev = torch.device("cuda", 0)
    T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    ns, nf = pitch.shape
    n = 120 << lm
    d_freq, d_tr = T(freq, np.float32), T(tr, np.uint8)
    d_pp, d_pg, d_pt = T(pitch, np.int32), T(gain, np.float32), T(taps, np.int32)
    d_si = T(pst, np.float32)
    d_so = torch.zeros_like(d_si)
    d_ov, d_h, d_m = T(ov, np.float32), T(hist, np.float32), T(dm, np.float32)
    d_out = torch.zeros((ns, nf * n, ch), device=dev)
    d_pcm = torch.empty((ns * ch, nf * n), device=dev)
    d_work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    torch.cuda.synchronize(dev)
    ctx.set_option(nyq.binding.OPT_CHAIN_FUSED, form)
    ctx.set_option(nyq.binding.OPT_CHAIN_WINDOW, window)
    ctx.set_option(nyq.binding.OPT_CHAIN_OVERLAP, overlap)
    try:
        ctx.celt_chain_dev(lm, d_freq.data_ptr(), d_tr.data_ptr(), d_pp.data_ptr(), d_pg.data_ptr(), d_pt.data_ptr(), d_si.data_ptr(),
                           d_so.data_ptr(), d_ov.data_ptr(), d_h.data_ptr(), d_m.data_ptr(), d_out.data_ptr(), d_pcm.data_ptr(),
                           d_work.data_ptr(), ns, nf, ch)
        ctx.synchronize()
    finally:
        ctx.set_option(nyq.binding.OPT_CHAIN_FUSED, ONE_LAUNCH)          # the default
        ctx.set_option(nyq.binding.OPT_CHAIN_WINDOW, 0)
        ctx.set_option(nyq.binding.OPT_CHAIN_OVERLAP, 0)
    return d_out.cpu().numpy(), d_so.cpu().numpy(), d_ov.cpu().numpy(), d_h.cpu().numpy(), d_m.cpu().numpy()


def _case(rng, ns, nf, ptr, ch=2, lm=3):
    n = 120 << lm
    freq = (rng.standard_normal((ns, nf, ch, n)) * 30).astype(np.float32)
    tr = (rng.uniform(size=(ns, nf)) < ptr).astype(np.uint8)
    ov = (rng.standard_normal((ns * ch, 60)) * 30).astype(np.float32)
    hist = (rng.standard_normal((ns * ch, 1088)) * 30).astype(np.float32)
    pitch = rng.integers(15, 1023, (ns, nf)).astype(np.int32)
    short = rng.uniform(size=(ns, nf)) < 0.6
    pitch[short] = rng.integers(15, 70, int(short.sum()))
    gain = (rng.integers(0, 9, (ns, nf)) * 0.09375).astype(np.float32)
    gain[rng.uniform(size=(ns, nf)) < 0.3] = 0
    taps = rng.integers(0, 3, (ns, nf)).astype(np.int32)
    pst = np.stack([[rng.integers(15, 1023), rng.integers(15, 1023), 0.28125, 0.375, 1, 2] for _ in range(ns)]).astype(np.float32)
    dm = (rng.standard_normal(ns * ch) * 10).astype(np.float32)
    return freq, tr, pitch, gain, taps, pst, ov, hist, dm


@pytest.mark.parametrize("ns,nf,ptr", [(1, 1, 0.0), (1, 2, 1.0), (2, 3, 0.5), (3, 17, 0.1), (5, 40, 0.03), (8, 33, 1.0), (33, 20, 0.3),
                                       (2, 130, 0.05), (300, 6, 0.2)])
def test_one_launch_chain_vs_oracle_and_two_kernel_chain(ctx, ctx_ab, oracle, ns, nf, ptr):
    """The one-launch kernel (the product's default for 20 ms stereo frames) against the oracle (synthesis, then post-filter
    + de-emphasis) and against the two-kernel chain (another factorisation of the transform and another order of the
    de-emphasis sums: equal to rounding); round 2's fused kernel (A/B build) bit for bit against the two kernels."""
    ctx.set_tables(*oracle.tables()[:2])
    ctx_ab.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(4200 + ns * 100 + nf)
    freq, tr, pitch, gain, taps, pst, ov, hist, dm = _case(rng, ns, nf, ptr)
    ch, n = 2, 960
    wp, ws = oracle.celt_synth(3, freq, tr, ov, nthreads=4)
    want, filt, wst, wdm = oracle.celt_post(3, np.concatenate([hist.reshape(ns, ch, 1088), wp.reshape(ns, ch, nf * n)], axis=2), 1088,
                                            pitch, gain, taps, pst, dm)
    out, gst, gov, gh, gdm = _run_chain(ctx, 3, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=ONE_LAUNCH)
    assert rel_rms(out, want) <= 1e-5, (ns, nf, ptr)
    assert np.array_equal(gst, wst)
    assert rel_rms(gov, ws) <= 1e-6
    assert rel_rms(gh, filt[:, :, -1088:].reshape(ns * ch, 1088)) <= 1e-5
    assert rel_rms(gdm, wdm) <= 1e-5
    out2, gst2, gov2, gh2, gdm2 = _run_chain(ctx, 3, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=TWO_KERNELS)
    assert rel_rms(out, out2) <= 3e-6 and np.abs(out - out2).max() <= 1e-5 * np.abs(out2).max()
    assert np.array_equal(gst, gst2)
    assert rel_rms(gov, gov2) <= 1e-6 and rel_rms(gh, gh2) <= 3e-6
    if ns <= 33:
        out3, gst3, gov3, gh3, gdm3 = _run_chain(ctx_ab, 3, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=FUSED_R2)
        assert rel_rms(out3, out2) <= 1e-6 and np.abs(out3 - out2).max() <= 2e-6
        assert np.array_equal(gst3, gst2)


@pytest.mark.parametrize("form,lm,h", [(ONE_LAUNCH, 3, 11), (ONE_LAUNCH, 3, 1), (ONE_LAUNCH, 3, 16), (FUSED_R2, 3, 11), (TWO_KERNELS, 3, 11),
                                       (TWO_KERNELS, 3, 1), (TWO_KERNELS, 2, 7), (TWO_KERNELS, 1, 5), (TWO_KERNELS, 0, 3)])
def test_chain_continues_from_its_own_state(ctx, ctx_ab, oracle, form, lm, h):
    """Two calls with the state of the first handed to the second == one call over both halves -- the one-launch kernel,
    round 2's fused kernel and the two-kernel chain (whose post-filter keeps 1040 samples of history in LDS and assembles
    the 1088 of the hand-over from both of its buffers), every frame size, a first call as short as one frame."""
    ctx = ctx_ab if form == FUSED_R2 else ctx
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(77 + lm)
    ns, nf = 6, 24
    freq, tr, pitch, gain, taps, pst, ov, hist, dm = _case(rng, ns, nf, 0.1, lm=lm)
    whole = _run_chain(ctx, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, 2, form=form)
    a = _run_chain(ctx, lm, freq[:, :h], tr[:, :h], pitch[:, :h], gain[:, :h], taps[:, :h], pst, ov, hist, dm, 2, form=form)
    b = _run_chain(ctx, lm, freq[:, h:], tr[:, h:], pitch[:, h:], gain[:, h:], taps[:, h:], a[1], a[2], a[3], a[4], 2, form=form)
    got = np.concatenate([a[0], b[0]], axis=1)
    if form != TWO_KERNELS or h % 16 == 0:
        assert np.array_equal(got, whole[0])
        for x, y in zip(b[1:], whole[1:]):
            assert np.array_equal(x, y)
    else:
        # the synthesis stage chains heads in-wave inside 16-frame chunks and fixes the others up afterwards: a cut that
        # is not a multiple of the chunk length moves a few heads from one form to the other (same value, other rounding)
        assert rel_rms(got, whole[0]) <= 1e-6
        assert np.array_equal(b[1], whole[1])
        for x, y in zip(b[2:], whole[2:]):
            assert rel_rms(x, y) <= 1e-6


def test_other_shapes_run_the_two_kernel_chain(ctx, oracle):
    """Frame sizes / channel counts the one-launch kernel does not cover go through synth + post inside the same entry."""
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(5)
    for lm, ch in ((2, 2), (3, 1), (0, 3)):
        ns, nf, n = 3, 9, 120 << lm
        freq, tr, pitch, gain, taps, pst, ov, hist, dm = _case(rng, ns, nf, 0.2, ch=ch, lm=lm)
        wp, ws = oracle.celt_synth(lm, freq, tr, ov, nthreads=2)
        want, filt, wst, wdm = oracle.celt_post(lm, np.concatenate([hist.reshape(ns, ch, 1088), wp.reshape(ns, ch, nf * n)], axis=2), 1088,
                                                pitch, gain, taps, pst, dm)
        out, gst, gov, gh, gdm = _run_chain(ctx, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=ONE_LAUNCH)
        assert rel_rms(out, want) <= 1e-5, (lm, ch)
        assert np.array_equal(gst, wst)


def test_product_build_refuses_the_ab_forms(ctx):
    """The measured-and-rejected kernel forms are not in the product library: asking for them is an error, not a silent
    fallback, and nothing is chosen through the process environment any more."""
    import libnyquist_amd as nyq
    assert ctx.lib.nyq_ab_forms_built() == 0
    with pytest.raises(nyq.NyqError):
        ctx.set_option(nyq.binding.OPT_CHAIN_FUSED, FUSED_R2)
    with pytest.raises(nyq.NyqError):
        ctx.set_option(nyq.binding.OPT_POST_FORM, nyq.binding.POST_FORM_WAVE_PER_CHANNEL)
    with pytest.raises(nyq.NyqError):
        ctx.set_option(nyq.binding.OPT_CHAIN_OVERLAP, 1)
    ctx.set_option(nyq.binding.OPT_POST_FORM, nyq.binding.POST_FORM_PIPELINE)
    assert ctx.get_option(nyq.binding.OPT_CHAIN_FUSED) == ONE_LAUNCH


@pytest.mark.parametrize("lm,ch,ns,nf,window,with_state", [(3, 2, 5, 200, 64, True), (3, 2, 3, 131, 64, False), (2, 1, 4, 300, 128, True),
                                                          (1, 2, 3, 257, 64, False), (0, 3, 2, 330, 64, True), (3, 2, 2, 128, 64, True)])
def test_windowed_chain_is_bit_identical_to_one_window(ctx, ctx_ab, oracle, lm, ch, ns, nf, window, with_state):
    """nyq_celt_chain_dev over time windows (NYQ_OPT_CHAIN_WINDOW: synthesis and post-filter alternate over windows of 64 k
    frames, the time-domain frames of a window living in the first part of d_pcm only): the samples and every piece of
    decoder state equal the one-window call bit for bit -- with the caller's state buffers and, where the caller passes
    none (fresh decoder), through the temporaries carved from d_work -- and the oracle within 1e-5."""
    import libnyquist_amd as nyq
    import torch
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(31 * lm + nf)
    n = 120 << lm
    freq, tr, pitch, gain, taps, pst, ov, hist, dm = _case(rng, ns, nf, 0.05, ch=ch, lm=lm)
    if with_state:
        one = _run_chain(ctx, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=TWO_KERNELS, window=0)
        win = _run_chain(ctx, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=TWO_KERNELS, window=window)
        for a, b in zip(one, win):
            assert np.array_equal(a, b)
        # ... and with the post-filter of window k on a second stream beside the synthesis of window k + 1 (A/B build only)
        ctx_ab.set_tables(*oracle.tables()[:2])
        lap = _run_chain(ctx_ab, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=TWO_KERNELS, window=window, overlap=1)
        for a, b in zip(one, lap):
            assert np.array_equal(a, b)
        wp, ws = oracle.celt_synth(lm, freq, tr, ov, nthreads=4)
        want, filt, wst, wdm = oracle.celt_post(lm, np.concatenate([hist.reshape(ns, ch, 1088), wp.reshape(ns, ch, nf * n)], axis=2), 1088,
                                                pitch, gain, taps, pst, dm)
        assert rel_rms(win[0], want) <= 1e-5 and np.array_equal(win[1], wst)
        return
    dev = torch.device("cuda", 0)
    T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    d_freq, d_tr = T(freq, np.float32), T(tr, np.uint8)
    d_pp, d_pg, d_pt = T(pitch, np.int32), T(gain, np.float32), T(taps, np.int32)
    outs = []
    for w in (0, window):
        d_out = torch.zeros((ns, nf * n, ch), device=dev)
        d_pcm = torch.empty((ns * ch, nf * n), device=dev)
        d_work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
        torch.cuda.synchronize(dev)
        ctx.set_option(nyq.binding.OPT_CHAIN_WINDOW, w)
        ctx.set_option(nyq.binding.OPT_CHAIN_FUSED, TWO_KERNELS)
        try:
            ctx.celt_chain_dev(lm, d_freq.data_ptr(), d_tr.data_ptr(), d_pp.data_ptr(), d_pg.data_ptr(), d_pt.data_ptr(), 0, 0, 0, 0, 0,
                               d_out.data_ptr(), d_pcm.data_ptr(), d_work.data_ptr(), ns, nf, ch)
            ctx.synchronize()
        finally:
            ctx.set_option(nyq.binding.OPT_CHAIN_WINDOW, 0)
            ctx.set_option(nyq.binding.OPT_CHAIN_FUSED, ONE_LAUNCH)
        outs.append(d_out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])


class _OutDesc(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("base", _C.c_void_p), ("first", _C.c_longlong), ("last", _C.c_longlong), ("t0", _C.c_longlong), ("cstride", _C.c_int),
                ("coff", _C.c_int * 2), ("gain", _C.c_float)]


@pytest.mark.parametrize("lm,ch,form", [(3, 2, ONE_LAUNCH), (3, 2, TWO_KERNELS), (2, 2, ONE_LAUNCH), (3, 1, ONE_LAUNCH), (0, 1, ONE_LAUNCH)])
def test_mapped_output_equals_the_dense_output_rearranged(ctx, oracle, lm, ch, form):
    """Row f3: with an output record per elementary stream (nyq_out_desc) the kernels' store phase writes sample ts of
    channel k to base[(ts - first) * cstride + coff[k]] * gain for first <= ts < last and nowhere else -- the channel mapping
    of opus_multistream_decoder.c:305-331, the pre-skip / end trim and the header gain without a pass on the host.  Checked
    bit for bit against the dense output of the same call, for the one-launch kernel and the post-filter pipeline (stereo
    and mono instances), with a stream that has no record (dense), a channel that is not written (-1) and a time-slice
    offset t0."""
    import ctypes as C
    import libnyquist_amd as nyq
    import torch
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(900 + 10 * lm + ch)
    ns, nf, n = 5, 9, 120 << lm
    freq, tr, pitch, gain, taps, pst, ov, hist, dm = _case(rng, ns, nf, 0.2, ch=ch, lm=lm)
    dense = _run_chain(ctx, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=form)[0]      # [ns][nf*n][ch]
    dev = torch.device("cuda", 0)
    T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    cstride, total = 5, nf * n
    files = torch.full((ns, total + 64, cstride), -7.0, device=dev)
    desc = (_OutDesc * ns)()
    want = np.full((ns, total + 64, cstride), -7.0, np.float32)
    for s in range(ns):
        t0 = 1000 * s                                            # the call is a slice that starts at stream sample t0
        first, last = t0 + 37 + s, t0 + total - 11 * s           # (first not a multiple of 2 or 4: unaligned windows)
        g = np.float32(1.0 if s == 0 else 0.5 + 0.1 * s)
        c0, c1 = (3, 1) if s % 2 == 0 else (0, -1)               # (odd streams: the second channel is dropped)
        if s == 3:
            desc[s].base = None                                  # no record: this stream goes to the dense output
            continue
        desc[s].base = files[s].data_ptr()
        desc[s].first, desc[s].last, desc[s].t0, desc[s].cstride, desc[s].gain = first, last, t0, cstride, float(g)
        desc[s].coff[0], desc[s].coff[1] = c0, (c1 if ch == 2 else -1)
        w = dense[s, first - t0:last - t0]
        want[s, :last - first, c0] = w[:, 0] * g
        if ch == 2 and c1 >= 0:
            want[s, :last - first, c1] = w[:, 1] * g
    d_desc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
    d_freq, d_tr = T(freq, np.float32), T(tr, np.uint8)
    d_pp, d_pg, d_pt = T(pitch, np.int32), T(gain, np.float32), T(taps, np.int32)
    d_si = T(pst, np.float32)
    d_so = torch.zeros_like(d_si)
    d_ov, d_h, d_m = T(ov, np.float32), T(hist, np.float32), T(dm, np.float32)
    d_out = torch.full((ns, nf * n, ch), 3.0, device=dev)
    d_pcm = torch.empty((ns * ch, nf * n), device=dev)
    d_work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    torch.cuda.synchronize(dev)
    V = lambda t: C.c_void_p(t.data_ptr())
    ctx.set_option(nyq.binding.OPT_CHAIN_FUSED, form)
    try:
        ctx._ck(ctx.lib.nyq_celt_chain_mapped_dev(ctx.h, lm, V(d_freq), V(d_tr), V(d_pp), V(d_pg), V(d_pt), V(d_si), V(d_so), V(d_ov), V(d_h),
                                                  V(d_m), V(d_out), V(d_desc), V(d_pcm), V(d_work), ns, nf, ch))
        ctx.synchronize()
    finally:
        ctx.set_option(nyq.binding.OPT_CHAIN_FUSED, ONE_LAUNCH)
    got = files.cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(d_out[3].cpu().numpy(), dense[3])      # the stream without a record, dense as ever


@pytest.mark.parametrize("form", [ONE_LAUNCH, TWO_KERNELS])
def test_unchanged_tap_set_shortcut_on_sb_reverie_parameters(ctx, oracle, form):
    """From a stream's second frame on, the first 120 samples of every 20 ms frame are filtered with old == current tap
    set (celt_decoder_clean.c:678-683), where the kernels run the constant filter in place of the reference's cross-fade
    (nyq_post_pipe.hpp, pipe_comb_call: the weights (1-f) g + f g add up to g).  Pinned here on the REAL parameter sequence
    of sb-reverie.opus (tests/golden/sb_reverie_pf_params.npz: 7 windows of 160 frames, 76 % of the frames filtered,
    periods 15 .. 1022) against the oracle, which cross-fades like the reference: an explicit bound on the largest
    sample difference, not only on the RMS."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sb_reverie_pf_params.npz"))
    ns, nf, ch, n = 7, 160, 2, 960
    idx = ((np.arange(ns) * 1571) % (len(z["pf_pitch"]) - nf))[:, None] + np.arange(nf)[None, :]
    pitch = z["pf_pitch"][idx].astype(np.int32)
    gain = z["pf_gain_q"][idx].astype(np.float32) * np.float32(0.09375)
    taps = z["pf_tapset"][idx].astype(np.int32)
    tr = z["transient"][idx].astype(np.uint8)
    same = (pitch[:, 1:] == pitch[:, :-1]) & (gain[:, 1:] == gain[:, :-1]) & (taps[:, 1:] == taps[:, :-1]) & (gain[:, 1:] != 0)
    assert (gain != 0).mean() > 0.5 and same.sum() > 50         # (plenty of frames on either branch of the shortcut)
    rng = np.random.default_rng(777)
    freq = (rng.standard_normal((ns, nf, ch, n)) * 30).astype(np.float32)
    ov = (rng.standard_normal((ns * ch, 60)) * 30).astype(np.float32)
    hist = (rng.standard_normal((ns * ch, 1088)) * 30).astype(np.float32)
    pst = np.stack([[pitch[s, 0], pitch[s, 0], gain[s, 0], gain[s, 0], taps[s, 0], taps[s, 0]] for s in range(ns)]).astype(np.float32)
    dm = (rng.standard_normal(ns * ch) * 10).astype(np.float32)
    ctx.set_tables(*oracle.tables()[:2])
    wp, ws = oracle.celt_synth(3, freq, tr, ov, nthreads=4)
    want, filt, wst, wdm = oracle.celt_post(3, np.concatenate([hist.reshape(ns, ch, 1088), wp.reshape(ns, ch, nf * n)], axis=2), 1088,
                                            pitch, gain, taps, pst, dm)
    out, gst, gov, gh, gdm = _run_chain(ctx, 3, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=form)
    peak = float(np.abs(want).max())
    worst = float(np.abs(out - want).max())
    print(f"form {form}: largest sample difference {worst:.3e} at a peak of {peak:.3e} ({worst / peak:.2e} of it), rel rms {rel_rms(out, want):.2e}")
    assert worst <= 1e-6 * peak                                  # (measured: 1.9e-7 of the peak, both forms)
    assert rel_rms(out, want) <= 2e-6
    assert np.array_equal(gst, wst)


@pytest.mark.parametrize("lm,ch,ns,nf,with_state,window", [(3, 2, 3, 700, True, 128), (3, 2, 2, 333, False, 64), (3, 1, 2, 450, True, 192),
                                                            (1, 2, 3, 900, True, 256), (3, 2, 16, 520, True, 0)])
def test_host_call_on_few_long_streams_runs_in_time_windows(ctx, oracle, lm, ch, ns, nf, with_state, window):
    """nyq_celt_frames_to_pcm on a few LONG streams cuts the call in time windows (upload / kernels / download of
    consecutive windows at once, states carried on the device: nyq_imdct.hip frames_to_pcm_core): bit for bit what one
    device-resident launch over the whole length gives, decoder state out included.  NYQ_OPT_HOST_WINDOW forces the window
    length on cases small enough for the test tier (3 .. 6 windows, the last one ragged); window 0 = the built-in choice on
    a case large enough for it to cut (16 streams x 520 frames = 64 MB of freq[]: three windows of 192)."""
    import libnyquist_amd as nyq
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(900 + lm * 10 + ch)
    freq, tr, pitch, gain, taps, pst, ov, hist, dm = _case(rng, ns, nf, 0.05, ch=ch, lm=lm)
    if not with_state:
        pst, ov, hist, dm = np.zeros_like(pst), np.zeros_like(ov), np.zeros_like(hist), np.zeros_like(dm)
    want, wst, wov, wh, wdm = _run_chain(ctx, lm, freq, tr, pitch, gain, taps, pst, ov, hist, dm, ch, form=ONE_LAUNCH)
    state = np.concatenate([ov.ravel(), hist.ravel(), dm.ravel(), pst.ravel()]).astype(np.float32) if with_state else None
    ctx.set_option(nyq.binding.OPT_HOST_WINDOW, window)
    try:
        got = ctx.celt_frames_to_pcm(lm, freq, tr, pitch, gain, taps, ch, state=state)
    finally:
        ctx.set_option(nyq.binding.OPT_HOST_WINDOW, 0)
    assert np.array_equal(got, want)
    if with_state:
        nsc = ns * ch
        assert np.array_equal(state[:nsc * 60].reshape(nsc, 60), wov)
        assert np.array_equal(state[nsc * 60:nsc * 1148].reshape(nsc, 1088), wh)
        assert np.array_equal(state[nsc * 1148:nsc * 1149], wdm)
        assert np.array_equal(state[nsc * 1149:].reshape(ns, 6), wst)


@pytest.mark.parametrize("lm,ch", [(3, 2), (3, 1), (2, 2)])
def test_host_call_in_time_windows_with_output_records(ctx, oracle, lm, ch):
    """The windowed host call with an output record per stream (what a multistream file of long streams runs through): every
    window's launch gets the records with t0 moved to the window's first sample.  The files' device buffers must equal, bit
    for bit, what the unwindowed call writes."""
    import ctypes as C
    import libnyquist_amd as nyq
    import torch
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(77 + lm + ch)
    ns, nf, n = 3, 300, 120 << lm
    freq, tr, pitch, gain, taps, pst, ov, hist, dm = _case(rng, ns, nf, 0.05, ch=ch, lm=lm)
    dev = torch.device("cuda", 0)
    total = nf * n
    results = []
    for window in (0, 128):
        files = torch.full((ns, total, 3), -7.0, device=dev)
        desc = (_OutDesc * ns)()
        for s in range(ns):
            desc[s].base = files[s].data_ptr()
            desc[s].first, desc[s].last, desc[s].t0 = 100 + 7 * s, total - 50 * s, 0
            desc[s].cstride, desc[s].gain = 3, 0.5 + 0.25 * s
            desc[s].coff[0], desc[s].coff[1] = 2, (0 if ch == 2 else -1)
        state = np.concatenate([ov.ravel(), hist.ravel(), dm.ravel(), pst.ravel()]).astype(np.float32)
        torch.cuda.synchronize(dev)
        ctx.set_option(nyq.binding.OPT_HOST_WINDOW, window)
        try:
            P = lambda a: a.ctypes.data_as(C.c_void_p)
            rc = ctx.lib.nyq_celt_frames_to_pcm_mapped(ctx.h, lm, P(freq), P(tr), P(pitch), P(gain), P(taps), None, C.cast(desc, C.c_void_p),
                                                       P(state), ns, nf, ch, nf)
        finally:
            ctx.set_option(nyq.binding.OPT_HOST_WINDOW, 0)
        assert rc == 0, ctx.lib.nyq_last_error(ctx.h)
        ctx.synchronize()
        results.append((files.cpu().numpy(), state))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    assert (results[0][0][:, :, 1] == -7.0).all() and (results[0][0][0, :total - 100, 2] != -7.0).all()
