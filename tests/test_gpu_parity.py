"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the oracle, the committed
golden fixtures and size-independent properties.  Tolerances: north_star's 1e-5 RMS on the
bundled unit-scale IFFT vectors (absolute) and 1e-5 RELATIVE RMS on IMDCT outputs
(decoder-scale inputs have rms ~30); measured errors are ~2e-7."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, abs_rms, rel_rms

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    import libnyquist_amd as nyq
    c = nyq.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_ref_tables(ref_tables):
    """context carrying the reference's own static tables (what the drop-in shims upload)"""
    import libnyquist_amd as nyq
    c = nyq.Context(0)
    c.set_tables(ref_tables["trig"], ref_tables["window"])
    yield c
    c.close()


def test_native_library_is_the_one_loaded(ctx):
    import libnyquist_amd as nyq
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(nyq.LIB_PATH) in maps
    cus, name = ctx.device_info()
    assert cus > 0 and "gfx950" in name


def test_builtin_tables_match_reference(ctx, ref_tables):
    t, w = ctx.get_tables()
    assert np.abs(t - ref_tables["trig"]).max() <= 6e-8
    assert np.abs(w - ref_tables["window"]).max() <= 6e-8


# ---- config C2: bundled IFFT vectors ------------------------------------------------
@pytest.mark.parametrize("nfft,reps", [(60, 65536), (480, 4096)])
def test_bundled_ifft_vectors_replicated(ctx, nfft, reps):
    x = np.fromfile(os.path.join(GOLDEN, f"ifft_input_N{nfft}.bin"), np.float32)
    want = np.fromfile(os.path.join(GOLDEN, f"ifft_output_N{nfft}.bin"), np.float32)
    y = ctx.ifft_batch(nfft, np.tile(x, (reps, 1)))
    assert y.shape == (reps, 2 * nfft)
    err = np.sqrt(np.mean((y.astype(np.float64) - want[None, :]) ** 2, axis=1))
    assert err.max() <= TOL                      # every row within north_star's 1e-5 RMS
    assert err.max() <= 2e-7                      # and in fact at float rounding level
    assert np.array_equal(y[0], y[-1]) and np.array_equal(y[0], y[reps // 2 + 3])   # batch-position independent


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_ifft_vs_reference_fixture_and_oracle(ctx, oracle, shift):
    z = np.load(os.path.join(GOLDEN, "ref_ifft_shared.npz"))
    x, want = z[f"x{shift}"], z[f"y{shift}"]
    nfft = 480 >> shift
    assert rel_rms(ctx.ifft_batch(nfft, x), want) <= TOL
    rng = np.random.default_rng(shift)
    xr = rng.uniform(-1, 1, (1001, 2 * nfft)).astype(np.float32)    # ragged: not a multiple of 4
    assert rel_rms(ctx.ifft_batch(nfft, xr), oracle.ifft_batch(nfft, xr, shared=True, nthreads=4)) <= 1e-6


# ---- full IMDCT -------------------------------------------------------------------------
@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_imdct_vs_reference_fixtures(ctx_ref_tables, shift):
    z = np.load(os.path.join(GOLDEN, f"ref_imdct_s{shift}.npz"))
    x, carry, want = z["x"], z["carry"], z["out"]
    n2 = 960 >> shift
    fin, tail = ctx_ref_tables.imdct_batch(shift, x, carry)
    assert rel_rms(fin, want[:, :n2]) <= TOL
    assert rel_rms(tail, want[:, n2:]) <= TOL
    for r in range(x.shape[0]):                     # row by row, incl. silence and impulses
        ref = want[r]
        got = np.concatenate([fin[r], tail[r]])
        scale = max(np.sqrt(np.mean(ref.astype(np.float64) ** 2)), 1e-3)
        assert abs_rms(got, ref) <= TOL * scale, r
    assert rel_rms(fin, want[:, :n2]) <= 1e-6       # measured level


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
@pytest.mark.parametrize("rows", [1, 3, 4, 5, 63, 1000, 4099])
def test_imdct_vs_oracle_ragged(ctx, oracle, shift, rows):
    n2 = 960 >> shift
    rng = np.random.default_rng(1000 * shift + rows)
    x = (rng.standard_normal((rows, n2)) * 30).astype(np.float32)
    carry = (rng.standard_normal((rows, 60)) * 30).astype(np.float32)
    ctx.set_tables(*oracle.tables()[:2])
    for cy in (carry, None):
        fin, tail = ctx.imdct_batch(shift, x, cy)
        wf, wt = oracle.imdct_batch(shift, x, cy, nthreads=4)
        assert rel_rms(fin, wf) <= 1e-6
        assert rel_rms(tail, wt) <= 1e-6
    fin2, none = ctx.imdct_batch(shift, x, carry, want_tail=False)
    assert none is None and np.array_equal(fin2, ctx.imdct_batch(shift, x, carry)[0])


@pytest.mark.parametrize("pinned", [False, True])
def test_host_path_pipelined_pieces_vs_oracle(ctx, oracle, pinned):
    """The host-buffer entry points cut big batches into 32 MB pieces that flow through three streams
    (upload / kernel / download): several pieces with a ragged last one, pageable and pinned buffers."""
    import libnyquist_amd as nyq
    rng = np.random.default_rng(77)
    ctx.set_tables(*oracle.tables()[:2])
    rows = 20011                                            # 3 pieces of 8740 rows at nfft 480
    x = (nyq.pinned_empty if pinned else np.empty)((rows, 960), np.float32)
    x[:] = rng.standard_normal((rows, 960)) * 30
    carry = (rng.standard_normal((rows, 60)) * 30).astype(np.float32)
    fin, tail = ctx.imdct_batch(0, x, carry, pinned=pinned)
    wf, wt = oracle.imdct_batch(0, x, carry, nthreads=8)
    assert rel_rms(fin, wf) <= 1e-6 and rel_rms(tail, wt) <= 1e-6
    for lo in (0, 8739, 8740, 17480, rows - 1):            # piece edges, row by row
        assert rel_rms(fin[lo], wf[lo]) <= 1e-6 and rel_rms(tail[lo], wt[lo]) <= 1e-6
    nchains, length = 301, 40                               # 46 MB: two pieces of whole chains
    xc = (rng.standard_normal((nchains * length, 960)) * 30).astype(np.float32)
    c0 = (rng.standard_normal((nchains, 60)) * 30).astype(np.float32)
    pcm, tails = ctx.imdct_chain(0, xc, c0, nchains=nchains, pinned=pinned)
    for c in (0, 1, 227, 228, 229, nchains - 1):
        wp, wt1 = oracle.imdct_chain(0, xc[c * length:(c + 1) * length], c0[c])
        assert rel_rms(pcm[c * length:(c + 1) * length], wp) <= 1e-6
        assert rel_rms(tails[c], wt1) <= 1e-6
    small, _ = ctx.imdct_batch(3, x[:5, :120].copy(), None)   # and a batch far below one piece
    assert rel_rms(small, oracle.imdct_batch(3, x[:5, :120].copy(), None)[0]) <= 1e-6


def test_empty_batch_and_bad_arguments(ctx):
    import libnyquist_amd as nyq
    fin, tail = ctx.imdct_batch(0, np.zeros((0, 960), np.float32))
    assert fin.shape == (0, 960) and tail.shape == (0, 60)
    assert ctx.ifft_batch(60, np.zeros((0, 120), np.float32)).shape == (0, 120)
    with pytest.raises(nyq.NyqError) as e:
        ctx._ck(ctx.lib.nyq_imdct_batch(ctx.h, 4, None, None, None, None, 1))
    assert e.value.code == -1
    with pytest.raises(nyq.NyqError):
        ctx._ck(ctx.lib.nyq_ifft_batch(ctx.h, 64, None, None, 1))
    with pytest.raises(nyq.NyqError):
        ctx._ck(ctx.lib.nyq_imdct_batch(ctx.h, 0, None, None, None, None, 1))     # NULL buffers


@pytest.mark.parametrize("shift", [0, 3])
def test_chain_vs_oracle(ctx, oracle, shift):
    n2 = 960 >> shift
    rng = np.random.default_rng(shift + 50)
    nchains, length = 5, 37
    x = (rng.standard_normal((nchains * length, n2)) * 30).astype(np.float32)
    c0 = (rng.standard_normal((nchains, 60)) * 30).astype(np.float32)
    ctx.set_tables(*oracle.tables()[:2])
    pcm, tails = ctx.imdct_chain(shift, x, c0, nchains=nchains)
    for c in range(nchains):
        wp, wt = oracle.imdct_chain(shift, x[c * length:(c + 1) * length], c0[c])
        assert rel_rms(pcm[c * length:(c + 1) * length], wp) <= 1e-6
        assert rel_rms(tails[c], wt) <= 1e-6
    pcm0, _ = ctx.imdct_chain(shift, x, None, nchains=nchains)      # NULL seed = zeros
    wp0, _ = oracle.imdct_chain(shift, x[:length], None)
    assert rel_rms(pcm0[:length], wp0) <= 1e-6


def test_chain_vs_reference_mixed_fixture(ctx_ref_tables):
    """long/short mixed channel (ref_chain.npz): each homogeneous run is one chain call."""
    z = np.load(os.path.join(GOLDEN, "ref_chain.npz"))
    kinds, freq = "".join(z["kinds"]), z["freq"]
    carry = z["carry_in"].copy()
    out = []
    for f, k in enumerate(kinds):
        if k == "L":
            p, t = ctx_ref_tables.imdct_chain(0, freq[f][None, :], carry[None, :])
        else:
            p, t = ctx_ref_tables.imdct_chain(3, freq[f].reshape(120, 8).T.copy(), carry[None, :])
        carry = t[0]
        out.append(p.reshape(-1))
    assert rel_rms(np.concatenate(out), z["pcm"]) <= 1e-6
    assert rel_rms(carry, z["tail"]) <= 1e-6


# ---- the reference's operator names -----------------------------------------------------
def test_dropin_shims_vs_reference_strided_fixture(ref_tables):
    import libnyquist_amd as nyq
    lib = nyq.load()
    z = np.load(os.path.join(GOLDEN, "ref_imdct_strided.npz"))
    trig = np.ascontiguousarray(ref_tables["trig"])
    win = np.ascontiguousarray(ref_tables["window"])
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    fp2 = C.c_void_p * 2
    X, c0, syn = z["X"], z["carry0"], z["syn"]
    sine3 = np.float32(2) * np.float32(3.141592653) * np.float32(.125) / np.float32(240)
    for f in range(X.shape[0]):
        mem = np.zeros((2, 960 + 60), np.float32)
        mem[:, :60] = c0[f]
        xin = np.ascontiguousarray(X[f])
        for b in range(8):      # compute_inv_mdcts, celt_decoder_clean.c:292-300
            ins = fp2(xin[0, b:].ctypes.data, xin[1, b:].ctypes.data)
            outs = fp2(mem[0, 120 * b:].ctypes.data, mem[1, 120 * b:].ctypes.data)
            lib.processMDCTCudaB1C2(ins, outs, P(trig), 240, 3, 8, sine3, 120, P(win))
        assert rel_rms(mem, syn[f]) <= 1e-6
    XL, cl, synl = z["XL"], z["carryL"], z["synL"]
    sine0 = np.float32(2) * np.float32(3.141592653) * np.float32(.125) / np.float32(1920)
    for f in range(XL.shape[0]):
        for c in range(2):
            mem = np.zeros(960 + 60, np.float32)
            mem[:60] = cl[f, c]
            lib.processMDCTCuda(P(np.ascontiguousarray(XL[f, c])), P(mem), P(trig), 1920, 0, 1, sine0, 120, P(win))
            assert rel_rms(mem, synl[f, c]) <= 1e-6
    lib.printCudaVersion()
    lib.cleanupCudaBuffers()


def test_dropin_b8c2_is_eight_rows_and_the_handler_may_call_back(ref_tables, oracle):
    """processMDCTCudaB8C2 (cuda/mdct_cuda.hpp:96-98: declared with eight row pointers, never called by the reference): eight
    clt_mdct_backward rows per call, equal to the oracle's.  And the error handler runs with no library lock held: a handler
    that calls cleanupCudaBuffers() and another operator (what an integrator would do on a failure) returns."""
    import libnyquist_amd as nyq
    lib = nyq.load()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    trig = np.ascontiguousarray(ref_tables["trig"])
    win = np.ascontiguousarray(ref_tables["window"])
    rng = np.random.default_rng(88)
    sine1 = np.float32(2) * np.float32(3.141592653) * np.float32(.125) / np.float32(960)
    x = (rng.standard_normal((8, 480)) * 30).astype(np.float32)
    mem = np.zeros((8, 480 + 60), np.float32)
    mem[:, :60] = (rng.standard_normal((8, 60)) * 30).astype(np.float32)
    want_f, want_t = oracle.imdct_batch(1, x, mem[:, :60].copy())
    fp8 = C.c_void_p * 8
    lib.processMDCTCudaB8C2(fp8(*[x[r].ctypes.data for r in range(8)]), fp8(*[mem[r].ctypes.data for r in range(8)]), P(trig), 960, 1, 1,
                            sine1, 120, P(win))
    assert rel_rms(mem, np.concatenate([want_f, want_t], axis=1)) <= 1e-6
    seen = []
    HANDLER = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p)

    def on_error(who, what):
        seen.append((who, what))
        lib.cleanupCudaBuffers()                                   # would deadlock if the shim's mutex were still held
        m2 = np.zeros(480 + 60, np.float32)
        lib.processMDCTCuda(P(x[0]), P(m2), P(trig), 960, 1, 1, sine1, 120, P(win))
        seen.append(rel_rms(m2[:480], want_f[0] * 0 + oracle.imdct_batch(1, x[:1], None)[0][0]))

    cb = HANDLER(on_error)
    lib.nyq_shim_set_error_handler.argtypes = [HANDLER]
    lib.nyq_shim_set_error_handler(cb)
    try:
        before = mem.copy()
        lib.processMDCTCuda(P(x[0]), P(mem[0]), P(trig), 1000, 1, 1, sine1, 120, P(win))   # N does not belong to the mode
        assert seen and seen[0][0] == b"processMDCTCuda" and np.array_equal(mem, before)
        assert seen[1] <= 1e-6
    finally:
        lib.nyq_shim_set_error_handler(C.cast(None, HANDLER))
    lib.cleanupCudaBuffers()


def test_dropin_shims_tables_by_content_and_threads(ref_tables, oracle):
    """The reference-named entry points keep process-wide state (one context, the uploaded tables).  (1) Tables are
    identified by CONTENT: a caller that refills the SAME buffers with another window gets results for the new window
    (pointer identity would serve stale tables).  (2) Calls from several threads at once are serialised by the library:
    every thread gets the result of its own input."""
    import threading
    import libnyquist_amd as nyq
    lib = nyq.load()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    trig = np.ascontiguousarray(ref_tables["trig"]).copy()
    win = np.ascontiguousarray(ref_tables["window"]).copy()
    sine0 = np.float32(2) * np.float32(3.141592653) * np.float32(.125) / np.float32(1920)
    rng = np.random.default_rng(8)
    x = (rng.standard_normal(960) * 30).astype(np.float32)
    carry = (rng.standard_normal(60) * 30).astype(np.float32)

    def call(xx):
        mem = np.zeros(960 + 60, np.float32)
        mem[:60] = carry
        lib.processMDCTCuda(P(xx), P(mem), P(trig), 1920, 0, 1, sine0, 120, P(win))
        return mem

    a = call(x)
    win[:] = win[::-1].copy()                                  # same buffer, other content
    b = call(x)
    want_b = np.zeros(960 + 60, np.float32)
    want_b[:60] = carry
    from oracle.pyoracle import Oracle
    orig = tuple(np.array(t, copy=True) for t in oracle.tables())
    try:                                                       # (the C restatement's tables are process-wide: put them back)
        Oracle((trig, win, orig[2])).imdct(x, want_b, 0)
    finally:
        Oracle(orig)
    assert rel_rms(b, want_b) <= 1e-6 and rel_rms(a[:120], b[:120]) > 1e-3   # (only the mirrored head depends on the window)
    win[:] = ref_tables["window"]
    xs = [(rng.standard_normal(960) * 30).astype(np.float32) for _ in range(6)]
    want = [call(xx) for xx in xs]
    bad = []

    def worker(k):
        for _ in range(40):
            if not np.array_equal(call(xs[k]), want[k]):
                bad.append(k)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad
    lib.cleanupCudaBuffers()


# ---- device-resident API at BASELINE.json's full size: properties ------------------------
def test_full_size_device_resident_properties(ctx, oracle):
    """2^20 rows of nfft-480 (config C3) through nyq_imdct_batch_dev: a sampled parity check,
    linearity, and batch-position independence -- properties that do not need a full-size oracle."""
    import torch
    dev = torch.device("cuda", 0)
    rows = 1 << 20
    ctx.set_tables(*oracle.tables()[:2])
    s = torch.cuda.current_stream(dev)
    ctx.set_stream(s.cuda_stream)
    try:
        g = torch.Generator(device=dev)
        g.manual_seed(480)
        a = torch.rand((rows, 960), generator=g, device=dev) * 2 - 1
        b = torch.rand((rows, 960), generator=g, device=dev) * 2 - 1
        fa, ta = torch.empty_like(a), torch.empty((rows, 60), device=dev)
        fb, tb = torch.empty_like(a), torch.empty((rows, 60), device=dev)
        fs, ts = torch.empty_like(a), torch.empty((rows, 60), device=dev)
        ctx.imdct_batch_dev(0, a.data_ptr(), 0, fa.data_ptr(), ta.data_ptr(), rows)
        ctx.imdct_batch_dev(0, b.data_ptr(), 0, fb.data_ptr(), tb.data_ptr(), rows)
        ab = a + b
        ctx.imdct_batch_dev(0, ab.data_ptr(), 0, fs.data_ptr(), ts.data_ptr(), rows)
        torch.cuda.synchronize(dev)
        # linearity over the whole batch
        num = torch.sqrt(torch.mean((fs - (fa + fb)).double() ** 2))
        den = torch.sqrt(torch.mean(fs.double() ** 2))
        assert float(num / den) <= 2e-6
        assert float(torch.sqrt(torch.mean((ts - (ta + tb)).double() ** 2)) / torch.sqrt(torch.mean(ts.double() ** 2))) <= 2e-6
        # sampled rows against the oracle (first, last, strided)
        idx = torch.cat([torch.arange(0, 1024), torch.arange(rows - 1024, rows), torch.arange(0, rows, 509)[:2048]]).to(dev)
        xs = a[idx].cpu().numpy()
        wf, wt = oracle.imdct_batch(0, xs, None, nthreads=4)
        assert rel_rms(fa[idx].cpu().numpy(), wf) <= 1e-6
        assert rel_rms(ta[idx].cpu().numpy(), wt) <= 1e-6
        # same row anywhere in the batch gives the same bits
        a2 = a.clone()
        a2[rows - 7] = a[3]
        ctx.imdct_batch_dev(0, a2.data_ptr(), 0, fb.data_ptr(), tb.data_ptr(), rows)
        torch.cuda.synchronize(dev)
        assert torch.equal(fb[rows - 7], fa[3]) and torch.equal(tb[rows - 7], ta[3])
        # misaligned device pointer is rejected, not launched
        import libnyquist_amd as nyq
        with pytest.raises(nyq.NyqError):
            ctx.imdct_batch_dev(0, a.data_ptr() + 4, 0, fa.data_ptr(), ta.data_ptr(), 16)
    finally:
        ctx.reset_stream()


def test_chain_dev_matches_host_chain(ctx, oracle):
    import torch
    dev = torch.device("cuda", 0)
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(9)
    nchains, length = 64, 50
    x = (rng.standard_normal((nchains * length, 960)) * 30).astype(np.float32)
    d_x = torch.from_numpy(x).to(dev)
    d_pcm = torch.empty_like(d_x)
    d_tail = torch.empty((nchains, 60), device=dev)
    d_work = torch.empty((nchains, length + 1, 60), device=dev)
    torch.cuda.synchronize(dev)      # ctx runs on its own stream here: order it after torch's copies
    ctx.imdct_chain_dev(0, d_x.data_ptr(), 0, d_pcm.data_ptr(), d_tail.data_ptr(), d_work.data_ptr(), nchains, length)
    ctx.synchronize()
    wp, wt = oracle.imdct_chain(0, x[:length], None)
    assert rel_rms(d_pcm[:length].cpu().numpy(), wp) <= 1e-6
    wp, wt = oracle.imdct_chain(0, x[-length:], None)
    assert rel_rms(d_pcm[-length:].cpu().numpy(), wp) <= 1e-6
    assert rel_rms(d_tail[-1].cpu().numpy(), wt) <= 1e-6


# ---- frame sequences: the compute_inv_mdcts replacement ---------------------------------------
def test_frame_synth_vs_reference_fixture(ctx_ref_tables):
    z = np.load(os.path.join(GOLDEN, "ref_synth.npz"))
    pcm, st = ctx_ref_tables.celt_synth(3, z["freq"], z["transient"], z["state_in"], channels=2)
    assert rel_rms(pcm, z["pcm"]) <= TOL and rel_rms(pcm, z["pcm"]) <= 1e-6
    assert rel_rms(st, z["state_out"]) <= 1e-6
    c = np.load(os.path.join(GOLDEN, "ref_chain.npz"))
    tr = np.array([[k == "S" for k in "".join(c["kinds"])]], np.uint8)
    pcm, st = ctx_ref_tables.celt_synth(3, c["freq"][None, :, None, :], tr, c["carry_in"][None, :], channels=1)
    assert rel_rms(pcm.reshape(-1), c["pcm"]) <= 1e-6 and rel_rms(st[0], c["tail"]) <= 1e-6


@pytest.mark.parametrize("lm", [3, 2, 1, 0])
def test_frame_synth_vs_oracle(ctx, oracle, lm):
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(40 + lm)
    n = 120 << lm
    for ns, nf, ch, ptr in ((1, 1, 1, 0.0), (3, 50, 2, 0.1), (2, 33, 8, 0.5), (5, 4, 2, 1.0), (64, 21, 2, 0.03), (2, 7, 1, 0.0)):
        freq = (rng.standard_normal((ns, nf, ch, n)) * 30).astype(np.float32)
        tr = (rng.uniform(size=(ns, nf)) < ptr).astype(np.uint8)
        st = (rng.standard_normal((ns * ch, 60)) * 30).astype(np.float32)
        pcm, so = ctx.celt_synth(lm, freq, tr, st, channels=ch)
        wp, ws = oracle.celt_synth(lm, freq, tr, st, nthreads=4)
        assert rel_rms(pcm, wp) <= 1e-6, (ns, nf, ch, ptr)
        assert rel_rms(so, ws) <= 1e-6
        pcm0, none = ctx.celt_synth(lm, freq, None, None, channels=ch)      # no transients, zero state
        wp0, _ = oracle.celt_synth(lm, freq, None, None, nthreads=4)
        assert none is None and rel_rms(pcm0, wp0) <= 1e-6


def test_frame_synth_matches_sb_reverie_mix_at_scale(ctx, oracle):
    """config C4 shape: many stereo streams with the measured sb-reverie.opus frame mix (2.8 %
    transient), device resident; checked on sampled streams + against the chain operator."""
    import torch
    dev = torch.device("cuda", 0)
    ctx.set_tables(*oracle.tables()[:2])
    ns, nf, ch = 256, 200, 2
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    freq = torch.randn((ns, nf, ch, 960), generator=g, device=dev) * 30
    tr = (torch.rand((ns, nf), generator=g, device=dev) < 0.028).to(torch.uint8)
    pcm = torch.empty((ns, ch, nf * 960), device=dev)
    state = torch.zeros((ns * ch, 60), device=dev)
    work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    torch.cuda.synchronize(dev)
    ctx.celt_synth_dev(3, freq.data_ptr(), tr.data_ptr(), pcm.data_ptr(), state.data_ptr(), work.data_ptr(), ns, nf, ch)
    ctx.synchronize()
    for s in (0, 17, 255):
        wp, ws = oracle.celt_synth(3, freq[s:s + 1].cpu().numpy(), tr[s:s + 1].cpu().numpy(), np.zeros((ch, 60), np.float32), nthreads=2)
        assert rel_rms(pcm[s].cpu().numpy(), wp[0]) <= 1e-6
        assert rel_rms(state[s * ch:(s + 1) * ch].cpu().numpy(), ws) <= 1e-6


def test_frame_synth_on_real_decoder_frames(ctx_ref_tables):
    """Real freq[] of test_data/short.opus frames 64..127 (captured from the reference decoder):
    the GPU path must reproduce the decoder's own out_syn."""
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    pcm, st = ctx_ref_tables.celt_synth(3, z["freq"], z["transient"], z["state_in"], channels=2)
    assert rel_rms(pcm, z["pcm"]) <= TOL and rel_rms(pcm, z["pcm"]) <= 1e-6
    assert rel_rms(st, z["state_out"]) <= 1e-6
    # per-frame check, transient frames included
    want = z["pcm"].reshape(2, -1, 960)
    got = pcm.reshape(2, -1, 960)
    for f in range(want.shape[1]):
        assert rel_rms(got[:, f], want[:, f]) <= TOL, f


def test_config4_thousand_streams_sb_reverie_pattern(ctx, oracle):
    """BASELINE config 4 shape: 1000 concurrent stereo streams following the REAL transient map of
    test_data/sb-reverie.opus (first 64 frames of the map per stream, streams start at different
    offsets), synthetic decoder-scale freq[]; sampled streams checked against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    ctx.set_tables(*oracle.tables()[:2])
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    tmap = z["sb_reverie_transient"]
    ns, nf, ch = 1000, 64, 2
    offs = (np.arange(ns) * 11) % (len(tmap) - nf)
    tr_np = np.stack([tmap[o:o + nf] for o in offs]).astype(np.uint8)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    freq = torch.randn((ns, nf, ch, 960), generator=g, device=dev) * 30
    tr = torch.from_numpy(tr_np).to(dev)
    pcm = torch.empty((ns, ch, nf * 960), device=dev)
    work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    torch.cuda.synchronize(dev)
    ctx.celt_synth_dev(3, freq.data_ptr(), tr.data_ptr(), pcm.data_ptr(), 0, work.data_ptr(), ns, nf, ch)
    ctx.synchronize()
    hit = [int(i) for i in np.nonzero(tr_np.sum(1))[0][:3]] + [0, ns - 1]
    for s in hit:
        wp, _ = oracle.celt_synth(3, freq[s:s + 1].cpu().numpy(), tr_np[s:s + 1], None, nthreads=2)
        assert rel_rms(pcm[s].cpu().numpy(), wp[0]) <= 1e-6, s


# ---- post-filter + de-emphasis + interleave (out_syn -> AudioData samples) ----------------------
def _post_on_gpu(ctx, lm, pcm, pf_pitch, pf_gain, pf_tapset, pf_state, hist, deemph, channels):
    import torch
    dev = torch.device("cuda", 0)
    T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    ns = pf_pitch.shape[0]
    nf = pf_pitch.shape[1]
    n = 120 << lm
    d_pcm = T(pcm, np.float32)
    d_pp, d_pg, d_pt = T(pf_pitch, np.int32), T(pf_gain, np.float32), T(pf_tapset, np.int32)
    d_si = T(pf_state, np.float32)
    d_so = torch.zeros_like(d_si)
    d_h, d_m = T(hist, np.float32), T(deemph, np.float32)
    d_out = torch.empty((ns, nf * n, channels), device=dev)
    torch.cuda.synchronize(dev)
    ctx.celt_post_dev(lm, d_pcm.data_ptr(), d_pp.data_ptr(), d_pg.data_ptr(), d_pt.data_ptr(), d_si.data_ptr(),
                      d_so.data_ptr(), d_h.data_ptr(), d_m.data_ptr(), d_out.data_ptr(), ns, nf, channels)
    ctx.synchronize()
    return d_out.cpu().numpy(), d_so.cpu().numpy(), d_h.cpu().numpy(), d_m.cpu().numpy()


def test_post_chain_on_real_decoder_frames(ctx_ref_tables):
    """freq[] -> nyq_celt_synth -> nyq_celt_post_dev must give the reference decoder's FINAL PCM
    (AudioData::samples) for frames 64..127 of test_data/short.opus."""
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    pcm, _ = ctx_ref_tables.celt_synth(3, z["freq"], z["transient"], z["state_in"], channels=2)
    out, pst, hist, dm = _post_on_gpu(ctx_ref_tables, 3, pcm, z["pf_pitch"], z["pf_gain"], z["pf_tapset"],
                                      z["pf_state_in"], z["hist_in"], z["deemph_in"], 2)
    assert rel_rms(out, z["final"]) <= TOL
    assert rel_rms(out, z["final"]) <= 2e-6          # measured level
    assert np.abs(out - z["final"]).max() <= 2e-6     # samples are in [-1, 1)
    assert np.array_equal(pst, z["pf_state_out"])
    assert rel_rms(dm, z["deemph_out"]) <= 1e-5


@pytest.mark.parametrize("lm", [3, 2, 1, 0])
def test_post_vs_oracle_random_parameters(ctx, oracle, lm):
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(70 + lm)
    n = 120 << lm
    for ns, nf, ch in ((1, 3, 1), (3, 17, 2), (2, 9, 6)):
        pcm = (rng.standard_normal((ns, ch, nf * n)) * 300).astype(np.float32)
        hist = (rng.standard_normal((ns, ch, 1088)) * 300).astype(np.float32)
        pitch = rng.integers(15, 1023, (ns, nf)).astype(np.int32)
        pitch[:, ::5] = rng.integers(15, 24, pitch[:, ::5].shape)          # short periods: narrow wave steps
        gain = (rng.integers(0, 9, (ns, nf)) * 0.09375).astype(np.float32)  # (qg+1)*3/32 or 0 = off
        gain[:, 1::4] = 0
        taps = rng.integers(0, 3, (ns, nf)).astype(np.int32)
        st = np.stack([[rng.integers(15, 1023), rng.integers(15, 1023), 0.28125, 0.375, 1, 2] for _ in range(ns)]).astype(np.float32)
        dm = (rng.standard_normal(ns * ch) * 100).astype(np.float32)
        buf = np.concatenate([hist, pcm], axis=2)
        want, filt, wst, wdm = oracle.celt_post(lm, buf, 1088, pitch, gain, taps, st, dm)
        out, pst, gh, gdm = _post_on_gpu(ctx, lm, pcm, pitch, gain, taps, st, hist.reshape(ns * ch, 1088), dm, ch)
        assert rel_rms(out, want) <= 1e-5, (ns, nf, ch)
        assert np.array_equal(pst, wst)
        assert rel_rms(gh, filt[:, :, -1088:].reshape(ns * ch, 1088)) <= 1e-5
        assert rel_rms(gdm, wdm) <= 1e-5


def test_device_entry_points_capture_into_a_hip_graph(ctx, oracle):
    """The `_dev` entry points neither allocate nor synchronise, so a decode step (frame synthesis +
    post-filter + a row batch) can be captured once into a hipGraph and replayed on new data: the replay
    must equal the eager result bit for bit."""
    import torch
    dev = torch.device("cuda", 0)
    ctx.set_tables(*oracle.tables()[:2])
    ns, nf, ch, n = 24, 40, 2, 960
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    freq = torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30
    trans = (torch.rand((ns, nf), generator=g, device=dev) < 0.1).to(torch.uint8)
    pitch = torch.randint(15, 1000, (ns, nf), generator=g, device=dev, dtype=torch.int32)
    gain = (torch.randint(0, 9, (ns, nf), generator=g, device=dev) * 0.09375).float()
    tap = torch.randint(0, 3, (ns, nf), generator=g, device=dev, dtype=torch.int32)
    pcm = torch.empty((ns, ch, nf * n), device=dev)
    out = torch.empty((ns, nf * n, ch), device=dev)
    state = torch.zeros((ns * ch, 60), device=dev)
    work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    rows = torch.randn((1000, 240), generator=g, device=dev)
    fin, tail = torch.empty_like(rows), torch.empty((1000, 60), device=dev)

    def step():
        ctx.celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), state.data_ptr(), work.data_ptr(), ns, nf, ch)
        ctx.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch)
        ctx.imdct_batch_dev(2, rows.data_ptr(), 0, fin.data_ptr(), tail.data_ptr(), rows.shape[0])

    side = torch.cuda.Stream(dev)
    ctx.set_stream(side.cuda_stream)
    try:
        with torch.cuda.stream(side):
            step()                                          # warm: occupancy queries are cached per context
            side.synchronize()
            state.zero_()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                step()
            # new data in the same buffers, replay, then the same step eagerly
            freq.copy_(torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30)
            rows.copy_(torch.randn((1000, 240), generator=g, device=dev))
            state.zero_()
            graph.replay()
            side.synchronize()
            got = (out.clone(), fin.clone(), tail.clone(), state.clone())
            state.zero_()
            out.zero_()
            fin.zero_()
            step()
            side.synchronize()
        assert torch.equal(got[0], out) and torch.equal(got[1], fin) and torch.equal(got[2], tail) and torch.equal(got[3], state)
        assert float(out.abs().max()) > 0
    finally:
        ctx.reset_stream()


def test_frames_to_pcm_pipelined_pieces_equal_single_calls(ctx, oracle):
    """nyq_celt_frames_to_pcm cuts a big batch into pieces of whole streams (upload / kernels / download on
    three streams).  Streams are independent, so any stream's PCM and decoder state must equal, bit for bit,
    what a call on that stream alone returns -- across piece borders too."""
    import libnyquist_amd as nyq
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(5)
    ns, nf, ch, n = 500, 20, 2, 960                 # 77 MB of freq: three pieces of 219 streams
    freq = (rng.standard_normal((ns, nf, ch, n)) * 30).astype(np.float32)
    tr = (rng.random((ns, nf)) < 0.1).astype(np.uint8)
    pp = rng.integers(15, 1000, (ns, nf)).astype(np.int32)
    pg = (rng.integers(0, 9, (ns, nf)) * 0.09375).astype(np.float32)
    pt = rng.integers(0, 3, (ns, nf)).astype(np.int32)
    nst = int(ctx.lib.nyq_celt_state_floats(ns, ch))
    state = np.zeros(nst, np.float32)
    out = ctx.celt_frames_to_pcm(3, freq, tr, pp, pg, pt, ch, state=state)
    assert np.isfinite(out).all() and np.abs(out).max() > 0
    nsc = ns * ch
    # (layout of nyq_celt_state_floats: overlap | history | de-emphasis | post-filter parameters | the entropy stage's state, unused here)
    ov, hi, de, pf, _ = np.split(state, [nsc * 60, nsc * 60 + nsc * 1088, nsc * 60 + nsc * 1088 + nsc, nsc * 60 + nsc * 1088 + nsc + ns * 6])
    for k in (0, 1, 218, 219, 220, 437, 438, ns - 1):
        st1 = np.zeros(int(ctx.lib.nyq_celt_state_floats(1, ch)), np.float32)
        o1 = ctx.celt_frames_to_pcm(3, freq[k:k + 1], tr[k:k + 1], pp[k:k + 1], pg[k:k + 1], pt[k:k + 1], ch, state=st1)
        assert np.array_equal(o1[0], out[k])
        o_ov, o_hi, o_de, o_pf, _ = np.split(st1, [ch * 60, ch * 60 + ch * 1088, ch * 60 + ch * 1088 + ch, ch * 60 + ch * 1088 + ch + 6])
        assert np.array_equal(o_ov, ov[k * ch * 60:(k + 1) * ch * 60])
        assert np.array_equal(o_hi, hi[k * ch * 1088:(k + 1) * ch * 1088])
        assert np.array_equal(o_de, de[k * ch:(k + 1) * ch])
        assert np.array_equal(o_pf, pf[k * 6:(k + 1) * 6])
    # continuing from the returned state == decoding the concatenation in one go (stream 219, first of piece 2)
    k = 219
    both = ctx.celt_frames_to_pcm(3, np.concatenate([freq[k:k + 1], freq[k + 1:k + 2]], axis=1), np.concatenate([tr[k:k + 1], tr[k + 1:k + 2]], axis=1),
                                  np.concatenate([pp[k:k + 1], pp[k + 1:k + 2]], axis=1), np.concatenate([pg[k:k + 1], pg[k + 1:k + 2]], axis=1),
                                  np.concatenate([pt[k:k + 1], pt[k + 1:k + 2]], axis=1), ch)
    st1 = np.zeros(int(ctx.lib.nyq_celt_state_floats(1, ch)), np.float32)
    ctx.celt_frames_to_pcm(3, freq[k:k + 1], tr[k:k + 1], pp[k:k + 1], pg[k:k + 1], pt[k:k + 1], ch, state=st1)
    second = ctx.celt_frames_to_pcm(3, freq[k + 1:k + 2], tr[k + 1:k + 2], pp[k + 1:k + 2], pg[k + 1:k + 2], pt[k + 1:k + 2], ch, state=st1)
    assert rel_rms(second[0], both[0, nf * n:]) <= 1e-6


def test_frames_to_pcm_window_slices_equal_one_call(ctx, oracle):
    """nyq_celt_frames_to_pcm_window on consecutive time slices of per-stream arrays, decoder state carried from
    slice to slice, reproduces one call over the whole length: bit for bit (PCM and final state) when the slice
    boundaries fall on multiples of 64 frames -- the in-wave carry chains of the synthesis kernels restart at the
    same frames either way -- and to within an ulp of the TDAC mirror for any other slicing.  This is how the
    batch decoder walks long streams in bounded memory."""
    import ctypes as C
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(11)
    ns, nf, ch, n = 20, 200, 2, 960
    freq = (rng.standard_normal((ns, nf, ch, n)) * 30).astype(np.float32)
    tr = (rng.random((ns, nf)) < 0.1).astype(np.uint8)
    pp = rng.integers(15, 1000, (ns, nf)).astype(np.int32)
    pg = (rng.integers(0, 9, (ns, nf)) * 0.09375).astype(np.float32)
    pt = rng.integers(0, 3, (ns, nf)).astype(np.int32)
    nst = int(ctx.lib.nyq_celt_state_floats(ns, ch))
    st_all = np.zeros(nst, np.float32)
    want = ctx.celt_frames_to_pcm(3, freq, tr, pp, pg, pt, ch, state=st_all)
    p = lambda a, off: C.c_void_p(a.ctypes.data + off)

    def sliced(lengths):
        out = np.zeros((ns, nf * n, ch), np.float32)
        st = np.zeros(nst, np.float32)
        f0 = 0
        for length in lengths:
            rc = ctx.lib.nyq_celt_frames_to_pcm_window(ctx.h, 3, p(freq, f0 * ch * n * 4), p(tr, f0), p(pp, f0 * 4), p(pg, f0 * 4),
                                                       p(pt, f0 * 4), p(out, f0 * n * ch * 4), p(st, 0), ns, length, ch, nf)
            assert rc == 0
            f0 += length
        assert f0 == nf
        return out, st

    out, st = sliced((64, 128, 8))
    assert np.array_equal(out, want) and np.array_equal(st, st_all)
    out, st = sliced((32, 1, 40, 27, 100))                # ragged: chains restart elsewhere
    assert np.abs(out - want).max() <= 2e-6 and rel_rms(out, want) <= 1e-6
    assert rel_rms(st, st_all) <= 1e-5
    with pytest.raises(Exception):
        ctx._ck(ctx.lib.nyq_celt_frames_to_pcm_window(ctx.h, 3, p(freq, 0), p(tr, 0), p(pp, 0), p(pg, 0), p(pt, 0), p(out, 0), p(st, 0),
                                                      ns, 64, ch, 32))


@pytest.fixture(scope="module")
def ctx_ab():
    """context of the tools' A/B build (tools/libnyq_imdct_ab.so): the product's kernels plus the round-1 post-filter forms"""
    import libnyquist_amd as nyq
    c = nyq.Context(0, ab=True)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_shapes_synth_then_post_vs_oracle(ctx, ctx_ab, oracle, seed):
    """Randomised shapes through both GPU stages against the oracle: frame size, channel count, stream and frame
    counts (around the kernels' group / chain / slice boundaries), transient density, incoming state, post-filter
    parameters with short periods and switched-off frames, both stereo post-filter modes."""
    ctx.set_tables(*oracle.tables()[:2])
    rng = np.random.default_rng(9000 + seed)
    lm = int(rng.integers(0, 4))
    n = 120 << lm
    ch = int(rng.choice([1, 2, 2, 2, 3, 5]))
    ns = int(rng.choice([1, 2, 5, 17, 33]))
    nf = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 70]))
    ptr = float(rng.choice([0.0, 0.03, 0.3, 1.0]))
    freq = (rng.standard_normal((ns, nf, ch, n)) * 30).astype(np.float32)
    tr = (rng.uniform(size=(ns, nf)) < ptr).astype(np.uint8)
    st = (rng.standard_normal((ns * ch, 60)) * 30).astype(np.float32) if seed % 2 else None
    pcm, so = ctx.celt_synth(lm, freq, tr, st, channels=ch)
    wp, ws = oracle.celt_synth(lm, freq, tr, st, nthreads=4)
    assert rel_rms(pcm, wp) <= 1e-6, (lm, ch, ns, nf, ptr)
    if st is not None:
        assert rel_rms(so, ws) <= 1e-6
    hist = (rng.standard_normal((ns, ch, 1088)) * 30).astype(np.float32)
    pitch = rng.integers(15, 1023, (ns, nf)).astype(np.int32)
    short = rng.uniform(size=(ns, nf)) < 0.5
    pitch[short] = rng.integers(15, 70, int(short.sum()))
    gain = (rng.integers(0, 9, (ns, nf)) * 0.09375).astype(np.float32)
    gain[rng.uniform(size=(ns, nf)) < 0.3] = 0
    taps = rng.integers(0, 3, (ns, nf)).astype(np.int32)
    pst = np.stack([[rng.integers(15, 1023), rng.integers(15, 1023), 0.28125, 0.375, 1, 2] for _ in range(ns)]).astype(np.float32)
    dm = (rng.standard_normal(ns * ch) * 10).astype(np.float32)
    want, filt, wst, wdm = oracle.celt_post(lm, np.concatenate([hist, wp.reshape(ns, ch, nf * n)], axis=2), 1088, pitch, gain, taps, pst, dm)
    import libnyquist_amd as nyq
    B = nyq.binding
    ctx_ab.set_tables(*oracle.tables()[:2])
    # the product's one form (workgroup pipeline) on the product library, then the A/B build's three forms
    forms = [(ctx, B.POST_FORM_PIPELINE), (ctx_ab, B.POST_FORM_PIPELINE), (ctx_ab, B.POST_FORM_WAVE_PER_CHANNEL)]
    if ch == 2:
        forms.append((ctx_ab, B.POST_FORM_WAVE_PER_PAIR))
    for cx, mode in forms:
        cx.set_option(B.OPT_POST_FORM, mode)
        try:
            out, gst, gh, gdm = _post_on_gpu(cx, lm, wp.reshape(ns, ch, nf * n), pitch, gain, taps, pst, hist.reshape(ns * ch, 1088), dm, ch)
        finally:
            cx.set_option(B.OPT_POST_FORM, B.POST_FORM_PIPELINE)
        assert rel_rms(out, want) <= 1e-5, (lm, ch, ns, nf, mode)
        assert np.array_equal(gst, wst)
        assert rel_rms(gh, filt[:, :, -1088:].reshape(ns * ch, 1088)) <= 1e-5
        assert rel_rms(gdm, wdm) <= 1e-5
