"""CPU tests: pin the oracle (oracle/nyq_oracle.c) before anything trusts it.

Pins, in order of authority:
  1. the reference's own bundled golden vectors  test_data/ifft_{input,output}_N{60,480}.bin
     (copied as data to tests/golden/) -- IFFT stage, tolerance 1e-5 abs RMS (north_star);
  2. outputs of the reference itself (oracle/_ref, fixtures by oracle/gen_golden.py) -- bit-exact;
  3. independent float64 known answers (direct DFT, closed-form IMDCT of SURVEY.md section 3.2).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, abs_rms, rel_rms
from oracle.pyoracle import HALF_OV, Oracle, n2_of


def _bin(name):
    return np.fromfile(os.path.join(GOLDEN, name), dtype=np.float32)


@pytest.mark.parametrize("nfft", [60, 480])
def test_bundled_ifft_vectors(oracle, nfft):
    """kiss_fft.c:749-771 test_opus_ifft shape: own twiddles, unscaled inverse."""
    x = _bin(f"ifft_input_N{nfft}.bin")
    want = _bin(f"ifft_output_N{nfft}.bin")
    assert x.size == 2 * nfft and want.size == 2 * nfft
    got = oracle.ifft_own(nfft, x)
    assert abs_rms(got, want) <= 1e-5
    # measured against the reference build in the survey: 5.4e-9 / 2.9e-8
    assert abs_rms(got, want) <= 1e-7


@pytest.mark.parametrize("nfft", [60, 480])
def test_bundled_vectors_are_unscaled_inverse_dft(nfft):
    """Known answer independent of any FFT code: y[n] = sum_k x[k] e^{+2 pi i k n / nfft}."""
    x = _bin(f"ifft_input_N{nfft}.bin").astype(np.float64).view(np.complex128)
    want = _bin(f"ifft_output_N{nfft}.bin").astype(np.float64).view(np.complex128)
    k = np.arange(nfft)
    y = np.exp(2j * np.pi * np.outer(k, k) / nfft) @ x
    assert np.sqrt(np.mean(np.abs(y - want) ** 2)) <= 1e-6


def test_tables_match_reference(oracle, ref_tables):
    """Generated tables vs the reference's static tables (static_modes_float.h): <= 1 ulp."""
    o = Oracle()  # default (formula) tables -- shares the library's global state
    try:
        t, w, tw = o.tables()
        assert np.abs(t - ref_tables["trig"]).max() <= 6e-8
        assert np.abs(w - ref_tables["window"]).max() <= 6e-8
        assert np.abs(tw - ref_tables["tw"]).max() <= 6e-8
        assert (t != ref_tables["trig"]).sum() <= 30
    finally:
        Oracle((ref_tables["trig"], ref_tables["window"], ref_tables["tw"]))  # re-pin


def test_plans_match_reference(oracle, ref_tables):
    """Digit-reversal tables and factor lists (static_modes_float.h:343-471)."""
    off = 0
    for s in range(4):
        n = 480 >> s
        perm, radix, rest = oracle.plan(s)
        assert np.array_equal(perm, ref_tables["bitrev"][off:off + n])
        fac = ref_tables["factors"][s]
        assert list(fac[: 2 * len(radix): 2]) == list(radix)
        assert list(fac[1: 2 * len(radix): 2]) == list(rest)
        off += n


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_imdct_bit_exact_vs_reference(oracle, shift):
    z = np.load(os.path.join(GOLDEN, f"ref_imdct_s{shift}.npz"))
    x, carry, want = z["x"], z["carry"], z["out"]
    n2 = n2_of(shift)
    for r in range(x.shape[0]):
        buf = np.zeros(n2 + HALF_OV, np.float32)
        buf[:HALF_OV] = carry[r]
        oracle.imdct(x[r], buf, shift, 1)
        assert np.array_equal(buf, want[r]), f"row {r}"
    fin, tail = oracle.imdct_batch(shift, x, carry, nthreads=2)
    assert np.array_equal(fin, want[:, :n2])
    assert np.array_equal(tail, want[:, n2:])


def test_imdct_strided_bit_exact_vs_reference(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_imdct_strided.npz"))
    X, c0, syn = z["X"], z["carry0"], z["syn"]
    for f in range(X.shape[0]):
        for c in range(2):
            mem = np.zeros(960 + HALF_OV, np.float32)
            mem[:HALF_OV] = c0[f, c]
            for b in range(8):
                oracle.imdct(X[f, c, b:], mem[120 * b: 120 * b + 180], 3, 8)
            assert np.array_equal(mem, syn[f, c])
    XL, cl, synl = z["XL"], z["carryL"], z["synL"]
    for f in range(XL.shape[0]):
        for c in range(2):
            mem = np.zeros(960 + HALF_OV, np.float32)
            mem[:HALF_OV] = cl[f, c]
            oracle.imdct(XL[f, c], mem, 0, 1)
            assert np.array_equal(mem, synl[f, c])
    for s, B in ((1, 2), (2, 4)):
        n2 = n2_of(s)
        mem = np.zeros(960 + HALF_OV, np.float32)
        mem[:HALF_OV] = z[f"g{s}_carry"]
        for b in range(B):
            oracle.imdct(z[f"g{s}_x"][b:], mem[n2 * b: n2 * b + n2 + HALF_OV], s, B)
        assert np.array_equal(mem, z[f"g{s}_out"])


def test_ifft_shared_bit_exact_vs_reference(oracle):
    z = np.load(os.path.join(GOLDEN, "ref_ifft_shared.npz"))
    for s in range(4):
        x, want = z[f"x{s}"], z[f"y{s}"]
        for r in range(x.shape[0]):
            assert np.array_equal(oracle.ifft_shared(s, x[r]), want[r])
        assert np.array_equal(oracle.ifft_batch(480 >> s, x, shared=True, nthreads=2), want)


def test_chain_vs_reference(oracle):
    """Consecutive blocks of one channel, long and short mixed (celt_decoder_clean.c:625,641)."""
    z = np.load(os.path.join(GOLDEN, "ref_chain.npz"))
    kinds, freq = "".join(z["kinds"]), z["freq"]
    carry = z["carry_in"].copy()
    pcm = []
    for f, k in enumerate(kinds):
        if k == "L":
            p, carry = oracle.imdct_chain(0, freq[f][None, :], carry)
        else:
            blocks = freq[f].reshape(120, 8).T.copy()  # block b = X[b::8]
            p, carry = oracle.imdct_chain(3, blocks, carry)
        pcm.append(p.reshape(-1))
    assert np.array_equal(np.concatenate(pcm), z["pcm"])
    assert np.array_equal(carry, z["tail"])


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_closed_form_imdct(oracle, shift):
    """raw[j] = y[N/4 + j], y[n] = sum_k X[k] cos(2pi/N (n + 1/2 + N/4)(k + 1/2)).
    The reference's approximate rotations deviate from this by 2e-7 (nfft 480) .. 1.2e-5 (nfft 60)."""
    N = 1920 >> shift
    n2 = N // 2
    rng = np.random.default_rng(shift)
    x = rng.uniform(-1, 1, n2).astype(np.float32)
    fin, tail = oracle.imdct_batch(shift, x[None, :], None)
    n = np.arange(N // 4, N // 4 + n2)[:, None]
    k = np.arange(n2)[None, :]
    raw = (np.cos(2 * np.pi / N * (n + 0.5 + N / 4) * (k + 0.5)) @ x.astype(np.float64))
    got_raw = np.concatenate([fin[0, 120:], tail[0]])  # out[120 .. N2+60) = raw[60 .. N2)
    tol = {0: 1e-6, 1: 3e-6, 2: 1e-5, 3: 4e-5}[shift]
    assert rel_rms(got_raw, raw[60:]) <= tol
    # zero carry: out[i] = -w[i] raw[59-i], out[119-i] = w[119-i] raw[59-i]
    w = oracle.tables()[1].astype(np.float64)
    i = np.arange(60)
    head = np.empty(120)
    head[i] = -w[i] * raw[59 - i]
    head[119 - i] = w[119 - i] * raw[59 - i]
    assert rel_rms(fin[0, :120], head) <= tol


def test_linearity_and_edge_cases(oracle):
    rng = np.random.default_rng(7)
    a = rng.uniform(-1, 1, (3, 960)).astype(np.float32)
    b = rng.uniform(-1, 1, (3, 960)).astype(np.float32)
    fa, ta = oracle.imdct_batch(0, a)
    fb, tb = oracle.imdct_batch(0, b)
    fs, ts = oracle.imdct_batch(0, a + b)
    assert rel_rms(fs, fa + fb) <= 1e-6
    assert rel_rms(ts, ta + tb) <= 1e-6
    z, zt = oracle.imdct_batch(0, np.zeros((2, 960), np.float32))
    assert not z.any() and not zt.any()
    e, et = oracle.imdct_batch(0, np.zeros((0, 960), np.float32))  # empty batch
    assert e.shape == (0, 960) and et.shape == (0, 60)


def test_frame_synth_bit_exact_vs_reference(oracle):
    """compute_inv_mdcts over frame sequences (celt_decoder_clean.c:264-312) incl. transient frames."""
    z = np.load(os.path.join(GOLDEN, "ref_synth.npz"))
    pcm, st = oracle.celt_synth(3, z["freq"], z["transient"], z["state_in"], nthreads=2)
    assert np.array_equal(pcm, z["pcm"])
    assert np.array_equal(st, z["state_out"])
    c = np.load(os.path.join(GOLDEN, "ref_chain.npz"))
    tr = np.array([[k == "S" for k in "".join(c["kinds"])]], np.uint8)
    pcm, st = oracle.celt_synth(3, c["freq"][None, :, None, :], tr, c["carry_in"][None, :])
    assert np.array_equal(pcm.reshape(-1), c["pcm"]) and np.array_equal(st[0], c["tail"])


def test_real_decoder_frames_bit_exact(oracle):
    """freq[] and out_syn captured from the reference decoder itself on test_data/short.opus
    (frames 64..127, four of them transient): the oracle's frame synthesis is bit-exact, and the
    captured run reproduces the reference's documented end-to-end numbers."""
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    pcm, st = oracle.celt_synth(3, z["freq"], z["transient"], z["state_in"])
    assert z["transient"].sum() == 4
    assert np.array_equal(pcm, z["pcm"])
    assert np.array_equal(st, z["state_out"])
    # examples/src/Main.cpp:146 pair for sb-reverie.opus: (int)sum == 403, size == 21472602
    assert int(z["sb_reverie"][0]) == 21472602 and int(float(z["sb_reverie_sum"])) == 403
    assert list(z["sb_reverie"][1:]) == [26764, 11184, 314]       # BASELINE.md section 2 call mix
    assert list(z["short_opus"]) == [421930, 554, 220, 8]


def test_real_decoder_post_chain_bit_exact(oracle):
    """freq[] -> IMDCT -> comb filter -> de-emphasis -> interleaved PCM: the oracle reproduces the
    reference decoder's final AudioData::samples for frames 64..127 of short.opus bit for bit
    (comb_filter celt.c:114-172 incl. the x86 association of comb_filter_const, deemphasis
    celt_decoder_clean.c:192-256, state handling :658-683)."""
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    pcm, _ = oracle.celt_synth(3, z["freq"], z["transient"], z["state_in"])
    buf = np.concatenate([z["hist_in"], pcm], axis=2)
    out, filt, pst, dm = oracle.celt_post(3, buf, 1088, z["pf_pitch"], z["pf_gain"], z["pf_tapset"],
                                          z["pf_state_in"], z["deemph_in"])
    assert np.array_equal(out, z["final"])
    assert np.array_equal(pst, z["pf_state_out"])
    assert np.array_equal(dm, z["deemph_out"])
    assert (z["pf_gain"] > 0).sum() > 30 and (z["pf_gain"] == 0).sum() > 0     # filter on and off both occur
