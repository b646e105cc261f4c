"""GPU tier: the band shapes of CELT frames built ON THE DEVICE from symbol records (nyq_celt_shape_lm_dev,
nyq_shape_kernel.hpp) against the host entropy stage's own freq[] (CeltDecoder::decode -- itself pinned to the reference
decoder by tests/test_opus_corpus.py and tests/test_host_decoder.py on the CPU tier): every one-stream file of the corpus
(2.5 / 5 / 10 / 20 ms frames) and the two bundled files, every frame; then the whole way to PCM
(nyq_celt_symbols_to_pcm_mapped) against the freq[] path."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_decoder import entropy_decode, load_host

pytestmark = pytest.mark.gpu

FILES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))
               if not os.path.basename(p).startswith(("surround", "unsupported_"))) + \
    [os.path.join(GOLDEN, "short.opus"), os.path.join(GOLDEN, "sb-reverie.opus")]


@pytest.fixture(scope="module")
def host():
    H = load_host()
    H.nyqh_symbol_bytes.argtypes = [C.c_int]
    H.nyqh_symbol_bytes.restype = C.c_long
    H.nyqh_symbol_bytes_lm.argtypes = [C.c_int, C.c_int]
    H.nyqh_symbol_bytes_lm.restype = C.c_long
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    H.nyqh_decode_to_symbols.argtypes = [C.c_char_p, C.c_long, C.c_long, u8] + list(H.nyqh_decode_to_freq.argtypes[4:])
    return H


@pytest.fixture(scope="module")
def ctx():
    import libnyquist_amd as nyq
    c = nyq.Context(0)
    yield c
    c.close()


def symbols(H, raw, max_frames, channels, lm=3):
    rec = H.nyqh_symbol_bytes_lm(channels, lm)
    sym = np.zeros((max_frames, rec), np.uint8)
    flags = np.zeros((max_frames, 4), np.int32)
    gain = np.zeros(max_frames, np.float32)
    rng = np.zeros(max_frames, np.uint32)
    info = np.zeros(8, np.int64)
    rc = H.nyqh_decode_to_symbols(raw, len(raw), max_frames, sym, flags, gain, rng, info)
    return rc, sym, flags, gain, rng, info


def test_packed_records_equal_slotted_records(host, ctx):
    """Records packed back to back (what the batch decoder stages: a record is as long as its content) through
    nyq_celt_symbols_packed_to_pcm_mapped against the same records one per slot: bit for bit, on a length that takes the
    time-window path (sb-reverie's first 1500 frames) and on a short one; the packed form is about half the bytes."""
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    u32 = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    host.nyqh_decode_to_symbols_packed.argtypes = [C.c_char_p, C.c_long, C.c_long, u8, u32] + list(host.nyqh_decode_to_freq.argtypes[4:])
    for name, cap in (("sb-reverie.opus", 1500), ("short.opus", 220)):
        raw = open(os.path.join(GOLDEN, name), "rb").read()
        rc, sym, flags, gain, rng, info = symbols(host, raw, cap, 2)
        nf = int(info[2])
        assert rc == 0 and nf == cap
        packed = np.zeros_like(sym)
        off = np.zeros(cap + 1, np.uint32)
        flags2, gain2, rng2, info2 = np.zeros((cap, 4), np.int32), np.zeros(cap, np.float32), np.zeros(cap, np.uint32), np.zeros(8, np.int64)
        assert host.nyqh_decode_to_symbols_packed(raw, len(raw), cap, packed, off, flags2, gain2, rng2, info2) == 0
        assert int(info2[2]) == nf and np.array_equal(rng, rng2)
        used = int(off[nf]) * 16
        assert 0.3 * sym[:nf].size <= used <= 0.7 * sym[:nf].size
        tr, pp, pt = (np.ascontiguousarray(flags[:nf, k]) for k in range(3))
        args = (tr.astype(np.uint8), pp.astype(np.int32), gain[:nf].copy(), pt.astype(np.int32))
        want = ctx.celt_symbols_to_pcm(sym[:nf], *args, 1, nf, 2)
        got = ctx.celt_symbols_packed_to_pcm(packed, off, packed.size, *args, 1, nf, 2)
        assert np.array_equal(got, want)


def test_record_size_is_one_number_on_both_sides(host, ctx):
    for ch in (1, 2):
        assert host.nyqh_symbol_bytes(ch) == ctx.lib.nyq_celt_symbol_bytes(ch) == 3072 + ch * 3840
        for lm in range(4):
            n = host.nyqh_symbol_bytes_lm(ch, lm)
            assert n == ctx.lib.nyq_celt_symbol_bytes_lm(ch, lm) and n % 16 == 0 and n >= 32 + ch * (120 << lm) * 4
    assert ctx.lib.nyq_celt_symbol_bytes(3) == 0 and ctx.lib.nyq_celt_symbol_bytes_lm(2, 4) == 0


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-5])
def test_device_band_shapes_equal_the_host_entropy_stage(host, ctx, path):
    import torch
    raw = open(path, "rb").read()
    rc, _, _, _, _, info = entropy_decode(host, raw, max_frames=4, channels=2, n=960)     # (first: the file's shape)
    assert rc == 0
    ch, n = int(info[0]), int(info[3])
    lm = {120: 0, 240: 1, 480: 2, 960: 3}[n]
    cap = 12000 if "reverie" in path else 4200
    rc, freq, flags, gain, rng, info = entropy_decode(host, raw, max_frames=cap, channels=ch, n=n)
    assert rc == 0
    nf = int(info[2])
    freq = freq[:nf]
    rc, sym, flags2, gain2, rng2, info2 = symbols(host, raw, cap, ch, lm)
    assert rc == 0 and int(info2[2]) == nf
    assert np.array_equal(rng[:nf], rng2[:nf]) and np.array_equal(flags[:nf], flags2[:nf])
    host_built = int(info2[6])
    print(f"{os.path.basename(path)}: {nf} frames, {host_built} travel as host-built freq[]")
    # (anti-collapse frames -- the corpus is full of clicks -- and packets whose channel count differs from the stream's)
    if os.path.basename(path) == "sb-reverie.opus":
        assert host_built <= nf // 20                         # music: a few percent
    dev = torch.device("cuda", 0)
    d_sym = torch.from_numpy(sym[:nf]).to(dev)
    d_freq = torch.full((nf, ch, n), float("nan"), device=dev)
    torch.cuda.synchronize(dev)
    ctx.celt_shape_dev(d_sym.data_ptr(), d_freq.data_ptr(), 1, nf, ch, lm=lm)
    ctx.synchronize()
    got = d_freq.cpu().numpy()
    assert np.isfinite(got).all()
    # reductions (norms, the stereo merge's dot products) are summed in another order on the device: a few ulp per band,
    # relative to the band's own peak -- 2e-6 of the frame's peak bounds it with room
    peak = np.abs(freq).reshape(nf, -1).max(1)
    err = np.abs(got - freq).reshape(nf, -1).max(1)
    bad = np.nonzero(err > 2e-6 * np.maximum(peak, 1.0))[0]
    assert bad.size == 0, (bad[:8], err[bad[:8]], peak[bad[:8]])


def test_symbols_to_pcm_equals_the_freq_path(host, ctx):
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    rc, freq, flags, gain, rng, info = entropy_decode(host, raw, max_frames=400, channels=2, n=960)
    nf = int(info[2])
    rc2, sym, *_ = symbols(host, raw, 400, 2)
    assert rc == 0 and rc2 == 0
    tr, pp, pt = (np.ascontiguousarray(flags[:nf, k]) for k in range(3))
    want = ctx.celt_frames_to_pcm(3, freq[None, :nf], tr[None], pp[None], gain[None, :nf], pt[None], 2)
    got = ctx.celt_symbols_to_pcm(sym[:nf], tr.astype(np.uint8), pp.astype(np.int32), gain[:nf], pt.astype(np.int32), 1, nf, 2)
    assert np.abs(got - want).max() <= 1e-6                   # samples are in [-1, 1)


@pytest.mark.timeout(90)
def test_damaged_records_leave_the_kernel_bounded(host, ctx):
    """Records come from this project's own entropy stage, but the C ABI takes them from any caller: with bytes of the
    operation list, the vector records and the leaves overwritten at random, nyq_celt_shape_dev must still return (every
    loop of the kernel is bounded by a checked field; what it writes for such a frame is unspecified, the frames beside it
    are untouched)."""
    import torch
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    rc, sym, flags, gain, rng, info = symbols(host, raw, 400, 2)
    nf = int(info[2])
    assert rc == 0 and nf == 220
    # 32 copies of the stream in one launch: 7040 frames over the resident grid, so every wave goes through several frames,
    # damaged ones and sound ones in turn (what a damaged frame leaves behind in its wave's working set must not matter)
    good = np.tile(sym[:nf], (32, 1))
    nf *= 32
    bad = good.copy()
    r = np.random.default_rng(5)
    for f in range(0, nf, 2):                                  # every other frame damaged, the others as decoded
        for _ in range(40):
            pos = int(r.integers(32, bad.shape[1]))
            bad[f, pos] = r.integers(0, 256)
        if f % 8 == 0:
            bad[f, 200:3072] = r.integers(0, 256, 2872, dtype=np.uint8)          # operations and vectors: noise
        if f % 8 == 4:
            bad[f, 3072:3072 + 4000] = r.integers(0, 256, 4000, dtype=np.uint8)  # further in: noise
    dev = torch.device("cuda", 0)
    d_good, d_bad = torch.from_numpy(good).to(dev), torch.from_numpy(bad).to(dev)
    want = torch.zeros((nf, 2, 960), device=dev)
    got = torch.zeros((nf, 2, 960), device=dev)
    torch.cuda.synchronize(dev)
    ctx.celt_shape_dev(d_good.data_ptr(), want.data_ptr(), 1, nf, 2)
    ctx.celt_shape_dev(d_bad.data_ptr(), got.data_ptr(), 1, nf, 2)
    ctx.synchronize()
    assert torch.equal(got[1::2], want[1::2])


def test_symbols_to_pcm_in_time_windows(host, ctx):
    """nyq_celt_symbols_to_pcm_mapped cut in time windows (NYQ_OPT_HOST_WINDOW = 64: four windows over short.opus) against
    one window: the shape kernel and the chain run window by window, states carried on the device -- bit for bit."""
    import libnyquist_amd as nyq
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    rc, sym, flags, gain, rng, info = symbols(host, raw, 400, 2)
    nf = int(info[2])
    assert rc == 0
    tr, pp, pt = (np.ascontiguousarray(flags[:nf, k]) for k in range(3))
    args = (tr.astype(np.uint8), pp.astype(np.int32), gain[:nf].copy(), pt.astype(np.int32))
    want = ctx.celt_symbols_to_pcm(sym[:nf], *args, 1, nf, 2)
    ctx.set_option(nyq.binding.OPT_HOST_WINDOW, 64)
    try:
        got = ctx.celt_symbols_to_pcm(sym[:nf], *args, 1, nf, 2)
    finally:
        ctx.set_option(nyq.binding.OPT_HOST_WINDOW, 0)
    assert np.array_equal(got, want)
