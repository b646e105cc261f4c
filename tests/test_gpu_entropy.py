"""GPU tier: the entropy stage ON THE DEVICE (nyq_celt_entropy_dev: a frame per lane, csrc/nyq_entropy_core.hpp, then the energy
pass, a wave per stream) against the host decoder's symbol records (CeltDecoder::decodeSymbols) on every frame of every
one-stream file of the corpus: operation, vector and leaf lists byte for byte, head fields, log gains bit for bit, final range,
post-filter parameters and flags; then the band shapes built from the device's records against those built from the host's."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_decoder import load_host

pytestmark = pytest.mark.gpu

FILES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))
               if not os.path.basename(p).startswith(("surround", "unsupported_"))) + \
    [os.path.join(GOLDEN, "short.opus"), os.path.join(GOLDEN, "sb-reverie.opus")]

DESC = np.dtype([("offset", "<u4"), ("len", "<u2"), ("channels", "u1"), ("start", "u1"), ("end", "u1"), ("pad", "u1", 3)])
INFO = np.dtype([("range_final", "<u4"), ("pf_pitch", "<i2"), ("pf_tapset", "u1"), ("pf_gain_index", "u1"), ("flags", "u1"),
                 ("lm", "u1"), ("channels", "u1"), ("start", "u1"), ("end", "u1"), ("pad", "u1", 3)])
HEAD = np.dtype([("seed", "<u4"), ("nleaves", "<u2"), ("nvecs", "<u2"), ("nops", "<u2"), ("flags", "u1"), ("spread", "u1"),
                 ("start", "u1"), ("end", "u1"), ("channels", "u1"), ("lm", "u1"), ("reserved", "<u4", 4)])
TOO_LARGE, ERROR, TRANSIENT, SILENCE = 32, 16, 1, 2


@pytest.fixture(scope="module")
def host():
    H = load_host()
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    H.nyqh_symbol_bytes_lm.argtypes = [C.c_int, C.c_int]
    H.nyqh_symbol_bytes_lm.restype = C.c_long
    H.nyqh_decode_to_symbols.argtypes = [C.c_char_p, C.c_long, C.c_long, u8] + list(H.nyqh_decode_to_freq.argtypes[4:])
    H.nyqh_entropy_tables.argtypes = [C.c_void_p, C.c_long]
    H.nyqh_entropy_tables.restype = C.c_long
    H.nyqh_frame_table.argtypes = [C.c_char_p, C.c_long, C.c_long, u8, C.c_long, C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    return H


@pytest.fixture(scope="module")
def ctx():
    import libnyquist_amd as nyq
    c = nyq.Context(0)
    yield c
    c.close()


def device_records(host, ctx, raw, copies=1, full_slot=True):
    """(lm, channels, nf, records [copies * nf][slot] uint8, info [copies * nf], device tensor of the records); full_slot: slots
    that hold any frame (nyq_celt_entropy_slot_bytes), else the host records' slot (busy frames come back TOO_LARGE)"""
    import torch
    need = host.nyqh_entropy_tables(None, 0)
    assert need == ctx.lib.nyq_celt_entropy_tables_bytes()
    tables = np.zeros(need, np.uint8)
    assert host.nyqh_entropy_tables(tables.ctypes.data, need) == need
    cap = 12000
    payload = np.zeros(cap * 1275 // 4, np.uint8)
    desc = np.zeros(cap, DESC)
    finfo = np.zeros(8, np.int64)
    assert host.nyqh_frame_table(raw, len(raw), cap, payload, payload.size, desc.ctypes.data, finfo) == 0
    ch, nf, frame = int(finfo[0]), int(finfo[2]), int(finfo[3])
    lm = {120: 0, 240: 1, 480: 2, 960: 3}[frame]
    slot = ctx.lib.nyq_celt_entropy_slot_bytes(ch, lm) if full_slot else ctx.lib.nyq_celt_symbol_bytes_lm(ch, lm)
    dev = torch.device("cuda", 0)
    d_tab = torch.from_numpy(tables).to(dev)
    d_pay = torch.from_numpy(payload[:max(int(finfo[4]), 1)].copy()).to(dev)
    d_desc = torch.from_numpy(np.tile(desc[:nf], copies).view(np.uint8)).to(dev)
    d_sym = torch.zeros((copies * nf, slot), dtype=torch.uint8, device=dev)
    d_info = torch.zeros((copies * nf, INFO.itemsize), dtype=torch.uint8, device=dev)
    d_energy = torch.zeros((copies * nf, 672), dtype=torch.uint8, device=dev)
    d_state = torch.zeros((copies, 43 * 3 * 4), dtype=torch.uint8, device=dev)
    assert INFO.itemsize == 16 and DESC.itemsize == 12
    torch.cuda.synchronize(dev)
    ctx.celt_entropy_dev(lm, d_tab.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_desc.data_ptr(), copies, nf, ch, d_sym.data_ptr(), d_info.data_ptr(),
                         d_energy.data_ptr(), d_state.data_ptr(), True, slot)
    ctx.synchronize()
    return lm, ch, nf, d_sym.cpu().numpy(), d_info.cpu().numpy().view(INFO).reshape(-1), d_sym


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_device_entropy_stage_equals_the_host_decoder(host, ctx, path):
    import torch
    raw = open(path, "rb").read()
    lm, ch, nf, recs, info, d_sym = device_records(host, ctx, raw)
    assert not (info["flags"] & TOO_LARGE).any()                      # (the full slot holds any frame)
    slot = ctx.lib.nyq_celt_symbol_bytes_lm(ch, lm)                   # the host records' slot
    sym = np.zeros((nf, slot), np.uint8)
    flags = np.zeros((nf, 4), np.int32)
    gain = np.zeros(nf, np.float32)
    rng = np.zeros(nf, np.uint32)
    hinfo = np.zeros(8, np.int64)
    assert host.nyqh_decode_to_symbols(raw, len(raw), nf, sym, flags, gain, rng, hinfo) == 0 and int(hinfo[2]) == nf
    # what the frame says beside its record
    assert np.array_equal(info["range_final"], rng)
    assert np.array_equal((info["flags"] & TRANSIENT) != 0, flags[:, 0] != 0)
    assert np.array_equal(info["pf_pitch"].astype(np.int32), flags[:, 1]) and np.array_equal(info["pf_tapset"].astype(np.int32), flags[:, 2])
    assert np.array_equal(np.float32(.09375) * info["pf_gain_index"].astype(np.float32), gain)
    assert not (info["flags"] & ERROR).any()
    hh = sym[:, :32].copy().view(HEAD).reshape(-1)
    dh = recs[:, :32].copy().view(HEAD).reshape(-1)
    compared = too_large = 0
    for f in range(nf):
        if hh["flags"][f] & 1:                                      # built by the host there (more leaves than a host record holds)
            too_large += 1
            continue
        if hh["nops"][f] == 0:                                      # a silent frame
            assert dh["nops"][f] == 0
            continue
        for k in ("seed", "nleaves", "nvecs", "nops", "flags", "spread", "start", "end", "channels", "lm"):
            assert hh[k][f] == dh[k][f], (f, k)
        no, nv, nl = int(hh["nops"][f]), int(hh["nvecs"][f]), int(hh["nleaves"][f])
        o_ops, o_vecs = int(dh["reserved"][f][0]) & 0xffff, int(dh["reserved"][f][0]) >> 16
        o_leaves, o_level = int(dh["reserved"][f][1]) & 0xffff, int(dh["reserved"][f][1]) >> 16
        h, d = sym[f], recs[f]
        assert np.array_equal(h[200:200 + 16 * no], d[o_ops:o_ops + 16 * no]), f
        assert np.array_equal(h[200 + 16 * no:200 + 16 * no + 24 * nv], d[o_vecs:o_vecs + 24 * nv]), f
        hl = 200 + 16 * no + 24 * nv
        assert np.array_equal(h[hl:hl + 40 * nl], d[o_leaves:o_leaves + 40 * nl]), f
        C_, end = int(hh["channels"][f]), int(hh["end"][f])
        hg, dg = h[32:200].view(np.float32), d[32:200].view(np.float32)
        for c in range(C_):
            assert np.array_equal(hg[21 * c:21 * c + end].view(np.uint32), dg[21 * c:21 * c + end].view(np.uint32)), (f, c)
        if hh["flags"][f] & 2:                                      # anti-collapse levels: a double exponential on either side
            hv = h[hl + 40 * nl:hl + 40 * nl + 168].view(np.float32)
            dv = d[o_level:o_level + 168].view(np.float32)
            for c in range(C_):
                assert np.allclose(hv[21 * c:21 * c + end], dv[21 * c:21 * c + end], rtol=2e-6, atol=0), (f, c)
        compared += 1
    print(f"{os.path.basename(path)}: {nf} frames, {compared} records equal, {too_large} have no host record (built on the host there)")
    assert compared >= (nf * 3) // 4 or lm < 3 or "256k" in path
    # the band shapes from the device's records against those from the host's
    dev = torch.device("cuda", 0)
    n = 120 << lm
    d_host = torch.from_numpy(sym).to(dev)
    want = torch.zeros((nf, ch, n), device=dev)
    got = torch.zeros((nf, ch, n), device=dev)
    torch.cuda.synchronize(dev)
    ctx.celt_shape_dev(d_host.data_ptr(), want.data_ptr(), 1, nf, ch, lm=lm)
    ctx.celt_shape_slots_dev(lm, d_sym.data_ptr(), recs.shape[1], got.data_ptr(), 1, nf, ch)
    ctx.synchronize()
    # (every frame: where the host built freq[] itself -- its record carries it -- the device's record stands against the host's
    # floats, a few ulp per band apart as in test_gpu_shape; elsewhere both sides ran the same kernel on equal records)
    w, g = want.cpu().numpy(), got.cpu().numpy()
    assert np.isfinite(g).all()
    peak = np.abs(w).reshape(len(w), -1).max(1)
    err = np.abs(g - w).reshape(len(w), -1).max(1)
    built = (hh["flags"] & 1) != 0
    assert (err[~built] <= 1e-6 * np.maximum(peak[~built], 1.0)).all()
    assert (err[built] <= 2e-6 * np.maximum(peak[built], 1.0)).all()


def bytes_to_pcm_on_device(ctx, lm, ch, copies, nf, d_tab, d_pay, d_desc, bufs=None):
    """frames' bytes -> interleaved PCM without leaving the device: entropy stage, per-frame arrays, band shapes, synthesis +
    post-filter (fresh streams).  Returns the PCM tensor [copies][nf * (120 << lm)][ch] and the infos' tensor."""
    import torch
    dev = torch.device("cuda", 0)
    n = 120 << lm
    slot = ctx.lib.nyq_celt_entropy_slot_bytes(ch, lm)
    tot = copies * nf
    if bufs is None:
        bufs = dict(
            sym=torch.zeros((tot, slot), dtype=torch.uint8, device=dev), info=torch.zeros((tot, 16), dtype=torch.uint8, device=dev),
            energy=torch.zeros((tot, 672), dtype=torch.uint8, device=dev), state=torch.zeros((copies, 43 * 3 * 4), dtype=torch.uint8, device=dev),
            tr=torch.zeros(tot, dtype=torch.uint8, device=dev), pp=torch.zeros(tot, dtype=torch.int32, device=dev),
            pg=torch.zeros(tot, dtype=torch.float32, device=dev), pt=torch.zeros(tot, dtype=torch.int32, device=dev),
            freq=torch.zeros((tot, ch, n), device=dev), out=torch.zeros((copies, nf * n, ch), device=dev),
            pcm=torch.empty((copies * ch, nf * n), device=dev), work=torch.empty(ctx.celt_synth_work_floats(copies, nf, ch), device=dev))
        torch.cuda.synchronize(dev)
    b = bufs
    ctx.celt_entropy_dev(lm, d_tab.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_desc.data_ptr(), copies, nf, ch, b["sym"].data_ptr(), b["info"].data_ptr(),
                         b["energy"].data_ptr(), b["state"].data_ptr(), True, slot)
    ctx.celt_entropy_split_dev(b["info"].data_ptr(), tot, b["tr"].data_ptr(), b["pp"].data_ptr(), b["pg"].data_ptr(), b["pt"].data_ptr())
    ctx.celt_shape_slots_dev(lm, b["sym"].data_ptr(), slot, b["freq"].data_ptr(), copies, nf, ch)
    ctx.celt_chain_dev(lm, b["freq"].data_ptr(), b["tr"].data_ptr(), b["pp"].data_ptr(), b["pg"].data_ptr(), b["pt"].data_ptr(), 0, 0, 0, 0, 0,
                       b["out"].data_ptr(), b["pcm"].data_ptr(), b["work"].data_ptr(), copies, nf, ch)
    return b


def frame_table(host, raw, cap=12000):
    payload = np.zeros(cap * 1275 // 4, np.uint8)
    desc = np.zeros(cap, DESC)
    finfo = np.zeros(8, np.int64)
    assert host.nyqh_frame_table(raw, len(raw), cap, payload, payload.size, desc.ctypes.data, finfo) == 0
    nf = int(finfo[2])
    return int(finfo[0]), nf, {120: 0, 240: 1, 480: 2, 960: 3}[int(finfo[3])], payload[:max(int(finfo[4]), 1)].copy(), desc[:nf].copy()


def test_frames_too_large_for_a_host_sized_slot_are_flagged(host, ctx):
    """With the host records' slot (slot_bytes = 0) a busy frame does not fit: it is flagged and its record is a silent frame."""
    raw = open(os.path.join(GOLDEN, "corpus", "st_20ms_256k_cbr.opus"), "rb").read()
    lm, ch, nf, recs, info, _ = device_records(host, ctx, raw, full_slot=False)
    big = (info["flags"] & TOO_LARGE) != 0
    assert 5 <= big.sum() < nf
    dh = recs[:, :32].copy().view(HEAD).reshape(-1)
    assert (dh["nops"][big] == 0).all() and (dh["nops"][~big] > 0).all()


@pytest.mark.parametrize("name", ["short.opus", "corpus/mono_20ms_64k.opus", "corpus/st_10ms_20k_cx0.opus", "corpus/st_2p5ms_128k.opus",
                                  "corpus/st_20ms_256k_cbr.opus", "corpus/st_5ms_96k.opus"])
def test_bytes_to_pcm_on_the_device_equals_the_host_record_path(host, ctx, name):
    """The whole decode of a stream's frames on the device -- bytes in, PCM out, nothing but packet parsing on the host -- against
    today's product path (host entropy stage -> symbol records -> nyq_celt_symbols_to_pcm_mapped)."""
    import torch
    raw = open(os.path.join(GOLDEN, name), "rb").read()
    ch, nf, lm, payload, desc = frame_table(host, raw)
    need = host.nyqh_entropy_tables(None, 0)
    tables = np.zeros(need, np.uint8)
    host.nyqh_entropy_tables(tables.ctypes.data, need)
    dev = torch.device("cuda", 0)
    d_tab, d_pay = torch.from_numpy(tables).to(dev), torch.from_numpy(payload).to(dev)
    d_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    b = bytes_to_pcm_on_device(ctx, lm, ch, 1, nf, d_tab, d_pay, d_desc)
    ctx.synchronize()
    got = b["out"].cpu().numpy()
    info = b["info"].cpu().numpy().view(INFO).reshape(-1)
    assert not (info["flags"] & (TOO_LARGE | ERROR)).any()
    slot = ctx.lib.nyq_celt_symbol_bytes_lm(ch, lm)
    sym = np.zeros((nf, slot), np.uint8)
    flags = np.zeros((nf, 4), np.int32)
    gain = np.zeros(nf, np.float32)
    rng = np.zeros(nf, np.uint32)
    hinfo = np.zeros(8, np.int64)
    assert host.nyqh_decode_to_symbols(raw, len(raw), nf, sym, flags, gain, rng, hinfo) == 0 and int(hinfo[2]) == nf
    tr, pp, pt = (np.ascontiguousarray(flags[:, k]) for k in range(3))
    want = ctx.celt_symbols_to_pcm(sym, tr.astype(np.uint8), pp.astype(np.int32), gain, pt.astype(np.int32), 1, nf, ch, lm=lm)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6                       # samples are in [-1, 1)


@pytest.mark.timeout(120)
def test_random_bytes_leave_the_entropy_stage_bounded(host, ctx):
    """Frames of random bytes and random lengths (and descriptors that point outside the payload): the kernels return, the entropy
    stage says the same about every frame twice, and every record it leaves passes through the shape kernel and the chain
    (whose own checks bound them).  What such frames SOUND like is unspecified, as it is in the reference: records that are valid
    field by field may still name fold sources no band wrote, and a wave's working set holds what its previous frame left."""
    import torch
    need = host.nyqh_entropy_tables(None, 0)
    tables = np.zeros(need, np.uint8)
    host.nyqh_entropy_tables(tables.ctypes.data, need)
    r = np.random.default_rng(11)
    ns, nf = 64, 128
    payload = r.integers(0, 256, 1 << 20, dtype=np.uint8)
    desc = np.zeros(ns * nf, DESC)
    desc["offset"] = r.integers(0, payload.size + 4000, ns * nf)
    desc["len"] = r.integers(0, 1400, ns * nf)
    desc["channels"] = r.integers(0, 4, ns * nf)
    desc["start"] = r.integers(0, 3, ns * nf) * (r.uniform(size=ns * nf) < .1)
    desc["end"] = r.choice([13, 17, 19, 21, 21, 21, 30, 0], ns * nf)
    dev = torch.device("cuda", 0)
    d_tab, d_pay = torch.from_numpy(tables).to(dev), torch.from_numpy(payload).to(dev)
    d_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    for lm, ch in ((3, 2), (0, 1), (2, 2)):
        outs = []
        for rep in range(2):
            b = bytes_to_pcm_on_device(ctx, lm, ch, ns, nf, d_tab, d_pay, d_desc)
            ctx.synchronize()
            outs.append((b["info"].cpu().numpy().copy(), b["out"].cpu().numpy().copy()))
        assert np.array_equal(outs[0][0], outs[1][0])
        assert outs[0][1].shape == outs[1][1].shape


def byte_slots(ctx, payload, desc):
    """the host-buffer form's input: a frame's bytes at the start of its slot, and a word per frame"""
    slot = int(ctx.lib.nyq_celt_byte_slot())
    nf = len(desc)
    fb = np.zeros((nf, slot), np.uint8)
    for i in range(nf):
        o, n = int(desc["offset"][i]), int(desc["len"][i])
        fb[i, :n] = payload[o:o + n]
    words = (desc["len"].astype(np.uint32) | desc["channels"].astype(np.uint32) << 16 | desc["end"].astype(np.uint32) << 24).astype(np.uint32)
    return fb, words


@pytest.mark.parametrize("name", ["short.opus", "corpus/mono_20ms_64k.opus", "corpus/st_10ms_96k.opus", "corpus/st_20ms_256k_cbr.opus"])
def test_bytes_to_pcm_host_buffer_form(host, ctx, name):
    """nyq_celt_bytes_to_pcm_mapped (frames' bytes in host memory, everything else on the device) against the host-record path:
    in one call, in time windows (NYQ_OPT_HOST_WINDOW), and cut in two calls with the state carried between them."""
    import libnyquist_amd as nyq
    raw = open(os.path.join(GOLDEN, name), "rb").read()
    ch, nf, lm, payload, desc = frame_table(host, raw)
    need = host.nyqh_entropy_tables(None, 0)
    tables = np.zeros(need, np.uint8)
    host.nyqh_entropy_tables(tables.ctypes.data, need)
    ctx.set_entropy_tables(tables)
    fb, words = byte_slots(ctx, payload, desc)
    slot = ctx.lib.nyq_celt_symbol_bytes_lm(ch, lm)
    sym = np.zeros((nf, slot), np.uint8)
    flags, gain, rng, hinfo = np.zeros((nf, 4), np.int32), np.zeros(nf, np.float32), np.zeros(nf, np.uint32), np.zeros(8, np.int64)
    assert host.nyqh_decode_to_symbols(raw, len(raw), nf, sym, flags, gain, rng, hinfo) == 0 and int(hinfo[2]) == nf
    tr, pp, pt = (np.ascontiguousarray(flags[:, k]) for k in range(3))
    want = ctx.celt_symbols_to_pcm(sym, tr.astype(np.uint8), pp.astype(np.int32), gain, pt.astype(np.int32), 1, nf, ch, lm=lm)
    got = ctx.celt_bytes_to_pcm(lm, fb, words, 1, nf, ch)
    assert np.abs(got - want).max() <= 2e-6
    if nf >= 128:
        ctx.set_option(nyq.binding.OPT_HOST_WINDOW, 64)
        try:
            win = ctx.celt_bytes_to_pcm(lm, fb, words, 1, nf, ch)
        finally:
            ctx.set_option(nyq.binding.OPT_HOST_WINDOW, 0)
        assert np.abs(win - want).max() <= 2e-6
    # two calls, the second one continuing the first (state: synthesis, post-filter and the entropy stage's energies / range)
    cut = (nf // 2) // 8 * 8
    state = np.zeros(ctx.lib.nyq_celt_state_floats(1, ch), np.float32)
    a = ctx.celt_bytes_to_pcm(lm, fb[:cut], words[:cut], 1, cut, ch, state=state)
    b = ctx.celt_bytes_to_pcm(lm, fb[cut:], words[cut:], 1, nf - cut, ch, state=state)
    assert np.abs(np.concatenate([a, b], axis=1) - want).max() <= 2e-6
    est = state[-129:].view(np.uint32)
    assert est[126] == rng[-1] and est[128] == 0                  # final range of the last frame; no frame in error
