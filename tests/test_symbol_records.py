"""CPU tier: the SYMBOL records the entropy stage hands to the GPU (include/nyq_imdct.h: nyq_sym_head / op / vec / leaf), read
back field by field.  Every record of every one-stream corpus file (all four frame sizes) must satisfy what
celt_shape_kernel's validation pass demands of it (csrc/nyq_shape_kernel.hpp: no offset outside its bound, every leaf inside
its vector), be exactly as long as its counts say, end the range coder where the reference encoder did, and -- where the
frame travels as host-built freq[] -- carry decode()'s own output."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_decoder import entropy_decode, load_host

HEAD = np.dtype([("seed", "<u4"), ("nleaves", "<u2"), ("nvecs", "<u2"), ("nops", "<u2"), ("flags", "u1"), ("spread", "u1"),
                 ("start", "u1"), ("end", "u1"), ("channels", "u1"), ("lm", "u1"), ("reserved", "<u4", 4)])
OP = np.dtype([("kind", "u1"), ("band", "u1"), ("a", "<i2"), ("b", "<i2"), ("n", "<i2"), ("f0", "<f4"), ("f1", "<f4")])
VEC = np.dtype([("x", "<i2"), ("n", "<i2"), ("fold", "<i2"), ("out", "<i2"), ("nb_tree", "<i2"), ("leaf0", "<i2"), ("leaf1", "<i2"),
                ("sel", "u1"), ("recombine", "u1"), ("time_divide", "u1"), ("b_tree", "u1"), ("b_in", "u1"), ("band", "u1"),
                ("cm_ch", "u1"), ("fill_mode", "u1"), ("fill_lo", "u1"), ("fill_hi", "u1")])
LEAF = np.dtype([("off", "<i2"), ("n", "<i2"), ("k", "<i2"), ("blocks", "u1"), ("kind", "u1"), ("gain", "<f4"), ("fold_off", "<i2"),
                 ("shift", "u1"), ("pad", "u1"), ("index", "<u4"), ("abs", "<i2"), ("pad2", "<i2"), ("img", "<u2", 8)])
FILES = sorted(p for p in glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))
               if not os.path.basename(p).startswith(("surround", "unsupported_", "twosize")))


@pytest.fixture(scope="module")
def host():
    H = load_host()
    H.nyqh_symbol_bytes_lm.argtypes = [C.c_int, C.c_int]
    H.nyqh_symbol_bytes_lm.restype = C.c_long
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    u32 = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    H.nyqh_decode_to_symbols_packed.argtypes = [C.c_char_p, C.c_long, C.c_long, u8, u32] + list(H.nyqh_decode_to_freq.argtypes[4:])
    return H


def test_record_layout_sizes():
    assert (HEAD.itemsize, OP.itemsize, VEC.itemsize, LEAF.itemsize) == (32, 16, 24, 40)


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-5])
def test_records_are_what_the_kernel_accepts(host, path):
    digest = np.load(os.path.join(GOLDEN, "corpus_digest.npz"))
    name = os.path.basename(path)[:-5]
    raw = open(path, "rb").read()
    ch, frame = (int(v) for v in digest[name + "/meta"][:2])
    lm = {120: 0, 240: 1, 480: 2, 960: 3}[frame]
    want_rng = digest[name + "/ranges"]
    cap = len(want_rng) + 4
    slot = host.nyqh_symbol_bytes_lm(ch, lm)
    rec = np.zeros(cap * slot, np.uint8)
    off = np.zeros(cap + 1, np.uint32)
    flags, gain, rng, info = np.zeros((cap, 4), np.int32), np.zeros(cap, np.float32), np.zeros(cap, np.uint32), np.zeros(8, np.int64)
    assert host.nyqh_decode_to_symbols_packed(raw, len(raw), cap, rec, off, flags, gain, rng, info) == 0
    nf = int(info[2])
    assert nf == len(want_rng) and np.array_equal(rng[:nf], want_rng)            # the symbol phase alone reads every bit
    rc, freq, *_ = entropy_decode(host, raw, max_frames=cap, channels=ch, n=frame)
    assert rc == 0
    N, built_on_host, on_device = frame, 0, 0
    for f in range(nf):
        r = rec[int(off[f]) * 16:int(off[f + 1]) * 16]
        assert 32 <= r.size <= slot and r.size % 16 == 0
        h = r[:32].view(HEAD)[0]
        if h["flags"] & 1:                                                        # host-built: head | freq[]
            built_on_host += 1
            assert r.size == 32 + ch * N * 4
            assert np.array_equal(r[32:].view("<f4").reshape(ch, N), freq[f])
            continue
        if h["nops"] == 0:                                                        # silence: a zero head
            assert r.size == 32 and not r.any()
            continue
        on_device += 1
        nops, nvecs, nleaves = int(h["nops"]), int(h["nvecs"]), int(h["nleaves"])
        assert nops <= 113 and nvecs <= 44 and nleaves <= 192 and h["lm"] == lm and h["channels"] in (1, 2)
        assert h["start"] <= 20 and h["start"] <= h["end"] <= 21
        body = 200 + 16 * nops + 24 * nvecs + 40 * nleaves + (168 if h["flags"] & 2 else 0)
        assert r.size == (body + 15) // 16 * 16
        ops = r[200:200 + 16 * nops].view(OP)
        vecs = r[200 + 16 * nops:200 + 16 * nops + 24 * nvecs].view(VEC)
        lv = r[200 + 16 * nops + 24 * nvecs:200 + 16 * nops + 24 * nvecs + 40 * nleaves].view(LEAF)
        # leaves
        assert (lv["n"] >= 1).all() and (lv["off"] >= 0).all() and (lv["off"].astype(int) + lv["n"] <= 176).all()
        assert ((lv["blocks"] >= 1) & (lv["blocks"] <= 16)).all() and (lv["shift"] <= 15).all() and (lv["kind"] <= 1).all()
        assert (lv["abs"] >= 0).all() and (lv["abs"].astype(int) + lv["n"] <= 2 * N).all()
        p = lv[lv["kind"] == 0]
        assert (p["n"] >= 2).all() and (p["k"] >= 1).all() and (p["k"] <= 176).all() and (p["n"] % p["blocks"] == 0).all()
        assert (lv[lv["kind"] == 1]["index"] == 0).all()
        # vectors, and every leaf inside its vector; the vectors' leaf ranges tile the leaf list in order
        assert ((vecs["n"] >= 2) & (vecs["n"] <= 176)).all() and (vecs["x"] >= 0).all() and (vecs["x"].astype(int) + vecs["n"] <= 2 * N).all()
        assert (vecs["recombine"] <= 3).all() and (vecs["time_divide"] <= 3).all() and (vecs["b_in"] >= 1).all() and (vecs["b_in"] <= 8).all()
        assert (vecs["nb_tree"].astype(int) * vecs["b_tree"] == vecs["n"]).all() and (vecs["n"] % vecs["b_in"] == 0).all()
        assert (vecs["fill_lo"] <= vecs["fill_hi"]).all() and (vecs["fill_hi"] <= 21).all() and (vecs["fill_mode"] <= 3).all()
        assert (vecs["band"] <= 20).all() and (vecs["fold"].astype(int) + vecs["n"] <= 800).all() and (vecs["out"].astype(int) + vecs["n"] <= 800).all()
        nxt = 0
        for v in vecs:
            assert v["leaf0"] == nxt and v["leaf0"] < v["leaf1"] <= nleaves and v["leaf1"] - v["leaf0"] <= 16
            nxt = int(v["leaf1"])
            mine = lv[v["leaf0"]:v["leaf1"]]
            assert (mine["off"].astype(int) + mine["n"] <= v["n"]).all() and (mine["abs"] == v["x"] + mine["off"]).all()
            assert (mine["fold_off"].astype(int) + mine["n"] <= v["n"]).all()
            assert int(mine["n"].sum()) == v["n"]                                 # the leaves of a split tree partition their vector
        assert nxt == nleaves
        # operations
        assert (ops["kind"] <= 5).all()
        vec_ops = ops[ops["kind"] == 0]
        assert np.array_equal(vec_ops["a"], np.arange(nvecs))                     # every vector once, in order
        for o in ops:
            k = int(o["kind"])
            if k == 1:
                assert 0 <= o["a"] < 2 * N and o["b"] < 800 and o["band"] <= 20 and abs(o["f0"]) == 1.0
            elif k == 2:
                assert o["a"] >= 0 and o["b"] >= 0 and o["a"] + 2 <= 2 * N and o["b"] + 2 <= 2 * N
            elif k in (3, 4):
                assert o["a"] >= 0 and o["n"] >= 0 and o["a"] + o["n"] <= 2 * N
            elif k == 5:
                assert 0 <= o["a"] <= 800
    # (frames whose record would outgrow its slot travel as freq[]: most of a 256 kbit/s stereo stream, few elsewhere)
    assert on_device > 0 and (built_on_host <= nf // 2 or "256k" in name)
