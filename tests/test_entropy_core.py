"""CPU tier: the frame-per-lane entropy stage (libnyquist_amd/csrc/nyq_entropy_core.hpp -- the text the GPU kernel compiles)
built for the HOST and run over every one-stream file of the corpus against the host decoder's symbol records
(tools/scripts/entropy_core_check.cpp: lists byte for byte, head fields, log gains and anti-collapse levels after the energy
pass, final range, post-filter parameters).  The host decoder itself is pinned to the reference by test_opus_corpus.py."""
import glob
import os
import subprocess

from conftest import GOLDEN, ROOT as REPO


def test_entropy_core_equals_the_host_decoder_on_the_corpus(tmp_path):
    exe = str(tmp_path / "entropy_core_check")
    host = os.path.join(REPO, "libnyquist_amd", "host")
    srcs = [os.path.join(REPO, "tools", "scripts", "entropy_core_check.cpp")] + [os.path.join(host, f) for f in ("celt_mode.cpp", "celt_decoder.cpp", "opus_stream.cpp")]
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + host, "-I" + os.path.join(REPO, "include"), "-o", exe] + srcs, check=True)
    files = [os.path.join(GOLDEN, "short.opus"), os.path.join(GOLDEN, "sb-reverie.opus")] + sorted(
        p for p in glob.glob(os.path.join(GOLDEN, "corpus", "*.opus")) if not os.path.basename(p).startswith(("surround", "unsupported_")))
    out = subprocess.run([exe] + files, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    frames = int(last.split()[1])
    assert frames >= 13000 and " 0 differ" in last, last


def test_frame_table_and_tables_of_the_host_library():
    """nyqh_frame_table (what a caller of nyq_celt_entropy_dev stages: bytes back to back + a descriptor per frame) walks the same
    frames as the host decoder's dump; nyqh_entropy_tables fills the same block twice."""
    import ctypes as C

    import numpy as np

    from test_host_decoder import load_host
    H = load_host()
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    H.nyqh_entropy_tables.argtypes = [C.c_void_p, C.c_long]
    H.nyqh_entropy_tables.restype = C.c_long
    H.nyqh_frame_table.argtypes = [C.c_char_p, C.c_long, C.c_long, u8, C.c_long, C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    need = H.nyqh_entropy_tables(None, 0)
    assert 40000 < need < 80000
    a, b = np.zeros(need, np.uint8), np.full(need, 255, np.uint8)
    assert H.nyqh_entropy_tables(a.ctypes.data, need) == need and H.nyqh_entropy_tables(b.ctypes.data, need) == need
    assert np.array_equal(a, b) and H.nyqh_entropy_tables(a.ctypes.data, need - 1) == -1
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    desc = np.zeros(400 * 12, np.uint8)
    payload = np.zeros(400 * 1275, np.uint8)
    info = np.zeros(8, np.int64)
    assert H.nyqh_frame_table(raw, len(raw), 400, payload, payload.size, desc.ctypes.data, info) == 0
    assert list(info[:4]) == [2, 312, 220, 960]                     # channels, pre-skip, 20 ms frames (the closing 2.5 ms frame ends the walk), size
    d = desc[:220 * 12].view(np.dtype([("offset", "<u4"), ("len", "<u2"), ("channels", "u1"), ("start", "u1"), ("end", "u1"), ("pad", "u1", 3)]))
    assert (d["channels"] == 2).all() and (d["end"] == 21).all() and (d["start"] == 0).all()
    assert np.array_equal(d["offset"][1:], np.cumsum(d["len"].astype(np.int64))[:-1]) and int(d["len"].sum()) == int(info[4])
    assert H.nyqh_frame_table(raw, len(raw), 400, payload, 100, desc.ctypes.data, info) == -12
    assert H.nyqh_frame_table(raw[:50], 50, 400, payload, payload.size, desc.ctypes.data, info) == -10
