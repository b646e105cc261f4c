"""CPU tier: the C++ host side above the C ABI (libnyquist_amd/host): mode tables, Ogg/Opus packet
layer and the CPU entropy stage (own CELT frame decoder), checked against data captured from the
reference decoder.  The GPU stages are exercised in tests/test_gpu_host.py."""
import ctypes as C
import os
import subprocess

import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

_f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32 = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i16 = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def load_host():
    import libnyquist_amd as nyq
    nyq.build()
    subprocess.run(["make", "-C", os.path.join(ROOT, "libnyquist_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    H = C.CDLL(os.path.join(ROOT, "libnyquist_amd", "libnyquist_host.so"))
    H.nyqh_mode_tables.argtypes = [_i16, _i16, _u8, C.POINTER(C.c_int), _u8]
    H.nyqh_decode_to_freq.argtypes = [C.c_char_p, C.c_long, C.c_long, _f32, _i32, _f32, _u32, _i64]
    H.nyqh_nyquistio_load.argtypes = [C.c_char_p, C.c_void_p, C.c_long, _i64]
    H.nyqh_nyquistio_load.restype = C.c_long
    H.nyqh_nyquistio_load_buffer.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, _i64]
    H.nyqh_nyquistio_load_buffer.restype = C.c_long
    H.nyqh_batch_decode.argtypes = [C.c_char_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_long,
                                    np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
    H.nyqh_batch_decode.restype = C.c_long
    H.nyqh_batch_decode_timed.argtypes = H.nyqh_batch_decode.argtypes
    H.nyqh_batch_decode_timed.restype = C.c_long
    H.nyqh_batch_decode_files.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.c_long, C.c_int, C.POINTER(C.c_long), C.c_void_p, C.c_long]
    H.nyqh_batch_decode_files.restype = C.c_long
    H.nyqh_batch_load_devices.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_long), C.c_long, C.POINTER(C.c_int), C.c_int,
                                          C.POINTER(C.c_long), C.c_void_p, C.c_long]
    H.nyqh_batch_load_devices.restype = C.c_long
    H.nyqh_capi_device_count.restype = C.c_int
    H.nyqh_set_default_device.argtypes = [C.c_int]
    H.nyqh_set_default_device.restype = None
    H.nyqh_set_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    H.nyqh_set_devices.restype = None
    H.nyqh_decoder_pool_counts.argtypes = [C.POINTER(C.c_long)]
    H.nyqh_decoder_pool_counts.restype = None
    H.nyqh_last_error.restype = C.c_char_p
    return H


@pytest.fixture(scope="module")
def host():
    return load_host()


def entropy_decode(H, raw, max_frames=400, channels=2, n=960):
    freq = np.zeros((max_frames, channels, n), np.float32)
    flags = np.zeros((max_frames, 4), np.int32)
    gain = np.zeros(max_frames, np.float32)
    rng = np.zeros(max_frames, np.uint32)
    info = np.zeros(6, np.int64)
    rc = H.nyqh_decode_to_freq(raw, len(raw), max_frames, freq, flags, gain, rng, info)
    return rc, freq, flags, gain, rng, info


def test_tell_frac_table_equals_the_squaring_loop(host):
    """ec_tell_frac (entcode.c:69-93) as a linear guess plus one table comparison: the same integer as the reference's three
    squarings for every value of the range's top 16 bits at every width (589,806 cases)."""
    host.nyqh_tell_frac_self_check.restype = C.c_long
    assert host.nyqh_tell_frac_self_check() == 0


def test_mode_tables_match_reference(host, ref_tables):
    """logN, pulse cache and caps are COMPUTED by the host library (rate.c:73-245 restated); they
    must equal the reference's generated static tables (static_modes_float.h:36-97)."""
    logn = np.zeros(21, np.int16)
    cidx = np.zeros(105, np.int16)
    cbits = np.zeros(1024, np.uint8)
    ccaps = np.zeros(168, np.uint8)
    nb = C.c_int()
    assert host.nyqh_mode_tables(logn, cidx, cbits, C.byref(nb), ccaps) == 0
    assert np.array_equal(logn, ref_tables["logN"])
    assert np.array_equal(cidx, ref_tables["cache_index"])
    assert nb.value == ref_tables["cache_bits"].size == 392
    assert np.array_equal(cbits[: nb.value], ref_tables["cache_bits"])
    assert np.array_equal(ccaps, ref_tables["cache_caps"])


def test_entropy_stage_matches_reference_decoder_on_short_opus(host):
    """Every one of the 220 frames of test_data/short.opus: transient flag and post-filter parameters
    identical, freq[] equal to what the reference decoder computed (digests: sum, energy, peak per
    channel; full data for frames 64..127)."""
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    d = np.load(os.path.join(GOLDEN, "short_opus_digest.npz"))
    rc, freq, flags, gain, rng, info = entropy_decode(host, raw)
    assert rc == 0
    ch, preskip, nframes, fsize, granule, npackets = (int(v) for v in info)
    assert (ch, preskip, nframes, fsize) == (2, 312, 220, 960) and npackets == 221
    assert np.array_equal(flags[:nframes, 0], d["transient"].astype(np.int32))
    assert np.array_equal(flags[:nframes, 1], d["pf_pitch"])
    assert np.array_equal(flags[:nframes, 2], d["pf_tapset"])
    assert np.array_equal(gain[:nframes], d["pf_gain"])
    f64 = freq[:nframes].astype(np.float64)
    peak = np.maximum(d["freq_peak"], 1e-3)
    assert np.abs(f64.sum(axis=2) - d["freq_sum"]).max() <= 1e-3
    assert (np.abs((f64 ** 2).sum(axis=2) - d["freq_sq"]) / np.maximum(d["freq_sq"], 1e-6)).max() <= 1e-5
    assert (np.abs(np.abs(f64).max(axis=2) - d["freq_peak"]) / peak).max() <= 1e-5
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    lo, hi = (int(v) for v in z["window"])
    ref = z["freq"][0]
    err = np.abs(freq[lo:hi] - ref).max() / np.abs(ref).max()
    assert err <= 1e-6
    assert (freq[lo:hi] == ref).mean() > 0.8          # most coefficients are bit-identical


def test_bad_crc_captures_are_skipped_like_libogg_does(host):
    """libogg (ogg_sync_pageseek) treats a capture pattern whose checksum fails as NOT a page and resynchronises; the
    reference therefore loads files with stray "OggS" bytes in trailing junk and with damaged pages of OTHER logical
    streams.  Same here: identical frames to the clean file.  A lost page of the SELECTED stream is a hole in the page
    sequence: the reference gives up (OP_HOLE -> OpusDecoder.cpp:108-112), and so does this build."""
    import struct
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from oggopus import ogg_crc, page
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    rc0, freq0, flags0, gain0, rng0, info0 = entropy_decode(host, raw)
    assert rc0 == 0 and info0[2] == 220
    # (1) trailing junk with a fake capture pattern and a plausible header
    junk = b"\x00" * 7 + b"OggS" + bytes([0, 0]) + struct.pack("<qIII", 12345, 0xDEAD, 7, 0x1234) + bytes([2, 10, 10]) + b"x" * 20 + b"OggS"
    rc1, freq1, flags1, gain1, rng1, info1 = entropy_decode(host, raw + junk)
    assert rc1 == 0 and np.array_equal(rng1, rng0) and np.array_equal(freq1, freq0)
    # (2) a page of a foreign logical stream with a WRONG checksum in the middle of the file
    foreign = bytearray(page(0x0BADBEEF, 3, 999, b"not our stream" * 10, 0))
    foreign[22] ^= 0xFF
    pos, pages = 0, []
    while pos + 27 <= len(raw) and raw[pos:pos + 4] == b"OggS":
        nseg = raw[pos + 26]
        ln = 27 + nseg + sum(raw[pos + 27:pos + 27 + nseg])
        pages.append(raw[pos:pos + ln])
        pos += ln
    mixed = b"".join(pages[:5]) + bytes(foreign) + b"".join(pages[5:])
    rc2, freq2, flags2, gain2, rng2, info2 = entropy_decode(host, mixed)
    assert rc2 == 0 and np.array_equal(rng2, rng0) and np.array_equal(freq2, freq0)
    # (3) one of OUR audio pages damaged (checksum no longer matches): the page is dropped, the hole is an error
    hurt = bytearray(raw)
    off = sum(len(p) for p in pages[:6]) + 27 + pages[6][26] + 40      # inside the body of page 6
    hurt[off] ^= 0x55
    assert ogg_crc(bytes(pages[6])) is not None
    rc3 = entropy_decode(host, bytes(hurt))[0]
    assert rc3 == -10


def test_malformed_inputs_are_rejected(host):
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    assert entropy_decode(host, b"not an ogg file at all" * 10)[0] == -10
    assert entropy_decode(host, raw[:40])[0] == -10
    broken = bytearray(raw)
    i = broken.find(b"OpusHead")
    broken[i:i + 8] = b"OpusHaed"
    assert entropy_decode(host, bytes(broken))[0] == -10
    # a truncated file still decodes its complete packets
    rc, *_ , info = entropy_decode(host, raw[: len(raw) // 2])
    assert rc == 0 and 50 < int(info[2]) < 220


def test_plugin_surface_rejects_unknown_extension(host):
    info = np.zeros(4, np.int64)
    assert host.nyqh_nyquistio_load(b"/tmp/whatever.flac", None, 0, info) == -2      # UnsupportedExtensionEx
    assert host.nyqh_nyquistio_load(b"/nonexistent/file.opus", None, 0, info) == -1  # std::runtime_error


def test_device_list_from_the_environment_is_parsed_strictly(host):
    """NYQ_DEVICES / NYQ_DEVICE select the GPUs of the C batch entry points (tests and tools); anything that is not a list of
    non-negative integers is an error with a message -- round 2 parsed with atoi, so junk silently meant device 0."""
    import ctypes as C
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    files = (C.c_char_p * 1)(raw)
    sizes = (C.c_long * 1)(len(raw))
    ns = (C.c_long * 1)()
    for bad in ("0,x", "zero", "0,,1", "-1", "0,"):
        os.environ["NYQ_DEVICES"] = bad
        try:
            assert host.nyqh_batch_decode_files(files, sizes, 1, 1, ns, None, 0) == -1, bad
            assert b"device list" in host.nyqh_last_error(), (bad, host.nyqh_last_error())
        finally:
            del os.environ["NYQ_DEVICES"]
