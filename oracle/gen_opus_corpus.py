#!/usr/bin/env python3
"""oracle/gen_opus_corpus.py -- TEST INFRASTRUCTURE.  Builds a small corpus of CELT-only Ogg Opus files with
the REFERENCE's own encoder (opus_encode_float of the reference build in oracle/_ref/libref_decode.so,
OPUS_APPLICATION_RESTRICTED_LOWDELAY = CELT only) over the parameter space the three bundled files do not
reach: frame sizes 2.5 / 5 / 10 / 20 ms, mono and stereo, 12 ... 256 kbit/s, CBR / VBR, band-limited modes;
and records, per file, what the reference decoder makes of it:
  * the encoder's final range-coder state after every packet (OPUS_GET_FINAL_RANGE) -- the Opus conformance
    hook: a bit-exact entropy decoder ends every frame in exactly that state;
  * the reference's NyquistIO::Load output: length, sum, sum of squares and every 5th sample.
Outputs: tests/golden/corpus/<name>.opus and tests/golden/corpus_digest.npz.  Run from the repo root in the
build container (needs oracle/_ref, i.e. /root/reference); the generated files are data and are committed."""
import ctypes as C
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oggopus  # noqa: E402

R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_decode.so"))
R.opus_encoder_create.restype = C.c_void_p
R.opus_encoder_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
R.opus_encode_float.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_char_p, C.c_int]
R.opus_encoder_destroy.argtypes = [C.c_void_p]
R.ref_decode_pcm.restype = C.c_long
R.ref_decode_pcm.argtypes = [C.c_char_p, C.c_long, C.POINTER(C.c_float), C.c_long, C.POINTER(C.c_long)]

RESTRICTED_LOWDELAY = 2051
SET_BITRATE, SET_VBR, SET_BANDWIDTH, SET_COMPLEXITY, GET_LOOKAHEAD, GET_FINAL_RANGE, SET_VBR_CONSTRAINT = 4002, 4006, 4008, 4010, 4027, 4031, 4020
BW = {"nb": 1101, "wb": 1103, "swb": 1104, "fb": 1105}


def ctl_set(enc, req, val):
    rc = R.opus_encoder_ctl(C.c_void_p(enc), C.c_int(req), C.c_int(val))
    assert rc == 0, (req, val, rc)


def ctl_get(enc, req, ctype=C.c_int):
    v = ctype(0)
    rc = R.opus_encoder_ctl(C.c_void_p(enc), C.c_int(req), C.byref(v))
    assert rc == 0, (req, rc)
    return v.value


def signal(seconds, channels, seed):
    """tones + a sweep + castanet-like clicks (transient frames) + a silent gap + a noise tail"""
    rng = np.random.default_rng(seed)
    n = int(48000 * seconds)
    t = np.arange(n) / 48000.0
    chans = []
    for c in range(channels):
        f0 = 220.0 * (1 + 0.5 * c)
        x = 0.25 * np.sin(2 * np.pi * f0 * t) + 0.12 * np.sin(2 * np.pi * 3.01 * f0 * t + c)
        x += 0.15 * np.sin(2 * np.pi * (300 + 5000 * t / seconds) * t)
        for k in range(int(seconds * 6)):                      # clicks: sharp exponentially decaying bursts
            p = int(rng.integers(0, n - 2000))
            L = 1500
            x[p:p + L] += 0.6 * rng.standard_normal(L) * np.exp(-np.arange(L) / 120.0)
        g0, g1 = int(0.45 * n), int(0.55 * n)
        x[g0:g1] = 0.0                                          # digital silence
        x[int(0.8 * n):] += 0.05 * rng.standard_normal(n - int(0.8 * n))
        chans.append(x)
    return np.stack(chans, axis=1).astype(np.float32)          # [n][channels]


def encode(name, channels, frame, bitrate, vbr, bw, complexity, seconds, seed, application=RESTRICTED_LOWDELAY):
    err = C.c_int(0)
    enc = R.opus_encoder_create(48000, channels, application, C.byref(err))
    assert enc and err.value == 0
    ctl_set(enc, SET_BITRATE, bitrate)
    ctl_set(enc, SET_VBR, 1 if vbr else 0)
    if vbr:
        ctl_set(enc, SET_VBR_CONSTRAINT, 0)
    ctl_set(enc, SET_COMPLEXITY, complexity)
    if bw != "fb":
        ctl_set(enc, SET_BANDWIDTH, BW[bw])
    preskip = ctl_get(enc, GET_LOOKAHEAD)
    pcm = signal(seconds, channels, seed)
    total = pcm.shape[0]
    nfr = (total + preskip + frame - 1) // frame               # flush the encoder's look-ahead
    padded = np.zeros((nfr * frame, channels), np.float32)
    padded[:total] = pcm
    packets, ranges = [], []
    buf = C.create_string_buffer(4000)
    for i in range(nfr):
        blk = np.ascontiguousarray(padded[i * frame:(i + 1) * frame])
        nb = R.opus_encode_float(enc, blk.ctypes.data_as(C.POINTER(C.c_float)), frame, buf, 4000)
        assert nb > 0, nb
        packets.append(buf.raw[:nb])
        ranges.append(ctl_get(enc, GET_FINAL_RANGE, C.c_uint))
    R.opus_encoder_destroy(enc)
    raw = oggopus.mux_family0(packets, channels, preskip, frame, total)
    return raw, np.array(ranges, np.uint32)


def encode_surround(name, channels, frame, bitrate, seconds, seed):
    """channel mapping family 1 through the reference's surround encoder (opus_multistream_encoder.c): 5.1 =
    4 streams (2 coupled), 7.1 = 5 streams (3 coupled) -- the shape of BASELINE config 5's 8-channel file."""
    R.opus_multistream_surround_encoder_create.restype = C.c_void_p
    R.opus_multistream_surround_encoder_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                           C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    R.opus_multistream_encode_float.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_char_p, C.c_int]
    R.opus_multistream_encoder_destroy.argtypes = [C.c_void_p]
    streams, coupled, err = C.c_int(0), C.c_int(0), C.c_int(0)
    mapping = C.create_string_buffer(256)
    enc = R.opus_multistream_surround_encoder_create(48000, channels, 1, C.byref(streams), C.byref(coupled), mapping,
                                                     RESTRICTED_LOWDELAY, C.byref(err))
    assert enc and err.value == 0, err.value

    def mctl_set(req, val):
        assert R.opus_multistream_encoder_ctl(C.c_void_p(enc), C.c_int(req), C.c_int(val)) == 0

    mctl_set(SET_BITRATE, bitrate)
    look = C.c_int(0)
    assert R.opus_multistream_encoder_ctl(C.c_void_p(enc), C.c_int(GET_LOOKAHEAD), C.byref(look)) == 0
    preskip = look.value
    pcm = signal(seconds, channels, seed)
    pcm[:, 3 if channels > 3 else 0] *= 0.3                     # the LFE channel: quieter
    total = pcm.shape[0]
    nfr = (total + preskip + frame - 1) // frame
    padded = np.zeros((nfr * frame, channels), np.float32)
    padded[:total] = pcm
    packets = []
    buf = C.create_string_buffer(20000)
    for i in range(nfr):
        blk = np.ascontiguousarray(padded[i * frame:(i + 1) * frame])
        nb = R.opus_multistream_encode_float(enc, blk.ctypes.data_as(C.POINTER(C.c_float)), frame, buf, 20000)
        assert nb > 0, nb
        packets.append(buf.raw[:nb])
    R.opus_multistream_encoder_destroy(enc)
    head = (b"OpusHead" + bytes([1, channels]) + struct.pack("<HIh", preskip, 48000, 0) + bytes([1, streams.value, coupled.value])
            + mapping.raw[:channels])
    raw = oggopus.mux_packets(head, packets, preskip, frame, total, per_page=2)
    return raw, streams.value, coupled.value


def encode_two_sizes(seconds, seed, bitrate=64000):
    """stereo, 20 ms frames and then ONE closing 10 ms frame: long enough (10 s) for the batch decoder to walk
    it in time slices when 16 copies share a piece, with a later segment of another frame size on top"""
    err = C.c_int(0)
    enc = R.opus_encoder_create(48000, 2, RESTRICTED_LOWDELAY, C.byref(err))
    assert enc and err.value == 0
    ctl_set(enc, SET_BITRATE, bitrate)
    preskip = ctl_get(enc, GET_LOOKAHEAD)
    pcm = signal(seconds, 2, seed)
    total = pcm.shape[0]
    n20 = (total + preskip) // 960                               # whole 20 ms frames, then 10 ms ones for the rest
    rest = total + preskip - n20 * 960
    n10 = (rest + 479) // 480
    padded = np.zeros((n20 * 960 + n10 * 480, 2), np.float32)
    padded[:total] = pcm
    packets, sizes = [], []
    buf = C.create_string_buffer(4000)
    pos = 0
    for frame in [960] * n20 + [480] * n10:
        blk = np.ascontiguousarray(padded[pos:pos + frame])
        nb = R.opus_encode_float(enc, blk.ctypes.data_as(C.POINTER(C.c_float)), frame, buf, 4000)
        assert nb > 0
        packets.append(buf.raw[:nb])
        sizes.append(frame)
        pos += frame
    R.opus_encoder_destroy(enc)
    head = b"OpusHead" + bytes([1, 2]) + struct.pack("<HIh", preskip, 48000, 0) + bytes([0])
    return oggopus.mux_packets_sized(head, packets, sizes, preskip, total), n20, n10


SURROUND = [
    ("surround71_20ms_320k", 8, 960, 320000, 1.0),
    ("surround51_10ms_192k", 6, 480, 192000, 0.8),
]

CORPUS = [
    # name             ch frame  bitrate vbr   bw   cx  seconds
    ("st_20ms_128k",    2, 960, 128000, True, "fb", 10, 1.2),
    ("st_20ms_32k",     2, 960,  32000, True, "fb", 10, 1.2),
    ("st_20ms_12k_cbr", 2, 960,  12000, False, "fb", 5, 1.0),
    ("st_10ms_96k",     2, 480,  96000, True, "fb", 10, 1.0),
    ("st_5ms_96k",      2, 240,  96000, True, "fb", 10, 0.8),
    ("st_2p5ms_128k",   2, 120, 128000, True, "fb", 10, 0.6),
    ("st_20ms_256k_cbr", 2, 960, 256000, False, "fb", 10, 1.0),
    ("mono_20ms_64k",   1, 960,  64000, True, "fb", 10, 1.2),
    ("mono_20ms_16k",   1, 960,  16000, True, "fb", 10, 1.0),
    ("mono_10ms_24k_cbr", 1, 480, 24000, False, "fb", 3, 1.0),
    ("mono_5ms_64k",    1, 240,  64000, True, "fb", 10, 0.8),
    ("mono_2p5ms_48k",  1, 120,  48000, True, "fb", 10, 0.6),
    ("st_20ms_48k_swb", 2, 960,  48000, True, "swb", 10, 1.0),
    ("st_20ms_32k_wb",  2, 960,  32000, True, "wb", 10, 1.0),
    ("mono_20ms_16k_nb", 1, 960, 16000, True, "nb", 10, 1.0),
    ("st_10ms_20k_cx0", 2, 480,  20000, True, "fb", 0, 1.0),
]


def main():
    out_dir = os.path.join(ROOT, "tests", "golden", "corpus")
    os.makedirs(out_dir, exist_ok=True)
    dig = {}
    for k, (name, ch, frame, br, vbr, bw, cx, secs) in enumerate(CORPUS):
        raw, ranges = encode(name, ch, frame, br, vbr, bw, cx, secs, 1000 + k)
        open(os.path.join(out_dir, name + ".opus"), "wb").write(raw)
        info = (C.c_long * 3)()
        n = R.ref_decode_pcm(raw, len(raw), None, 0, info)
        assert n > 0, name
        pcm = np.zeros(n, np.float32)
        assert R.ref_decode_pcm(raw, len(raw), pcm.ctypes.data_as(C.POINTER(C.c_float)), n, info) == n
        dig[name + "/ranges"] = ranges
        dig[name + "/meta"] = np.array([ch, frame, n, len(raw)], np.int64)
        dig[name + "/sum"] = np.array([pcm.astype(np.float64).sum(), (pcm.astype(np.float64) ** 2).sum()])
        dig[name + "/every5"] = pcm[::5].copy()
        print(f"{name}: {len(raw)} bytes, {len(ranges)} packets, {n} samples, rms {np.sqrt((pcm.astype(np.float64)**2).mean()):.4f}")
    # a SILK stream (VOIP application at 12 kbit/s): NOT decodable by this library by design -- the error path
    raw, _ = encode("silk_voip_12k", 1, 960, 12000, True, "fb", 5, 0.4, 3000, application=2048)
    open(os.path.join(out_dir, "unsupported_silk_voip_12k.opus"), "wb").write(raw)
    # stereo-coded packets under a MONO OpusHead: legal (the stereo flag is per packet, RFC 6716 section 3.1), the
    # decoder decodes both channels and mixes them down (celt_decoder_clean.c:648-652)
    k = [c[0] for c in CORPUS].index("st_20ms_32k")
    cname, cch, cframe, cbr, cvbr, cbw, ccx, csecs = CORPUS[k]
    raw_st, ranges_st = encode(cname, cch, cframe, cbr, cvbr, cbw, ccx, csecs, 1000 + k)
    pk, _, _ = oggopus.read_packets(raw_st)
    head_st, audio = pk[0], pk[2:]
    preskip_st = struct.unpack("<H", head_st[10:12])[0]
    raw = oggopus.mux_family0(audio, 1, preskip_st, cframe, int(48000 * csecs))
    name = "monohead_st_20ms_32k"
    open(os.path.join(out_dir, name + ".opus"), "wb").write(raw)
    info = (C.c_long * 3)()
    n = R.ref_decode_pcm(raw, len(raw), None, 0, info)
    assert n > 0 and info[0] == 1, (n, info[0])
    pcm = np.zeros(n, np.float32)
    assert R.ref_decode_pcm(raw, len(raw), pcm.ctypes.data_as(C.POINTER(C.c_float)), n, info) == n
    dig[name + "/ranges"] = ranges_st
    dig[name + "/meta"] = np.array([1, cframe, n, len(raw)], np.int64)
    dig[name + "/sum"] = np.array([pcm.astype(np.float64).sum(), (pcm.astype(np.float64) ** 2).sum()])
    dig[name + "/every5"] = pcm[::5].copy()
    print(f"{name}: {len(raw)} bytes, {len(audio)} stereo-coded packets under a mono header, {n} samples")
    # the same stereo stream with an output gain of -4.5 dB in the header (Q7.8 = -1152; RFC 7845 section 5.1,
    # applied by opusfile's default OP_HEADER_GAIN and by OPUS_SET_GAIN, opus_decoder_clean.c:700-712)
    head_gain = head_st[:16] + struct.pack("<h", -1152) + head_st[18:]
    raw = oggopus.mux_packets(head_gain, audio, preskip_st, cframe, int(48000 * csecs))
    name = "gain_st_20ms_32k"
    open(os.path.join(out_dir, name + ".opus"), "wb").write(raw)
    n = R.ref_decode_pcm(raw, len(raw), None, 0, info)
    assert n > 0 and info[0] == 2, (n, info[0])
    pcm = np.zeros(n, np.float32)
    assert R.ref_decode_pcm(raw, len(raw), pcm.ctypes.data_as(C.POINTER(C.c_float)), n, info) == n
    dig[name + "/ranges"] = ranges_st
    dig[name + "/meta"] = np.array([2, cframe, n, len(raw)], np.int64)
    dig[name + "/sum"] = np.array([pcm.astype(np.float64).sum(), (pcm.astype(np.float64) ** 2).sum()])
    dig[name + "/every5"] = pcm[::5].copy()
    print(f"{name}: {len(raw)} bytes, header gain -4.5 dB, {n} samples, rms {np.sqrt((pcm.astype(np.float64)**2).mean()):.4f}")
    raw, n20, n10 = encode_two_sizes(10.0, 4000)
    name = "twosize_st_20ms_then_10ms_10s"
    open(os.path.join(out_dir, name + ".opus"), "wb").write(raw)
    info = (C.c_long * 3)()
    n = R.ref_decode_pcm(raw, len(raw), None, 0, info)
    assert n > 0 and n10 >= 1, (n, n10)
    pcm = np.zeros(n, np.float32)
    assert R.ref_decode_pcm(raw, len(raw), pcm.ctypes.data_as(C.POINTER(C.c_float)), n, info) == n
    dig[name + "/meta"] = np.array([2, 960, n, len(raw), n20, n10], np.int64)
    dig[name + "/sum"] = np.array([pcm.astype(np.float64).sum(), (pcm.astype(np.float64) ** 2).sum()])
    dig[name + "/every5"] = pcm[::5].copy()
    print(f"{name}: {len(raw)} bytes, {n20} x 20 ms + {n10} x 10 ms frames, {n} samples")
    for k, (name, ch, frame, br, secs) in enumerate(SURROUND):
        raw, nstreams, ncoupled = encode_surround(name, ch, frame, br, secs, 2000 + k)
        open(os.path.join(out_dir, name + ".opus"), "wb").write(raw)
        info = (C.c_long * 3)()
        n = R.ref_decode_pcm(raw, len(raw), None, 0, info)
        assert n > 0, name
        pcm = np.zeros(n, np.float32)
        assert R.ref_decode_pcm(raw, len(raw), pcm.ctypes.data_as(C.POINTER(C.c_float)), n, info) == n
        dig[name + "/meta"] = np.array([ch, frame, n, len(raw), nstreams, ncoupled], np.int64)
        dig[name + "/sum"] = np.array([pcm.astype(np.float64).sum(), (pcm.astype(np.float64) ** 2).sum()])
        dig[name + "/every5"] = pcm[::5].copy()
        print(f"{name}: {len(raw)} bytes, {nstreams} streams ({ncoupled} coupled), {n} samples, rms {np.sqrt((pcm.astype(np.float64)**2).mean()):.4f}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "corpus_digest.npz"), **dig)


if __name__ == "__main__":
    main()
