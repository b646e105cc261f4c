"""Corpus of CELT-only Ogg Opus files made with the REFERENCE's encoder (oracle/gen_opus_corpus.py): frame
sizes 2.5 / 5 / 10 / 20 ms, mono / stereo, 12 ... 256 kbit/s, CBR / VBR, narrowband ... fullband.
  * CPU tier: the host entropy decoder must leave the range coder, after EVERY frame, in exactly the state the
    reference encoder recorded (OPUS_GET_FINAL_RANGE -- the Opus conformance criterion for the bit-exact
    half of the decoder: a single mis-decoded symbol anywhere in a frame changes it);
  * GPU tier: NyquistIO::Load of every file (CPU entropy stage + MI355X IMDCT / post-filter / de-emphasis)
    against the reference decoder's PCM: the committed digest, and sample for sample when the reference
    build (oracle/_ref/libref_decode.so) is next to the tests."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rel_rms
from test_host_decoder import entropy_decode, load_host

NAMES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))
               if not os.path.basename(p).startswith("unsupported_"))
FAMILY0 = [n for n in NAMES if not n.startswith(("surround", "twosize"))]   # one elementary stream, one frame size


@pytest.fixture(scope="module")
def host():
    return load_host()


@pytest.fixture(scope="module")
def digest():
    return np.load(os.path.join(GOLDEN, "corpus_digest.npz"))


def test_corpus_is_complete(digest):
    assert len(FAMILY0) == 18 and len(NAMES) == 21
    sizes = {int(digest[n + "/meta"][1]) for n in FAMILY0}
    assert sizes == {120, 240, 480, 960}                    # every CELT frame size
    assert {int(digest[n + "/meta"][0]) for n in NAMES} == {1, 2, 6, 8}
    # 7.1 = 5 streams, 3 coupled: the shape SURVEY section 8(d) expects of BASELINE config 5's missing 8-channel file
    assert [int(v) for v in digest["surround71_20ms_320k/meta"][4:6]] == [5, 3]


@pytest.mark.parametrize("name", FAMILY0)
def test_entropy_decoder_final_range_matches_reference_encoder(host, digest, name):
    raw = open(os.path.join(GOLDEN, "corpus", name + ".opus"), "rb").read()
    ch, frame, nsamp, nbytes = (int(v) for v in digest[name + "/meta"][:4])
    assert nbytes == len(raw)
    want = digest[name + "/ranges"]
    rc, freq, flags, gain, rng, info = entropy_decode(host, raw, max_frames=len(want) + 4, channels=ch, n=frame)
    assert rc == 0
    assert int(info[0]) == ch and int(info[2]) == len(want)
    assert np.array_equal(rng[:len(want)], want)
    assert np.isfinite(freq[:len(want)]).all()
    assert (flags[:len(want), 3] == {120: 0, 240: 1, 480: 2, 960: 3}[frame]).all()


@pytest.mark.parametrize("name", FAMILY0)
def test_entropy_stage_freq_matches_the_reference_decoder(host, digest, name):
    """freq[] of EVERY frame of every corpus file against what the reference decoder handed to its IMDCT
    (corpus_freq_digest.npz, oracle/gen_corpus_freq_digest.py: per clt_mdct_backward call the block's sum, energy and
    eight picked coefficients).  The final-range test above pins the symbols; this one pins the float half of the
    entropy stage (folding, spreading, resolution changes, stereo merging, denormalisation) on the CPU tier."""
    fd = np.load(os.path.join(GOLDEN, "corpus_freq_digest.npz"))
    raw = open(os.path.join(GOLDEN, "corpus", name + ".opus"), "rb").read()
    ch, frame = (int(v) for v in digest[name + "/meta"][:2])
    nf = len(digest[name + "/ranges"])
    rc, freq, flags, gain, rng, info = entropy_decode(host, raw, max_frames=nf + 4, channels=ch, n=frame)
    assert rc == 0
    M = frame // 120
    blocks = []                                              # the reference's call order (celt_decoder_clean.c:286-311)
    for f in range(nf):
        if flags[f, 0] and M == 8 and ch == 2:
            blocks += [freq[f, c, b::M] for b in range(M) for c in range(ch)]
        elif flags[f, 0]:
            blocks += [freq[f, c, b::M] for c in range(ch) for b in range(M)]
        else:
            blocks += [freq[f, c] for c in range(ch)]
    shape, sums, pick = fd[name + "/shape"], fd[name + "/sums"], fd[name + "/pick"]
    assert len(blocks) == len(shape)
    scale = max(float(np.abs(freq[:nf]).max()), 1.0)
    for k, x in enumerate(blocks):
        assert x.size == shape[k, 2]
        x64 = x.astype(np.float64)
        assert abs(x64.sum() - sums[k, 0]) <= 2e-6 * scale * x.size, (k, x64.sum(), sums[k, 0])
        assert abs((x64 ** 2).sum() - sums[k, 1]) <= 1e-5 * max(sums[k, 1], 1.0), k
        assert np.abs(x[(np.arange(8) * x.size) // 16] - pick[k]).max() <= 2e-6 * scale, k


def _load(host, raw):
    info = np.zeros(8, np.int64)
    n = host.nyqh_nyquistio_load_buffer(raw, len(raw), None, 0, info)
    assert n > 0
    out = np.zeros(n, np.float32)
    assert host.nyqh_nyquistio_load_buffer(raw, len(raw), out.ctypes.data_as(C.c_void_p), n, info) == n
    return out, info


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_plugin_surface_decodes_corpus_like_the_reference(host, digest, name):
    raw = open(os.path.join(GOLDEN, "corpus", name + ".opus"), "rb").read()
    ch, frame, nsamp, _ = (int(v) for v in digest[name + "/meta"][:4])
    got, info = _load(host, raw)
    assert int(info[0]) == ch and int(info[1]) == 48000
    assert got.size == nsamp
    want5 = digest[name + "/every5"]
    assert rel_rms(got[::5], want5) <= 1e-5
    assert np.abs(got[::5] - want5).max() <= 4e-6           # samples are in [-1, 1)
    s, ss = digest[name + "/sum"]
    assert abs(float(got.astype(np.float64).sum()) - s) <= 1e-3 * max(1.0, abs(s)) + 2e-2
    assert abs(float((got.astype(np.float64) ** 2).sum()) - ss) <= 1e-4 * ss
    ref = os.path.join(ROOT, "oracle", "_ref", "libref_decode.so")
    if os.path.exists(ref):                                  # the reference itself, sample for sample
        R = C.CDLL(ref)
        R.ref_decode_pcm.restype = C.c_long
        R.ref_decode_pcm.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
        full = np.zeros(nsamp, np.float32)
        assert R.ref_decode_pcm(raw, len(raw), full.ctypes.data_as(C.c_void_p), nsamp, None) == nsamp
        assert rel_rms(got, full) <= 1e-5
        assert np.abs(got - full).max() <= 4e-6


def test_silk_stream_is_refused_not_misdecoded(host):
    """SILK / hybrid packets are outside the CELT path this library accelerates: the entropy stage reports
    them (TOC configuration < 16) instead of producing audio."""
    raw = open(os.path.join(GOLDEN, "corpus", "unsupported_silk_voip_12k.opus"), "rb").read()
    rc = entropy_decode(host, raw, max_frames=8, channels=1, n=960)[0]
    assert rc == -11


@pytest.mark.gpu
def test_batch_with_a_silk_stream_fails_only_that_stream(host):
    raw_bad = open(os.path.join(GOLDEN, "corpus", "unsupported_silk_voip_12k.opus"), "rb").read()
    info = np.zeros(8, np.int64)
    assert host.nyqh_nyquistio_load_buffer(raw_bad, len(raw_bad), None, 0, info) == -1     # std::runtime_error
    raw_ok = open(os.path.join(GOLDEN, "corpus", "mono_20ms_64k.opus"), "rb").read()
    assert host.nyqh_nyquistio_load_buffer(raw_ok, len(raw_ok), None, 0, info) == 57600
