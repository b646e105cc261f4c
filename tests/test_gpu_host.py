"""GPU tier: the plugin surface end to end.  NyquistIO::Load() of the MI355X build (CPU entropy stage
+ batched GPU IMDCT / post-filter / de-emphasis) against the reference decoder's own output for the
bundled test file, and the batched multi-stream loader."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from test_host_decoder import load_host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    import torch  # noqa: F401  (same HIP runtime for every library of the process)
    return load_host()


def test_nyquistio_load_matches_reference_decoder(host):
    """BASELINE config 1 on the GPU path: decode test_data/short.opus via NyquistIO and compare with
    the reference's AudioData (size 421930; every sample within 2e-6)."""
    d = np.load(os.path.join(GOLDEN, "short_opus_digest.npz"))
    path = os.path.join(GOLDEN, "short.opus").encode()
    info = np.zeros(4, np.int64)
    n = host.nyqh_nyquistio_load(path, None, 0, info)
    assert n == int(d["samples"]) == 421930
    assert list(info) == [2, 48000, 64, 4]          # channelCount, sampleRate, frameSize = 2*32, lengthSeconds = 210965 // 48000
    buf = np.zeros(n, np.float32)
    assert host.nyqh_nyquistio_load(path, buf.ctypes.data_as(C.c_void_p), n, info) == n
    got = buf.reshape(-1, 2)
    want = d["final"]
    assert got.shape == want.shape                   # incl. the closing 2.5 ms frame and the end trim
    assert np.abs(got - want).max() <= 2e-6
    # the reference's own end-to-end check is a float sum of all samples (examples/src/Main.cpp:137-154)
    assert abs(float(got.sum(dtype=np.float64)) - float(want.sum(dtype=np.float64))) <= 1e-3


def test_batched_streams_are_identical_to_single(host):
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    n = 421930
    first = np.zeros(n, np.float32)
    last = np.zeros(n, np.float32)
    stats = np.zeros(4, np.float64)
    got = host.nyqh_batch_decode(raw, len(raw), 48, 8, first.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), n, stats)
    assert got == n
    assert np.array_equal(first, last)               # batch position does not matter
    d = np.load(os.path.join(GOLDEN, "short_opus_digest.npz"))
    assert np.abs(first - d["final"].reshape(-1)).max() <= 2e-6
    assert stats[2] == 48 * 221                      # 220 frames of 20 ms and one closing 2.5 ms frame per stream


def test_example_program_reports_reference_numbers():
    """libnyquist_amd/nyq_decode = the reference's examples/src/Main.cpp for this build."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "libnyquist_amd", "nyq_decode")
    r = subprocess.run([exe, os.path.join(GOLDEN, "short.opus")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "len: 421930" in r.stdout and "channels: 2 rate: 48000" in r.stdout
    d = np.load(os.path.join(GOLDEN, "short_opus_digest.npz"))
    want = float(np.float32(0) + d["final"].reshape(-1).astype(np.float32).sum(dtype=np.float32))
    got = float(r.stdout.split("sum:")[1].split()[0])
    assert abs(got - want) < 0.05          # float accumulation order differs; Main.cpp compares (int)sum


def test_multistream_four_channels_matches_reference(host):
    """Channel mapping family 1 (opus_multistream_decoder.c:184-331): a 4-channel file with two coupled
    streams and mapping [2,0,3,1], muxed from short.opus' packets; ground truth = the reference decoder's
    output for the very same file (digests in multistream_digest.npz)."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import oggopus
    d = np.load(os.path.join(GOLDEN, "multistream_digest.npz"))
    pk, _gr, last = oggopus.read_packets(open(os.path.join(GOLDEN, "short.opus"), "rb").read())
    aud = pk[2:]
    r = int(d["rotate"])
    ms = oggopus.mux_family1([aud, aud[r:220] + aud[:r] + aud[220:]], 2, [int(v) for v in d["mapping"]], 312,
                             [(i + 1) * 960 for i in range(220)] + [last])
    info = np.zeros(4, np.int64)
    n = host.nyqh_nyquistio_load_buffer(ms, len(ms), None, 0, info)
    assert n == int(d["samples"]) and int(info[0]) == 4
    buf = np.zeros(n, np.float32)
    assert host.nyqh_nyquistio_load_buffer(ms, len(ms), buf.ctypes.data_as(C.c_void_p), n, info) == n
    got = buf.reshape(-1, 4)
    assert np.abs(got[:9600] - d["head"]).max() <= 2e-6
    assert np.abs(got[-2000:] - d["tail"]).max() <= 2e-6
    nb = got.shape[0] // 960
    bs = got[: nb * 960].astype(np.float64).reshape(nb, 960, 4).sum(axis=1)
    bq = (got[: nb * 960].astype(np.float64) ** 2).reshape(nb, 960, 4).sum(axis=1)
    assert np.abs(bs - d["block_sum"]).max() <= 2e-3
    assert (np.abs(bq - d["block_sq"]) / np.maximum(d["block_sq"], 1e-9)).max() <= 1e-4
    # stream A's channels are the stereo file's (mapping 0 -> out 1, 1 -> out 3)
    st = np.load(os.path.join(GOLDEN, "short_opus_digest.npz"))["final"]
    assert np.abs(got[:, 1] - st[:, 0]).max() <= 2e-6 and np.abs(got[:, 3] - st[:, 1]).max() <= 2e-6


@pytest.mark.gpu
def test_corrupted_files_never_crash_the_batch_path(host):
    """Random byte damage anywhere in real files (container, headers, packet framing, entropy-coded payload):
    NyquistIO::Load either decodes SOMETHING of plausible size or reports an error -- no crash, no hang, no
    out-of-range sample count.  (The CPU tier fuzzes the parser and the entropy decoder alone; this adds the
    scan / layout / GPU pieces / trimming around them.)"""
    rng = np.random.default_rng(2024)
    srcs = ["corpus/st_20ms_32k.opus", "corpus/mono_5ms_64k.opus", "corpus/surround51_10ms_192k.opus", "short.opus"]
    info = np.zeros(8, np.int64)
    ok = bad = 0
    for it in range(120):
        raw = bytearray(open(os.path.join(GOLDEN, srcs[it % len(srcs)]), "rb").read())
        mode = it % 3
        if mode == 0:                                   # flip a few bytes anywhere
            for _ in range(int(rng.integers(1, 6))):
                raw[int(rng.integers(0, len(raw)))] = int(rng.integers(0, 256))
        elif mode == 1:                                 # truncate
            raw = raw[: int(rng.integers(1, len(raw)))]
        else:                                           # damage inside the first audio pages only
            lo = min(len(raw) - 1, 120)
            for _ in range(int(rng.integers(1, 12))):
                raw[int(rng.integers(lo, min(len(raw), lo + 4000)))] ^= 1 << int(rng.integers(0, 8))
        n = host.nyqh_nyquistio_load_buffer(bytes(raw), len(raw), None, 0, info)
        assert n == -1 or n == -2 or 0 <= n <= 8 * 48000 * 10
        ok += n >= 0
        bad += n < 0
    assert ok > 0 and bad > 0          # both outcomes occur: the damage is neither always fatal nor always ignored


@pytest.mark.gpu
@pytest.mark.parametrize("name,want_int_sum", [("sb-reverie.opus", 403), ("sb-reverie-60ms-frames.opus", 719)])
def test_reference_ctest_criteria(host, name, want_int_sum):
    """The reference's own integration tests (CMakeLists.txt:199-217 run examples/src/Main.cpp on these files):
    pass iff samples.size() == 21472602 and (int) of the float sum, accumulated channel after channel, is 403
    (20 ms packets) / 719 (60 ms packets) -- Main.cpp:136-148.  Same files, same criterion, through this build's
    NyquistIO::Load (223.7 s of stereo audio: 11184 frames on the CPU entropy stage, then the GPU stages)."""
    raw = open(os.path.join(GOLDEN, name), "rb").read()
    info = np.zeros(8, np.int64)
    n = host.nyqh_nyquistio_load_buffer(raw, len(raw), None, 0, info)
    assert n == 21472602
    out = np.zeros(n, np.float32)
    assert host.nyqh_nyquistio_load_buffer(raw, len(raw), out.ctypes.data_as(C.c_void_p), n, info) == n
    assert int(info[0]) == 2 and int(info[1]) == 48000
    pcm = out.reshape(-1, 2)
    s = np.float32(0)
    for c in range(2):                                   # sequential float32 accumulation, channel-major
        s = np.cumsum(np.concatenate([[s], pcm[:, c]]).astype(np.float32), dtype=np.float32)[-1]
    assert int(s) == want_int_sum
    assert np.isfinite(out).all() and np.abs(out).max() <= 1.5
    # ... and against the reference decoder's own PCM, ALWAYS: block digests made here from the reference build
    # (oracle/gen_reverie_digest.py -> tests/golden/sb_reverie_digest.npz), plus, where the reference build itself
    # travelled with the snapshot (oracle/_ref/libref_decode.so), every sample.
    dg = np.load(os.path.join(GOLDEN, "sb_reverie_digest.npz"))
    k = "a" if name == "sb-reverie.opus" else "b"
    assert str(dg[k + "_file"]) == name
    assert np.abs(pcm[:9600] - dg[k + "_head"]).max() <= 4e-6 and np.abs(pcm[-2000:] - dg[k + "_tail"]).max() <= 4e-6
    assert np.abs(pcm[::997] - dg[k + "_every997"]).max() <= 4e-6
    nb = dg[k + "_block_sum"].shape[0]
    blk = pcm[: nb * 4800].astype(np.float64).reshape(nb, 4800, 2)
    assert np.abs(blk.sum(axis=1) - dg[k + "_block_sum"]).max() <= 4800 * 4e-6
    assert np.abs((blk ** 2).sum(axis=1) - dg[k + "_block_sq"]).max() <= 1e-5 * max(1.0, float(dg[k + "_block_sq"].max()))
    ref = os.path.join(ROOT, "oracle", "_ref", "libref_decode.so")
    if os.path.exists(ref):
        R = C.CDLL(ref)
        R.ref_decode_pcm.restype = C.c_long
        R.ref_decode_pcm.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
        full = np.zeros(n, np.float32)
        assert R.ref_decode_pcm(raw, len(raw), full.ctypes.data_as(C.c_void_p), n, None) == n
        assert np.abs(out - full).max() <= 4e-6
        d = out.astype(np.float64) - full
        assert np.sqrt((d ** 2).mean()) <= 1e-5 * np.sqrt((full.astype(np.float64) ** 2).mean())


@pytest.mark.gpu
def test_reference_decoder_built_with_use_cuda_runs_on_this_library():
    """INTEGRATION.md section 1, executed: the reference's OWN decoder (src/OpusDependencies.c + src/*.cpp) compiled with
    plain gcc and -DUSE_CUDA -- so that third_party/opus/celt/mdct.c:219-254 forwards every clt_mdct_backward[_B1_C2] call
    to processMDCTCuda[B1C2] (cuda/mdct_cuda.hpp:89-94) -- and linked against libnyq_imdct.so (oracle/Makefile target
    libref_decode_dropin.so).  short.opus through THAT build must give the plain reference build's 421,930 samples
    within 1e-5 relative RMS: long frames (B1_C2, stride 1) and transient frames (eight stride-8 B1_C2 calls) alike.
    Both are reference builds made in the build container; they travel with the snapshot."""
    base = os.path.join(ROOT, "oracle", "_ref")
    plain, dropin = os.path.join(base, "libref_decode.so"), os.path.join(base, "libref_decode_dropin.so")
    if not (os.path.exists(plain) and os.path.exists(dropin)):
        pytest.skip("reference builds not in the snapshot (make -C oracle needs /root/reference)")
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    n = 421930
    got = {}
    for key, path in (("plain", plain), ("dropin", dropin)):
        R = C.CDLL(path)
        R.ref_decode_pcm.restype = C.c_long
        R.ref_decode_pcm.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
        a = np.zeros(n, np.float32)
        assert R.ref_decode_pcm(raw, len(raw), a.ctypes.data_as(C.c_void_p), n, None) == n, key
        got[key] = a
    d = got["dropin"].astype(np.float64) - got["plain"]
    rel = np.sqrt((d ** 2).mean()) / np.sqrt((got["plain"].astype(np.float64) ** 2).mean())
    assert rel <= 1e-5, rel
    assert np.abs(d).max() <= 4e-6
    assert np.abs(got["dropin"]).max() > 0.05           # (real audio came out, not silence)


@pytest.mark.gpu
def test_config4_thousand_real_streams(host):
    """BASELINE config 4 at its stated stream count on REAL packets: 1000 copies of short.opus (221 frames each,
    1.7 GB of decoded audio) as one batch through nyqh_batch_decode -- the first and the last stream (and the frame
    count) must equal the reference decoder's output for the file (short_opus_digest.npz)."""
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    d = np.load(os.path.join(GOLDEN, "short_opus_digest.npz"))
    n = 421930
    first, last = np.zeros(n, np.float32), np.zeros(n, np.float32)
    stats = np.zeros(6, np.float64)
    got = host.nyqh_batch_decode_timed(raw, len(raw), 1000, 16, first.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), n, stats)
    assert got == n
    assert int(stats[2]) == 1000 * 221
    want = d["final"].reshape(-1)
    assert np.abs(first - want).max() <= 2e-6 and np.abs(last - want).max() <= 2e-6
    assert np.array_equal(first, last)


@pytest.mark.gpu
def test_long_streams_sliced_differently_decode_bit_identically(host):
    """A 224 s stream is walked by the GPU in time slices whose length depends on how many streams share a piece
    (one file alone: slices of 4352 frames; three files on two threads: pieces of two streams, 2176 frames).  The
    slices end on multiples of 64 frames, so the synthesis kernels chain the same frames in-wave either way and
    the decoded audio must not depend on the batch it was decoded in."""
    raw = open(os.path.join(GOLDEN, "sb-reverie.opus"), "rb").read()
    n = 21472602
    info = np.zeros(8, np.int64)
    alone = np.zeros(n, np.float32)
    assert host.nyqh_nyquistio_load_buffer(raw, len(raw), alone.ctypes.data_as(C.c_void_p), n, info) == n
    first, last = np.zeros(n, np.float32), np.zeros(n, np.float32)
    stats = np.zeros(4, np.float64)
    got = host.nyqh_batch_decode(raw, len(raw), 3, 2, first.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), n, stats)
    assert got == n and stats[2] == 3 * 11184
    assert np.array_equal(first, alone) and np.array_equal(last, alone)


@pytest.mark.gpu
def test_sliced_piece_with_a_later_segment(host):
    """Sixteen copies of a 10 s stream (500 frames of 20 ms and one closing 10 ms frame) on sixteen threads share
    one piece, which is long enough to be walked in time slices; the closing frame is a later segment that starts
    from the decoder state the last slice leaves behind.  Same samples as the file decoded alone (one slice)."""
    raw = open(os.path.join(GOLDEN, "corpus", "twosize_st_20ms_then_10ms_10s.opus"), "rb").read()
    n = 960000
    info = np.zeros(8, np.int64)
    alone = np.zeros(n, np.float32)
    assert host.nyqh_nyquistio_load_buffer(raw, len(raw), alone.ctypes.data_as(C.c_void_p), n, info) == n
    first, last = np.zeros(n, np.float32), np.zeros(n, np.float32)
    stats = np.zeros(4, np.float64)
    got = host.nyqh_batch_decode(raw, len(raw), 16, 16, first.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), n, stats)
    assert got == n and stats[2] == 16 * 501
    assert np.array_equal(first, alone) and np.array_equal(last, alone)


@pytest.mark.gpu
def test_mixed_batch_every_file_as_if_decoded_alone(host):
    """One batch of everything at once: mono and stereo, all four frame sizes, 5.1 and 7.1 multistream files, files
    that close with a frame of another size, very different lengths (so groups are padded and a short stream with a
    later segment needs its state recomputed), a SILK file that must fail alone, several copies of some files.
    Every file must come out exactly as from NyquistIO::Load on that file alone."""
    import glob
    paths = sorted(glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))) + [os.path.join(GOLDEN, "short.opus")]
    paths = paths + paths[::3] + [os.path.join(GOLDEN, "short.opus")] * 3
    raws = [open(p, "rb").read() for p in paths]
    alone = []
    info = np.zeros(8, np.int64)
    for r in raws:
        n = host.nyqh_nyquistio_load_buffer(r, len(r), None, 0, info)
        if n < 0:
            alone.append(None)
            continue
        a = np.zeros(n, np.float32)
        assert host.nyqh_nyquistio_load_buffer(r, len(r), a.ctypes.data_as(C.c_void_p), n, info) == n
        alone.append(a)
    assert sum(a is None for a in alone) == sum("unsupported_" in p for p in paths) >= 1      # only the SILK file fails
    cnt = len(raws)
    files = (C.c_char_p * cnt)(*raws)
    sizes = (C.c_long * cnt)(*[len(r) for r in raws])
    ns = (C.c_long * cnt)()
    cap = sum(a.size for a in alone if a is not None)
    for threads in (16, 3):
        out = np.zeros(cap, np.float32)
        total = host.nyqh_batch_decode_files(files, sizes, cnt, threads, ns, out.ctypes.data_as(C.c_void_p), cap)
        assert total == cap
        pos = 0
        for i, a in enumerate(alone):
            if a is None:
                assert ns[i] == -1
                continue
            assert ns[i] == a.size, paths[i]
            assert np.array_equal(out[pos:pos + a.size], a), paths[i]
            pos += a.size


@pytest.mark.gpu
def test_two_device_shards_on_one_gpu_equal_the_one_device_batch(host):
    """The real multi-device path on hardware: a device LIST {0, 0} makes two device shards -- two sets of six feeder
    threads and contexts, two page-locked staging arenas, streams dealt s mod 2 -- on the box's one GPU, through both
    entry points (nyqh_batch_decode_files with the list set, and nqr::BatchLoad(out, buffers, {0, 0})).  Every file must
    equal NyquistIO::Load of that file alone bit for bit, the decoder must report two devices, and a list with a device
    that does not exist must be refused up front."""
    import glob
    paths = sorted(glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))) + [os.path.join(GOLDEN, "short.opus")]
    paths = [p for p in paths if "unsupported_" not in p]
    paths = paths + paths[::3] + [os.path.join(GOLDEN, "short.opus")] * 3
    raws = [open(p, "rb").read() for p in paths]
    info = np.zeros(8, np.int64)
    alone = []
    for r in raws:
        n = host.nyqh_nyquistio_load_buffer(r, len(r), None, 0, info)
        a = np.zeros(n, np.float32)
        assert host.nyqh_nyquistio_load_buffer(r, len(r), a.ctypes.data_as(C.c_void_p), n, info) == n
        alone.append(a)
    cnt = len(raws)
    files = (C.c_char_p * cnt)(*raws)
    sizes = (C.c_long * cnt)(*[len(r) for r in raws])
    ns = (C.c_long * cnt)()
    cap = sum(a.size for a in alone)

    def check(out):
        pos = 0
        for i, a in enumerate(alone):
            assert ns[i] == a.size, paths[i]
            assert np.array_equal(out[pos:pos + a.size], a), paths[i]
            pos += a.size

    two = (C.c_int * 2)(0, 0)
    host.nyqh_set_devices(two, 2)
    try:
        out = np.zeros(cap, np.float32)
        assert host.nyqh_batch_decode_files(files, sizes, cnt, 8, ns, out.ctypes.data_as(C.c_void_p), cap) == cap, host.nyqh_last_error()
        assert host.nyqh_capi_device_count() == 2
        check(out)
        raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
        first, last = np.zeros(421930, np.float32), np.zeros(421930, np.float32)
        stats = np.zeros(6, np.float64)
        assert host.nyqh_batch_decode_timed(raw, len(raw), 64, 8, first.ctypes.data_as(C.c_void_p), last.ctypes.data_as(C.c_void_p), 421930, stats) == 421930
        assert stats[5] == 2 and np.array_equal(first, last)
    finally:
        host.nyqh_set_devices(two, 0)
    out = np.zeros(cap, np.float32)
    assert host.nyqh_batch_load_devices(files, sizes, cnt, two, 2, ns, out.ctypes.data_as(C.c_void_p), cap) == cap, host.nyqh_last_error()
    check(out)
    bad = (C.c_int * 2)(0, 97)
    assert host.nyqh_batch_load_devices(files, sizes, cnt, bad, 2, ns, None, 0) == -1
    assert b"does not exist" in host.nyqh_last_error()


@pytest.mark.gpu
def test_concurrent_loads_from_several_threads(host):
    """NyquistIO::Load from six host threads at once (each call leases its own decoder = its own GPU contexts and
    staging memory from the per-device pool): same samples as the sequential loads."""
    import glob
    import threading
    paths = [p for p in sorted(glob.glob(os.path.join(GOLDEN, "corpus", "*.opus"))) if "unsupported" not in p] + [os.path.join(GOLDEN, "short.opus")]
    raws = [open(p, "rb").read() for p in paths]
    want = []
    info = np.zeros(8, np.int64)
    for r in raws:
        n = host.nyqh_nyquistio_load_buffer(r, len(r), None, 0, info)
        a = np.zeros(n, np.float32)
        assert host.nyqh_nyquistio_load_buffer(r, len(r), a.ctypes.data_as(C.c_void_p), n, info) == n
        want.append(a)
    errors = []

    def worker(tid):
        inf = np.zeros(8, np.int64)
        for rep in range(4):
            for k in range(tid, len(raws), 3):
                out = np.zeros(want[k].size, np.float32)
                n = host.nyqh_nyquistio_load_buffer(raws[k], len(raws[k]), out.ctypes.data_as(C.c_void_p), out.size, inf)
                if n != want[k].size or not np.array_equal(out, want[k]):
                    errors.append((tid, rep, paths[k]))

    counts = (C.c_long * 2)()
    host.nyqh_decoder_pool_counts(counts)
    made0, gone0 = counts[0], counts[1]
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    # decoders are pooled: six loading threads make at most six decoders between them and tear none down (round 2 tore
    # down and rebuilt four of them -- 24 contexts -- per Load while the other threads were inside their GPU calls)
    host.nyqh_decoder_pool_counts(counts)
    assert counts[0] - made0 <= 6 and counts[1] - gone0 == 0, (counts[0] - made0, counts[1] - gone0)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"NYQ_HOST_PACKED": "1"}, {"NYQ_HOST_SYMBOLS": "0"}], ids=["packed-records", "freq-not-symbols"])
def test_staging_forms_decode_to_the_same_samples(host, env):
    """The batch decoder's two measurement switches, read once at a decoder's construction (so: a process of its own): symbol
    records packed back to back with one gather launch per window (NYQ_HOST_PACKED=1), and freq[] built on the host instead of
    symbol records (NYQ_HOST_SYMBOLS=0).  Packed and slotted records are the same records: bit for bit.  Host-built freq[]
    differs from the device's band shapes by the order of a few sums: 2e-6 of full scale."""
    import subprocess
    import sys
    raw = open(os.path.join(GOLDEN, "short.opus"), "rb").read()
    n = 421930
    first = np.zeros(n, np.float32)
    stats = np.zeros(4, np.float64)
    assert host.nyqh_batch_decode(raw, len(raw), 40, 8, first.ctypes.data_as(C.c_void_p), None, n, stats) == n
    code = ("import sys, ctypes as C, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_host_decoder import load_host\n"
            "H = load_host(); raw = open(%r, 'rb').read(); n = 421930\n"
            "a = np.zeros(n, np.float32); b = np.zeros(n, np.float32); st = np.zeros(4, np.float64)\n"
            "assert H.nyqh_batch_decode(raw, len(raw), 40, 8, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), n, st) == n\n"
            "assert np.array_equal(a, b); a.tofile(sys.argv[1])\n") % (ROOT, os.path.join(ROOT, "tests"), os.path.join(GOLDEN, "short.opus"))
    out = os.path.join("/tmp", "nyq_staging_form_%s.bin" % "_".join(env))
    r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    other = np.fromfile(out, np.float32)
    os.unlink(out)
    if "NYQ_HOST_PACKED" in env:
        assert np.array_equal(other, first)
    else:
        assert np.abs(other - first).max() <= 2e-6


def test_device_entropy_mode_of_the_batch_decoder_decodes_to_the_same_samples(host):
    """NYQ_DEVICE_ENTROPY=1 (read at a decoder's construction: a process of its own): streams of one frame size hand the GPU their
    frames' BYTES -- the host only walks the packets -- and the files come out as through the host entropy stage: long streams in
    time slices (sb-reverie), mono / stereo / 10 ms / 256 kbit/s corpus files, the 7.1 file's five elementary streams, and streams
    that change frame size (short.opus' closing 2.5 ms frame, 20 ms then 10 ms): every segment's entropy stage on the device, the
    stage's state carried from segment to segment."""
    import subprocess
    import sys
    names = ["sb-reverie.opus", "short.opus", "corpus/st_20ms_32k.opus", "corpus/mono_20ms_64k.opus", "corpus/st_10ms_96k.opus",
             "corpus/st_20ms_256k_cbr.opus", "corpus/surround71_20ms_320k.opus", "corpus/mono_2p5ms_48k.opus",
             "corpus/twosize_st_20ms_then_10ms_10s.opus", "corpus/surround51_10ms_192k.opus", "corpus/monohead_st_20ms_32k.opus"]
    code = ("import sys, ctypes as C, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_host_decoder import load_host\n"
            "H = load_host()\n"
            "H.nyqh_nyquistio_load_buffer.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, np.ctypeslib.ndpointer(np.int64)]\n"
            "H.nyqh_nyquistio_load_buffer.restype = C.c_long\n"
            "H.nyqh_device_entropy_frames.restype = C.c_long\n"
            "outs = []\n"
            "for p in sys.argv[2:]:\n"
            "    raw = open(p, 'rb').read(); info = np.zeros(4, np.int64)\n"
            "    n = H.nyqh_nyquistio_load_buffer(raw, len(raw), None, 0, info)\n"
            "    assert n > 0, (p, n)\n"
            "    a = np.zeros(n, np.float32)\n"
            "    assert H.nyqh_nyquistio_load_buffer(raw, len(raw), a.ctypes.data_as(C.c_void_p), n, info) == n\n"
            "    outs.append(a)\n"
            "np.concatenate(outs).tofile(sys.argv[1]); print('frames', H.nyqh_device_entropy_frames(), [len(o) for o in outs])\n") % (ROOT, os.path.join(ROOT, "tests"))
    paths = [os.path.join(GOLDEN, n) for n in names]
    res = {}
    for tag, env in (("host", {"NYQ_DEVICE_ENTROPY": "0"}), ("device", {"NYQ_DEVICE_ENTROPY": "1"})):
        out = os.path.join("/tmp", "nyq_device_entropy_%s.bin" % tag)
        r = subprocess.run([sys.executable, "-c", code, out] + paths, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("frames")][-1]
        res[tag] = (np.fromfile(out, np.float32), int(line.split()[1]), eval(line.split(" ", 2)[2]))
        os.unlink(out)
    assert res["host"][1] == 0 and res["device"][1] >= 2 * 11184                # (two loads per file; padding frames count too)
    assert res["host"][2] == res["device"][2]
    a, b = res["host"][0], res["device"][0]
    lens = res["host"][2]
    at = 0
    for n, name in zip(lens, names):
        d = np.abs(a[at:at + n] - b[at:at + n]).max()
        print(name, n, d)
        assert d <= 2e-6, name
        at += n
