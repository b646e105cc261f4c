"""GPU tier: the plugin surface end to end.  NyquistIO::Load() of the MI355X build (CPU entropy stage
+ batched GPU IMDCT / post-filter / de-emphasis) against the reference decoder's own output for the
bundled test file, and the batched multi-stream loader."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN
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
