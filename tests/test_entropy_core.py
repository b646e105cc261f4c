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
