"""The batch decoder's scheduler (decode threads publishing progress, feeder threads, time slices with carried
state, ordered hand-over of the samples, memory-bounded sub-batches) under ThreadSanitizer, against a CPU stand-in
for the GPU library (tests/sched/fake_gpu.cpp: its "PCM" depends on every frame, every parameter and the carried
state, so a slice delivered out of order or with the wrong state shows).  The real entropy decoder runs; no GPU."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

SCHED = os.path.join(ROOT, "tests", "sched")


@pytest.fixture(scope="module")
def harness():
    subprocess.run(["make", "-C", SCHED], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(SCHED, "sched_check_tsan")


def run(harness, files, **env):
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", **env)
    r = subprocess.run([harness] + [os.path.join(GOLDEN, f) for f in files], env=e, capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert "0 mismatches" in r.stdout
    return r.stdout


def test_mixed_batches_small_budget(harness):
    """mono / stereo / 5.1 / two frame sizes / a failing SILK file, 8-3-1 threads, sub-batches of 40 MB"""
    out = run(harness, ["corpus/st_20ms_32k.opus", "corpus/mono_5ms_64k.opus", "corpus/surround51_10ms_192k.opus",
                        "corpus/twosize_st_20ms_then_10ms_10s.opus", "short.opus", "corpus/unsupported_silk_voip_12k.opus",
                        "corpus/st_2p5ms_128k.opus"], NYQ_BATCH_BYTES="40000000")
    assert "6 batches" in out


def test_two_devices_equal_one_device(harness):
    """One BatchOpusDecoder over two devices of the stand-in (streams shard s mod 2, a feeder set and a staging arena
    per device, results through the sink form out of pooled buffers): bit-identical to the one-device results, both
    devices used, no race."""
    out = run(harness, ["corpus/st_20ms_32k.opus", "corpus/mono_5ms_64k.opus", "corpus/surround51_10ms_192k.opus",
                        "corpus/twosize_st_20ms_then_10ms_10s.opus", "short.opus", "corpus/st_2p5ms_128k.opus"],
              SCHED_THREADS="4", SCHED_REPS="1", SCHED_DEVICES="0,1", NYQ_BATCH_BYTES="60000000")
    assert "two devices: 18 of 18 files delivered" in out
    assert "2 batches" in out


def test_long_streams_in_time_slices(harness):
    """a 224 s stream is walked in time slices next to short ones"""
    out = run(harness, ["sb-reverie.opus", "corpus/st_20ms_32k.opus", "corpus/twosize_st_20ms_then_10ms_10s.opus", "short.opus"],
              SCHED_THREADS="4", SCHED_REPS="1")      # (one copy of each: two 224 s streams under TSan took 25-250 s here,
    assert "1 batches" in out                      # depending on the kernel's page-fault mood -- sys time, not ours)


def test_damaged_files_under_address_sanitizer():
    """Random damage (byte flips, truncation, bit flips in the audio pages) of real files -- among them a mono-header
    file with stereo-coded packets, a 5.1 file and one that changes its frame size -- through the whole batch path
    against the fake GPU, built with AddressSanitizer + UBSan.  (This is the harness that caught the decoder writing
    both channels of a stereo-coded packet into a mono stream's buffer.)"""
    subprocess.run(["make", "-C", SCHED, "asan"], check=True, stdout=subprocess.DEVNULL)
    files = ["corpus/monohead_st_20ms_32k.opus", "corpus/st_20ms_32k.opus", "corpus/mono_5ms_64k.opus", "corpus/surround51_10ms_192k.opus",
             "corpus/twosize_st_20ms_then_10ms_10s.opus", "short.opus", "corpus/st_2p5ms_128k.opus", "corpus/mono_20ms_16k_nb.opus"]
    r = subprocess.run([os.path.join(SCHED, "fuzz_check_asan"), "120", "7"] + [os.path.join(GOLDEN, f) for f in files],
                       capture_output=True, text=True, timeout=900)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    assert "120 iterations" in r.stdout


LEASE_FILES = ["corpus/st_20ms_32k.opus", "corpus/mono_5ms_64k.opus", "corpus/surround51_10ms_192k.opus",
               "corpus/twosize_st_20ms_then_10ms_10s.opus", "short.opus", "corpus/st_2p5ms_128k.opus"]


@pytest.mark.parametrize("san", ["tsan", "asan"])
@pytest.mark.parametrize("mode,threads,reps,files", [("plain", 6, 3, LEASE_FILES), ("gpu-faults", 6, 3, LEASE_FILES),
                                                     ("thread-faults", 6, 3, LEASE_FILES), ("churn", 22, 2, LEASE_FILES[:2])])
def test_concurrent_loads_lease_pool(san, mode, threads, reps, files):
    """Six threads inside nqr::NyquistIO::Load at once -- the situation of the one process abort seen on a GPU box
     -- against the fake GPU: the real plugin surface, lease pool, scheduler and entropy decoder under
    ThreadSanitizer and AddressSanitizer.  plain: bit-exact, no decoder torn down; gpu-faults: every 29th GPU call and every
    17th context creation fail -> std::runtime_error out of Load, nothing else, survivors bit-exact, no context leaked;
    thread-faults: thread starts fail on and off -> the batch runs on the threads it gets or throws, no joinable thread is
    ever destroyed (that is std::terminate -> abort); churn: 22 threads against a pool of 16 -> the surplus decoders are
    retired and destroyed only at a moment without active leases (the stand-in counts contexts destroyed while a GPU call is
    in flight: none), still bit-exact."""
    subprocess.run(["make", "-C", SCHED, "lease"], check=True, stdout=subprocess.DEVNULL)
    e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = subprocess.run([os.path.join(SCHED, f"lease_check_{san}"), mode, str(threads), str(reps)] + [os.path.join(GOLDEN, f) for f in files],
                       env=e, capture_output=True, text=True, timeout=900)
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-800:], r.stderr[-1500:])
    assert r.stdout.strip().endswith("OK"), r.stdout[-800:]
