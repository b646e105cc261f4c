"""pytest configuration: markers, shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU needed).
`-m gpu`      : parity tests proper -- HIP path through the C-ABI vs the oracle / golden fixtures.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock guards, outside the parity tier (run with -m perf on the GPU box)")


# The 2-rank rehearsal of bench.py (tests/test_sharding.py::test_bench_two_ranks_on_one_gpu) has to be STARTED before this
# process touches the GPU: a process that has initialised the HIP runtime must not spawn-and-exec GPU programs on the
# pool's boxes.  So the GPU tier launches it here, at session start, as a background child, and the test only collects
# its output.  (The CPU tier never starts it.)
REHEARSAL = {"proc": None, "out": None, "err": None, "out4": None, "err4": None}


def pytest_sessionstart(session):
    import subprocess
    expr = (session.config.getoption("-m") or "").strip()
    if expr != "gpu" or os.environ.get("NYQ_NO_REHEARSAL"):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:          # (counting devices does not initialise the runtime)
            return
    except Exception:
        return
    port = 29600 + (os.getpid() % 300)
    REHEARSAL["out"] = open(os.path.join("/tmp", "nyq_rehearsal_stdout.txt"), "w+")
    REHEARSAL["err"] = open(os.path.join("/tmp", "nyq_rehearsal_stderr.txt"), "w+")
    REHEARSAL["out4"], REHEARSAL["err4"] = "/tmp/nyq_rehearsal4_stdout.txt", "/tmp/nyq_rehearsal4_stderr.txt"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def launch(n, rows, p):
        return (f"{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node {n} --master-addr 127.0.0.1 --master-port {p} "
                f"{os.path.join(ROOT, 'bench.py')} --gpus {n} --rehearse-on-one-gpu --dist-backend gloo --rows {rows} --steps 2 --warmup 1 "
                f"--cpu-seconds 2")
    # two ranks on the one GPU, then FOUR (port / memory / thread-split problems that only show at N > 2): one after the
    # other in one shell child, so that with this process at most five touch the GPU at a time (the pool allows six)
    REHEARSAL["proc"] = subprocess.Popen(
        ["bash", "-c", launch(2, 65536, port) + "; rc2=$?; " + launch(4, 16384, port + 301) + f" > {REHEARSAL['out4']} 2> {REHEARSAL['err4']}; "
         "rc4=$?; echo \"rehearsal rc2=$rc2 rc4=$rc4\" >&2; exit $((rc2 + rc4))"],
        cwd=ROOT, env=env, stdout=REHEARSAL["out"], stderr=REHEARSAL["err"])


def pytest_sessionfinish(session, exitstatus):
    p = REHEARSAL["proc"]
    if p is not None and p.poll() is None:
        p.kill()                                   # (exactly the child started above)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ref_tables():
    z = np.load(os.path.join(GOLDEN, "ref_tables.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle(ref_tables):
    """The C restatement, pinned to the reference's static tables (bit-exact mode)."""
    from oracle.pyoracle import Oracle, build
    build()
    return Oracle((ref_tables["trig"], ref_tables["window"], ref_tables["tw"]))


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def abs_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
