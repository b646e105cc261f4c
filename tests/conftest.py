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
