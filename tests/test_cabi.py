"""CPU tier: the C-ABI library builds, loads and exports every symbol include/nyq_imdct.h declares;
without a GPU the product fails loudly instead of falling back."""
import os
import re

import pytest

import libnyquist_amd as nyq
from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "nyq_imdct.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|char|size_t)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_and_binding_agree():
    declared = _declared_functions()
    assert len(declared) >= 19
    assert sorted(nyq.EXPORTS) == declared


def test_library_exports_every_declared_symbol():
    lib = nyq.load()
    for name in _declared_functions():
        assert hasattr(lib, name), name


def test_no_torch_or_oracle_in_product_sources():
    """The product path must not reach the oracle or any CPU fallback."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "libnyquist_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "nyq_oracle" not in src and "liboracle" not in src, f
                assert "lane_emu" not in src, f


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nyq.NyqError) as e:
        nyq.Context(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_header_is_plain_c():
    """include/nyq_imdct.h is the FFI contract: it must compile as strict C99 on its own (no C++, no torch types)."""
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
        f.write('#include "nyq_imdct.h"\nint main(void) { return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", f.name], capture_output=True, text=True)
    os.unlink(f.name)
    assert r.returncode == 0, r.stderr


def test_shim_error_handler_instead_of_abort():
    """The reference-named void operators cannot return a status; by default a failure ends the process as the reference's
    offload does (mdct_cuda.cu:11-19).  With nyq_shim_set_error_handler installed the handler hears about it and the call
    returns with `output` untouched: on a box without a GPU every call fails ("no HIP device"), and a call outside the static
    48 kHz mode fails on any box -- in both cases the process must live and the output must be as it was."""
    import ctypes as C
    import numpy as np
    lib = nyq.load()
    heard = []
    CB = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p)
    cb = CB(lambda who, what: heard.append((who.decode(), what.decode())))
    lib.nyq_shim_set_error_handler.argtypes = [CB]
    lib.nyq_shim_set_error_handler.restype = None
    lib.nyq_shim_set_error_handler(cb)
    try:
        trig, win = np.zeros(481, np.float32), np.zeros(120, np.float32)
        x = np.ones(960, np.float32)
        out = np.full(960 + 60, 7.0, np.float32)
        fp = lambda a: a.ctypes.data_as(C.c_void_p)
        lib.processMDCTCuda(fp(x), fp(out), fp(trig), 1000, 0, 1, C.c_float(0.0), 120, fp(win))     # N is not 1920 >> shift
        assert heard and heard[-1][0] == "processMDCTCuda" and "unsupported call" in heard[-1][1]
        assert np.all(out == 7.0)
        import torch
        if not torch.cuda.is_available():
            lib.processMDCTCuda(fp(x), fp(out), fp(trig), 1920, 0, 1, C.c_float(0.0), 120, fp(win))
            assert "no HIP device" in heard[-1][1] and np.all(out == 7.0)
    finally:
        lib.nyq_shim_set_error_handler(CB(0))
