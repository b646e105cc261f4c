"""CPU tier: replay the HIP kernels' lane program (the same __host__ __device__ headers hipcc
compiles for gfx950) on the CPU and compare with the oracle.  Checks the prime-factor index
maps, the stage tasks and the TDAC placement without a GPU.  tests/emu is test-only code."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rel_rms

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu")], check=True, stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "liblane_emu.so"))
    L.emu_dft.argtypes = [C.c_int, _f32p]
    L.emu_imdct_batch.argtypes = [C.c_int, _f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, _f32p, _f32p]
    L.emu_ifft_batch.argtypes = [C.c_int, _f32p, _f32p, C.c_long]
    return L


@pytest.mark.parametrize("r", [2, 3, 4, 5, 8, 15, 16, 32, 64])
def test_register_dft(emu, r):
    rng = np.random.default_rng(r)
    x = rng.standard_normal(2 * r).astype(np.float32)
    y = x.copy()
    assert emu.emu_dft(r, y) == 0
    k = np.arange(r)
    want = np.exp(2j * np.pi * np.outer(k, k) / r) @ x.astype(np.float64).view(np.complex128)
    got = y.astype(np.float64).view(np.complex128)
    assert np.abs(got - want).max() <= 4e-6 * np.sqrt(r)


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
@pytest.mark.parametrize("rows", [1, 4, 9])
def test_lane_program_vs_oracle(emu, oracle, shift, rows):
    n2 = 960 >> shift
    rng = np.random.default_rng(100 * shift + rows)
    x = (rng.standard_normal((rows, n2)) * 30).astype(np.float32)
    carry = (rng.standard_normal((rows, 60)) * 30).astype(np.float32)
    trig, win, _ = oracle.tables()
    for cy in (carry, None):
        fin = np.zeros((rows, n2), np.float32)
        tail = np.zeros((rows, 60), np.float32)
        emu.emu_imdct_batch(shift, x.reshape(-1), None if cy is None else cy.ctypes.data_as(C.c_void_p),
                            fin.reshape(-1), tail.ctypes.data_as(C.c_void_p), rows, trig, win)
        wf, wt = oracle.imdct_batch(shift, x, cy)
        assert rel_rms(fin, wf) <= 1e-6
        assert rel_rms(tail, wt) <= 1e-6


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_lane_program_vs_reference_fixtures(emu, ref_tables, shift):
    z = np.load(os.path.join(GOLDEN, f"ref_imdct_s{shift}.npz"))
    x, carry, want = z["x"], z["carry"], z["out"]
    rows, n2 = x.shape
    fin = np.zeros((rows, n2), np.float32)
    tail = np.zeros((rows, 60), np.float32)
    emu.emu_imdct_batch(shift, x.reshape(-1), carry.ctypes.data_as(C.c_void_p), fin.reshape(-1),
                        tail.ctypes.data_as(C.c_void_p), rows, ref_tables["trig"], ref_tables["window"])
    assert rel_rms(np.concatenate([fin, tail], 1), want) <= 1e-6


@pytest.mark.parametrize("nfft", [60, 480])
def test_lane_program_bundled_ifft_vectors(emu, nfft):
    x = np.fromfile(os.path.join(GOLDEN, f"ifft_input_N{nfft}.bin"), np.float32)
    want = np.fromfile(os.path.join(GOLDEN, f"ifft_output_N{nfft}.bin"), np.float32)
    xs = np.tile(x, (6, 1))
    ys = np.zeros_like(xs)
    assert emu.emu_ifft_batch(nfft, xs.reshape(-1), ys.reshape(-1), 6) == 0
    for r in range(6):
        assert np.sqrt(np.mean((ys[r].astype(np.float64) - want) ** 2)) <= 1e-5   # north_star tolerance
        assert np.sqrt(np.mean((ys[r].astype(np.float64) - want) ** 2)) <= 2e-7


@pytest.mark.parametrize("lm", [3, 2, 1, 0])
def test_frame_synth_lane_program_vs_oracle(emu, oracle, lm):
    """In-wave chaining of long frames, tail-ring chaining of short blocks, fix-up of the rest."""
    emu.emu_celt_synth.argtypes = [C.c_int, _f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, C.c_long, C.c_int,
                                   _f32p, _f32p]
    trig, win, _ = oracle.tables()
    rng = np.random.default_rng(lm)
    n = 120 << lm
    for ns, nf, ch, ptr in ((1, 1, 1, 0.0), (2, 9, 2, 0.3), (1, 37, 3, 0.2), (3, 4, 2, 1.0), (2, 33, 1, 0.0), (1, 16, 2, 0.1)):
        freq = (rng.standard_normal((ns, nf, ch, n)) * 30).astype(np.float32)
        tr = (rng.uniform(size=(ns, nf)) < ptr).astype(np.uint8)
        st = (rng.standard_normal((ns * ch, 60)) * 30).astype(np.float32)
        for use_state in (True, False):
            pcm = np.zeros((ns, ch, nf * n), np.float32)
            st2 = st.copy()
            rc = emu.emu_celt_synth(lm, freq.reshape(-1), tr.ctypes.data_as(C.c_void_p), pcm.reshape(-1),
                                    st2.ctypes.data_as(C.c_void_p) if use_state else None, ns, nf, ch, trig, win)
            assert rc == 0
            wp, ws = oracle.celt_synth(lm, freq, tr, st if use_state else None)
            assert rel_rms(pcm, wp) <= 1e-6
            if use_state:
                assert rel_rms(st2, ws) <= 1e-6


def test_frame_synth_lane_program_vs_reference_fixture(emu, ref_tables):
    emu.emu_celt_synth.argtypes = [C.c_int, _f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, C.c_long, C.c_int,
                                   _f32p, _f32p]
    z = np.load(os.path.join(GOLDEN, "ref_synth.npz"))
    freq, tr = np.ascontiguousarray(z["freq"]), np.ascontiguousarray(z["transient"])
    ns, nf, ch, n = freq.shape
    pcm = np.zeros((ns, ch, nf * n), np.float32)
    st = z["state_in"].copy()
    emu.emu_celt_synth(3, freq.reshape(-1), tr.ctypes.data_as(C.c_void_p), pcm.reshape(-1),
                       st.ctypes.data_as(C.c_void_p), ns, nf, ch, ref_tables["trig"], ref_tables["window"])
    assert rel_rms(pcm, z["pcm"]) <= 1e-6
    assert rel_rms(st, z["state_out"]) <= 1e-6


def test_frame_synth_lane_program_on_real_decoder_frames(emu, ref_tables):
    emu.emu_celt_synth.argtypes = [C.c_int, _f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, C.c_long, C.c_int,
                                   _f32p, _f32p]
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    freq, tr = np.ascontiguousarray(z["freq"]), np.ascontiguousarray(z["transient"])
    ns, nf, ch, n = freq.shape
    pcm = np.zeros((ns, ch, nf * n), np.float32)
    st = z["state_in"].copy()
    emu.emu_celt_synth(3, freq.reshape(-1), tr.ctypes.data_as(C.c_void_p), pcm.reshape(-1),
                       st.ctypes.data_as(C.c_void_p), ns, nf, ch, ref_tables["trig"], ref_tables["window"])
    assert rel_rms(pcm, z["pcm"]) <= 1e-6
    assert rel_rms(st, z["state_out"]) <= 1e-6


@pytest.mark.parametrize("nf,ptr", [(1, 0.0), (5, 0.0), (1, 1.0), (3, 1.0), (24, 0.3), (70, 0.1)])
def test_fused_transform_wave_program_vs_oracle(emu, oracle, nf, ptr):
    """The transform wave of the one-launch frames -> PCM kernel (nyq_fuse_lanes.hpp: one wave, both channels of a 20 ms
    frame, in place in two 3840-byte regions; 480 = 2 x 16 x 15 long program, 16 x nfft-60 transient program, tails in
    registers) replayed on the CPU: long / transient / mixed sequences, state in and out."""
    emu.emu_fuse_synth.argtypes = [_f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, _f32p, _f32p]
    trig, win, _ = oracle.tables()
    rng = np.random.default_rng(nf)
    freq = (rng.standard_normal((1, nf, 2, 960)) * 30).astype(np.float32)
    tr = (rng.uniform(size=(1, nf)) < ptr).astype(np.uint8)
    st = (rng.standard_normal((2, 60)) * 30).astype(np.float32)
    for use_state in (True, False):
        pcm = np.zeros((1, 2, nf * 960), np.float32)
        st2 = st.copy()
        assert emu.emu_fuse_synth(freq.reshape(-1), tr.ctypes.data_as(C.c_void_p), pcm.reshape(-1),
                                  st2.ctypes.data_as(C.c_void_p) if use_state else None, nf, trig, win) == 0
        wp, ws = oracle.celt_synth(3, freq, tr, st if use_state else None)
        assert rel_rms(pcm, wp) <= 1e-6
        if use_state:
            assert rel_rms(st2, ws) <= 1e-6


def test_fused_transform_wave_program_on_real_decoder_frames(emu, ref_tables):
    emu.emu_fuse_synth.argtypes = [_f32p, C.c_void_p, _f32p, C.c_void_p, C.c_long, _f32p, _f32p]
    z = np.load(os.path.join(GOLDEN, "real_opus_frames.npz"))
    freq, tr = np.ascontiguousarray(z["freq"]), np.ascontiguousarray(z["transient"])
    ns, nf, ch, n = freq.shape
    assert ch == 2 and n == 960
    for s in range(ns):
        pcm = np.zeros((ch, nf * n), np.float32)
        st = np.ascontiguousarray(z["state_in"][2 * s:2 * s + 2])
        emu.emu_fuse_synth(np.ascontiguousarray(freq[s]).reshape(-1), np.ascontiguousarray(tr[s]).ctypes.data_as(C.c_void_p),
                           pcm.reshape(-1), st.ctypes.data_as(C.c_void_p), nf, ref_tables["trig"], ref_tables["window"])
        assert rel_rms(pcm, z["pcm"][s]) <= 1e-6
        assert rel_rms(st, z["state_out"][2 * s:2 * s + 2]) <= 1e-6
