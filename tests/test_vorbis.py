"""Vorbis inverse MDCT: libvorbis' mdct_backward for block sizes 64..8192.
CPU tier: the closed-form oracle against outputs of libvorbis' own mdct.c (compiled standalone,
fixtures ref_vorbis.npz), and the HIP lane program replayed on the CPU.  GPU tier: the kernel
through the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, rel_rms

SIZES = [64, 128, 256, 512, 1024, 2048, 4096, 8192]
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


@pytest.mark.parametrize("n", SIZES)
def test_oracle_vs_libvorbis(oracle, n):
    z = np.load(os.path.join(GOLDEN, "ref_vorbis.npz"))
    assert rel_rms(oracle.vorbis_imdct(n, z[f"x{n}"]), z[f"y{n}"]) <= 1e-6


@pytest.mark.parametrize("n", SIZES)
def test_lane_program_vs_libvorbis(oracle, n):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu")], check=True, stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "tests", "emu", "liblane_emu.so"))
    L.emu_vorbis_imdct.argtypes = [C.c_int, _f32p, _f32p, C.c_long, _f32p, _f32p]
    z = np.load(os.path.join(GOLDEN, "ref_vorbis.npz"))
    n4 = n // 4
    i = np.arange(n4)
    rot = np.stack([np.cos(2 * np.pi * (i + 0.125) / n), np.sin(2 * np.pi * (i + 0.125) / n)], 1).astype(np.float32).reshape(-1)
    tw = np.stack([np.cos(2 * np.pi * i / n4), np.sin(2 * np.pi * i / n4)], 1).astype(np.float32).reshape(-1)
    rng = np.random.default_rng(n)
    x = np.concatenate([z[f"x{n}"], rng.uniform(-1, 1, (18, n // 2)).astype(np.float32)])   # 21 rows: ragged groups
    y = np.zeros((x.shape[0], n), np.float32)
    assert L.emu_vorbis_imdct(n, x.reshape(-1), y.reshape(-1), x.shape[0], rot, tw) == 0
    assert rel_rms(y[:3], z[f"y{n}"]) <= 1e-6
    assert rel_rms(y, oracle.vorbis_imdct(n, x)) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES)
def test_gpu_vorbis_imdct(oracle, n):
    import libnyquist_amd as nyq
    ctx = nyq.Context(0)
    z = np.load(os.path.join(GOLDEN, "ref_vorbis.npz"))
    assert rel_rms(ctx.vorbis_imdct_batch(n, z[f"x{n}"]), z[f"y{n}"]) <= 1e-5      # vs libvorbis itself
    rng = np.random.default_rng(3 * n)
    for rows in (1, 5, 257):
        x = (rng.standard_normal((rows, n // 2)) * 30).astype(np.float32)
        y = ctx.vorbis_imdct_batch(n, x)
        take = min(rows, 16)
        assert rel_rms(y[:take], oracle.vorbis_imdct(n, x[:take])) <= 1e-6
        # structure of the output: outer quarters mirror the middle half (mdct.c:455-489)
        q = n // 4
        assert np.array_equal(y[:, :q], -y[:, q:2 * q][:, ::-1])
        assert np.array_equal(y[:, 3 * q:], y[:, 2 * q:3 * q][:, ::-1])
    with pytest.raises(nyq.NyqError):
        ctx.vorbis_imdct_batch(96, np.zeros((1, 48), np.float32))
    ctx.close()


@pytest.mark.gpu
def test_gpu_mixed_codec_batch(oracle):
    """BASELINE config 5 shape (synthetic, as SURVEY section 8(d) prescribes): CELT nfft-480 / nfft-60 rows and
    Vorbis 2048 / 256 blocks issued back to back on one context and one stream."""
    import libnyquist_amd as nyq
    import torch
    dev = torch.device("cuda", 0)
    ctx = nyq.Context(0)
    ctx.set_tables(*oracle.tables()[:2])
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    a = torch.randn((4096, 960), generator=g, device=dev) * 30
    b = torch.randn((8192, 120), generator=g, device=dev) * 30
    v1 = torch.randn((4096, 1024), generator=g, device=dev) * 30
    v2 = torch.randn((8192, 128), generator=g, device=dev) * 30
    fa, ta = torch.empty_like(a), torch.empty((4096, 60), device=dev)
    fb, tb = torch.empty_like(b), torch.empty((8192, 60), device=dev)
    o1, o2 = torch.empty((4096, 2048), device=dev), torch.empty((8192, 256), device=dev)
    torch.cuda.synchronize(dev)
    ctx.imdct_batch_dev(0, a.data_ptr(), 0, fa.data_ptr(), ta.data_ptr(), 4096)
    ctx.vorbis_imdct_batch_dev(2048, v1.data_ptr(), o1.data_ptr(), 4096)
    ctx.imdct_batch_dev(3, b.data_ptr(), 0, fb.data_ptr(), tb.data_ptr(), 8192)
    ctx.vorbis_imdct_batch_dev(256, v2.data_ptr(), o2.data_ptr(), 8192)
    ctx.synchronize()
    assert rel_rms(fa[:8].cpu().numpy(), oracle.imdct_batch(0, a[:8].cpu().numpy())[0]) <= 1e-6
    assert rel_rms(fb[-8:].cpu().numpy(), oracle.imdct_batch(3, b[-8:].cpu().numpy())[0]) <= 1e-6
    assert rel_rms(o1[-4:].cpu().numpy(), oracle.vorbis_imdct(2048, v1[-4:].cpu().numpy())) <= 1e-6
    assert rel_rms(o2[:4].cpu().numpy(), oracle.vorbis_imdct(256, v2[:4].cpu().numpy())) <= 1e-6
    ctx.close()

