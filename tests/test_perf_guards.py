"""Performance guards -- NOT part of the parity tier.  Marker `perf` only: `pytest -m gpu` (the parity run, usually with
-x) never selects them, so a noisy box cannot cut the parity tests off; run them with `pytest -m perf` on a GPU box
(tools/final_prof.sh does).  Without a GPU they skip."""
import numpy as np
import pytest

pytestmark = pytest.mark.perf


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="module")
def ctx():
    if not _gpu():
        pytest.skip("needs a GPU")
    import libnyquist_amd as nyq
    c = nyq.Context(0)
    yield c
    c.close()


def test_post_filter_kernel_is_placed_at_once_behind_other_kernels(ctx):
    """The post-filter pipeline's workgroups live for the whole launch; with a register / LDS footprint that fits a CU
    exactly, some of them were placed 0.8 ms late whenever another kernel had run before (1.6 instead of 1.0 ms,
    DESIGN 4.4a).  Guard: the kernel behind a synthesis call must not take much longer than behind itself."""
    import torch
    dev = torch.device("cuda", 0)
    ns, nf, ch, n = 1024, 64, 2, 960
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    freq = torch.randn((ns, nf, ch, n), generator=g, device=dev) * 30.0
    trans = (torch.rand((ns, nf), generator=g, device=dev) < 0.028).to(torch.uint8)
    pitch = torch.randint(15, 80, (ns, nf), generator=g, device=dev, dtype=torch.int32)
    gain = (torch.rand((ns, nf), generator=g, device=dev) < 0.7).float() * (torch.randint(1, 9, (ns, nf), generator=g, device=dev) * 0.09375).float()
    tap = torch.randint(0, 3, (ns, nf), generator=g, device=dev, dtype=torch.int32)
    out = torch.empty((ns, nf * n, ch), device=dev)
    pcm = torch.empty((ns * ch, nf * n), device=dev)
    work = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    try:
        def synth():
            ctx.celt_synth_dev(3, freq.data_ptr(), trans.data_ptr(), pcm.data_ptr(), 0, work.data_ptr(), ns, nf, ch)

        def post_ms(before):
            ts = []
            for _ in range(7):
                before()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                ctx.celt_post_dev(3, pcm.data_ptr(), pitch.data_ptr(), gain.data_ptr(), tap.data_ptr(), 0, 0, 0, 0, out.data_ptr(), ns, nf, ch)
                b.record(stream)
                torch.cuda.synchronize(dev)
                ts.append(a.elapsed_time(b))
            return float(np.median(ts))
        synth()
        post_ms(lambda: None)
        alone, behind = 1e9, 1e9
        for _ in range(3):
            alone = min(alone, post_ms(lambda: None))
            behind = min(behind, post_ms(synth))
        # measured spread of this kernel under the profiler: 0.945-1.216 ms (profiles/r02_g_kernel_stats.csv); a misplaced
        # launch takes 1.6-1.9 ms.  Best of three rounds against 1.45 x: noise does not trip it, the placement defect does.
        assert behind <= 1.45 * alone, (alone, behind)
    finally:
        ctx.reset_stream()


def test_entropy_stage_on_the_device_stays_far_above_the_host_stage(ctx):
    """The entropy stage as a kernel pair: 32 streams of sb-reverie.opus' frames, bytes resident in HBM.  Measured
    24 M frames/s (8.7 ms a frame per lane + 6.4 ms the per-stream energy pass); sixteen host threads do 2.5-2.8 M.  Guard: half."""
    import ctypes as C
    import os
    import time

    import torch

    from conftest import GOLDEN
    from test_host_decoder import load_host
    H = load_host()
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    H.nyqh_entropy_tables.argtypes = [C.c_void_p, C.c_long]
    H.nyqh_entropy_tables.restype = C.c_long
    H.nyqh_frame_table.argtypes = [C.c_char_p, C.c_long, C.c_long, u8, C.c_long, C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    raw = open(os.path.join(GOLDEN, "sb-reverie.opus"), "rb").read()
    need = H.nyqh_entropy_tables(None, 0)
    tables = np.zeros(need, np.uint8)
    assert H.nyqh_entropy_tables(tables.ctypes.data, need) == need
    payload, desc, info = np.zeros(11200 * 320, np.uint8), np.zeros(11200 * 12, np.uint8), np.zeros(8, np.int64)
    assert H.nyqh_frame_table(raw, len(raw), 11200, payload, payload.size, desc.ctypes.data, info) == 0
    nf, ns = int(info[2]), 32
    dev = torch.device("cuda", 0)
    slot = int(ctx.lib.nyq_celt_entropy_slot_bytes(2, 3))
    d_tab, d_pay = torch.from_numpy(tables).to(dev), torch.from_numpy(payload[:int(info[4])].copy()).to(dev)
    d_desc = torch.from_numpy(np.tile(desc[:nf * 12], ns)).to(dev)
    Z = lambda shape: torch.zeros(shape, dtype=torch.uint8, device=dev)
    d_sym, d_info, d_energy, d_state = Z((ns * nf, slot)), Z((ns * nf, 16)), Z((ns * nf, 672)), Z((ns, 516))
    torch.cuda.synchronize(dev)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        ctx.celt_entropy_dev(3, d_tab.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_desc.data_ptr(), ns, nf, 2, d_sym.data_ptr(), d_info.data_ptr(),
                             d_energy.data_ptr(), d_state.data_ptr(), True, slot)
        ctx.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    rate = ns * nf / best
    print(f"entropy stage on the device: {best * 1e3:.2f} ms per {ns * nf} frames = {rate / 1e6:.1f} M frames/s")
    assert rate >= 12e6
