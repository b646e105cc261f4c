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
