"""The N>1 path of bench.py.
CPU tier: bench.shard_rows (the partition bench.main applies to the global batch) and the barrier / MAX-over-ranks
reduction with two gloo ranks; the compute leg is stubbed by the oracle because this tier has no GPU.
GPU tier: bench.py itself under torch.distributed.run with two ranks sharing the box's one GPU (started by
tests/conftest.py before this process touches the GPU): the real kernels, one context per rank, the gloo fallback,
the per-rank file-decode leg, the whole-job aggregation."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import json

from conftest import REHEARSAL, ROOT

sys.path.insert(0, ROOT)
from bench import shard_rows  # noqa: E402


@pytest.mark.parametrize("total,world", [(10, 1), (10, 2), (11, 2), (1 << 20, 8), (7, 8), (0, 2)])
def test_shard_rows_partitions_exactly(total, world):
    spans = [shard_rows(total, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.pyoracle import Oracle
    orc = Oracle()
    total = 37
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (total, 960)).astype(np.float32)      # same on every rank
    lo, hi = shard_rows(total, world, rank)
    fin, tail = orc.imdct_batch(0, x[lo:hi], None)
    # bench.py's timing reduction: barrier, then MAX over ranks
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # gather only to let rank 0 verify coverage (NOT on the product's data path)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, fin, tail))
    if rank == 0:
        full_fin = np.concatenate([p[2] for p in parts])
        full_tail = np.concatenate([p[3] for p in parts])
        wf, wt = orc.imdct_batch(0, x, None)
        q.put((float(t.item()), bool(np.array_equal(full_fin, wf) and np.array_equal(full_tail, wt)),
               [(p[0], p[1]) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    tmax, same, spans = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert tmax == 1.5                      # max over ranks, not rank 0's own time
    assert same                             # shards tile the batch exactly
    assert spans == [(0, 19), (19, 37)]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --rehearse-on-one-gpu --dist-backend gloo
    --rows 65536 --steps 2 --cpu-seconds 2` must print ONE JSON line with n_gpus 2, both shards in the global batch, a parity figure from
    the oracle, a whole-job Opus chain rate and the per-rank file-decode leg aggregated over both ranks."""
    p = REHEARSAL["proc"]
    if p is None:
        pytest.skip("rehearsal child not started (needs `-m gpu` and a GPU at session start)")
    p.wait(timeout=900)                        # (the child runs the two-rank launch, then the four-rank one)
    REHEARSAL["out"].seek(0)
    REHEARSAL["err"].seek(0)
    out, err = REHEARSAL["out"].read(), REHEARSAL["err"].read()
    assert "rc2=0" in err, (out[-800:], err[-2500:])
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-1500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak"
    assert j["config"]["global_rows"] == 2 * 65536 and j["config"]["rows_per_gpu"] == 65536
    assert j["config"]["rank0_rows"] == [0, 65536]
    assert j["value"] > 0 and j["parity_rel_rms_vs_oracle"] <= 1e-5
    assert j["opus_frame_synthesis"]["n_gpus"] == 2
    assert j["opus_frame_synthesis"]["whole_job_stereo_frames_per_sec_synthesis_plus_post_filter"] > 0
    leg = j["opus_file_decode"]
    assert leg["whole_job"]["n_gpus"] == 2 and leg["whole_job"]["files"] == 2 * leg["files"]
    assert leg["whole_job"]["slowest_rank_wall_seconds"] >= leg["wall_seconds"] * 0.999
    # the multi-rank line is complete: the reference's CPU path timed in the same run, the process
    # group that really came up, how many ranks answered, and every rank's own kernel time / roofline fraction
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "IMDCT/s"
    assert j["dist_backend"] == "gloo" and j["ranks_seen"] == 2
    assert len(j["roofline"]["per_rank_kernel_avg_ms"]) == 2 and len(j["roofline"]["per_rank_frac"]) == 2
    assert j["roofline"]["kernel_avg_ms"] == max(j["roofline"]["per_rank_kernel_avg_ms"])
    assert abs(j["roofline"]["frac"] - min(j["roofline"]["per_rank_frac"])) < 1e-9


@pytest.mark.gpu
def test_bench_four_ranks_on_one_gpu():
    """The same launch with FOUR ranks (`--gpus 4 --rows 16384`, behind the two-rank run in the same child): one line,
    n_gpus 4, the four shards tile the global batch, four per-rank kernel times, the file-decode leg aggregated over four
    ranks with every rank's host-thread share and GPU-busy fraction on the line (what a 1 -> 8 curve will be read with)."""
    p = REHEARSAL["proc"]
    if p is None:
        pytest.skip("rehearsal child not started (needs `-m gpu` and a GPU at session start)")
    p.wait(timeout=900)
    out, err = open(REHEARSAL["out4"]).read(), open(REHEARSAL["err4"]).read()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out[-1500:], err[-2500:])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 4 and j["ranks_seen"] == 4 and j["dist_backend"] == "gloo"
    assert j["config"]["global_rows"] == 4 * 16384 and j["config"]["rank0_rows"] == [0, 16384]
    assert j["value"] > 0 and j["value_cold"] > 0 and j["parity_rel_rms_vs_oracle"] <= 1e-5
    assert len(j["roofline"]["per_rank_kernel_avg_ms"]) == 4
    assert j["opus_frame_synthesis"]["n_gpus"] == 4
    wj = j["opus_file_decode"]["whole_job"]
    assert wj["n_gpus"] == 4 and wj["files"] == 4 * j["opus_file_decode"]["files"]
    assert len(wj["per_rank_wall_seconds"]) == 4 and len(wj["per_rank_gpu_busy_fraction"]) == 4
    assert all(t >= 1 for t in wj["per_rank_host_threads"]) and wj["host_threads_total"] == sum(wj["per_rank_host_threads"])
    assert all(0.0 < b <= 1.0 for b in wj["per_rank_gpu_busy_fraction"])
    assert j["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_rccl_process_group_comes_up_on_this_box():
    """bench.py's N > 1 runs use torch.distributed's "nccl" backend (= RCCL on ROCm) for the barrier, the MAX of the times and
    the gather of the per-rank kernel times -- never for data.  One-GPU boxes cannot run two RCCL ranks (RCCL refuses two ranks
    on one device), but a ONE-rank group exercises the same initialisation, a GPU all-reduce, an all-gather and a barrier: if
    this passes, `dist_backend` of a multi-GPU bench line will read "nccl" and not the gloo fallback.  Runs in a child
    process with a time limit (a communicator that does not come up must not hang the tier)."""
    import subprocess
    code = r'''
import datetime, os, sys
import torch, torch.distributed as dist
port = 29800 + os.getpid() % 150
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                        timeout=datetime.timedelta(seconds=60), device_id=torch.device("cuda", 0))
t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = [torch.zeros_like(t)]
dist.all_gather(g, t)
ones = torch.ones(1, device="cuda", dtype=torch.float64)
dist.all_reduce(ones)
dist.barrier()
torch.cuda.synchronize()
print("RCCL", dist.get_backend(), float(t.item()), float(g[0].item()), int(ones.item()))
dist.destroy_process_group()
'''
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def record(outcome, detail=""):
        # whatever happens is kept (gpurun_out/ is merged back): a communicator that did not start is to be looked at, not lost
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump({"probe": "one-rank RCCL process group", "outcome": outcome, "detail": detail[-2000:]},
                      open(os.path.join(ROOT, "gpurun_out", "rccl_probe.json"), "w"))
        except OSError:
            pass
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired as e:
        record("timeout after 240 s", (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""))
        pytest.skip("the one-rank RCCL group did not come up within 240 s on this box (an environment probe, not a parity test)")
    record("ok" if r.returncode == 0 else f"rc {r.returncode}", r.stdout + r.stderr)
    if r.returncode != 0:
        # a capability probe of the box, not of the product: report, do not cut the parity tier off under -x
        pytest.skip(f"the one-rank RCCL group did not come up on this box: rc {r.returncode}, {r.stderr[-600:]}")
    assert "RCCL nccl 1.5 1.5 1" in r.stdout, r.stdout
