#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched CELT inverse-MDCT path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W          (N > 1)

Workload (BASELINE.json configs[2]): per GPU 2^20 rows of
nfft-480 IMDCTs (MDCT N = 1920: 960 float32 coefficients in, 960 finished samples + 60
tail floats out), X ~ U(-1,1) from a fixed seed, generated on the device and resident in
HBM before the timed region; shift 0, stride 1, overlap 120, carry-in zeros.
One "step" = one pass of nyq_imdct_batch_dev over the whole batch.  Rows shard
embarrassingly across GPUs (each rank owns its own 2^20 rows: weak scaling, no data-path
collective; torch.distributed is used only for the barrier and the max-over-ranks time).

Before its W warm-up steps every leg runs its own operator back to back for --preroll-ms (40) of GPU time, untimed, so that
the K timed steps see the steady clocks of a running service and not the 10-20 ms ramp after idle (DESIGN.md section 6;
config.clock_preroll_ms in the line).  Rank 0 prints ONE JSON line; see README/DESIGN.md for the fields.  `roofline.achieved`
= 7680 algorithmic bytes x rows / mean kernel duration (HIP events on the kernel's own
stream); `cpu_baseline` = the reference's own clt_mdct_backward (oracle/_ref, kind
"reference") or this repo's C restatement (kind "port") timed on this host's cores over a
bounded sample of the same rows.

Secondary keys on the same line (rank 0; the last two at N = 1 only, never part of `value`):
  opus_frame_synthesis  1024 streams x 256 stereo frames per GPU (every rank): nyq_celt_synth_dev and nyq_celt_post_dev alone, and
                        nyq_celt_chain_dev (freq[] -> PCM in ONE launch; the whole-job Opus frames/s comes from the slowest
                        rank's chain time)
  host_boundary         nyq_imdct_batch on pinned HOST buffers (PCIe both ways) and the per-call latency of
                        the reference's own offload interface (processMDCTCuda)
  opus_file_decode      256 Ogg Opus files through the plugin surface next to the reference's NyquistIO::Load
                        on the same host threads (that leg's cpu_baseline); .long_streams (32 x the 224 s file of BASELINE
                        config 4), .surround_7_1 (config 5's shape: 8 channels in 5 elementary streams),
                        .band_shapes_on_device (the entropy stage's two halves on real packets: host symbol stage
                        frames/s/thread, celt_shape_kernel frames/s), .entropy_stage_on_device (the entropy stage itself as a
                        kernel, a frame per lane: frames/s, and bytes -> PCM device-resident)
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_IMDCT = 7680          # 960 f32 in + 960 f32 out
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N2 = 960
HALF_OV = 60


def shard_rows(total_rows, world, rank):
    """Contiguous row range [lo, hi) of `rank` when `total_rows` are split over `world` ranks."""
    base, rem = divmod(total_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def csrc_digest():
    """sha1 over the kernel sources (csrc/*.hpp, *.hip): measurements taken with other kernels must not be replayed."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "libnyquist_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(rows):
    """roofline.traffic: HBM bytes per launch of the headline kernel from the PMC passes recorded in
    profiles/traffic_latest.json (tools/pmc_summary.py) -- only if that file was produced with THESE kernel sources
    and this row count; otherwise null, with the reason."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    src = {"file": "profiles/traffic_latest.json", "csrc_sha16_now": csrc_digest()}
    if not os.path.exists(tpath):
        return None, dict(src, status="absent")
    try:
        tj = json.load(open(tpath))
    except Exception as e:
        return None, dict(src, status=f"unreadable: {e!r}")
    src.update({k: tj.get(k) for k in ("csrc_sha16", "git_head", "kernel", "measured_on", "rows")})
    if tj.get("rows") != rows:
        return None, dict(src, status="other row count")
    if tj.get("csrc_sha16") != src["csrc_sha16_now"]:
        return None, dict(src, status="stale: kernel sources changed since the PMC passes")
    return tj.get("hbm_bytes_per_launch"), dict(src, status="measured with these kernel sources (separate --pmc FETCH_SIZE / WRITE_SIZE passes)")


def cpu_share():
    """Host cores this process may actually use: affinity mask clipped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, int(q / p + 0.5)))
    except Exception:
        pass
    return n


def band_shapes_leg(ctx, dev, rep=8):
    """The entropy stage's two halves on real packets (tests/golden/sb-reverie.opus, 11184 stereo 20 ms frames): the host half
    that reads every bit of a frame and stops at the symbol record (one thread, frames/s), and the device half that builds
    freq[] from the records (celt_shape_kernel, `rep` copies of the stream in one launch)."""
    import torch
    H = ctypes.CDLL(os.path.join(ROOT, "libnyquist_amd", "libnyquist_host.so"))
    H.nyqh_symbol_bytes.argtypes = [ctypes.c_int]
    H.nyqh_symbol_bytes.restype = ctypes.c_long
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u32 = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    i64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    H.nyqh_decode_to_symbols.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_long, u8, i32, f32, u32, i64]
    raw = open(os.path.join(ROOT, "tests", "golden", "sb-reverie.opus"), "rb").read()
    cap, rec = 11200, int(H.nyqh_symbol_bytes(2))
    sym = np.zeros((cap, rec), np.uint8)
    flags, gain, rng, info = np.zeros((cap, 4), np.int32), np.zeros(cap, np.float32), np.zeros(cap, np.uint32), np.zeros(8, np.int64)
    best_host = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        if H.nyqh_decode_to_symbols(raw, len(raw), cap, sym, flags, gain, rng, info) != 0:
            raise RuntimeError("nyqh_decode_to_symbols failed")
        best_host = min(best_host, time.perf_counter() - t0)
    nf = int(info[2])
    d_sym = torch.from_numpy(sym[:nf]).to(dev).repeat(rep, 1).contiguous()
    d_freq = torch.empty((rep * nf, 2, 960), device=dev)
    torch.cuda.synchronize(dev)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        ctx.celt_shape_dev(d_sym.data_ptr(), d_freq.data_ptr(), rep, nf, 2)
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    n = rep * nf
    return {"file": "sb-reverie.opus", "frames": nf, "frames_built_on_the_host": int(info[6]), "record_bytes": rec,
            "host_symbol_stage_frames_per_sec_per_thread": nf / best_host, "host_symbol_stage_includes": "Ogg + packet parsing, one thread",
            "device_frames_per_launch": n, "device_ms_per_launch": best * 1e3, "device_frames_per_sec": n / best,
            "device_GBps_records_in_plus_freq_out": n * (rec + 7680) / best / 1e9,
            "note": "latency-bound integer / LDS work (DESIGN 4.10), not a roofline kernel: what matters is device_frames_per_sec against "
                    "host threads x host_symbol_stage_frames_per_sec_per_thread"}


def device_entropy_leg(ctx, dev, streams=32):
    """The entropy stage ON THE DEVICE (nyq_celt_entropy_dev: a frame per lane, then the energy pass, a wave per stream) on
    `streams` copies of sb-reverie.opus' 11184 frames as independent streams, bytes resident in HBM: frames/s of the stage, and of
    bytes -> PCM through it (entropy + band shapes + synthesis + post-filter, device-resident).  The batch decoder uses it under
    NYQ_DEVICE_ENTROPY=1 (leg long_streams_128_host_vs_device_entropy); by default it runs the entropy stage on the host."""
    import torch
    H = ctypes.CDLL(os.path.join(ROOT, "libnyquist_amd", "libnyquist_host.so"))
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    H.nyqh_entropy_tables.argtypes = [ctypes.c_void_p, ctypes.c_long]
    H.nyqh_entropy_tables.restype = ctypes.c_long
    H.nyqh_frame_table.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_long, u8, ctypes.c_long, ctypes.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    raw = open(os.path.join(ROOT, "tests", "golden", "sb-reverie.opus"), "rb").read()
    need = H.nyqh_entropy_tables(None, 0)
    tables = np.zeros(need, np.uint8)
    if H.nyqh_entropy_tables(tables.ctypes.data, need) != need:
        raise RuntimeError("nyqh_entropy_tables failed")
    cap = 11200
    payload = np.zeros(cap * 320, np.uint8)
    desc = np.zeros(cap * 12, np.uint8)
    finfo = np.zeros(8, np.int64)
    t0 = time.perf_counter()
    if H.nyqh_frame_table(raw, len(raw), cap, payload, payload.size, desc.ctypes.data, finfo) != 0:
        raise RuntimeError("nyqh_frame_table failed")
    walk = time.perf_counter() - t0
    nf, nbytes = int(finfo[2]), int(finfo[4])
    slot = int(ctx.lib.nyq_celt_entropy_slot_bytes(2, 3))           # holds any frame (the records never leave the device)
    tot = streams * nf
    d_tab = torch.from_numpy(tables).to(dev)
    d_pay = torch.from_numpy(payload[:nbytes].copy()).to(dev)
    d_desc = torch.from_numpy(np.tile(desc[:nf * 12], streams)).to(dev)
    Z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    d_sym, d_info, d_energy, d_state = Z((tot, slot), torch.uint8), Z((tot, 16), torch.uint8), Z((tot, 672), torch.uint8), Z((streams, 516), torch.uint8)
    d_tr, d_pp, d_pg, d_pt = Z(tot, torch.uint8), Z(tot, torch.int32), Z(tot, torch.float32), Z(tot, torch.int32)
    d_freq, d_out = Z((tot, 2, 960), torch.float32), Z((streams, nf * 960, 2), torch.float32)
    d_pcm = torch.empty((streams * 2, nf * 960), device=dev)
    d_work = torch.empty(ctx.celt_synth_work_floats(streams, nf, 2), device=dev)
    torch.cuda.synchronize(dev)

    def entropy():
        ctx.celt_entropy_dev(3, d_tab.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_desc.data_ptr(), streams, nf, 2, d_sym.data_ptr(),
                             d_info.data_ptr(), d_energy.data_ptr(), d_state.data_ptr(), True, slot)

    def rest():
        ctx.celt_entropy_split_dev(d_info.data_ptr(), tot, d_tr.data_ptr(), d_pp.data_ptr(), d_pg.data_ptr(), d_pt.data_ptr())
        ctx.celt_shape_slots_dev(3, d_sym.data_ptr(), slot, d_freq.data_ptr(), streams, nf, 2)
        ctx.celt_chain_dev(3, d_freq.data_ptr(), d_tr.data_ptr(), d_pp.data_ptr(), d_pg.data_ptr(), d_pt.data_ptr(), 0, 0, 0, 0, 0,
                           d_out.data_ptr(), d_pcm.data_ptr(), d_work.data_ptr(), streams, nf, 2)

    best_e = best_all = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        entropy()
        ctx.synchronize()
        t1 = time.perf_counter()
        rest()
        ctx.synchronize()
        t2 = time.perf_counter()
        if rep:
            best_e, best_all = min(best_e, t1 - t0), min(best_all, t2 - t0)
    flags = d_info.cpu().numpy()[:, 8]
    return {"file": "sb-reverie.opus", "streams": streams, "frames": tot, "payload_bytes_per_frame": nbytes / nf, "record_slot_bytes": slot,
            "entropy_stage_ms": best_e * 1e3, "entropy_stage_frames_per_sec": tot / best_e,
            "bytes_to_pcm_device_resident_ms": best_all * 1e3, "bytes_to_pcm_device_resident_frames_per_sec": tot / best_all,
            "frames_too_large_for_a_record": int((flags & 32 != 0).sum()), "frames_in_error": int((flags & 16 != 0).sum()),
            "host_packet_walk_ms_per_file_one_thread": walk * 1e3, "checksum_stream0": float(d_out[0].double().abs().sum().item()),
            "note": "parity: tests/test_gpu_entropy.py (records equal to the host decoder's on every corpus frame; bytes -> PCM within 1e-6 "
                    "of the host-record path); compare entropy_stage_frames_per_sec with opus_file_decode.long_streams.frames_per_sec, "
                    "which the host entropy stage bounds"}


def device_entropy_files_leg(streams=128, threads=None):
    """File-level decode of `streams` x sb-reverie.opus through the batch decoder with the entropy stage on the host (today's
    default) and on the device (NYQ_DEVICE_ENTROPY=1, read when a decoder is constructed: each form in a process of its own,
    tools/e2e_bench.py), back to back."""
    import subprocess
    out = {}
    for tag, v in (("host_entropy_stage", "0"), ("device_entropy_stage", "1")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py"), str(streams), str(threads or cpu_share()), "sb-reverie.opus"],
                           env=dict(os.environ, NYQ_DEVICE_ENTROPY=v), capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-500:])
        j = json.loads(r.stdout.strip().splitlines()[-1])
        out[tag] = {"wall_s": j["wall_s_of_the_call"], "frames_per_sec": j["frames_per_s"], "host_stage_s": j["breakdown"]["cpu_entropy_s"],
                    "after_host_stage_s": j["breakdown"]["after_cpu_s"]}
    out["streams"] = streams
    out["speedup"] = out["host_entropy_stage"]["wall_s"] / out["device_entropy_stage"]["wall_s"]
    out["pcm_GBps_out_device_form"] = streams * 21472602 * 4 / out["device_entropy_stage"]["wall_s"] / 1e9
    out["note"] = "with the entropy stage on the device the job waits for the PCM's way out (PCIe down + hand-over), not for the host"
    return out


def opus_file_decode_leg(count=256, fname="short.opus", n=421930, threads=None, device=0, channels=2):
    """File-level decode of `count` copies of tests/golden/<fname> (short.opus: 220 stereo 20 ms CELT frames + one
    closing 2.5 ms frame, 123 kbit/s; sb-reverie.opus: 11184 frames = 224 s, BASELINE config 4's file) as ONE
    batch through libnyquist_host (CPU entropy stage in
    threads, IMDCT/post-filter pieces on the GPU behind it, PCIe included), and the same files through the
    reference's own NyquistIO::Load on the same number of host threads (oracle/_ref/libref_decode.so: the
    cpu_baseline of this leg)."""
    threads = threads or cpu_share()
    raw = open(os.path.join(ROOT, "tests", "golden", fname), "rb").read()
    H = ctypes.CDLL(os.path.join(ROOT, "libnyquist_amd", "libnyquist_host.so"))
    H.nyqh_batch_decode_stats8.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_long, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
    H.nyqh_batch_decode_stats8.restype = ctypes.c_long
    H.nyqh_set_devices.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    H.nyqh_set_devices.restype = None
    H.nyqh_set_devices((ctypes.c_int * 1)(device), 1)      # this rank's GPU (an explicit call: the environment is not touched)
    first = np.zeros(n, np.float32)
    stats = np.zeros(8, np.float64)
    H.nyqh_batch_decode_stats8(raw, len(raw), count, threads, first.ctypes.data_as(ctypes.c_void_p), None, n, stats)   # contexts, pinned staging, pooled buffers
    t0 = time.perf_counter()
    got = H.nyqh_batch_decode_stats8(raw, len(raw), count, threads, first.ctypes.data_as(ctypes.c_void_p), None, n, stats)
    wall = time.perf_counter() - t0          # a clock AROUND the call; every result is consumed and released inside it
    if got != n:
        raise RuntimeError(f"nyqh_batch_decode returned {got}")
    cpu_s, tail_s, frames, thr, wall_inside, ndev, gpu_call_s, feeders = stats
    leg = {"file": fname, "files": count, "channels": channels, "frames": int(frames), "host_threads": int(thr), "devices": int(ndev),
           # how busy the GPU side was kept: time the feeder threads spent inside GPU calls (uploads, kernels, downloads),
           # summed, over wall x feeder threads -- the rest of the wall clock the GPU waited for the host's entropy stage
           "gpu_feeder_threads": int(feeders), "gpu_call_seconds_summed_over_feeders": float(gpu_call_s),
           "gpu_busy_fraction": float(gpu_call_s / (wall * feeders)) if feeders else None,
           "seconds": wall, "wall_seconds": wall, "wall_seconds_measured_inside_the_library": float(wall_inside),
           "files_per_sec": count / wall, "frames_per_sec": frames / wall,
           "breakdown_cpu_entropy_stage_s": cpu_s, "breakdown_not_hidden_gpu_and_trim_s": tail_s,
           "wall_over_stage_sum": wall / (cpu_s + tail_s),
           "x_realtime": count * (n / channels / 48000.0) / wall, "checksum_file0": float(first.astype(np.float64).sum())}
    rp = os.path.join(ROOT, "oracle", "_ref", "libref_decode.so")
    if os.path.exists(rp):
        R = ctypes.CDLL(rp)
        R.ref_decode_bench.restype = ctypes.c_double
        R.ref_decode_bench.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.POINTER(ctypes.c_long),
                                       ctypes.POINTER(ctypes.c_double)]
        ns, ck = ctypes.c_long(0), ctypes.c_double(0)
        R.ref_decode_bench(raw, len(raw), min(count, 2 * threads), threads, ctypes.byref(ns), ctypes.byref(ck))   # warm
        secs = R.ref_decode_bench(raw, len(raw), count, threads, ctypes.byref(ns), ctypes.byref(ck))
        leg["cpu_baseline"] = {"kind": "reference", "cores": threads, "seconds": secs, "files_per_sec": count / secs,
                               "samples_per_file": int(ns.value), "checksum_file0": ck.value,
                               "sample": f"{count} in-memory copies of {fname} through the reference's NyquistIO::Load, {threads} threads"}
        leg["vs_cpu_baseline"] = secs / wall          # wall clock against wall clock
    return leg


def cpu_baseline(x_sample, seconds=12.0):
    """Time the CPU path on `x_sample` ([rows][960] float32) for about `seconds` of wall time:
    one thread per usable core, each looping over its own slice until the deadline."""
    from oracle import pyoracle
    ncores = cpu_share()
    rows = x_sample.shape[0]
    per = max(rows // ncores, 64)
    slices = [np.ascontiguousarray(x_sample[i * per:(i + 1) * per]) for i in range(ncores)]
    slices = [s for s in slices if s.shape[0] == per]
    if pyoracle.ref_available():
        ref = pyoracle.Ref()
        kind = "reference"
        what = "reference clt_mdct_backward built from the reference's own sources (oracle/_ref)"

        def run(sl):
            ref.bench(sl, 0, 1)             # ctypes releases the GIL
    else:
        orc = pyoracle.Oracle()
        kind = "port"
        what = "oracle/nyq_oracle.c restatement"

        def run(sl):
            orc.imdct_batch(0, sl, None, nthreads=1, want_tail=True)
    run(slices[0][:64])
    done = [0] * len(slices)
    deadline = time.perf_counter() + seconds

    def work(i):
        while time.perf_counter() < deadline:
            run(slices[i])
            done[i] += slices[i].shape[0]

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(slices))]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    total = sum(done)
    return {"value": total / wall, "unit": "IMDCT/s", "cores": len(slices), "kind": kind,
            "sample": f"{total} nfft-480 rows in {wall:.1f} s: {len(slices)} threads (one per usable host core), each looping over "
                      f"its own {per}-row slice of the bench batch; {what}",
            "per_core": total / wall / len(slices), "seconds": wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1 << 20, help="rows per GPU (default 2^20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the pinned host-buffer (PCIe) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--threads-per-rank", type=int, default=0,
                    help="host threads of each rank's file-decode leg (default: this process's usable cores / world size)")
    ap.add_argument("--preroll-ms", type=float, default=40.0,
                    help="GPU time each leg's operator runs untimed before its warm-up, to measure at steady clocks (0 = none)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-time reduction (nccl = RCCL)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="map every rank to cuda:0 (multi-rank rehearsal on a 1-GPU box; use with --dist-backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the IMDCT path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = args.dist_backend
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
            except Exception as e:      # the data path needs no collective: a CPU process group is enough for the barrier
                print(f"bench.py: RCCL process group failed ({e!r}); using gloo for barrier / max-time", file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group("gloo")
        args.dist_backend = backend
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    import libnyquist_amd as nyq
    ctx = nyq.Context(local_rank)
    stream = torch.cuda.Stream(dev)          # kernels, data generation and events share this stream
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    cus, devname = ctx.device_info()

    # the job is world x args.rows rows; this rank's share (weak scaling: the global batch grows with the GPU count)
    lo, hi = shard_rows(world * args.rows, world, rank)
    rows = hi - lo
    gen = torch.Generator(device=dev)
    gen.manual_seed(480 + rank)
    x = torch.rand((rows, N2), generator=gen, device=dev, dtype=torch.float32).mul_(2.0).sub_(1.0)
    fin = torch.empty((rows, N2), device=dev, dtype=torch.float32)
    tail = torch.empty((rows, HALF_OV), device=dev, dtype=torch.float32)

    def step():
        ctx.imdct_batch_dev(0, x.data_ptr(), 0, fin.data_ptr(), tail.data_ptr(), rows)

    def barrier():
        if world > 1:
            dist.barrier()

    def preroll(fn, ms=None):
        """Run `fn` back to back for about `ms` of GPU time before a leg's warm-up: after idle (host-side set-up, a CPU
        leg, data generation with small kernels) the GPU's clocks take 10-20 ms of load to come back up -- the kernel
        trace of profiles/r03_l shows the same launch taking 1.21, 1.19, 1.16 ... 0.88 ms over the first dozen launches of a
        leg -- and a throughput figure should be the steady state a decode service runs in.  Untimed; reported as
        config.clock_preroll_ms."""
        ms = args.preroll_ms if ms is None else ms
        if ms <= 0:
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        spent = 0.0
        while spent < ms:
            e0.record(stream)
            for _ in range(4):
                fn()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            spent += e0.elapsed_time(e1)

    def timed_pass():
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(stream)
            step()
            b.record(stream)
        torch.cuda.synchronize(dev)
        barrier()
        return time.perf_counter() - t0, [a.elapsed_time(b) for a, b in ev]

    # (1) the same K steps WITHOUT the clock pre-roll (W warm-up steps only, as rounds 1-2 measured): reported as value_cold /
    # kernel_avg_ms_cold so that rounds stay comparable; (2) pre-roll, warm-up, and the K steps `value` is computed from
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    elapsed_cold, kern_ms_cold = timed_pass()
    preroll(step)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    elapsed, kern_ms = timed_pass()

    # measured on-node device copy: the SAME bytes the kernel launch reads (x) and writes (fin), moved by the library's tuned
    # plain copies (nyq_device_copy_dev: grid-stride and chunk-per-wave float4 forms, the survey of tools/copybench.hip), every
    # form timed at steady clocks, the best reported -- the practical ceiling of a kernel that writes as much as it reads
    copy_gbs, copy_form, copy_all = None, None, None
    if rank == 0:
        try:
            nb = x.numel() * 4
            forms = ctx.lib.nyq_device_copy_forms()
            cp = lambda f: ctx._ck(ctx.lib.nyq_device_copy_dev(ctx.h, fin.data_ptr(), x.data_ptr(), nb, f))
            preroll(lambda: cp(0))
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            copy_all = {}
            for rnd in range(3):                # forms interleaved, best of three rounds of four launches each
                for f in range(forms):
                    cp(f)
                    c0.record(stream)
                    for _ in range(4):
                        cp(f)
                    c1.record(stream)
                    torch.cuda.synchronize(dev)
                    g = 4 * 2 * nb / (c0.elapsed_time(c1) * 1e-3) / 1e9
                    nm = ctx.lib.nyq_device_copy_form_name(f).decode()
                    copy_all[nm] = max(copy_all.get(nm, 0.0), g)
            copy_form = max(copy_all, key=copy_all.get)
            copy_gbs = copy_all[copy_form]
            if not torch.equal(fin, x):
                raise RuntimeError("device copy produced a different buffer")
        except Exception as e:
            copy_all = {"error": repr(e)}
        step()                                  # restore fin for the parity check below
        torch.cuda.synchronize(dev)

    # secondary figure (not the headline metric): Opus 20 ms stereo frames/s through the frame-sequence
    # operator nyq_celt_synth_dev on the measured sb-reverie.opus frame mix (2.8 % transient frames,
    # BASELINE.md section 2), 1024 concurrent streams x 256 frames, device resident.
    # Opus frames/s: the device-resident frames -> PCM chain (frame synthesis, then post-filter + de-emphasis +
    # interleave) on every rank's own streams; the whole-job rate uses the slowest rank's time, like `value`
    synth = None
    if True:
        ns, nf, ch = 1024, 256, 2
        gs = torch.Generator(device=dev)
        gs.manual_seed(4)
        sfreq = torch.randn((ns, nf, ch, N2), generator=gs, device=dev) * 30.0
        strans = (torch.rand((ns, nf), generator=gs, device=dev) < 0.028).to(torch.uint8)
        spcm = torch.empty((ns, ch, nf * N2), device=dev)
        sstate = torch.zeros((ns * ch, HALF_OV), device=dev)
        swork = torch.empty(ctx.celt_synth_work_floats(ns, nf, ch), device=dev)

        def sstep():
            ctx.celt_synth_dev(3, sfreq.data_ptr(), strans.data_ptr(), spcm.data_ptr(), sstate.data_ptr(),
                               swork.data_ptr(), ns, nf, ch)
        preroll(sstep)                          # (a leg that starts right after host-side work sees ramping clocks)
        for _ in range(3):
            sstep()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(stream)
        for _ in range(10):
            sstep()
        s1.record(stream)
        torch.cuda.synchronize(dev)
        sms = s0.elapsed_time(s1) / 10
        synth = {"stereo_frames_per_sec": ns * nf / (sms * 1e-3), "channel_frames_per_sec": ns * nf * ch / (sms * 1e-3),
                 "ms_per_call": sms, "algorithmic_GBps": ns * nf * ch * ALG_BYTES_PER_IMDCT / (sms * 1e-3) / 1e9,
                 "config": f"{ns} streams x {nf} frames x {ch} ch, LM 3, 2.8 % transient frames, chained carry"}
        # the stage after it: pitch post-filter + de-emphasis + interleave (nyq_celt_post_dev) on the same batch
        try:
            # what real streams look like (corpus + short.opus): the post-filter is on in about 70 % of
            # the frames and its period is short
            ppitch = torch.randint(15, 80, (ns, nf), generator=gs, device=dev, dtype=torch.int32)
            pon = (torch.rand((ns, nf), generator=gs, device=dev) < 0.7).float()
            pgain = pon * (torch.randint(1, 9, (ns, nf), generator=gs, device=dev) * 0.09375).float()
            ptap = torch.randint(0, 3, (ns, nf), generator=gs, device=dev, dtype=torch.int32)
            pout = torch.empty((ns, nf * N2, ch), device=dev)

            def pstep():
                ctx.celt_post_dev(3, spcm.data_ptr(), ppitch.data_ptr(), pgain.data_ptr(), ptap.data_ptr(), 0, 0, 0, 0,
                                  pout.data_ptr(), ns, nf, ch)
            preroll(pstep)
            for _ in range(3):
                pstep()
            s0.record(stream)
            for _ in range(10):
                pstep()
            s1.record(stream)
            torch.cuda.synchronize(dev)
            pms = s0.elapsed_time(s1) / 10
            synth["post_filter_ms_per_call"] = pms
            synth["post_filter_algorithmic_GBps"] = ns * nf * ch * ALG_BYTES_PER_IMDCT / (pms * 1e-3) / 1e9
            synth["post_filter_config"] = "70 % of the frames filtered, period uniform 15..79, gain 0.09..0.75, random tapset"
            # the chain as ONE operator (nyq_celt_chain_dev: freq[] -> interleaved PCM), timed as a unit
            def cstep():
                ctx.celt_chain_dev(3, sfreq.data_ptr(), strans.data_ptr(), ppitch.data_ptr(), pgain.data_ptr(), ptap.data_ptr(), 0, 0,
                                   sstate.data_ptr(), 0, 0, pout.data_ptr(), spcm.data_ptr(), swork.data_ptr(), ns, nf, ch)
            preroll(cstep)
            for _ in range(3):
                cstep()
            s0.record(stream)
            for _ in range(10):
                cstep()
            s1.record(stream)
            torch.cuda.synchronize(dev)
            synth["chain_ms_per_call"] = s0.elapsed_time(s1) / 10
            synth["chain_GBps_freq_in_plus_pcm_out"] = ns * nf * ch * ALG_BYTES_PER_IMDCT / (synth["chain_ms_per_call"] * 1e-3) / 1e9
            # the same chain on the parameters of a REAL stream: every stream a window of sb-reverie.opus's own post-filter
            # and transient sequence (BASELINE config 4's file; tests/golden/sb_reverie_pf_params.npz).  Reported beside the
            # synthetic mix above, which stays the figure the whole-job key is computed from.
            try:
                z = np.load(os.path.join(ROOT, "tests", "golden", "sb_reverie_pf_params.npz"))
                tot = len(z["pf_pitch"])
                idx = ((np.arange(ns) * 977) % (tot - nf))[:, None] + np.arange(nf)[None, :]
                rpitch = torch.from_numpy(z["pf_pitch"][idx].astype(np.int32)).to(dev)
                rgain = torch.from_numpy(z["pf_gain_q"][idx].astype(np.float32) * np.float32(0.09375)).to(dev)
                rtap = torch.from_numpy(z["pf_tapset"][idx].astype(np.int32)).to(dev)
                rtrans = torch.from_numpy(z["transient"][idx].astype(np.uint8)).to(dev)

                def rstep():
                    ctx.celt_chain_dev(3, sfreq.data_ptr(), rtrans.data_ptr(), rpitch.data_ptr(), rgain.data_ptr(), rtap.data_ptr(), 0, 0,
                                       sstate.data_ptr(), 0, 0, pout.data_ptr(), spcm.data_ptr(), swork.data_ptr(), ns, nf, ch)
                preroll(rstep)                  # (this leg follows host-side work -- np.load, index building, uploads: the
                for _ in range(3):              # GPU has been idle for tens of ms and its clocks are down)
                    rstep()
                s0.record(stream)
                for _ in range(10):
                    rstep()
                s1.record(stream)
                torch.cuda.synchronize(dev)
                rms = s0.elapsed_time(s1) / 10
                synth["chain_on_real_stream_parameters"] = {
                    "ms_per_call": rms, "stereo_frames_per_sec": ns * nf / (rms * 1e-3),
                    "parameters": "windows of sb-reverie.opus's own sequence: 82 % of the frames filtered, median period 435, 2.8 % transient"}
                del rpitch, rgain, rtap, rtrans
            except Exception as e:
                synth["chain_on_real_stream_parameters"] = {"error": repr(e)}
            del ppitch, pgain, ptap, pout
        except Exception as e:
            synth["post_filter_error"] = repr(e)
        del sfreq, spcm, swork
        chain_ms = synth.get("chain_ms_per_call", synth["ms_per_call"] + synth.get("post_filter_ms_per_call", float("inf")))
        if world > 1:
            tt = torch.tensor([chain_ms], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            chain_ms = float(tt.item())
        synth["whole_job_stereo_frames_per_sec_synthesis_plus_post_filter"] = (
            world * ns * nf / (chain_ms * 1e-3) if chain_ms != float("inf") else None)
        synth["n_gpus"] = world

    # The reference's FFI hands over HOST buffers (mdct.c:52-55).  nyq_imdct_batch on pinned buffers: upload,
    # kernel and download pipelined over three streams; the rate is PCIe-bound and is reported beside `value`,
    # never as it.
    host_boundary = None
    if rank == 0 and world == 1 and not args.no_host_leg:
        try:
            hrows = 1 << 17
            hx = nyq.pinned_empty((hrows, N2), np.float32)
            hx[:] = x[:hrows].cpu().numpy()
            hfin = nyq.pinned_empty((hrows, N2), np.float32)
            htail = nyq.pinned_empty((hrows, HALF_OV), np.float32)
            fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            call = lambda: ctx._ck(ctx.lib.nyq_imdct_batch(ctx.h, 0, fp(hx), None, fp(hfin), fp(htail), hrows))
            call()                                                    # scratch + copy streams warm
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                call()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            moved = hrows * (N2 * 4 + N2 * 4 + HALF_OV * 4)
            host_boundary = {"rows": hrows, "seconds": best, "imdct_per_sec": hrows / best,
                             "pcie_GBps_both_directions": moved / best / 1e9,
                             "note": "nyq_imdct_batch on pinned host buffers in and out (nyq_host_alloc), best of 5"}
            del hfin, htail
            # the reference's own per-call offload interface (mdct.c:219-254 -> processMDCTCuda): one row per call
            trig, window = ctx.get_tables()
            one_in = np.ascontiguousarray(x[0].cpu().numpy())
            one_out = np.zeros(N2 + HALF_OV, np.float32)
            L = ctx.lib
            call_args = (fp(one_in), fp(one_out), fp(trig), 1920, 0, 1, ctypes.c_float(2 * 3.141592653 * 0.125 / 1920), 120, fp(window))
            for _ in range(50):
                L.processMDCTCuda(*call_args)
            ncall = 2000
            t0 = time.perf_counter()
            for _ in range(ncall):
                L.processMDCTCuda(*call_args)
            host_boundary["dropin_processMDCTCuda_us_per_call"] = (time.perf_counter() - t0) / ncall * 1e6
            L.cleanupCudaBuffers()
            del hx
        except Exception as e:
            host_boundary = {"error": repr(e)}

    # File-level decode on EVERY rank: each decodes its own copies on its own GPU with its share of the host threads
    # (nyqh_set_devices selects the device of the C entry point); the whole-job rate uses the slowest rank's wall clock.
    file_leg = None
    if not args.no_host_leg:
        try:
            thr = args.threads_per_rank if args.threads_per_rank > 0 else max(1, cpu_share() // world)
            file_leg = opus_file_decode_leg(256 if world == 1 else 128, threads=thr, device=local_rank)
            if world == 1:
                # BASELINE config 4's file (224 s per stream): the GPU walks it in time slices behind the entropy stage
                file_leg["long_streams"] = opus_file_decode_leg(32, "sb-reverie.opus", 21472602, threads=thr, device=local_rank)
                # BASELINE config 5's shape: 7.1 surround (8 channels in 5 elementary streams, 3 coupled; the corpus file made
                # with the reference's surround encoder stands in for Rachel8ch.opus, which the reference mount lacks): channel
                # mapping, trim and gain happen in the kernels' store phase, one download per file (row f3)
                try:
                    file_leg["band_shapes_on_device"] = band_shapes_leg(ctx, dev)
                except Exception as e:  # noqa: BLE001
                    file_leg["band_shapes_on_device"] = {"error": str(e)}
                try:
                    file_leg["entropy_stage_on_device"] = device_entropy_leg(ctx, dev)
                except Exception as e:  # noqa: BLE001
                    file_leg["entropy_stage_on_device"] = {"error": str(e)}
                try:
                    file_leg["long_streams_128_host_vs_device_entropy"] = device_entropy_files_leg(128, thr)
                except Exception as e:  # noqa: BLE001
                    file_leg["long_streams_128_host_vs_device_entropy"] = {"error": str(e)}
                file_leg["surround_7_1"] = opus_file_decode_leg(128, os.path.join("corpus", "surround71_20ms_320k.opus"), 384000, threads=thr,
                                                                device=local_rank, channels=8)
        except Exception as e:
            file_leg = {"error": repr(e)} if file_leg is None else dict(file_leg, long_streams_error=repr(e))
        if world > 1:                        # (every rank takes part, whatever happened above)
            tt = torch.tensor([file_leg.get("wall_seconds", float("inf")), file_leg.get("gpu_busy_fraction") or 0.0,
                               float(file_leg.get("host_threads", 0))], device=red_dev, dtype=torch.float64)
            every = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(every, tt)
            walls = [float(e[0].item()) for e in every]
            if "files" in file_leg:
                # The file-level leg is bound by the host's bit-serial entropy stage: every rank decodes
                # with ITS share of the node's cores, so files/s follows the host threads, not the GPU count -- the per-rank
                # GPU-busy fractions say how much of each GPU the host could feed.
                file_leg["whole_job"] = {"n_gpus": world, "files": world * file_leg["files"], "slowest_rank_wall_seconds": max(walls),
                                         "files_per_sec": world * file_leg["files"] / max(walls),
                                         "per_rank_wall_seconds": walls,
                                         "per_rank_gpu_busy_fraction": [float(e[1].item()) for e in every],
                                         "per_rank_host_threads": [int(e[2].item()) for e in every],
                                         "host_threads_total": int(sum(e[2].item() for e in every))}

    my_kern_avg_ms = sum(kern_ms) / len(kern_ms)
    kern_avg_ms_cold = sum(kern_ms_cold) / len(kern_ms_cold)
    per_rank_kern_ms, ranks_seen = [my_kern_avg_ms], 1
    if world > 1:
        t = torch.tensor([elapsed, elapsed_cold, kern_avg_ms_cold], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_cold, kern_avg_ms_cold = float(t[0].item()), float(t[1].item()), float(t[2].item())
        k = torch.tensor([my_kern_avg_ms], device=red_dev, dtype=torch.float64)
        gathered = [torch.zeros_like(k) for _ in range(world)]
        dist.all_gather(gathered, k)                 # every rank's own mean kernel time (HIP events on its own stream)
        per_rank_kern_ms = [float(g.item()) for g in gathered]
        kern_avg_ms = max(per_rank_kern_ms)          # the roofline of the job is that of its slowest GPU
        ones = torch.ones(1, device=red_dev, dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)  # how many ranks the process group really joined
        ranks_seen = int(round(float(ones.item())))
    else:
        kern_avg_ms = my_kern_avg_ms

    if rank == 0:
        # parity spot check inside the bench: 4096 rows vs the oracle (relative RMS)
        from oracle.pyoracle import Oracle
        take = min(4096, rows)
        xs = x[:take].cpu().numpy()
        want_fin, want_tail = Oracle().imdct_batch(0, xs, None, nthreads=4)
        got_fin, got_tail = fin[:take].cpu().numpy(), tail[:take].cpu().numpy()
        num = np.sqrt(np.mean((got_fin.astype(np.float64) - want_fin) ** 2) + np.mean((got_tail.astype(np.float64) - want_tail) ** 2))
        den = np.sqrt(np.mean(want_fin.astype(np.float64) ** 2) + np.mean(want_tail.astype(np.float64) ** 2))
        parity = float(num / den)

        total = world * rows * args.steps
        value = total / elapsed
        achieved = ALG_BYTES_PER_IMDCT * rows / (kern_avg_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(rows)
        out = {
            "metric": "N=480 IMDCTs/sec (batched)",
            "value": value,
            "unit": "IMDCT/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # the same K steps before the clock pre-roll (W warm-up steps after set-up, as rounds 1-2 measured `value`)
            "value_cold": total / elapsed_cold,
            "ms_per_step_cold": elapsed_cold / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[2]: synthetic nfft-480 IMDCT batch (MDCT N=1920, shift 0, stride 1, overlap 120, zero carry)",
                       "rows_per_gpu": rows, "global_rows": world * args.rows, "rank0_rows": [lo, hi], "seed": 480,
                       "outputs": "960 finished samples + 60-float tail per row",
                       "parallelism": f"rows sharded over {world} GPU(s), no collective",
                       "clock_preroll_ms": args.preroll_ms,
                       "device": devname, "compute_units": cus},
            "opus_stereo_20ms_frames_per_sec": value / 2.0,
            "parity_rel_rms_vs_oracle": parity,
            "opus_frame_synthesis": synth,
            "host_boundary": host_boundary,
            "opus_file_decode": file_leg,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "nyq::imdct_rows_kernel<32, KCfg<1,false,0>>", "kernel_avg_ms": kern_avg_ms,
                         "kernel_median_ms_rank0": float(np.median(kern_ms)),
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_IMDCT * rows,
                         "kernel_avg_ms_cold": kern_avg_ms_cold,
                         # HBM bytes the launch really moves (PMC) over its time, beside the best plain copy of the same buffers
                         "kernel_traffic_GBps": (traffic / (kern_avg_ms * 1e-3) / 1e9 if traffic else None),
                         "measured_device_copy_GBps": copy_gbs,
                         "measured_device_copy_form": copy_form,
                         "measured_device_copy_all_forms_GBps": copy_all,
                         "frac_of_measured_copy": (achieved / copy_gbs if copy_gbs else None),
                         "per_rank_kernel_avg_ms": per_rank_kern_ms,
                         "per_rank_frac": [ALG_BYTES_PER_IMDCT * rows / (m * 1e-3) / 1e9 / HBM_PEAK_GBS for m in per_rank_kern_ms]},
            # which process group carried the barrier / MAX / gather (the data path has no collective), and how many ranks
            # answered an all-reduce of ones: "did RCCL see N ranks" can be read off the line
            "dist_backend": (args.dist_backend if world > 1 else None),
            "ranks_seen": ranks_seen,
        }
        if world > 1:                               # every rank is past its timed work: release them before the CPU leg
            dist.barrier()
        if not args.no_cpu_baseline:                # (any world size: the reference's CPU path timed in the same run)
            try:
                out["cpu_baseline"] = cpu_baseline(x[: 1 << 16].cpu().numpy(), args.cpu_seconds)
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            except Exception as e:          # never lose the GPU line to a host-side problem
                out["cpu_baseline"] = None
                out["cpu_baseline_error"] = repr(e)
        print(json.dumps(out), flush=True)

    if world > 1:
        if rank != 0:
            dist.barrier()                          # (rank 0 passed this barrier before its CPU leg)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
