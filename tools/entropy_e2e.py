"""tools/entropy_e2e.py [copies] [file] -- frames' BYTES to PCM with the entropy stage on the device: `copies` streams of the file's
frames; page-locked payload + descriptors up, nyq_celt_entropy_dev + split + nyq_celt_shape_lm_dev + nyq_celt_chain_dev on the
device, PCM down into page-locked memory; wall time of the whole (best of 4) and of the device part alone, beside the host's
packet walk (nyqh_frame_table, one thread, per file).  The batch decoder does not use this path yet."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402

torch.cuda.init()
import libnyquist_amd as nyq  # noqa: E402
import test_gpu_entropy as t  # noqa: E402
from test_host_decoder import load_host  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 32
name = sys.argv[2] if len(sys.argv) > 2 else "sb-reverie.opus"
H = load_host()
u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
H.nyqh_entropy_tables.argtypes = [C.c_void_p, C.c_long]
H.nyqh_entropy_tables.restype = C.c_long
H.nyqh_frame_table.argtypes = [C.c_char_p, C.c_long, C.c_long, u8, C.c_long, C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
ctx = nyq.Context(0)
raw = open(os.path.join(t.GOLDEN, name), "rb").read()
t0 = time.perf_counter()
ch, nf, lm, payload, desc = t.frame_table(H, raw)
walk = time.perf_counter() - t0
t0 = time.perf_counter()
ch, nf, lm, payload, desc = t.frame_table(H, raw)
walk = min(walk, time.perf_counter() - t0)
need = H.nyqh_entropy_tables(None, 0)
tables = np.zeros(need, np.uint8)
H.nyqh_entropy_tables(tables.ctypes.data, need)
dev = torch.device("cuda", 0)
n = 120 << lm
d_tab = torch.from_numpy(tables).to(dev)
# every stream has its own copy of the bytes (as independent files would)
h_pay = torch.from_numpy(np.tile(payload, copies)).pin_memory()
dd = np.tile(desc, copies)
dd["offset"] = (dd["offset"].astype(np.int64) + np.repeat(np.arange(copies, dtype=np.int64) * payload.size, nf)).astype(np.uint32)
h_desc = torch.from_numpy(dd.view(np.uint8)).pin_memory()
d_pay = torch.empty_like(h_pay, device=dev)
d_desc = torch.empty_like(h_desc, device=dev)
h_out = torch.empty((copies, nf * n, ch), dtype=torch.float32).pin_memory()
bufs = None
best = dev_best = 1e9
for rep in range(5):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    d_pay.copy_(h_pay, non_blocking=True)
    d_desc.copy_(h_desc, non_blocking=True)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    bufs = t.bytes_to_pcm_on_device(ctx, lm, ch, copies, nf, d_tab, d_pay, d_desc, bufs)
    ctx.synchronize()
    t2 = time.perf_counter()
    h_out.copy_(bufs["out"], non_blocking=True)
    torch.cuda.synchronize(dev)
    t3 = time.perf_counter()
    if rep:
        if t3 - t0 < best:
            best, parts = t3 - t0, (t1 - t0, t2 - t1, t3 - t2)
        dev_best = min(dev_best, t2 - t1)
info = bufs["info"].cpu().numpy().view(t.INFO).reshape(-1)
print(json.dumps({"file": name, "streams": copies, "frames": copies * nf, "channels": ch, "payload_MB": copies * payload.size / 1e6,
                  "pcm_MB": h_out.numel() * 4 / 1e6, "wall_ms": best * 1e3, "upload_ms": parts[0] * 1e3, "device_ms": parts[1] * 1e3,
                  "download_ms": parts[2] * 1e3, "device_ms_best": dev_best * 1e3, "frames_per_sec": copies * nf / best,
                  "host_packet_walk_ms_per_file_one_thread": walk * 1e3, "too_large": int((info["flags"] & 32 != 0).sum()),
                  "errors": int((info["flags"] & 16 != 0).sum()), "checksum_stream0": float(h_out[0].double().abs().sum())}))
