"""tools/entropy_time.py [copies] [file] -- the device's entropy stage (nyq_celt_entropy_dev: a frame per lane + the energy pass) on
`copies` x the file's frames as independent streams: wall time per call around a stream synchronise (best of 5), frames/s, and
the host decoder's symbol stage on one thread beside it."""
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

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "sb-reverie.opus"
H = load_host()
u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
H.nyqh_entropy_tables.argtypes = [C.c_void_p, C.c_long]
H.nyqh_entropy_tables.restype = C.c_long
H.nyqh_frame_table.argtypes = [C.c_char_p, C.c_long, C.c_long, u8, C.c_long, C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
ctx = nyq.Context(0)
raw = open(os.path.join(t.GOLDEN, name), "rb").read()
need = H.nyqh_entropy_tables(None, 0)
tables = np.zeros(need, np.uint8)
H.nyqh_entropy_tables(tables.ctypes.data, need)
cap = 12000
payload = np.zeros(cap * 1275 // 4, np.uint8)
desc = np.zeros(cap, t.DESC)
finfo = np.zeros(8, np.int64)
assert H.nyqh_frame_table(raw, len(raw), cap, payload, payload.size, desc.ctypes.data, finfo) == 0
ch, nf, frame = int(finfo[0]), int(finfo[2]), int(finfo[3])
lm = {120: 0, 240: 1, 480: 2, 960: 3}[frame]
slot = ctx.lib.nyq_celt_entropy_slot_bytes(ch, lm)
dev = torch.device("cuda", 0)
d_tab = torch.from_numpy(tables).to(dev)
d_pay = torch.from_numpy(payload[:int(finfo[4])].copy()).to(dev)
d_desc = torch.from_numpy(np.tile(desc[:nf], copies).view(np.uint8)).to(dev)
d_sym = torch.zeros((copies * nf, slot), dtype=torch.uint8, device=dev)
d_info = torch.zeros((copies * nf, 16), dtype=torch.uint8, device=dev)
d_energy = torch.zeros((copies * nf, 672), dtype=torch.uint8, device=dev)
d_state = torch.zeros((copies, 43 * 3 * 4), dtype=torch.uint8, device=dev)
d_freq = torch.zeros((copies * nf, ch, frame), device=dev)
torch.cuda.synchronize(dev)
best = 1e9
for rep in range(6):
    t0 = time.perf_counter()
    ctx.celt_entropy_dev(lm, d_tab.data_ptr(), d_pay.data_ptr(), d_pay.numel(), d_desc.data_ptr(), copies, nf, ch, d_sym.data_ptr(), d_info.data_ptr(),
                         d_energy.data_ptr(), d_state.data_ptr(), True, slot)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    if rep:
        best = min(best, dt)
bshape = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    ctx.celt_shape_slots_dev(lm, d_sym.data_ptr(), slot, d_freq.data_ptr(), copies, nf, ch)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    if rep:
        bshape = min(bshape, dt)
info = d_info.cpu().numpy().view(t.INFO).reshape(-1)
print(json.dumps({"file": name, "streams": copies, "frames": copies * nf, "payload_bytes_per_stream": int(finfo[4]),
                  "entropy_ms_per_call": best * 1e3, "entropy_frames_per_sec": copies * nf / best,
                  "shape_from_spread_records_ms": bshape * 1e3, "too_large": int((info["flags"] & 32 != 0).sum()),
                  "errors": int((info["flags"] & 16 != 0).sum())}))
