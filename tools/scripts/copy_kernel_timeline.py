#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (csv output): kernels and copies merged by start time,
for a window of the run.  usage: copy_kernel_timeline.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv> [skip] [n]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
ev = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][-60:], ""))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "")), r.get("Bytes", "")))
ev.sort()
print(len(ev), "events")
sel = ev[skip:skip + n]
t0 = sel[0][0]
for st, en, kind, name, nbytes in sel:
    dur = (en - st) / 1000.0
    rate = f"{int(nbytes) / (en - st):6.1f} GB/s" if nbytes and en > st else ""
    print(f"{(st - t0) / 1000.0:10.1f} us +{dur:8.1f}  {kind} {name:60s} {nbytes:>10s} {rate}")
