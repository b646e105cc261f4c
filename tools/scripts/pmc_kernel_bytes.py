#!/usr/bin/env python3
"""Per-kernel HBM bytes from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv): mean per launch, FETCH doubled
(gfx950 note of the guide), KiB -> bytes.  usage: pmc_kernel_bytes.py <fetch dir> <write dir> [name filter ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == counter:
                per[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
        for (_, name), v in per.items():
            acc[name].append(v)
    return acc


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
for name in sorted(set(f) | set(w)):
    if sys.argv[3:] and not any(k in name for k in sys.argv[3:]):
        continue
    fr = sum(f[name]) / max(1, len(f[name])) * 2 * 1024
    wr = sum(w[name]) / max(1, len(w[name])) * 1024
    print(f"{name[:90]:90s} launches {len(f[name]):3d}/{len(w[name]):3d}  read {fr / 1e9:7.3f} GB  write {wr / 1e9:7.3f} GB  total {(fr + wr) / 1e9:7.3f} GB")
