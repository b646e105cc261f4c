#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into the per-launch HBM traffic figure bench.py reports.

Usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <rows> <out.json>

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it
is doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import csv
import json
import sys


def mean_counter(path, name, kernel="imdct_rows_kernel"):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


def main():
    fetch_csv, write_csv, rows, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, nf = mean_counter(fetch_csv, "FETCH_SIZE")
    w, nw = mean_counter(write_csv, "WRITE_SIZE")
    read_b = 2.0 * f * 1024.0
    write_b = w * 1024.0
    d = {"rows": rows, "kernel": "imdct_rows_kernel<32>",
         "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w, "launches_averaged": [nf, nw],
         "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
         "hbm_bytes_per_launch": read_b + write_b,
         "algorithmic_bytes_per_launch": 7680 * rows,
         "ratio_to_algorithmic": (read_b + write_b) / (7680 * rows),
         "corrections": "FETCH_SIZE x2 (gfx950 wide-read under-count), KiB -> bytes x1024; separate --pmc passes"}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main()
