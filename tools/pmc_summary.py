#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-launch HBM traffic figures, with provenance.

Usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <rows> <out.json>

Emits one record per kernel of interest -- the one-launch frames -> PCM kernel `celt_chain_kernel` (round 4), the headline kernel `imdct_rows_kernel<32>` (what bench.py's
`roofline.traffic` quotes), and the two stages of the frames -> PCM chain, `synth_frames_kernel<32, 3>` (key `synth_long_kernel<32>`, its name before round 3's single launch) and the post-filter
kernel (`celt_post_pipe_kernel<3>`, or `celt_post_kernel<3,...>` when the round-1 form ran) -- each with its algorithmic
bytes and the ratio to them, plus the sha of the kernel sources the passes were taken with: bench.py replays the
headline figure only while `libnyquist_amd/csrc` still has that sha.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly
half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane
streaming stores.  The two counters come from SEPARATE --pmc passes.
"""
import csv
import datetime
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_counter(path, name, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def record(fetch_csv, write_csv, kernel, alg_bytes):
    f, nf = mean_counter(fetch_csv, "FETCH_SIZE", kernel)
    w, nw = mean_counter(write_csv, "WRITE_SIZE", kernel)
    if f is None or w is None:
        return None
    read_b, write_b = 2.0 * f * 1024.0, w * 1024.0
    return {"kernel": kernel, "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w, "launches_averaged": [nf, nw],
            "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b, "hbm_bytes_per_launch": read_b + write_b,
            "algorithmic_bytes_per_launch": alg_bytes, "ratio_to_algorithmic": (read_b + write_b) / alg_bytes}


def main():
    fetch_csv, write_csv, rows, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    from bench import csrc_digest
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    chain_units = 1024 * 256 * 2          # bench.py's frames -> PCM leg: streams x frames x channels
    recs = {"imdct_rows_kernel<32>": record(fetch_csv, write_csv, "imdct_rows_kernelILi32E", 7680 * rows) or
            record(fetch_csv, write_csv, "imdct_rows_kernel<32", 7680 * rows),
            "synth_long_kernel<32>": record(fetch_csv, write_csv, "synth_frames_kernelILi32ELi3", 7680 * chain_units) or
            record(fetch_csv, write_csv, "synth_frames_kernel<32, 3", 7680 * chain_units),
            "post_filter_kernel": record(fetch_csv, write_csv, "celt_post_pipe_kernel", 7680 * chain_units) or
            record(fetch_csv, write_csv, "celt_post_kernel", 7680 * chain_units),
            # round 4: freq[] -> interleaved PCM in ONE launch (3840 B in + 3840 B out per channel-frame)
            "celt_chain_kernel": record(fetch_csv, write_csv, "celt_chain_kernel", 7680 * chain_units)}
    head_rec = recs["imdct_rows_kernel<32>"]
    d = {"rows": rows, "kernel": "imdct_rows_kernel<32>", "csrc_sha16": csrc_digest(), "git_head": head,
         "measured_on": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"),
         "corrections": "FETCH_SIZE x2 (gfx950 wide-read under-count), KiB -> bytes x1024; separate --pmc passes",
         "note": "synth_long_kernel<32> = synth_frames_kernel<32, 3>, long-frame and transient-frame roles in one launch since round 3 (before: the long-frame part only; 2.8 % of the frames are transient and went "
                 "through synth_short_kernel): its algorithmic bytes are counted for all frames, so its ratio reads slightly low",
         "kernels": recs}
    if head_rec:
        d.update({k: head_rec[k] for k in ("FETCH_SIZE_KiB_raw", "WRITE_SIZE_KiB_raw", "launches_averaged", "read_bytes_per_launch",
                                           "write_bytes_per_launch", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch",
                                           "ratio_to_algorithmic")})
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main()
