#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run (the .db it writes): per kernel start/end relative to the first, and
the idle gap in front of each.  usage: trace_gaps.py results.db [first_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.stream_id, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[skip:skip + n]
t0 = rows[0][1]
last_end = None
for name, st, en, sid, qid in rows:
    short = name.split("(")[0]
    for a, b in (("_ZN3nyq", ""), ("ENS_4KCfgILi1ELb0ELi0ELi0EEEEEvNS_9SynthArgsEPKfS5_.kd", ""), ("EvNS_8PostArgsEPKf.kd", "")):
        short = short.replace(a, b)
    gap = (st - last_end) / 1000.0 if last_end is not None else 0.0
    print(f"{(st - t0) / 1000.0:10.1f} us  +{(en - st) / 1000.0:8.1f} us  gap {gap:7.1f}  q{qid} s{sid}  {short[:70]}")
    last_end = max(last_end or en, en)
