#!/usr/bin/env python
"""Kernel sequence of one sampler step (between two build_feats launches late in the trace) from a rocprofv3 rocpd database."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "build_feats" in r[0]]
a, b = idx[-3], idx[-2]
t0 = rows[a][1]
prev_end = None
for name, s, e in rows[a:b]:
    short = re.sub(r"\(.*", "", name).replace("void ", "")[:60]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.2f} gap  {(e - s) / 1e3:8.2f} us  {short}")
    prev_end = e
print(f"step span {(rows[b][1] - t0) / 1e3:.1f} us, kernel sum {sum((e - s) for _, s, e in rows[a:b]) / 1e3:.1f} us")
