#!/usr/bin/env python
"""FDIPT_DBG_TWICE aid: for kernels launched twice back to back, average duration of the first (cold) vs second (warm) launch."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
prev = None
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "")[:50]
    d = (e - s) / 1e3
    if prev is not None and prev[0] == short:
        a = agg.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1; a[1] += prev[1]; a[2] += d
        prev = None
    else:
        prev = (short, d)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:52s} pairs {a[0]:4d}  first {a[1]/a[0]:7.2f} us  second {a[2]/a[0]:7.2f} us")
