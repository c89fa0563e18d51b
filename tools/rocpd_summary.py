#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (markdown)."""
import re
import sqlite3
import sys


def main(db_path, out_path=None, top=40):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)
        a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        lines.append(f"| `{k[:110]}` | {a[0]} | {a[1]:.1f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100*a[1]/tot:.1f} |")
    lines.append(f"\ntotal kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches")
    txt = "\n".join(lines)
    if out_path:
        open(out_path, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
