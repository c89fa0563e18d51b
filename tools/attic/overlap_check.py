"""How much of the GPU time of a rocprofv3 kernel trace (rocpd database) has >= 2 kernels in flight?  (developer aid)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select start, end, name{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
rows = rows[len(rows) // 2:]  # second half: steady state
ev = []
for r in rows:
    ev.append((r[0], 1)); ev.append((r[1], -1))
ev.sort()
depth, last, busy1, busy2 = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
span = ev[-1][0] - ev[0][0]
print(f"span {span/1e6:.2f} ms, >=1 kernel {busy1/1e6:.2f} ms, >=2 kernels {busy2/1e6:.2f} ms ({100*busy2/max(busy1,1):.1f} %)")
if qcol:
    print("queues/streams:", sorted({r[3] for r in rows}))
if len(sys.argv) > 2:
    t0 = rows[0][0]
    for r in rows[:int(sys.argv[2])]:
        print(f"q{r[3]} {(r[0]-t0)/1e3:9.1f} -> {(r[1]-t0)/1e3:9.1f} us  {r[2][:50]}")
