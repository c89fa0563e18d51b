import re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", l)
        if m: d[m.group(1)[:46]] = (int(m.group(2)), float(m.group(4)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
print(f"{'kernel':48s} calls      A      B   B-A(us x calls/step)")
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 46
tot = 0
for k, (n, t) in sorted(a.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:24]:
    tb = b.get(k, (0, 0))[1]
    d = (tb - t) * n / steps
    tot += d if tb else 0
    print(f"{k:48s} {n:5d} {t:6.2f} {tb:6.2f} {d:8.1f}")
print("sum of differences per step (us):", round(tot, 1))
