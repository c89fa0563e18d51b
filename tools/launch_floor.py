#!/usr/bin/env python
"""What do N dependent kernel launches cost on this box when the kernels do nothing?  (the floor under a ~67-launch reverse step)
python tools/launch_floor.py [n=67]"""
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 67
x = torch.zeros(64, device="cuda")
def seq():
    for _ in range(n):
        x.add_(1.0)
for _ in range(3):
    seq()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g.capture_begin()
    seq()
    g.capture_end()
for name, fn in (("eager", seq), ("graph replay", g.replay)):
    for reps in (20, 200):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f"{name:13s} {reps:4d} x {n} one-wave kernels: GPU {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per sequence = {e0.elapsed_time(e1) / reps / n * 1e3:5.2f} us per launch; "
              f"host {host / reps * 1e6:8.1f} us per sequence")
