#!/usr/bin/env python
"""Per-step backbone error of a precision mode against the reference's teacher-forced trajectory fixtures (GPU): the numbers DESIGN.md section 5
and the documented-miss assertions of tests/test_gpu_round4.py quote.   python tools/step_margins.py [fp16|fp32] [fixture ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import _teacher_forced_steps  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
names = sys.argv[2:] or ["full_denovo_n300_T5_gain03", "full_denovo_n300_T5_gain03_seed11", "full_denovo_n300_T5_gain05", "full_denovo_n64_T20_gain03"]
for name in names:
    r = _teacher_forced_steps(name, prec)
    print(f"{prec} {name}: x_(t-1) worst {r[:, 1].max():.3e} A, x_0 prediction worst {r[:, 2].max():.3e} A; per step x_(t-1) [" +
          " ".join(f"{x:.2e}" for x in r[:, 1]) + "]  t = [" + " ".join(f"{x:.3f}" for x in r[:, 0]) + "]")
