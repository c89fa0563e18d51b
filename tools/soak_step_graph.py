#!/usr/bin/env python
"""Soak of the default loop (step-cursor launches replayed as HIP graphs) against the launch-by-launch loop: whole T-step trajectories of bench.py's
batch, every returned array compared bit for bit.   python tools/soak_step_graph.py [n_runs=5] [T=500] [N=300] [B=8]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from framedipt_amd import config, sharding  # noqa: E402
from framedipt_amd.diffusion import SE3Diffuser  # noqa: E402
from framedipt_amd.inference import inference_fn  # noqa: E402
from framedipt_amd.model import ScoreNetwork  # noqa: E402
from framedipt_amd.sampler import UnconditionalSampler  # noqa: E402

runs, T, N, B = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 5), (2, 500), (3, 300), (4, 8)))
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B * runs}), d, "cuda")
bad = 0
for r in range(runs):
    feats, tape = sharding.stack_items([sharding.seeded_item(ds, r * B + i, 11, d, T, 0.01) for i in range(B)])
    kw = dict(num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    ref = inference_fn(net, d, feats, graph=False, **kw)
    got = inference_fn(net, d, feats, **kw)
    diff = [k for k in ref if not np.array_equal(np.asarray(ref[k].cpu() if torch.is_tensor(ref[k]) else ref[k]),
                                                 np.asarray(got[k].cpu() if torch.is_tensor(got[k]) else got[k]))]
    fin = all(np.isfinite(np.asarray(v.cpu() if torch.is_tensor(v) else v)).all() for v in got.values())
    print(f"run {r}: T={T} N={N} B={B}  differing arrays: {diff or 'none'}  finite: {fin}", flush=True)
    bad += bool(diff) or not fin
print("SOAK", "FAILED" if bad else "ok", f"({runs} trajectories of {T} steps)")
sys.exit(1 if bad else 0)
