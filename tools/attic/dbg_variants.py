"""Compare the node trace of forward variants selected by environment switches (developer aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_parity import _net, _feats, load_golden

G = load_golden("fwd_full_denovo_n64.npz")
outs = {}
variants = {"default": {}, "sattn_unfused": {"FDIPT_SATTN_UNFUSED": "1"}, "no_tail": {"FDIPT_NO_TFMR_TAIL": "1"},
            "v2": {"FDIPT_ATTN_V2": "1"}, "v2_no_tail": {"FDIPT_ATTN_V2": "1", "FDIPT_NO_TFMR_TAIL": "1"}}
for tag, env in variants.items():
    for k in ("FDIPT_SATTN_UNFUSED", "FDIPT_NO_TFMR_TAIL", "FDIPT_ATTN_V2"):
        os.environ.pop(k, None)
    os.environ.update(env)
    net, _, conf = _net("full_denovo_n64", G, "bf16")
    out = net(_feats(G), trace=True)
    outs[tag] = out["trace_node"].cpu().numpy().copy()
ref = outs["no_tail"]
for tag, o in outs.items():
    print(tag, [round(float(np.linalg.norm(o[b] - ref[b]) / np.linalg.norm(ref[b])), 4) for b in range(1, 5)],
          "vs golden", [round(float(np.linalg.norm(o[b + 1] - G[f"tr_node_{b}"]) / np.linalg.norm(G[f"tr_node_{b}"])), 4) for b in range(4)])
