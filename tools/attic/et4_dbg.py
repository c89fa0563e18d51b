"""Developer helper: edge_transition4 vs edge_transition3 traces on the n64 golden (GPU)."""
import os, sys
sys.path.insert(0, "tests")
import numpy as np
from conftest import load_golden
import test_gpu_parity as T

G = load_golden("fwd_full_denovo_n64.npz")
outs = {}
for tag, var in (("v4", None), ("v3", "FDIPT_ET_V3")):
    os.environ.pop("FDIPT_ET_V3", None)
    if var:
        os.environ[var] = "1"
    net, _, conf = T._net("full_denovo_n64", G, "bf16")
    out = net(T._feats(G), trace=True)
    outs[tag] = out["trace_edge"].cpu().numpy().copy()
a, c = outs["v3"][1], outs["v4"][1]
print("shape", a.shape, "rel", np.linalg.norm(a - c) / np.linalg.norm(a))
err = np.abs(a - c)
B, N = a.shape[0], a.shape[1]
print("per b:", err.reshape(B, -1).max(1))
print("per i (b0):", np.round(err[0].max(axis=(1, 2)), 2))
print("per j (b0):", np.round(err[0].max(axis=(0, 2)), 2))
print("per ch (b0):", np.round(err[0].max(axis=(0, 1))[:32], 2))
print("sample v3", a[0, 0, 0, :8], "\n       v4", c[0, 0, 0, :8])
print("sample v3", a[0, 9, 5, :8], "\n       v4", c[0, 9, 5, :8])
