import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden
from test_gpu_parity import _net, _feats
G = load_golden("fwd_full_denovo_n64.npz")
def run(mask):
    os.environ["FDIPT_CHAIN_MASK"] = hex(mask)
    net, _, conf = _net("full_denovo_n64", G, "bf16")
    out = net(_feats(G), trace=True)
    return out["trace_node"].cpu().numpy().copy()
base = run(0)
names = ["TRANSITION","FFN","OUTPROJ","POST","INPROJ","SKIP","ETINIT","A1","AF","NE72","NE88","TORSION"]
for k,nm in enumerate(names):
    t = run(1 << k)
    errs = [float(np.linalg.norm(t[b]-base[b])/np.linalg.norm(base[b])) for b in range(5)]
    print(nm, ["%.3g" % e for e in errs])
