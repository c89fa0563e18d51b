"""Where does the throughput mode's psi / backbone error come from?  bf16 variants against the reference goldens (developer aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_parity import _net, _feats, load_golden, kabsch_free_rmsd

for gname in ("fwd_full_denovo_n64.npz",):
    G = load_golden(gname)
    variants = {"fp32": ("fp32", {}), "fp16": ("fp16", {})}  # (an fp32 torsion head alone: psi error -25 %; the node representation dominates)
    for tag, (prec, env) in variants.items():
        os.environ.pop("FDIPT_TORSION_F32", None)
        os.environ.update(env)
        net, _, conf = _net("full_denovo_n64", G, prec)
        out = net(_feats(G), trace=True)
        psi = out["psi"].cpu().numpy()
        ang = np.arctan2(psi[..., 0], psi[..., 1]); ref = np.arctan2(G["out_psi"][..., 0], G["out_psi"][..., 1])
        dpsi = np.abs(np.angle(np.exp(1j * (ang - ref))))
        node = out["trace_node"].cpu().numpy()[-1]
        refn = G["tr_node_3"] * G["in_res_mask"][..., None]
        print(f"{tag:18s} psi err rad: max {dpsi.max():.2e} mean {dpsi.mean():.2e} | last node rel {np.linalg.norm(node - refn) / np.linalg.norm(refn):.2e}"
              f" | atom37 rmsd {kabsch_free_rmsd(out['atom37'].cpu().numpy(), G['out_atom37']):.2e} | CA max {np.abs(out['rigids'].cpu().numpy()[..., 4:] - G['out_rigids'][..., 4:]).max():.2e}")
