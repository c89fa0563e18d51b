"""Which operand roundings of the half-precision mode cost how much accuracy?  (developer aid, CPU only)

Runs the NumPy oracle forward on a reference golden with fp16 operand rounding switched on for one group of products at
a time (activations AND weights rounded before the product, fp32 accumulation: what the MFMA path does), and prints the
error of psi / CA / last node representation against the unrounded oracle.  Errors of independent groups add in
quadrature, so the table says where split (hi + lo) operands pay.

    python tests/err_budget.py [golden] [fp16|bf16]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (lives under tests/: only tests, smoke() and the bench baseline may use the oracle)
import oracle.score_network as osn  # noqa: E402
from conftest import kabsch_free_rmsd, load_golden  # noqa: E402
from test_oracle_forward import _feats, _model  # noqa: E402

KIND = sys.argv[2] if len(sys.argv) > 2 else "fp16"


def rnd(x):
    x = np.asarray(x, dtype=np.float32)
    if KIND == "fp16":
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


GROUP_OF = [  # substring of the parameter name -> group
    ("node_embedder", "embed_node"), ("edge_embedder.0", None), ("edge_embedder", "embed_edge"),
    ("linear_q_points", "ipa_pts"), ("linear_kv_points", "ipa_pts"), ("linear_q", "ipa_qkv"), ("linear_kv", "ipa_qkv"),
    ("linear_b", "pair_bias"), ("down_z", "opair"), ("linear_out", "linear_out"), ("skip_embed", "skip"),
    ("seq_tfmr", "tfmr"), ("post_tfmr", "post"), ("node_transition", "transition"), ("bb_update", None),
    ("edge_transition", "et"), ("torsion_pred.linear_final", None), ("torsion_pred", "torsion"),
]
GROUPS = ["embed_node", "embed_edge", "ipa_pts", "ipa_qkv", "ipa_attn", "pair_bias", "opair", "linear_out", "skip", "tfmr",
          "tfmr_attn", "post", "transition", "et", "z_store", "torsion"]


class Net(osn.ScoreNetwork):
    on = frozenset()
    where = None  # "ipa" / "tfmr": which attention the np.matmul proxy is inside

    def _group(self, name):
        for sub, g in GROUP_OF:
            if sub in name:
                return g
        raise KeyError(name)

    def _lin(self, name, x):
        w, b = self.sd[name + ".weight"], self.sd[name + ".bias"]
        if self._group(name) in self.on:
            x, w = rnd(x), rnd(w)
        return osn.linear(x, w, b)

    def ipa(self, *a, **k):
        Net.where = "ipa_attn"
        try:
            return super().ipa(*a, **k)
        finally:
            Net.where = None

    def seq_tfmr(self, b, x, mask):
        Net.where = "tfmr_attn"
        # in_proj goes through osn.linear directly: round here
        try:
            return super().seq_tfmr(b, x, mask)
        finally:
            Net.where = None

    def embed(self, *a, **k):
        node, edge = super().embed(*a, **k)
        return node, (rnd(edge) if "z_store" in self.on else edge)

    def edge_transition(self, b, node, edge):
        z = super().edge_transition(b, node, edge)
        return rnd(z) if "z_store" in self.on else z


class NPProxy:
    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def matmul(a, b):
        if Net.where in Net.on:
            a, b = rnd(a), rnd(b)
        return np.matmul(a, b)


osn.np = NPProxy()
_lin0 = osn.linear


def _linear_hook(x, w, b):  # seq_tfmr's in_proj calls the module-level linear()
    if Net.where == "tfmr_attn" and "tfmr" in Net.on:
        x, w = rnd(x), rnd(w)
    return _lin0(x, w, b)


osn.linear = _linear_hook


def main():
    gname = sys.argv[1] if len(sys.argv) > 1 else "fwd_full_denovo_n64"
    G = load_golden(gname + ".npz")
    tables = dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))
    model, _ = _model(gname[4:], G, tables)
    model.__class__ = Net

    def run(on):
        Net.on = frozenset(on)
        model.trace = {}
        out = model(_feats(G))
        last = max(int(k[5:]) for k in model.trace if k.startswith("node_") and k[5:].isdigit())
        return out, model.trace[f"node_{last}"]

    ref, nref = run(())
    ang = lambda p: np.arctan2(p[..., 0], p[..., 1])  # noqa: E731
    print(f"{gname}  operand rounding: {KIND}")
    print(f"{'group':12s} {'psi max':>9s} {'psi rms':>9s} {'CA max':>9s} {'CA rms':>9s} {'node rel':>9s} {'bb rmsd':>9s}")
    tot = np.zeros(3)
    for g in GROUPS + ["ALL"]:
        out, nd = run(GROUPS if g == "ALL" else (g,))
        dpsi = np.abs(np.angle(np.exp(1j * (ang(out["psi"]) - ang(ref["psi"])))))
        dca = np.linalg.norm(out["rigids"][..., 4:] - ref["rigids"][..., 4:], axis=-1)
        rel = np.linalg.norm(nd - nref) / np.linalg.norm(nref)
        rm = kabsch_free_rmsd(out["atom37"], ref["atom37"])
        if g != "ALL":
            tot += np.array([np.sqrt((dpsi**2).mean()), np.sqrt((dca**2).mean()), rel]) ** 2
        print(f"{g:12s} {dpsi.max():9.2e} {np.sqrt((dpsi**2).mean()):9.2e} {dca.max():9.2e} {np.sqrt((dca**2).mean()):9.2e} "
              f"{rel:9.2e} {rm:9.2e}")
    print("quadrature sum of the groups: psi rms %.2e  CA rms %.2e  node rel %.2e" % tuple(np.sqrt(tot)))


if __name__ == "__main__":
    main()
