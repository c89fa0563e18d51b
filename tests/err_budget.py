"""Which operand roundings of the half-precision mode cost how much accuracy?  (developer aid, CPU only)

Runs the NumPy oracle forward on a reference golden with fp16 operand rounding switched on for one group of products at
a time (activations AND weights rounded before the product, fp32 accumulation: what the MFMA path does), and prints the
error of psi / CA / last node representation against the unrounded oracle.  Errors of independent groups add in
quadrature, so the table says where split (hi + lo) operands pay.

    python tests/err_budget.py [golden] [fp16|bf16] [bb_gain] [product|all] [rows]

``bb_gain`` overrides the BackboneUpdate weight scale of the fixture (0.3: frames move ~3 A per block, as trained weights
would); ``product`` restricts the table and the ALL row to the groups the fp16 product mode runs on plain (unsplit) fp16
operands — the node path runs on split operands there and does not round.
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
BB_GAIN = float(sys.argv[3]) if len(sys.argv) > 3 else None
PRODUCT = len(sys.argv) > 4 and sys.argv[4] == "product"


def rnd(x):
    x = np.asarray(x, dtype=np.float32)
    if KIND == "fp16":
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def rnd_w(w):
    """Weight rounding.  FDIPT_ERRB_WDIFF=1: error-diffusion rounding along the input dimension (the residual of element k is carried into
    element k + 1 before it is rounded, so every partial row sum of the rounded row is within half an ulp of the exact one)."""
    if os.environ.get("FDIPT_ERRB_WDIFF") != "1" or np.ndim(w) != 2:
        return rnd(w)
    w = np.asarray(w, dtype=np.float64)
    out = np.empty_like(w, dtype=np.float32)
    carry = np.zeros(w.shape[0])
    for k in range(w.shape[1]):
        t = w[:, k] + carry
        r = rnd(t.astype(np.float32)).astype(np.float64)
        out[:, k] = r
        carry = t - r
    return out


GROUP_OF = [  # substring of the parameter name -> group
    ("node_embedder", "embed_node"), ("edge_embedder.0", None), ("edge_embedder", "embed_edge"),
    ("linear_q_points", "ipa_pts"), ("linear_kv_points", "ipa_pts"), ("linear_q", "ipa_qkv"), ("linear_kv", "ipa_qkv"),
    ("linear_b", "pair_bias"), ("down_z", "opair"), ("linear_out", "linear_out"), ("skip_embed", "skip"),
    ("seq_tfmr", "tfmr"), ("post_tfmr", "post"), ("node_transition", "transition"), ("bb_update", None),
    # EdgeTransition per layer (round 5): et0 = initial_embed (per residue; the HIP path runs it on split operands), et1 / et2 = the two
    # trunk layers, etf = final_layer; "et" in a selection means all four
    ("edge_transition_0.initial_embed", "et0"), ("edge_transition_1.initial_embed", "et0"), ("edge_transition_2.initial_embed", "et0"),
    ("trunk.0", "et1"), ("trunk.2", "et2"), ("final_layer", "etf"),
    ("torsion_pred.linear_final", None), ("torsion_pred", "torsion"),
]
ET_SUB = ("et0", "et1", "et2", "etf")
GROUPS = ["embed_node", "embed_edge", "ipa_pts", "ipa_qkv", "ipa_attn", "pair_bias", "opair", "linear_out", "skip", "tfmr",
          "tfmr_attn", "post", "transition", "et", "z_store", "torsion"]


# groups whose products run on plain fp16 operands in the fp16 product mode (everything else is on split operands / fp32)
PRODUCT_GROUPS = ["embed_edge", "ipa_pts", "ipa_qkv", "ipa_attn", "pair_bias", "opair", "skip", "tfmr_attn", "et", "z_store"]


def want(group, block=None, idx=None):
    """(round activations?, round weights?) for a product of ``group`` in trunk block ``block`` (``idx``: which matmul of an
    attention).  Entries of Net.on: ``group[.x|.w][@block][#idx]`` — e.g. ``et.w@1`` rounds only the weights of the
    EdgeTransition after block 1, ``ipa_attn#0`` only the first matmul (Q K^T) of the IPA attention."""
    rx = rw = False
    for e in Net.on:
        g, i = (e.split("#") + [None])[:2]
        g, b = (g.split("@") + [None])[:2]
        g, part = (g.split(".") + ["xw"])[:2]
        if not (g == group or (g == "et" and group in ET_SUB)) or (b is not None and block is not None and int(b) != block) or (i is not None and idx is not None and int(i) != idx):
            continue
        rx, rw = rx or "x" in part, rw or "w" in part
    return rx, rw


class Net(osn.ScoreNetwork):
    on = frozenset()
    where = None  # "ipa" / "tfmr": which attention the np.matmul proxy is inside
    block = None
    mm_idx = 0

    def _group(self, name):
        for sub, g in GROUP_OF:
            if sub in name:
                return g
        raise KeyError(name)

    def _lin(self, name, x):
        w, b = self.sd[name + ".weight"], self.sd[name + ".bias"]
        g = self._group(name)
        if g is not None:
            digits = [int(t) for t in name.replace(".", "_").split("_") if t.isdigit()]
            blk = digits[0] if ("trunk" in name and digits) else None
            rx, rw = want(g, blk)
            x, w = (rnd(x) if rx else x), (rnd_w(w) if rw else w)
        return osn.linear(x, w, b)

    def ipa(self, b, *a, **k):
        Net.where, Net.block, Net.mm_idx = "ipa_attn", b, 0
        try:
            return super().ipa(b, *a, **k)
        finally:
            Net.where = None

    def seq_tfmr(self, b, x, mask):
        Net.where, Net.block, Net.mm_idx = "tfmr_attn", b, 0
        # in_proj goes through osn.linear directly: round here
        try:
            return super().seq_tfmr(b, x, mask)
        finally:
            Net.where = None

    def embed(self, *a, **k):
        node, edge = super().embed(*a, **k)
        return node, (rnd(edge) if want("z_store", 0)[0] else edge)

    def edge_transition(self, b, node, edge):
        z = super().edge_transition(b, node, edge)
        return rnd(z) if want("z_store", b + 1)[0] else z


class NPProxy:
    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def matmul(a, b):
        if Net.where is not None:
            # IPA: 0 = Q K^T, 1 = P V, 2 = P v_pts, 3 = P pair_z; sequence attention: per layer 0 = Q K^T, 1 = P V
            idx = Net.mm_idx if Net.where == "ipa_attn" else Net.mm_idx % 2
            Net.mm_idx += 1
            ra, rb = want(Net.where, Net.block, idx)
            a, b = (rnd(a) if ra else a), (rnd(b) if rb else b)
        return np.matmul(a, b)


osn.np = NPProxy()
_lin0 = osn.linear


def _linear_hook(x, w, b):  # seq_tfmr's in_proj calls the module-level linear()
    if Net.where == "tfmr_attn":
        rx, rw = want("tfmr", Net.block)
        x, w = (rnd(x) if rx else x), (rnd(w) if rw else w)
    return _lin0(x, w, b)


osn.linear = _linear_hook


def main():
    gname = sys.argv[1] if len(sys.argv) > 1 else "fwd_full_denovo_n64"
    G = load_golden(gname + ".npz")
    tables = dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))
    if BB_GAIN is not None:
        G = dict(G)
        G["bb_gain"] = BB_GAIN
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
    groups = [g for g in GROUPS if not PRODUCT or g in PRODUCT_GROUPS]
    if len(sys.argv) > 5:  # explicit rows: comma-separated selections, '+' joins selections into one row
        groups = sys.argv[5].split(",")
    print(f"{gname}  operand rounding: {KIND}  bb_gain: {float(G['bb_gain'])}" + ("  (groups the fp16 product mode rounds)" if PRODUCT else ""))
    print(f"{'group':12s} {'psi max':>9s} {'psi rms':>9s} {'CA max':>9s} {'CA rms':>9s} {'node rel':>9s} {'bb rmsd':>9s}")
    tot = np.zeros(3)
    for g in groups + ["ALL"]:
        out, nd = run(groups if g == "ALL" else g.split("+"))
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
