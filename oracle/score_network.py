"""ORACLE (test infrastructure, not product code): score-network forward in NumPy.

CPU restatement of ``framedipt/model/score_network.py`` and
``framedipt/model/ipa_pytorch.py`` (FrameDiff fork of IPA) plus
``framedipt/protein/all_atom.py:compute_backbone`` for SURVEY.md section 8 rows
a9-a20.  float32 arrays where the reference computes in float32.

Third-party arithmetic restated: ``torch.nn.TransformerEncoderLayer`` /
``MultiheadAttention`` / ``LayerNorm`` (reference pin pytorch 1.13.1,
``environment.yml:231``; call site ``ipa_pytorch.py:433-443``) following the
published post-norm encoder algorithm; key-padding semantics are those of the
torch>=2 inference fast path (SURVEY.md section 0 finding 9).
"""
from __future__ import annotations

import math

import numpy as np

from . import frames as fr

F32 = np.float32


def linear(x, w, b):
    return (x @ w.T + b).astype(F32, copy=False)


def layer_norm(x, g, b, eps=1e-5):
    x = x.astype(F32, copy=False)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    var = ((x - mu) ** 2).mean(-1, keepdims=True, dtype=F32)
    return ((x - mu) / np.sqrt(var + F32(eps)) * g + b).astype(F32, copy=False)


def relu(x):
    return np.maximum(x, 0)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=axis, keepdims=True)).astype(F32, copy=False)


def softplus(x):
    return np.log1p(np.exp(x))


# exp(-k ln(1e4)/15), k=0..15 exactly as torch float32 evaluates it (score_network.py:49-53); pinned by
# tests/golden/ops.npz["timestep_freqs"].  t*1e4*freq reaches 1e4 rad in float32, so a 1-ulp difference in a
# frequency moves sin/cos by 1e-4: the constants are part of the reference's behaviour.
TIMESTEP_FREQS = np.array([float.fromhex(h) for h in (
    "0x1.0000000000000p+0", "0x1.15142c0000000p-1", "0x1.2be4aa0000000p-2", "0x1.44960e0000000p-3",
    "0x1.5f4ff00000000p-4", "0x1.7c3d2e0000000p-5", "0x1.9b8c2e0000000p-6", "0x1.bd6f1a0000000p-7",
    "0x1.e21c500000000p-8", "0x1.04e74e0000000p-8", "0x1.1a62d60000000p-9", "0x1.31a3320000000p-10",
    "0x1.4acdb40000000p-11", "0x1.660aa40000000p-12", "0x1.8385b80000000p-13", "0x1.a36e2c0000000p-14")],
    dtype=np.float32)


def index_embedding(idx, embed_size=32, max_len=2056):
    """score_network.py:17-38 (float32 arithmetic as torch evaluates it)."""
    k = np.arange(embed_size // 2)
    denom = np.power(float(max_len), 2 * k / embed_size).astype(F32, copy=False)  # == torch float32 pow for these 16 values
    arg = ((np.asarray(idx).astype(F32, copy=False)[..., None] * F32(math.pi)).astype(F32, copy=False) / denom).astype(F32, copy=False)
    return np.concatenate([np.sin(arg), np.cos(arg)], axis=-1).astype(F32, copy=False)


def timestep_embedding(t, dim=32, max_positions=10000):
    """score_network.py:41-64."""
    t = np.asarray(t, dtype=F32) * F32(max_positions)
    half = dim // 2
    if dim == 32 and max_positions == 10000:
        emb = TIMESTEP_FREQS
    else:
        emb = np.exp(np.arange(half, dtype=F32) * F32(-math.log(max_positions) / (half - 1))).astype(F32, copy=False)
    emb = (t[:, None] * emb[None]).astype(F32, copy=False)
    return np.concatenate([np.sin(emb), np.cos(emb)], axis=1).astype(F32, copy=False)


def distogram(pos, min_bin, max_bin, num_bins):
    """framedipt/data/utils.py:541-550."""
    d = np.linalg.norm(pos[:, :, None, :] - pos[:, None, :, :], axis=-1)[..., None]
    lower = np.linspace(min_bin, max_bin, num_bins, dtype=np.float64).astype(F32, copy=False)
    upper = np.concatenate([lower[1:], np.array([1e8], dtype=F32)])
    return ((d > lower) * (d < upper)).astype(F32, copy=False)


class ScoreNetwork:
    """Callable feats -> outputs; ``sd`` is a name->ndarray state dict (reference names)."""

    def __init__(self, model_conf, diffuser, sd, inpainting=False, tables=None):
        self.mc = model_conf
        self.diffuser = diffuser
        self.sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
        self.inpainting = inpainting
        self.tables = tables  # residue-constant tables for compute_backbone
        self.trace = None  # optional dict collecting intermediates

    def _lin(self, name, x):
        return linear(x, self.sd[name + ".weight"], self.sd[name + ".bias"])

    def _ln(self, name, x):
        return layer_norm(x, self.sd[name + ".weight"], self.sd[name + ".bias"])

    # ------------------------------------------------------------- embedder
    def preprocess_aatype(self, aatype, fixed_mask):
        """framedipt/data/utils.py:565-610."""
        if aatype is None or (not self.inpainting and not self.mc.input_aatype):
            return None
        aatype = aatype.astype(np.int64)
        if not self.mc.input_aatype:
            aatype = np.where(fixed_mask.astype(bool), aatype, 20)
        return aatype

    def embed(self, seq_idx, t, fixed_mask, sc_ca, aatype):
        """score_network.py:129-197."""
        ec = self.mc.embed
        B, N = seq_idx.shape
        fm = fixed_mask[..., None].astype(F32, copy=False)
        te = np.tile(timestep_embedding(t, ec.index_embed_size)[:, None, :], (1, N, 1))
        if aatype is not None:
            oh = np.eye(21, dtype=F32)[aatype]
            eps_te = np.tile(timestep_embedding(np.ones_like(t) * 1e-5, ec.index_embed_size)[:, None, :], (1, N, 1))
            comb = np.where(fm.astype(bool), eps_te, te)
            pte = np.concatenate([oh, comb, fm], axis=-1)
        else:
            pte = np.concatenate([te, fm], axis=-1)
        node_feats = [pte, index_embedding(seq_idx, ec.index_embed_size)]
        cc = np.concatenate(
            [np.tile(pte[:, :, None, :], (1, 1, N, 1)), np.tile(pte[:, None, :, :], (1, N, 1, 1))], axis=-1
        ).reshape(B, N * N, -1)
        rel = (seq_idx[:, :, None] - seq_idx[:, None, :]).reshape(B, N * N)
        pair_feats = [cc, index_embedding(rel, ec.index_embed_size)]
        if ec.embed_self_conditioning:
            dg = distogram(sc_ca, ec.min_bin, ec.max_bin, ec.num_bins)
            pair_feats.append(dg.reshape(B, N * N, -1))
        p = "embedding_layer.node_embedder."
        x = np.concatenate(node_feats, axis=-1).astype(F32, copy=False)
        x = relu(self._lin(p + "0", x))
        x = relu(self._lin(p + "2", x))
        node = self._ln(p + "5", self._lin(p + "4", x))
        p = "embedding_layer.edge_embedder."
        y = np.concatenate(pair_feats, axis=-1).astype(F32, copy=False)
        y = relu(self._lin(p + "0", y))
        y = relu(self._lin(p + "2", y))
        edge = self._ln(p + "5", self._lin(p + "4", y)).reshape(B, N, N, -1)
        return node, edge

    # ------------------------------------------------------------------ IPA
    def ipa(self, b, s, z, quat, trans, mask):
        """ipa_pytorch.py:170-329. quat/trans [B,N,*] (trans in scaled units)."""
        ic = self.mc.ipa
        H, C, Pq, Pv = ic.no_heads, ic.c_hidden, ic.no_qk_points, ic.no_v_points
        B, N, _ = s.shape
        p = f"score_model.trunk.ipa_{b}."
        rot = fr.quat_to_rot(quat).astype(F32, copy=False)
        q = self._lin(p + "linear_q", s).reshape(B, N, H, C)
        kv = self._lin(p + "linear_kv", s).reshape(B, N, H, 2 * C)
        k, v = kv[..., :C], kv[..., C:]

        def pts(name, n_pts):
            x = self._lin(p + name, s)  # [B,N,H*n*3] as three planes
            x = np.stack(np.split(x, 3, axis=-1), axis=-1)  # [B,N,H*n,3]
            x = fr.rigid_apply(rot[:, :, None], trans[:, :, None], x).astype(F32, copy=False)
            return x.reshape(B, N, H, n_pts, 3)

        q_pts = pts("linear_q_points", Pq)
        kv_pts = pts("linear_kv_points", Pq + Pv)
        k_pts, v_pts = kv_pts[..., :Pq, :], kv_pts[..., Pq:, :]
        bz = self._lin(p + "linear_b", z)  # [B,N,N,H]
        a = np.matmul(q.transpose(0, 2, 1, 3), k.transpose(0, 2, 3, 1)).astype(F32, copy=False)
        a = a * F32(math.sqrt(1.0 / (3 * C)))
        a = a + F32(math.sqrt(1.0 / 3)) * bz.transpose(0, 3, 1, 2)
        disp = q_pts[:, :, None] - k_pts[:, None, :]  # [B,N,N,H,Pq,3]
        pt_att = (disp**2).sum(-1)
        hw = softplus(self.sd[p + "head_weights"]).astype(F32, copy=False).reshape(1, 1, 1, H, 1)
        hw = hw * F32(math.sqrt(1.0 / (3 * (Pq * 9.0 / 2))))
        pt_att = (pt_att * hw).sum(-1) * F32(-0.5)  # [B,N,N,H]
        sq = F32(1e5) * (mask[:, :, None] * mask[:, None, :] - 1)
        a = a + pt_att.transpose(0, 3, 1, 2) + sq[:, None]
        a = softmax(a.astype(F32, copy=False), axis=-1)  # [B,H,N,N]
        o = np.matmul(a, v.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3).reshape(B, N, H * C).astype(F32, copy=False)
        o_pt = np.matmul(a, v_pts.transpose(0, 2, 1, 3, 4).reshape(B, H, N, Pv * 3)).transpose(0, 2, 1, 3)
        o_pt = o_pt.reshape(B, N, H, Pv, 3).astype(F32, copy=False)  # [B,N,H,Pv,3]
        o_pt = fr.rigid_invert_apply(rot[:, :, None, None], trans[:, :, None, None], o_pt).astype(F32, copy=False)
        o_norm = np.sqrt((o_pt**2).sum(-1) + F32(1e-8)).reshape(B, N, H * Pv)
        o_pt = o_pt.reshape(B, N, H * Pv, 3)
        pair_z = self._lin(p + "down_z", z)  # [B,N,N,cz/4]
        o_pair = np.matmul(a.transpose(0, 2, 1, 3), pair_z).reshape(B, N, -1).astype(F32, copy=False)
        feats = np.concatenate([o, o_pt[..., 0], o_pt[..., 1], o_pt[..., 2], o_norm, o_pair], axis=-1)
        if self.trace is not None:
            self.trace[f"ipa_{b}_feats"] = feats
        return self._lin(p + "linear_out", feats.astype(F32, copy=False))

    def seq_tfmr(self, b, x, mask):
        """nn.TransformerEncoder(post-norm, ReLU, dropout 0), ipa_pytorch.py:433-443,536-538."""
        ic = self.mc.ipa
        nh = ic.seq_tfmr_num_heads
        B, N, D = x.shape
        hd = D // nh
        pad = F32(-1e30) * (1 - mask)[:, None, None, :]  # masked keys excluded (fast-path semantics)
        for l in range(ic.seq_tfmr_num_layers):
            p = f"score_model.trunk.seq_tfmr_{b}.layers.{l}."
            qkv = linear(x, self.sd[p + "self_attn.in_proj_weight"], self.sd[p + "self_attn.in_proj_bias"])
            q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, N, nh, hd) for i in range(3))
            att = np.matmul(q.transpose(0, 2, 1, 3), k.transpose(0, 2, 3, 1)).astype(F32, copy=False) * F32(1.0 / math.sqrt(hd))
            att = softmax(att + pad, axis=-1)
            o = np.matmul(att, v.transpose(0, 2, 1, 3)).transpose(0, 2, 1, 3).reshape(B, N, D).astype(F32, copy=False)
            o = self._lin(p + "self_attn.out_proj", o)
            x = self._ln(p + "norm1", x + o)
            f = self._lin(p + "linear2", relu(self._lin(p + "linear1", x)))
            x = self._ln(p + "norm2", x + f)
        return x

    def edge_transition(self, b, node, edge):
        """ipa_pytorch.py:84-102."""
        p = f"score_model.trunk.edge_transition_{b}."
        B, N, _ = node.shape
        ne = self._lin(p + "initial_embed", node)
        bias = np.concatenate(
            [np.tile(ne[:, :, None, :], (1, 1, N, 1)), np.tile(ne[:, None, :, :], (1, N, 1, 1))], axis=-1
        )
        x = np.concatenate([edge, bias], axis=-1).reshape(B * N * N, -1).astype(F32, copy=False)
        h = relu(self._lin(p + "trunk.0", x))
        h = relu(self._lin(p + "trunk.2", h))
        y = self._lin(p + "final_layer", h + x)
        return self._ln(p + "layer_norm", y).reshape(B, N, N, -1)

    def torsion(self, s):
        """ipa_pytorch.py:332-363 (linear_3 unused by the reference)."""
        p = "score_model.torsion_pred."
        x = self._lin(p + "linear_2", relu(self._lin(p + "linear_1", s))) + s
        un = self._lin(p + "linear_final", x)
        den = np.sqrt(np.maximum((un**2).sum(-1, keepdims=True), F32(1e-8)))
        return (un / den).astype(F32, copy=False)

    # -------------------------------------------------------------- forward
    def __call__(self, feats):
        """score_network.py:218-275 + ipa_pytorch.py:509-572."""
        ic = self.mc.ipa
        bb_mask = feats["res_mask"].astype(F32, copy=False)
        fixed_mask = feats["fixed_mask"].astype(F32, copy=False)
        edge_mask = bb_mask[..., None] * bb_mask[..., None, :]
        aatype = self.preprocess_aatype(feats.get("aatype"), fixed_mask)
        t = np.asarray(feats["t"], dtype=F32)
        node0, edge = self.embed(feats["seq_idx"], t, fixed_mask, feats["sc_ca_t"].astype(F32, copy=False), aatype)
        edge = edge * edge_mask[..., None]
        node0 = node0 * bb_mask[..., None]
        diffuse_mask = (1 - fixed_mask) * bb_mask
        rig = feats["rigids_t"].astype(F32, copy=False)
        q_init, t_init = rig[..., :4], rig[..., 4:]
        cs = F32(ic.coordinate_scaling)
        quat, trans = q_init.copy(), (t_init * cs).astype(F32, copy=False)
        node0 = node0 * bb_mask[..., None]
        node = node0 * bb_mask[..., None]
        if self.trace is not None:
            self.trace["node_init"], self.trace["edge_init"] = node.copy(), edge.copy()
        for b in range(ic.num_blocks):
            tr = "score_model.trunk."
            ipa = self.ipa(b, node, edge, quat, trans, bb_mask) * bb_mask[..., None]
            node = self._ln(f"{tr}ipa_ln_{b}", node + ipa)
            x = np.concatenate([node, self._lin(f"{tr}skip_embed_{b}", node0)], axis=-1)
            x = self.seq_tfmr(b, x, bb_mask)
            node = node + self._lin(f"{tr}post_tfmr_{b}", x)
            p = f"{tr}node_transition_{b}."
            h = relu(self._lin(p + "linear_1", node))
            h = relu(self._lin(p + "linear_2", h))
            node = self._ln(p + "ln", self._lin(p + "linear_3", h) + node)
            node = node * bb_mask[..., None]
            upd = self._lin(f"{tr}bb_update_{b}.linear", node * diffuse_mask[..., None])
            quat, trans = fr.compose_q_update_vec(quat, trans, upd, diffuse_mask[..., None])
            if b < ic.num_blocks - 1:
                edge = self.edge_transition(b, node, edge) * edge_mask[..., None]
            if self.trace is not None:
                self.trace[f"node_{b}"], self.trace[f"edge_{b}"] = node.copy(), edge.copy()
                self.trace[f"rigid_{b}"] = np.concatenate([quat, trans], -1)
        rot_score = self.diffuser.calc_rot_score(q_init, quat, t) * bb_mask[..., None]
        trans_u = (trans / cs).astype(F32, copy=False)
        trans_score = self.diffuser.calc_trans_score(t_init, trans_u, t[:, None, None]) * bb_mask[..., None]
        psi = self.torsion(node)
        gt_psi = feats["torsion_angles_sin_cos"][..., 2, :]
        dm = 1 - fixed_mask[..., None]
        psi = dm * psi + (1 - dm) * gt_psi
        rigids = np.concatenate([quat, trans_u], axis=-1).astype(F32, copy=False)
        atom37, atom14 = compute_backbone(quat, trans_u, psi, aatype, self.tables)
        return {"psi": psi, "rot_score": rot_score, "trans_score": trans_score, "rigids": rigids,
                "atom37": atom37, "atom14": atom14}


def compute_backbone(quat, trans, psi, aatype, tables, rot=None):
    """all_atom.py:147-176 -> feats.py:165-228 -> all_atom.py:108-144.

    quat [*,N,4] (or rot [*,N,3,3]), trans [*,N,3] Angstrom, psi [*,N,2] -> atom37 [*,N,37,3], atom14 [*,N,14,3].
    tables: dict(default_frames [21,8,4,4], group_idx [21,14], atom_mask [21,14], ideal_pos [21,14,3]).
    """
    if rot is None:
        rot = fr.quat_to_rot(quat.astype(F32, copy=False)).astype(F32, copy=False)
    trans = trans.astype(F32, copy=False)
    shp = trans.shape[:-1]
    if aatype is None:
        aatype = np.zeros(shp, dtype=np.int64)
    aatype = np.where(aatype == 20, 0, aatype)
    alpha = np.tile(psi.astype(F32, copy=False)[..., None, :], (1,) * len(shp) + (7, 1))
    d44 = tables["default_frames"].astype(F32, copy=False)[aatype]  # [*,N,8,4,4]
    dr, dt = d44[..., :3, :3], d44[..., :3, 3]
    bb = np.zeros(shp + (1, 2), dtype=F32)
    bb[..., 1] = 1
    al = np.concatenate([bb, alpha], axis=-2)  # [*,N,8,2]
    ar = np.zeros(shp + (8, 3, 3), dtype=F32)
    ar[..., 0, 0] = 1
    ar[..., 1, 1] = al[..., 1]
    ar[..., 1, 2] = -al[..., 0]
    ar[..., 2, 1] = al[..., 0]
    ar[..., 2, 2] = al[..., 1]
    fr_r = fr.rot_matmul(dr, ar).astype(F32, copy=False)  # default_r.compose(all_rots); trans of all_rots = 0
    fr_t = dt.copy()
    # chain chi2..chi4 onto chi1 (feats.py:204-212)
    R, T = [fr_r[..., i, :, :] for i in range(8)], [fr_t[..., i, :] for i in range(8)]
    for i in (5, 6, 7):
        r_new, t_new = fr.rigid_compose(R[i - 1], T[i - 1], R[i], T[i])
        R[i], T[i] = r_new.astype(F32, copy=False), t_new.astype(F32, copy=False)
    fr_r, fr_t = np.stack(R, axis=-3), np.stack(T, axis=-2)
    g_r, g_t = fr.rigid_compose(rot[..., None, :, :], trans[..., None, :], fr_r, fr_t)
    g_r, g_t = g_r.astype(F32, copy=False), g_t.astype(F32, copy=False)
    gi = tables["group_idx"][aatype]  # [*,N,14]
    oh = np.eye(8, dtype=F32)[gi]  # [*,N,14,8]
    a_r = np.einsum("...ag,...gij->...aij", oh, g_r).astype(F32, copy=False)
    a_t = np.einsum("...ag,...gi->...ai", oh, g_t).astype(F32, copy=False)
    ideal = tables["ideal_pos"].astype(F32, copy=False)[aatype]
    pos = (fr.rot_vec_mul(a_r, ideal) + a_t).astype(F32, copy=False) * tables["atom_mask"].astype(F32, copy=False)[aatype][..., None]
    a37 = np.zeros(shp + (37, 3), dtype=F32)
    a37[..., :3, :] = pos[..., :3, :]
    a37[..., 3, :] = pos[..., 4, :]
    a37[..., 4, :] = pos[..., 3, :]
    return a37, pos.astype(F32, copy=False)
