"""ORACLE (test infrastructure, not product code): the score-network forward on torch-CPU.

The NumPy restatement (``oracle/score_network.py``) is the parity checker; its element-wise passes are single-threaded, which makes it
a poor CPU *baseline* (4x slower than the reference's own torch-CPU loop on 8 cores).  This subclass evaluates the same forward —
same formulas, same order, float32 — with torch CPU ops, which thread over the host cores the way the reference's
``ScoreNetwork.forward`` (``framedipt/model/score_network.py:218-275``, ``framedipt/model/ipa_pytorch.py:105-572``) does.  Used by
``bench.py``'s ``cpu_baseline`` (k = all cores and k = 1) and pinned against the NumPy oracle in tests/test_oracle_forward.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import frames as fr
from . import score_network as osn


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


class TorchScoreNetwork(osn.ScoreNetwork):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.tsd = {k: _t(v) for k, v in self.sd.items()}

    # ------------------------------------------------------------------ small layers
    def _lin(self, name, x):
        return F.linear(_t(x), self.tsd[name + ".weight"], self.tsd[name + ".bias"]).numpy()

    def _tlin(self, name, x):
        return F.linear(x, self.tsd[name + ".weight"], self.tsd[name + ".bias"])

    def _ln(self, name, x):
        w = self.tsd[name + ".weight"]
        return F.layer_norm(_t(x), (w.shape[0],), w, self.tsd[name + ".bias"], 1e-5).numpy()

    def _tln(self, name, x):
        w = self.tsd[name + ".weight"]
        return F.layer_norm(x, (w.shape[0],), w, self.tsd[name + ".bias"], 1e-5)

    # ------------------------------------------------------------------ embedder (score_network.py:129-197)
    def embed(self, seq_idx, t, fixed_mask, sc_ca, aatype):
        ec = self.mc.embed
        B, N = seq_idx.shape
        fm = fixed_mask[..., None].astype(np.float32)
        te = np.tile(osn.timestep_embedding(t, ec.index_embed_size)[:, None, :], (1, N, 1))
        if aatype is not None:
            oh = np.eye(21, dtype=np.float32)[aatype]
            eps_te = np.tile(osn.timestep_embedding(np.ones_like(t) * 1e-5, ec.index_embed_size)[:, None, :], (1, N, 1))
            pte = np.concatenate([oh, np.where(fm.astype(bool), eps_te, te), fm], axis=-1)
        else:
            pte = np.concatenate([te, fm], axis=-1)
        pte_t = _t(pte)
        x = torch.cat([pte_t, _t(osn.index_embedding(seq_idx, ec.index_embed_size))], -1)
        p = "embedding_layer.node_embedder."
        x = F.relu(self._tlin(p + "0", x))
        x = F.relu(self._tlin(p + "2", x))
        node = self._tln(p + "5", self._tlin(p + "4", x))
        rel = seq_idx[:, :, None] - seq_idx[:, None, :]
        feats = [pte_t[:, :, None, :].expand(B, N, N, -1), pte_t[:, None, :, :].expand(B, N, N, -1),
                 _t(osn.index_embedding(rel, ec.index_embed_size))]
        if ec.embed_self_conditioning:
            feats.append(_t(osn.distogram(sc_ca, ec.min_bin, ec.max_bin, ec.num_bins)))
        y = torch.cat(feats, -1)
        p = "embedding_layer.edge_embedder."
        y = F.relu(self._tlin(p + "0", y))
        y = F.relu(self._tlin(p + "2", y))
        edge = self._tln(p + "5", self._tlin(p + "4", y))
        return node.numpy(), edge.numpy()

    # ------------------------------------------------------------------ IPA (ipa_pytorch.py:170-329)
    def ipa(self, b, s, z, quat, trans, mask):
        ic = self.mc.ipa
        H, C, Pq, Pv = ic.no_heads, ic.c_hidden, ic.no_qk_points, ic.no_v_points
        B, N, _ = s.shape
        p = f"score_model.trunk.ipa_{b}."
        rot = _t(fr.quat_to_rot(quat))
        tr = _t(trans)
        st, zt = _t(s), _t(z)
        q = self._tlin(p + "linear_q", st).view(B, N, H, C)
        kv = self._tlin(p + "linear_kv", st).view(B, N, H, 2 * C)
        k, v = kv[..., :C], kv[..., C:]

        def pts(name, n_pts):
            x = self._tlin(p + name, st)
            x = torch.stack(torch.split(x, x.shape[-1] // 3, dim=-1), dim=-1)           # [B,N,H*n,3]
            x = torch.einsum("bnij,bnpj->bnpi", rot, x) + tr[:, :, None, :]              # Rigid.apply
            return x.view(B, N, H, n_pts, 3)

        q_pts = pts("linear_q_points", Pq)
        kv_pts = pts("linear_kv_points", Pq + Pv)
        k_pts, v_pts = kv_pts[..., :Pq, :], kv_pts[..., Pq:, :]
        bz = self._tlin(p + "linear_b", zt)
        a = torch.matmul(q.permute(0, 2, 1, 3), k.permute(0, 2, 3, 1)) * math.sqrt(1.0 / (3 * C))
        a = a + math.sqrt(1.0 / 3) * bz.permute(0, 3, 1, 2)
        disp = q_pts[:, :, None] - k_pts[:, None, :]
        pt_att = (disp ** 2).sum(-1)
        hw = F.softplus(self.tsd[p + "head_weights"]).view(1, 1, 1, H, 1) * math.sqrt(1.0 / (3 * (Pq * 9.0 / 2)))
        pt_att = (pt_att * hw).sum(-1) * (-0.5)
        mt = _t(mask)
        sq = 1e5 * (mt[:, :, None] * mt[:, None, :] - 1)
        a = torch.softmax(a + pt_att.permute(0, 3, 1, 2) + sq[:, None], dim=-1)
        o = torch.matmul(a, v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, N, H * C)
        o_pt = torch.matmul(a, v_pts.permute(0, 2, 1, 3, 4).reshape(B, H, N, Pv * 3)).permute(0, 2, 1, 3).reshape(B, N, H, Pv, 3)
        o_pt = torch.einsum("bnji,bnhpj->bnhpi", rot, o_pt - tr[:, :, None, None, :])    # Rigid.invert_apply
        o_norm = torch.sqrt((o_pt ** 2).sum(-1) + 1e-8).reshape(B, N, H * Pv)
        o_pt = o_pt.reshape(B, N, H * Pv, 3)
        pair_z = self._tlin(p + "down_z", zt)
        o_pair = torch.matmul(a.permute(0, 2, 1, 3), pair_z).reshape(B, N, -1)
        feats = torch.cat([o, o_pt[..., 0], o_pt[..., 1], o_pt[..., 2], o_norm, o_pair], -1)
        return self._tlin(p + "linear_out", feats).numpy()

    # ------------------------------------------------------------------ sequence transformer (ipa_pytorch.py:433-443,536-538)
    def seq_tfmr(self, b, x, mask):
        ic = self.mc.ipa
        nh = ic.seq_tfmr_num_heads
        x = _t(x)
        B, N, D = x.shape
        hd = D // nh
        pad = -1e30 * (1 - _t(mask))[:, None, None, :]
        for l in range(ic.seq_tfmr_num_layers):
            p = f"score_model.trunk.seq_tfmr_{b}.layers.{l}."
            qkv = F.linear(x, self.tsd[p + "self_attn.in_proj_weight"], self.tsd[p + "self_attn.in_proj_bias"])
            q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(B, N, nh, hd) for i in range(3))
            att = torch.matmul(q.permute(0, 2, 1, 3), k.permute(0, 2, 3, 1)) * (1.0 / math.sqrt(hd))
            att = torch.softmax(att + pad, dim=-1)
            o = torch.matmul(att, v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, N, D)
            x = self._tln(p + "norm1", x + self._tlin(p + "self_attn.out_proj", o))
            f = self._tlin(p + "linear2", F.relu(self._tlin(p + "linear1", x)))
            x = self._tln(p + "norm2", x + f)
        return x.numpy()

    # ------------------------------------------------------------------ EdgeTransition (ipa_pytorch.py:84-102)
    def edge_transition(self, b, node, edge):
        p = f"score_model.trunk.edge_transition_{b}."
        B, N, _ = node.shape
        ne = self._tlin(p + "initial_embed", _t(node))
        x = torch.cat([_t(edge), ne[:, :, None, :].expand(B, N, N, -1), ne[:, None, :, :].expand(B, N, N, -1)], -1).reshape(B * N * N, -1)
        h = F.relu(self._tlin(p + "trunk.0", x))
        h = F.relu(self._tlin(p + "trunk.2", h))
        y = self._tlin(p + "final_layer", h + x)
        return self._tln(p + "layer_norm", y).reshape(B, N, N, -1).numpy()

    def torsion(self, s):
        p = "score_model.torsion_pred."
        st = _t(s)
        x = self._tlin(p + "linear_2", F.relu(self._tlin(p + "linear_1", st))) + st
        un = self._tlin(p + "linear_final", x)
        den = torch.sqrt(torch.clamp((un ** 2).sum(-1, keepdim=True), min=1e-8))
        return (un / den).numpy()
