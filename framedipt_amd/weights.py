"""Parameter inventory of the score network and deterministic synthetic weights.

``param_shapes`` restates the ``state_dict`` layout of the reference
``ScoreNetwork`` (``framedipt/model/score_network.py:67-216``,
``framedipt/model/ipa_pytorch.py:36-507``) so released checkpoints load by name.
No checkpoint ships with the reference (SURVEY.md section 0 finding 6), so tests and
``bench.py`` use ``synth_state_dict``: every tensor is drawn from a
``numpy.random.default_rng([seed, index])`` stream with a per-kind scale; the
zero-initialised ("final") layers get small non-zero weights so that the frame
update path is exercised.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


def node_feat_dim(model_conf, inpainting: bool) -> int:
    d = model_conf.embed.index_embed_size + 1
    if inpainting or model_conf.input_aatype:
        d += 21
    return d


def param_shapes(model_conf, inpainting: bool = False) -> "OrderedDict[str, tuple]":
    """Ordered name -> shape map == reference ``ScoreNetwork(...).state_dict()``."""
    ipa = model_conf.ipa
    emb = model_conf.embed
    cs, cz = model_conf.node_embed_size, model_conf.edge_embed_size
    d1 = node_feat_dim(model_conf, inpainting)
    node_in = d1 + emb.index_embed_size
    edge_in = 2 * d1 + emb.index_embed_size + (emb.num_bins if emb.embed_self_conditioning else 0)
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def lin(name, out_d, in_d):
        s[name + ".weight"] = (out_d, in_d)
        s[name + ".bias"] = (out_d,)

    def ln(name, d):
        s[name + ".weight"] = (d,)
        s[name + ".bias"] = (d,)

    e = "embedding_layer."
    lin(e + "node_embedder.0", cs, node_in)
    lin(e + "node_embedder.2", cs, cs)
    lin(e + "node_embedder.4", cs, cs)
    ln(e + "node_embedder.5", cs)
    lin(e + "edge_embedder.0", cz, edge_in)
    lin(e + "edge_embedder.2", cz, cz)
    lin(e + "edge_embedder.4", cz, cz)
    ln(e + "edge_embedder.5", cz)
    h, c, pq, pv = ipa.no_heads, ipa.c_hidden, ipa.no_qk_points, ipa.no_v_points
    d_t = ipa.c_s + ipa.c_skip
    for b in range(ipa.num_blocks):
        t = "score_model.trunk."
        p = f"{t}ipa_{b}."
        s[p + "head_weights"] = (h,)
        lin(p + "linear_q", h * c, ipa.c_s)
        lin(p + "linear_kv", 2 * h * c, ipa.c_s)
        lin(p + "linear_q_points", h * pq * 3, ipa.c_s)
        lin(p + "linear_kv_points", h * (pq + pv) * 3, ipa.c_s)
        lin(p + "linear_b", h, ipa.c_z)
        lin(p + "down_z", ipa.c_z // 4, ipa.c_z)
        lin(p + "linear_out", ipa.c_s, h * (ipa.c_z // 4 + c + pv * 4))
        lin(p + "linear_rbf", 1, 20)
        ln(f"{t}ipa_ln_{b}", ipa.c_s)
        lin(f"{t}skip_embed_{b}", ipa.c_skip, cs)
        for l in range(ipa.seq_tfmr_num_layers):
            q = f"{t}seq_tfmr_{b}.layers.{l}."
            s[q + "self_attn.in_proj_weight"] = (3 * d_t, d_t)
            s[q + "self_attn.in_proj_bias"] = (3 * d_t,)
            lin(q + "self_attn.out_proj", d_t, d_t)
            lin(q + "linear1", d_t, d_t)
            lin(q + "linear2", d_t, d_t)
            ln(q + "norm1", d_t)
            ln(q + "norm2", d_t)
        lin(f"{t}post_tfmr_{b}", ipa.c_s, d_t)
        for i in (1, 2, 3):
            lin(f"{t}node_transition_{b}.linear_{i}", ipa.c_s, ipa.c_s)
        ln(f"{t}node_transition_{b}.ln", ipa.c_s)
        lin(f"{t}bb_update_{b}.linear", 6, ipa.c_s)
        if b < ipa.num_blocks - 1:
            q = f"{t}edge_transition_{b}."
            bias_d = ipa.c_s // 2
            hid = 2 * bias_d + cz
            lin(q + "initial_embed", bias_d, ipa.c_s)
            lin(q + "trunk.0", hid, hid)
            lin(q + "trunk.2", hid, hid)
            lin(q + "final_layer", cz, hid)
            ln(q + "layer_norm", cz)
    for i in (1, 2, 3):
        lin(f"score_model.torsion_pred.linear_{i}", ipa.c_s, ipa.c_s)
    lin("score_model.torsion_pred.linear_final", 2, ipa.c_s)
    return s


_SOFTPLUS_INV_1 = 0.541324854612918  # reference framedipt/model/layers.py:209-212


BB_GAIN = 0.03  # keeps |log(R_pred^-1 R_t)| within ~5 sigma_min, where the IGSO(3) series is conditioned


def _gain(name: str, bb_gain: float = BB_GAIN) -> float:
    if "bb_update" in name:
        return bb_gain
    for tag in ("linear_out", "post_tfmr", "skip_embed", "node_transition", "final_layer",
                "linear_final", "torsion_pred.linear_3"):
        if tag in name and (tag != "node_transition" or name.endswith("linear_3.weight")):
            return 0.3
    return 1.0


def synth_tensor(name: str, shape: tuple, seed: int, index: int, bb_gain: float = BB_GAIN) -> np.ndarray:
    rng = np.random.default_rng([seed, index])
    x = rng.standard_normal(shape)
    if name.endswith("head_weights"):
        x = _SOFTPLUS_INV_1 + 0.1 * x
    elif len(shape) == 2:
        x = x * (_gain(name, bb_gain) / np.sqrt(shape[1]))
    elif name.endswith("bias"):
        x = 0.1 * x
    else:  # LayerNorm gamma
        x = 1.0 + 0.1 * x
    return x.astype(np.float32)


def synth_state_dict(shapes: "OrderedDict[str, tuple]", seed: int = 0,
                     bb_gain: float = BB_GAIN) -> "OrderedDict[str, np.ndarray]":
    """Deterministic synthetic weights for a name -> shape inventory (float32 numpy).

    ``bb_gain`` scales the BackboneUpdate weights: the default keeps the predicted frame within a few
    sigma of the noisy one at every t; larger values (stress fixtures) drive the IGSO(3) score series of
    the reference into its float32-noise regime ("conditioning of the rotation score").
    """
    return OrderedDict(
        (k, synth_tensor(k, tuple(shp), seed, i, bb_gain)) for i, (k, shp) in enumerate(shapes.items())
    )


def n_params(shapes) -> int:
    return int(sum(int(np.prod(s)) for s in shapes.values()))
