"""ORACLE (test infrastructure, not product code): the reverse-diffusion loop in NumPy.

CPU restatement of ``experiments/utils.py:inference_fn`` (:511-626),
``one_step_inference`` (:292-412), ``_set_t_feats`` (:166-190),
``self_conditioning`` (:193-254) and ``UnconditionalSampler.sample``
(``experiments/sampler.py:69-111``) for SURVEY.md section 8 rows a1-a3.
"""
from __future__ import annotations

import copy

import numpy as np

from . import frames as fr
from .score_network import compute_backbone

F32 = np.float32


def unconditional_feats(diffuser, n: int) -> dict:
    """experiments/sampler.py:69-111: de novo initial feature dict (batch dim 1)."""
    rot, trans = diffuser.sample_ref(n_samples=n)
    quat = fr.rot_to_quat(rot.astype(F32))
    rig = np.concatenate([quat, trans.astype(F32)], axis=-1).astype(F32)
    return {
        "res_mask": np.ones((1, n)),
        "seq_idx": np.arange(1, n + 1)[None],
        "fixed_mask": np.zeros((1, n)),
        "torsion_angles_sin_cos": np.zeros((1, n, 7, 2)),
        "sc_ca_t": np.zeros((1, n, 3)),
        "rigids_t": rig[None],
    }


def set_t_feats(feats, t, t_placeholder, diffuser):
    feats["t"] = (F32(t) * t_placeholder).astype(F32)
    rs, ts = diffuser.score_scaling(t)
    feats["rot_score_scaling"] = rs * t_placeholder
    feats["trans_score_scaling"] = ts * t_placeholder
    return feats


def one_step(model, diffuser, feats, t, min_t, dt, t_placeholder, center=True, noise_scale=1.0,
             embed_self_conditioning=True, aatype=None, noise=None, orthogonalize=False):
    """experiments/utils.py:292-412 with aux_traj=True. ``noise`` = (z_rot, z_trans) or None."""
    feats = set_t_feats(feats, t, t_placeholder, diffuser)
    fixed_mask = feats["fixed_mask"] * feats["res_mask"]
    diffuse_mask = (1 - feats["fixed_mask"]) * feats["res_mask"]
    out = model(feats)
    rigid_pred = out["rigids"]
    if t > min_t:
        if embed_self_conditioning:
            feats["sc_ca_t"] = rigid_pred[..., 4:]
        rig = feats["rigids_t"].astype(F32)
        z_rot, z_trans = noise if noise is not None else (None, None)
        rot1, tr1 = diffuser.reverse(
            rig[..., :4], rig[..., 4:], np.asarray(out["rot_score"]), np.asarray(out["trans_score"]),
            t, dt, diffuse_mask=diffuse_mask, center=center, noise_scale=noise_scale,
            z_rot=z_rot, z_trans=z_trans, orthogonalize=orthogonalize)
        q1 = fr.rot_to_quat(rot1)
    else:
        q1, tr1 = rigid_pred[..., :4].astype(F32), rigid_pred[..., 4:].astype(F32)
        rot1 = fr.quat_to_rot(q1).astype(F32)
    feats["rigids_t"] = np.concatenate([q1, tr1], axis=-1).astype(F32)
    aux_rigids = feats["rigids_t"].copy()
    gt_trans_0 = feats["rigids_t"][..., 4:]
    trans_pred_0 = diffuse_mask[..., None] * rigid_pred[..., 4:] + fixed_mask[..., None] * gt_trans_0
    psi = out["psi"]
    tables = model.tables
    bb_0 = compute_backbone(rigid_pred[..., :4], rigid_pred[..., 4:], psi, aatype, tables)[0]
    bb = compute_backbone(None, tr1, psi, aatype, tables, rot=rot1)[0]
    return feats, psi, bb, aux_rigids, bb_0, trans_pred_0


def inference_fn(model, diffuser, data_init, num_t, min_t, center=True, self_condition=True, noise_scale=1.0,
                 embed_self_conditioning=True, noise_tape=None, orthogonalize=False):
    """experiments/utils.py:511-626 (aux_traj=True).  ``noise_tape``: list of (z_rot, z_trans) per step."""
    feats = copy.deepcopy(data_init)
    aatype = model.preprocess_aatype(feats.get("aatype"), feats["fixed_mask"])
    B = feats["rigids_t"].shape[0]
    tp = np.ones((B,), dtype=F32)
    steps = np.linspace(min_t, 1.0, num_t)[::-1]
    dt = 1 / num_t
    all_rigids = [feats["rigids_t"].copy()]
    bbs, tr0, bb0 = [], [], []
    if embed_self_conditioning and self_condition:
        feats = set_t_feats(feats, steps[0], tp, diffuser)
        feats["sc_ca_t"] = model(feats)["rigids"][..., 4:]
    for i, t in enumerate(steps):
        nz = noise_tape[i] if (noise_tape is not None and i < len(noise_tape)) else None
        feats, psi, bb, aux, b0, t0 = one_step(model, diffuser, feats, t, min_t, dt, tp, center, noise_scale,
                                               embed_self_conditioning, aatype, nz, orthogonalize)
        bbs.append(bb)
        all_rigids.append(aux)
        bb0.append(b0)
        tr0.append(t0)
    return {
        "prot_traj": np.flip(np.stack(bbs), (0,)),
        "rigid_traj": np.flip(np.stack(all_rigids), (0,)),
        "trans_traj": np.flip(np.stack(tr0), (0,)),
        "psi_pred": psi[None],
        "rigid_0_traj": np.flip(np.stack(bb0), (0,)),
    }
