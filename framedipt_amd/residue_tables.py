"""Residue-constant tables consumed by ``fdipt_backbone_atoms``.

The four AlphaFold tables the backbone builder reads (``framedipt/protein/all_atom.py:10-16`` ->
``residue_constants.restype_rigid_group_default_frame`` [21,8,4,4], ``restype_atom14_rigid_group_positions``
[21,14,3], ``restype_atom14_mask`` [21,14], ``restype_atom14_to_rigid_group`` [21,14]) are shipped as the data file
``data/residue_tables.npz`` and packed here into the ``BackboneTables`` struct of ``csrc/frames.hip``.
"""
from __future__ import annotations

import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "residue_tables.npz")


def load() -> dict:
    return dict(np.load(_PATH))


def packed_bytes() -> np.ndarray:
    t = load()
    parts = [
        t["default_frames"].astype(np.float32).reshape(-1),
        t["ideal_pos"].astype(np.float32).reshape(-1),
        t["atom_mask"].astype(np.float32).reshape(-1),
    ]
    f = np.concatenate(parts).view(np.uint8)
    g = t["group_idx"].astype(np.int32).reshape(-1).view(np.uint8)
    return np.concatenate([f, g])
