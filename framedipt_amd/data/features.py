"""Inpainting feature builder: processed structure features -> what the conditional samplers read.

``process_csv_row`` follows ``framedipt/data/utils.py:745-890``: per chain keep the modelled residues (:692-741), concatenate
(:420-444), run the four OpenFold transforms the reference runs (``openfold/data/data_transforms.py``: ``atom37_to_frames`` :755-889,
``make_atom14_masks`` :572-645, ``make_atom14_positions`` :653-752, ``atom37_to_torsion_angles`` :922-1087) and renumber the
residues per chain with a gap of 200 between chains.  Host NumPy code - CPU work once per structure in the reference too - with the
reference's dtype flow: geometry in float64, frames rounded to float32 (``Rotation`` / ``Rigid`` force float32,
``openfold/utils/rigid_utils.py:325-329,898-899``).

The residue-constant index tables (atom orders, rigid-group base atoms, chi atoms, ...) are data: ``feature_tables.npz`` next to
this file, written by ``tests/golden/make_goldens_r2.py feature_tables`` from ``openfold/np/residue_constants.py``.
"""
from __future__ import annotations

import os
import pickle
import string

import numpy as np

_T = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "feature_tables.npz")))
ATOM_TYPES = [str(a) for a in _T["atom_types"]]
ATOM_ORDER = {a: i for i, a in enumerate(ATOM_TYPES)}
RESTYPE_3 = [str(a) for a in _T["restype_3"]]                       # the 20 standard residues in restype order
RESTYPE_3_TO_INDEX = {a: i for i, a in enumerate(RESTYPE_3)}
RESIDUE_GAP = 200                                                   # framedipt/data/utils.py:40

ALPHANUMERIC = string.ascii_letters + string.digits                 # framedipt/data/utils.py:36-38
CHAIN_TO_INT = {c: i for i, c in enumerate(ALPHANUMERIC)}


def chain_str_to_int(chain_str: str) -> int:  # framedipt/data/utils.py:243-249
    if len(chain_str) == 1:
        return CHAIN_TO_INT[chain_str]
    return sum(CHAIN_TO_INT[c] + i * len(ALPHANUMERIC) for i, c in enumerate(chain_str))


def map_to_new_str_name(index: int) -> str:  # framedipt/data/utils.py:252-272
    if index < 26:
        return chr(ord("A") + index)
    return map_to_new_str_name(index // 26 - 1) + chr(ord("A") + index % 26)


def concat_np_features(np_dicts, add_batch_dim: bool) -> dict:  # framedipt/data/utils.py:420-444
    keys: dict = {}
    for d in np_dicts:
        for k, v in d.items():
            if v is not None:
                keys.setdefault(k, []).append(v[None] if add_batch_dim else v)
    return {k: np.concatenate(v, axis=0) for k, v in keys.items()}


def parse_chain_feats(chain_feats: dict, scale_factor: float = 1.0) -> dict:  # framedipt/data/utils.py:513-524
    ca = ATOM_ORDER["CA"]
    chain_feats["bb_mask"] = chain_feats["atom_mask"][:, ca]
    center = np.sum(chain_feats["atom_positions"][:, ca], axis=0) / (np.sum(chain_feats["bb_mask"]) + 1e-5)
    pos = (chain_feats["atom_positions"] - center[None, None, :]) / scale_factor
    chain_feats["atom_positions"] = pos * chain_feats["atom_mask"][..., None]
    chain_feats["bb_positions"] = chain_feats["atom_positions"][:, ca]
    return chain_feats


def process_modeled_chain_features(features: dict, chain_id, min_idx: int, max_idx: int, rng=None, chain_max_len=None) -> dict:
    """framedipt/data/utils.py:692-741."""
    if chain_id is not None:
        m = features["chain_index"] == chain_id
        features = {k: v[m] for k, v in features.items()}
    idx = np.arange(min_idx, max_idx + 1, dtype=np.int64)
    n = max_idx + 1 - min_idx
    if chain_max_len is not None and n > chain_max_len:
        start = rng.integers(n - chain_max_len + 1) if rng is not None else np.random.randint(n - chain_max_len + 1)
        idx = idx[start:start + chain_max_len]
    return {k: v[idx] for k, v in features.items()}


# ------------------------------------------------------------------ geometry (openfold/utils/rigid_utils.py)
def from_3_points(p_neg_x_axis, origin, p_xy_plane, eps: float = 1e-8):
    """Rigid.from_3_points (rigid_utils.py:1233-1275): Gram-Schmidt in the input precision -> (rot float32 [...,3,3], trans float32)."""
    e0 = origin - p_neg_x_axis
    e1 = p_xy_plane - origin
    e0 = e0 / np.sqrt((e0 * e0).sum(-1, keepdims=True) + eps)
    e1 = e1 - e0 * (e0 * e1).sum(-1, keepdims=True)
    e1 = e1 / np.sqrt((e1 * e1).sum(-1, keepdims=True) + eps)
    e2 = np.cross(e0, e1)
    rot = np.stack([e0, e1, e2], axis=-1)  # columns
    return rot.astype(np.float32), origin.astype(np.float32)


def atom37_to_frames(aatype, pos, mask):
    """data_transforms.py:755-889 -> rigidgroups_gt_frames float32 [N,8,4,4] (+ gt_exists, group_exists [N,8])."""
    base = _T["rigidgroup_base_atom37_idx"][aatype]                          # [N,8,3]
    bp = pos[np.arange(len(aatype))[:, None, None], base]                   # [N,8,3,3] base-atom positions
    rot, trans = from_3_points(bp[..., 0, :], bp[..., 1, :], bp[..., 2, :], eps=1e-8)
    group_exists = _T["rigidgroup_mask"][aatype].astype(mask.dtype)
    gt_exists = mask[np.arange(len(aatype))[:, None, None], base].min(-1) * group_exists
    flip = np.ones((8, 3), dtype=np.float32)
    flip[0, 0] = flip[0, 2] = -1                                             # compose with diag(-1, 1, -1) on the backbone group
    rot = rot * flip[None, :, None, :]
    out = np.zeros((len(aatype), 8, 4, 4), dtype=np.float32)
    out[..., :3, :3], out[..., :3, 3], out[..., 3, 3] = rot, trans, 1
    return out, gt_exists, group_exists


def make_atom14_masks(aatype):
    """data_transforms.py:572-645 -> (atom14_atom_exists float32, residx_atom14_to_atom37 int64, residx_atom37_to_atom14 int64,
    atom37_atom_exists float32)."""
    return (_T["atom14_mask"][aatype].astype(np.float32), _T["atom14_to_atom37"][aatype].astype(np.int64),
            _T["atom37_to_atom14"][aatype].astype(np.int64), _T["atom37_mask"][aatype].astype(np.float32))


def make_atom14_positions(aatype, pos, mask, atom14_exists, idx14_to_37):
    """data_transforms.py:653-752 (the ground-truth part): atom14_gt_exists, atom14_gt_positions."""
    rows = np.arange(len(aatype))[:, None]
    gt_mask = atom14_exists * mask[rows, idx14_to_37]
    return gt_mask, gt_mask[..., None] * pos[rows, idx14_to_37]


def atom37_to_torsion_angles(aatype, pos, mask):
    """data_transforms.py:922-1087 -> torsion_angles_sin_cos [N,7,2], alt_torsion_angles_sin_cos, torsion_angles_mask [N,7]."""
    aatype = np.minimum(aatype, 20)
    n = len(aatype)
    prev_pos = np.concatenate([np.zeros((1, 37, 3), pos.dtype), pos[:-1]], 0)
    prev_mask = np.concatenate([np.zeros((1, 37), mask.dtype), mask[:-1]], 0)
    pre_omega = np.concatenate([prev_pos[:, 1:3], pos[:, :2]], -2)
    phi = np.concatenate([prev_pos[:, 2:3], pos[:, :3]], -2)
    psi = np.concatenate([pos[:, :3], pos[:, 4:5]], -2)
    pre_omega_mask = prev_mask[:, 1:3].prod(-1) * mask[:, :2].prod(-1)
    phi_mask = prev_mask[:, 2] * mask[:, :3].prod(-1)
    psi_mask = mask[:, :3].prod(-1) * mask[:, 4]
    chi_idx = _T["chi_atom_indices"][aatype]                                 # [N,4,4]
    rows = np.arange(n)[:, None, None]
    chis_pos = pos[rows, chi_idx]                                            # [N,4,4,3]
    chis_mask = _T["chi_angles_mask"][aatype].astype(mask.dtype) * mask[rows, chi_idx].prod(-1)
    tpos = np.concatenate([pre_omega[:, None], phi[:, None], psi[:, None], chis_pos], 1)  # [N,7,4,3]
    tmask = np.concatenate([pre_omega_mask[:, None], phi_mask[:, None], psi_mask[:, None], chis_mask], -1)
    rot, trans = from_3_points(tpos[..., 1, :], tpos[..., 2, :], tpos[..., 0, :], eps=1e-8)
    # Rigid.invert (rigid_utils.py:1039-1050) in float32, then .apply() promotes to the points' float64
    rot_inv = np.swapaxes(rot, -1, -2)
    trn_inv = -np.einsum("...ij,...j->...i", rot_inv, trans).astype(np.float32)
    rel = np.einsum("...ij,...j->...i", rot_inv.astype(np.float64), tpos[..., 3, :]) + trn_inv.astype(np.float64)
    sc = np.stack([rel[..., 2], rel[..., 1]], -1)
    sc = sc / np.sqrt((sc * sc).sum(-1, keepdims=True) + 1e-8)
    sc = sc * np.array([1.0, 1.0, -1.0, 1.0, 1.0, 1.0, 1.0])[None, :, None]
    mirror = np.concatenate([np.ones((n, 3)), 1.0 - 2.0 * _T["chi_pi_periodic"][aatype]], -1)
    return sc, sc * mirror[..., None], tmask


class _ArrayUnpickler(pickle.Unpickler):
    """Processed-structure pickles hold NumPy arrays, scalars and plain containers: nothing else may be constructed (unpickling
    runs the callables a file names)."""
    _OK = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
           ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode"),
           ("collections", "OrderedDict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"),
           ("numpy._core.numeric", "_frombuffer"), ("numpy.core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"processed-structure pickle names the global {module}.{name}: refused")


def read_pkl(path):
    with open(path, "rb") as f:
        return _ArrayUnpickler(f).load()


def process_csv_row(processed, process_monomer: bool = False, extract_single_chain: bool = False, rng=None, chain_max_len=None) -> dict:
    """framedipt/data/utils.py:745-890.  ``processed``: path of a processed-structure pickle or the dict itself.  Returns NumPy arrays
    with the reference's dtypes (``aatype`` int64, positions / masks float64, ``rigidgroups_0`` float32, ...)."""
    feats = dict(read_pkl(processed)) if isinstance(processed, (str, os.PathLike)) else dict(processed)
    ci = feats["chain_index"]
    first = np.unique(ci, return_index=True)[1]
    unique_chains = [ci[i] for i in sorted(first)]
    if process_monomer:
        modeled = feats.pop("modeled_idx")
        feats.pop("chains", None)
        feats = process_modeled_chain_features(feats, None, int(np.min(modeled)), int(np.max(modeled)), rng, None)
    else:
        los, his = feats.pop("min_modeled_idxs"), feats.pop("max_modeled_idxs")
        if extract_single_chain:
            k = rng.integers(len(los)) if rng is not None else np.random.randint(len(los))
            feats = process_modeled_chain_features(feats, unique_chains[k], los[k], his[k], rng, chain_max_len)
        else:
            feats = concat_np_features([process_modeled_chain_features(feats, c, lo, hi, rng, None)
                                        for c, lo, hi in zip(unique_chains, los, his)], False)
    aatype = np.asarray(feats["aatype"]).astype(np.int64)
    pos, mask = np.asarray(feats["atom_positions"], dtype=np.float64), np.asarray(feats["atom_mask"], dtype=np.float64)
    frames, _, _ = atom37_to_frames(aatype, pos, mask)
    a14_exists, idx14_to_37, _, _ = make_atom14_masks(aatype)
    _, a14_pos = make_atom14_positions(aatype, pos, mask, a14_exists, idx14_to_37)
    tors, _, _ = atom37_to_torsion_angles(aatype, pos, mask)
    chain_idx, res_idx = feats["chain_index"], feats["residue_index"]
    new_res_idx = np.zeros_like(res_idx)
    prev = 0
    for c in np.unique(chain_idx):  # residues renumbered per chain, RESIDUE_GAP between chains (utils.py:858-872)
        m = chain_idx == c
        new_res_idx[m] = prev + np.arange(m.sum())
        prev += m.sum() + RESIDUE_GAP
    return {"aatype": aatype, "seq_idx": new_res_idx, "chain_idx": chain_idx, "residx_atom14_to_atom37": idx14_to_37,
            "residue_index": feats["residue_index"], "res_mask": feats["bb_mask"], "atom37_pos": pos, "atom37_mask": mask,
            "atom14_pos": a14_pos, "rigidgroups_0": frames, "torsion_angles_sin_cos": tors}
