"""Sampler datasets with the surface of ``experiments/sampler.py``.

``UnconditionalSampler`` (:22-135) is complete.  ``ConditionalSampler`` keeps the x_T part of
``ConditionalSampler.__getitem__`` (:312-354) and takes already-processed per-structure features (the mmCIF /
OpenFold feature builder is CPU data preparation outside the hot path).
"""
from __future__ import annotations

import numpy as np
import torch

from .rigid import Rigid


class UnconditionalSampler(torch.utils.data.Dataset):
    """De novo sampling: one item per (length, sample index)."""

    def __init__(self, cfg, diffuser, device) -> None:
        self._cfg = cfg
        self._diffuser = diffuser
        self.device = device
        self.all_sampling_lengths = self.get_sampling_lengths()

    def get_sampling_lengths(self) -> np.ndarray:
        lengths = range(self._cfg.min_length, self._cfg.max_length + 1, self._cfg.length_step)
        return np.repeat(lengths, self._cfg.samples_per_length)

    def sample(self, sample_length: int) -> dict:
        """sampler.py:69-111: keys and dtypes as the reference (float64 masks, int64 indices, float32 frames)."""
        sample_length = int(sample_length)
        res_mask = np.ones(sample_length)
        ref_sample = self._diffuser.sample_ref(n_samples=sample_length, as_tensor_7=True)
        init_feats = {
            "res_mask": res_mask,
            "seq_idx": torch.arange(1, sample_length + 1),
            "fixed_mask": np.zeros_like(res_mask),
            "torsion_angles_sin_cos": np.zeros((sample_length, 7, 2)),
            "sc_ca_t": np.zeros((sample_length, 3)),
            **ref_sample,
        }
        init_feats = {k: v if torch.is_tensor(v) else torch.tensor(v) for k, v in init_feats.items()}
        return {k: v[None].to(self.device) for k, v in init_feats.items()}

    def __len__(self) -> int:
        return len(self.all_sampling_lengths)

    def __getitem__(self, item: int):
        sample_length = self.all_sampling_lengths[item]
        sample_i = item % self._cfg.samples_per_length
        return sample_length, sample_i, self.sample(sample_length)


class ConditionalSampler(torch.utils.data.Dataset):
    """Inpainting: items are built from processed feature dicts.

    ``structures``: list of ``(pdb_name, feats)`` where ``feats`` holds NumPy arrays ``rigids_0`` [N,7],
    ``diffuse_mask`` [N] (1 = redesign), ``aatype`` [N], ``seq_idx`` [N], ``chain_idx`` [N],
    ``torsion_angles_sin_cos`` [N,7,2], optionally ``res_mask`` [N].
    """

    def __init__(self, structures, diffuser, device, samples_per_structure: int = 1) -> None:
        self._structures = list(structures)
        self._diffuser = diffuser
        self.device = device
        self._n = samples_per_structure

    def __len__(self) -> int:
        return len(self._structures) * self._n

    def __getitem__(self, item: int):
        name, f = self._structures[item // self._n]
        sample_i = item % self._n
        n = f["rigids_0"].shape[0]
        dm = np.asarray(f["diffuse_mask"], dtype=np.float64)
        rigids_0 = torch.as_tensor(np.asarray(f["rigids_0"], dtype=np.float32), device=self.device)
        # sampler.py:330-336: x_T keeps the motif frames and replaces the diffused ones
        ref = self._diffuser.sample_ref(n_samples=n, impute=Rigid.from_tensor_7(rigids_0), diffuse_mask=dm,
                                        chain_index=f.get("chain_idx"), as_tensor_7=True)
        feats = {
            "res_mask": np.asarray(f.get("res_mask", np.ones(n)), dtype=np.float64),
            "seq_idx": np.asarray(f["seq_idx"], dtype=np.int64),
            "fixed_mask": 1 - dm,
            "torsion_angles_sin_cos": np.asarray(f["torsion_angles_sin_cos"], dtype=np.float64),
            "sc_ca_t": np.zeros((n, 3)),
            "aatype": np.asarray(f["aatype"], dtype=np.int64),
            "chain_idx": np.asarray(f.get("chain_idx", np.zeros(n)), dtype=np.int64),
            "rigids_0": np.asarray(f["rigids_0"], dtype=np.float32),
            **ref,
        }
        feats = {k: v if torch.is_tensor(v) else torch.tensor(v) for k, v in feats.items()}
        return name, sample_i, {k: v[None].to(self.device) for k, v in feats.items()}
