"""Sampler datasets with the surface of ``experiments/sampler.py``.

``UnconditionalSampler`` (:22-135), ``ConditionalSampler`` (:138-354) and ``TCRSampler`` (:357-467) keep the reference's
constructor signatures ``(cfg or data_conf, diffuser, device)`` and item protocol.  The mmCIF download / parsing step behind
``ConditionalSampler._init_metadata`` is CPU data preparation outside the hot path: the samplers read
its products (``processed/metadata.csv`` + per-structure feature files) or in-memory feature dicts.
"""
from __future__ import annotations

import os

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


def create_single_redacted_region(res_mask, rng, redact_min_len, redact_max_len):
    """framedipt/data/utils.py:613-654: one random contiguous window of a chain (all ones without length limits)."""
    if redact_min_len is None or redact_max_len is None:
        return np.ones_like(res_mask)
    modeled = np.where(res_mask != 0)[0]
    min_idx, max_idx = modeled[0], modeled[-1]
    modeled_len = max_idx - min_idx + 1
    diff_mask = np.zeros_like(res_mask)
    redact_max_len = min(redact_max_len, modeled_len)
    redact_min_len = min(redact_min_len, redact_max_len)
    length = rng.integers(low=redact_min_len, high=redact_max_len, endpoint=True)
    start_idx = rng.integers(low=min_idx, high=max_idx + 1 - length, endpoint=True)
    diff_mask[start_idx:start_idx + length] = 1
    return diff_mask


def create_redacted_regions(chain_idx, res_mask, rng, redact_min_len, redact_max_len):
    """framedipt/data/utils.py:657-689: one redacted window per chain, chains in np.unique order."""
    return np.concatenate([create_single_redacted_region(res_mask[chain_idx == c], rng, redact_min_len, redact_max_len)
                           for c in np.unique(chain_idx)])


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


class ConditionalSampler(torch.utils.data.Dataset):
    """Inpainting dataset with the constructor and item protocol of ``experiments/sampler.py:138-354``.

    ``ConditionalSampler(data_conf, diffuser, device)`` reads ``<download_dir>/processed/metadata.csv`` (columns
    ``pdb_name, processed_path, modeled_seq_len``), one row per structure.  Downloading and parsing mmCIF files
    (``_init_metadata`` of the reference, Biopython) is data preparation outside the hot path: the metadata file has to exist.
    ``processed_path`` names the per-structure features: a ``.npz`` / ``.pkl`` holding what ``data_utils.process_csv_row``
    returns (``aatype, seq_idx, chain_idx, res_mask, rigidgroups_0`` or ``rigids_0``, ``torsion_angles_sin_cos`` and any
    pass-through key such as ``atom37_pos``; optionally a precomputed ``diffuse_mask``).
    ``ConditionalSampler.from_features([(name, feats), ...], diffuser, device, samples=...)`` builds the same dataset from
    in-memory feature dicts (synthetic complexes, tests, bench.py).
    """

    def __init__(self, data_conf, diffuser, device) -> None:
        self._data_conf = data_conf
        self._features = None
        self._init_metadata()
        self._diffuser = diffuser
        self.device = device
        self.diffused_masks: dict = {}
        self.rng = np.random.default_rng(self._data_conf.seed)

    @classmethod
    def from_features(cls, structures, diffuser, device, samples: int = 1, seed: int = 123, redact_min_len=None,
                      redact_max_len=None):
        self = cls.__new__(cls)
        from .config import to_conf
        self._data_conf = to_conf({"samples": samples, "seed": seed,
                                   "redaction": {"redact_min_len": redact_min_len, "redact_max_len": redact_max_len}})
        self._features = [f for _, f in structures]
        self.metadata = [{"pdb_name": name, "processed_path": None,
                          "modeled_seq_len": int(np.asarray(f.get("rigids_0", f.get("rigidgroups_0"))).shape[0])}
                         for name, f in structures]
        self.pdb_csv, self.pdb_files = None, [name for name, _ in structures]
        self.all_chains_to_process = self.get_chains_to_process()
        self._diffuser, self.device = diffuser, device
        self.diffused_masks = {}
        self.rng = np.random.default_rng(seed)
        return self

    @property
    def diffuser(self):
        return self._diffuser

    @property
    def data_conf(self):
        return self._data_conf

    def get_chains_to_process(self):
        return [None] * len(self.pdb_files)

    def _init_metadata(self) -> None:
        import pathlib

        import pandas as pd

        from . import _lib
        download_dir = pathlib.Path(self.data_conf.download_dir)
        metadata_path = download_dir / "processed" / "metadata.csv"
        data_path = getattr(self.data_conf, "data_path", None)
        self.pdb_csv = pd.read_csv(data_path) if data_path and os.path.exists(str(data_path)) else None
        # sampler.py:189-222: the mmCIF files under <download_dir>/cifs (downloading them is the caller's job: no network access
        # is attempted) are parsed into processed pickles + metadata.csv unless that file exists already
        cifs = sorted((download_dir / "cifs").glob("*.cif"))
        if self.pdb_csv is not None and "pdb_id" in self.pdb_csv:
            ids = set(self.pdb_csv["pdb_id"].astype(str))
            cifs = [c for c in cifs if c.stem[:4] in ids]
        self.pdb_files = cifs
        if not metadata_path.exists() or getattr(self.data_conf, "overwrite", False):
            if not cifs:
                raise _lib.FdiptError(f"neither {metadata_path} nor mmCIF files under {download_dir / 'cifs'}: fetch the structures "
                                      "first (data_utils.download_cifs of the reference; this package makes no network access)")
            from .data import mmcif
            g = lambda k: getattr(self.data_conf, k, None)  # noqa: E731
            rows = mmcif.process_serially(cifs, download_dir / "processed", self.get_chains_to_process(), max_len=g("max_len"),
                                          min_len=g("min_len"), chain_max_len=g("chain_max_len"), chain_min_len=g("chain_min_len"),
                                          max_num_chains=g("max_num_chains"))
            (download_dir / "processed").mkdir(parents=True, exist_ok=True)
            pd.DataFrame(rows).to_csv(metadata_path, index=False)
        md = pd.read_csv(metadata_path)
        if self.pdb_csv is not None and "pdb_id" in self.pdb_csv:
            keep = set(self.pdb_csv["pdb_id"].astype(str))
            md = md[[str(n)[:4] in keep for n in md["pdb_name"]]].reset_index(drop=True)
        self.metadata = md.to_dict("records")
        self.pdb_files = [pathlib.Path(str(r["pdb_name"])) for r in self.metadata]
        self.all_chains_to_process = self.get_chains_to_process()

    def _chain_feats(self, example_idx: int) -> dict:
        if self._features is not None:
            return dict(self._features[example_idx])
        path = str(self.metadata[example_idx]["processed_path"])
        if path.endswith(".npz"):
            return dict(np.load(path, allow_pickle=False))
        from .data import features
        feats = features.read_pkl(path)
        if "rigidgroups_0" not in feats and "rigids_0" not in feats:  # a processed-structure pickle (sampler.py:284-289)
            feats = features.process_csv_row(feats, process_monomer=False, extract_single_chain=False, rng=self.rng)
        return feats

    def create_diffusion_mask(self, chain_feats, example_idx: int) -> np.ndarray:
        """sampler.py:224-254: cached per example; a precomputed ``diffuse_mask`` in the features wins."""
        if self.diffused_masks.get(example_idx) is not None:
            return self.diffused_masks[example_idx]
        if "diffuse_mask" in chain_feats:
            mask = np.asarray(_np(chain_feats["diffuse_mask"]), dtype=np.float64)
        else:
            rng = np.random.default_rng(example_idx)  # fixed seed for evaluation, sampler.py:245
            red = self.data_conf.redaction
            mask = create_redacted_regions(_np(chain_feats["chain_idx"]), _np(chain_feats["res_mask"]), rng,
                                           redact_min_len=red.redact_min_len, redact_max_len=red.redact_max_len)
        self.diffused_masks[example_idx] = mask
        return mask

    def __len__(self) -> int:
        return len(self.metadata) * self.data_conf.samples

    def __getitem__(self, idx: int):
        """sampler.py:267-354: (pdb_name, sample index, feature dict with a leading batch dimension on ``device``)."""
        example_idx, sample_idx = idx // self.data_conf.samples, idx % self.data_conf.samples
        row = self.metadata[example_idx]
        chain_feats = self._chain_feats(example_idx)
        if "rigidgroups_0" in chain_feats:
            g = torch.as_tensor(np.asarray(_np(chain_feats["rigidgroups_0"]), dtype=np.float32), device=self.device)
            gt_bb_rigid = Rigid.from_tensor_4x4(g)[:, 0]
        else:
            gt_bb_rigid = Rigid.from_tensor_7(torch.as_tensor(np.asarray(_np(chain_feats["rigids_0"]), dtype=np.float32),
                                                              device=self.device))
        n = gt_bb_rigid.shape[0]
        chain_feats.setdefault("res_mask", np.ones(n))
        chain_feats.setdefault("chain_idx", np.zeros(n, dtype=np.int64))
        diffused_mask = self.create_diffusion_mask(chain_feats, example_idx)
        if np.sum(diffused_mask) < 1:
            raise ValueError("Must be diffused")
        chain_feats.pop("diffuse_mask", None)
        chain_feats["fixed_mask"] = 1 - diffused_mask
        chain_feats["rigids_0"] = gt_bb_rigid.to_tensor_7()
        chain_feats["sc_ca_t"] = torch.zeros_like(gt_bb_rigid.get_trans())
        # t fixed to the final timestep: reference distribution on the diffused residues, motif kept (sampler.py:326-339)
        diff_feats_t = self.diffuser.sample_ref(n_samples=n, chain_index=_np(chain_feats["chain_idx"]), impute=gt_bb_rigid,
                                                diffuse_mask=diffused_mask, as_tensor_7=True)
        chain_feats.update(diff_feats_t)
        chain_feats["t"] = 1.0
        # (torch.tensor of a Python float is float32, of a NumPy array the array's dtype: tree.map_structure in sampler.py:341-344)
        final = {k: v if torch.is_tensor(v) else torch.tensor(v if isinstance(v, (float, int)) else np.asarray(v))
                 for k, v in chain_feats.items()}
        final = pad_feats(final, int(row["modeled_seq_len"]))
        return row["pdb_name"], sample_idx, {k: v[None].to(self.device) for k, v in final.items()}


UNPADDED_FEATS = ["t", "rot_score_scaling", "trans_score_scaling", "t_seq", "t_struct"]  # framedipt/data/utils.py:28-35
RIGID_FEATS = ["rigids_0", "rigids_t"]


def pad_feats(feats: dict, max_len: int) -> dict:
    """framedipt/data/utils.py:311-378 (use_torch=True): zero-pad dim 0 to ``max_len``, frames with identity rigids."""
    out = {}
    for k, v in feats.items():
        if k in UNPADDED_FEATS or v.dim() == 0:
            out[k] = v
            continue
        pad = max_len - v.shape[0]
        if pad < 0:
            raise ValueError(f"Invalid pad amount {pad}")
        if k in RIGID_FEATS:
            ident = torch.zeros(pad, 7, dtype=v.dtype, device=v.device)
            ident[:, 0] = 1
            out[k] = torch.cat([v, ident], dim=0)
        else:
            out[k] = torch.nn.functional.pad(v, [0, 0] * (v.dim() - 1) + [0, pad])
    return out


class TCRSampler(ConditionalSampler):
    """sampler.py:357-467: CDR3 loops of the TCR alpha / beta chains are redesigned.  The reference derives the loop masks with
    ANARCI numbering (``framedipt/protein/tcr.py``), an external tool absent here: the per-structure features must carry the
    CDR mask as ``diffuse_mask`` (or ``cdr_mask``)."""

    def __init__(self, data_conf, diffuser, device) -> None:
        super().__init__(data_conf=data_conf, diffuser=diffuser, device=device)

    def get_chains_to_process(self):
        """TCR alpha, beta (+ peptide, MHC chains when present) per structure, from the dataset CSV (sampler.py:385-428)."""
        if self.pdb_csv is None:
            return [None] * len(self.pdb_files)
        out = []
        for pdb_file in self.pdb_files:
            pdb_id = pdb_file.stem if hasattr(pdb_file, "stem") else str(pdb_file)
            if getattr(self.data_conf, "first_assembly", False):
                pdb_id = pdb_id[:4]
            hit = self.pdb_csv[self.pdb_csv["pdb_id"] == pdb_id]
            ex = (hit if len(hit) else self.pdb_csv[self.pdb_csv["pdb_id"] == pdb_id[:4]]).iloc[0]
            chains = [ex["tcr_alpha_chain"], ex["tcr_beta_chain"]]
            for col in ("peptide_chain", "mhc_alpha_chain", "mhc_beta_chain"):
                if col in ex and ex[col] is not None and isinstance(ex[col], str):
                    chains.append(ex[col])
            out.append(chains)
        return out

    def create_diffusion_mask(self, chain_feats, example_idx: int) -> np.ndarray:
        if self.diffused_masks.get(example_idx) is not None:
            return self.diffused_masks[example_idx]
        cdr = getattr(self.data_conf, "cdr_loops", None)
        if cdr is None or len(cdr) == 0:
            raise ValueError("CDR loops should be given in the config.")
        key = "diffuse_mask" if "diffuse_mask" in chain_feats else "cdr_mask" if "cdr_mask" in chain_feats else None
        if key is None:
            from . import _lib
            raise _lib.FdiptError("TCRSampler: the processed features carry no CDR mask (diffuse_mask / cdr_mask); the reference "
                                  "computes it with ANARCI (framedipt/protein/tcr.py), which is data preparation outside the hot path")
        mask = np.asarray(_np(chain_feats[key]), dtype=np.float64)
        self.diffused_masks[example_idx] = mask
        return mask
