"""On-disk outputs of a sampling run, in the formats the reference's evaluation reads.

  write_prot_to_pdb   framedipt/analysis/utils.py:76-157 (+ create_full_prot :18-73, framedipt/protein/protein.py:165-281 to_pdb)
  save_traj           experiments/inference.py:480-556
  save_diffusion_info experiments/utils.py:690-749 (+ get_diffused_region_per_chain :629-687)

Pure host code on NumPy arrays (what ``inference_fn`` returns): nothing here touches the GPU.  The PDB text is the fixed
80-column format (MODEL / ATOM / TER / ENDMDL records, chains renumbered from 0 and lettered A, B, ..., b-factor 100 = diffused
residue); byte-for-byte equality with files written by the reference is tested in tests/test_host_cpu.py.
"""
from __future__ import annotations

import os
import pathlib
import re

import numpy as np

# atom37 slot names and residue codes (the AlphaFold / OpenFold conventions the reference writes with)
ATOM37 = ("N", "CA", "C", "CB", "O", "CG", "CG1", "CG2", "OG", "OG1", "SG", "CD", "CD1", "CD2", "ND1", "ND2", "NE", "NE1", "NE2",
          "OD1", "OD2", "SD", "CE", "CE1", "CE2", "CE3", "NZ", "OE1", "OE2", "OH", "CH2", "CZ", "CZ2", "CZ3", "NH1", "NH2", "OXT")
RESTYPES = "ARNDCQEGHILKMFPSTWYV"
RES3 = ("ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR",
        "TRP", "TYR", "VAL", "UNK")
CHAIN_LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"
_ATOM_MASK_EPS = 1e-7


def _renumber(n, residue_index, chain_index):
    """create_full_prot: chains become 0, 1, ... in np.unique order of chain_index, residues restart at 0 in every chain; both
    stay 0 / arange(n) unless BOTH indices are given (the given residue_index values themselves are not used)."""
    res = np.arange(n)
    chain = np.zeros(n)
    if residue_index is not None and chain_index is not None:
        chain_index = np.asarray(chain_index)
        start = 0
        for i, c in enumerate(np.unique(chain_index)):
            length = int(np.sum(chain_index == c))
            chain[start:start + length] = i
            res[start:start + length] = np.arange(length)
            start += length
    return res.astype(int), chain.astype(int)


def _model_text(pos37, model, aatype, b_factors, residue_index, chain_index) -> str:
    """One MODEL ... ENDMDL block; atoms whose coordinates are (numerically) all zero are absent."""
    pos37 = np.asarray(pos37)
    if pos37.ndim != 3:
        raise ValueError(f"atom37 should be of dim 3, got {pos37.ndim}.")
    if pos37.shape[-1] != 3 or pos37.shape[-2] != 37:
        raise ValueError(f"atom37 should have shape [..., 37, 3], got {pos37.shape}.")
    n = pos37.shape[0]
    present = np.sum(np.abs(pos37), axis=-1) > _ATOM_MASK_EPS
    res, chain = _renumber(n, residue_index, chain_index)
    bf = np.zeros((n, 37)) if b_factors is None else np.asarray(b_factors)
    aa = np.zeros(n, dtype=np.int64) if aatype is None else np.asarray(aatype)
    if np.any(aa > 20) or np.any(aa < 0):
        raise ValueError("Invalid aatypes.")
    if chain.max() >= len(CHAIN_LETTERS):
        raise ValueError(f"The PDB format supports at most {len(CHAIN_LETTERS)} chains.")

    def ter(serial, i):
        return f"{'TER':<6}{serial:>5}      {RES3[aa[i]]:>3} {CHAIN_LETTERS[chain[i]]:>1}{res[i]:>4}"

    lines = [f"MODEL     {model}"]
    serial = 1
    for i in range(n):
        if i > 0 and chain[i] != chain[i - 1]:  # close the previous chain; the TER record takes a serial number
            lines.append(ter(serial, i - 1))
            serial += 1
        for j in np.nonzero(present[i])[0]:
            name = ATOM37[j]
            x, y, z = pos37[i, j]
            lines.append(f"{'ATOM':<6}{serial:>5} {name if len(name) == 4 else ' ' + name:<4}{'':>1}{RES3[aa[i]]:>3} "
                         f"{CHAIN_LETTERS[chain[i]]:>1}{res[i]:>4}{'':>1}   {x:>8.3f}{y:>8.3f}{z:>8.3f}{1.0:>6.2f}{bf[i, j]:>6.2f}"
                         f"          {name[0]:>2}{'':>2}")
            serial += 1
    lines.append(ter(serial, n - 1))
    lines.append("ENDMDL")
    return "\n".join(line.ljust(80) for line in lines) + "\n"


def write_prot_to_pdb(prot_pos, file_path, aatype=None, overwrite: bool = False, no_indexing: bool = False, b_factors=None,
                      residue_index=None, chain_index=None) -> pathlib.Path:
    """prot_pos [N,37,3] (one model) or [T,N,37,3] (models 1..T).  Unless ``no_indexing``, the file is ``<stem>_<k>.pdb`` with
    k = 1 + the largest index already present for that stem in the directory (1 when ``overwrite``)."""
    file_path = pathlib.Path(file_path)
    prot_pos = np.asarray(prot_pos)
    if overwrite:
        max_idx = 0
    else:
        file_dir = os.path.dirname(file_path)
        file_name = os.path.basename(file_path).strip(".pdb")  # (str.strip of the characters '.', 'p', 'd', 'b', as the reference)
        found = [re.findall(r"_(\d+).pdb", x) for x in os.listdir(file_dir) if file_name in x]
        max_idx = max([int(m[0]) for m in found if m] + [0])
    save_path = file_path if no_indexing else file_path.with_name(f"{file_path.stem}_{max_idx + 1}.pdb")
    if prot_pos.ndim == 4:
        text = "".join(_model_text(p, t + 1, aatype, b_factors, residue_index, chain_index) for t, p in enumerate(prot_pos))
    elif prot_pos.ndim == 3:
        text = _model_text(prot_pos, 1, aatype, b_factors, residue_index, chain_index)
    else:
        raise ValueError(f"Invalid positions shape {prot_pos.shape}")
    with open(save_path, "w", encoding="utf-8") as f:
        f.write(text)
        f.write("END")
    return save_path


def save_traj(bb_prot_traj, x0_traj, diffuse_mask, output_dir, sample_idx: int, aatype=None, residue_index=None, chain_index=None,
              save_backbone_trajectory: bool = True, save_pred_x0_trajectory: bool = True) -> dict:
    """Final sample (``sample_<i>_1.pdb`` = bb_prot_traj[0]), the reverse trajectory and the x_0 predictions; b-factor 100 marks
    the diffused residues.  The two flags are ``cfg.inference.save_backbone_trajectory / save_pred_x0_trajectory``."""
    output_dir = pathlib.Path(output_dir)
    b_factors = np.tile((np.asarray(diffuse_mask).astype(bool) * 100)[:, None], (1, 37))
    kw = dict(b_factors=b_factors, aatype=aatype, residue_index=residue_index, chain_index=chain_index)
    paths = {"sample_path": write_prot_to_pdb(np.asarray(bb_prot_traj)[0], output_dir / f"sample_{sample_idx}", **kw),
             "traj_path": output_dir / f"bb_traj_{sample_idx}", "x0_traj_path": output_dir / f"x0_traj_{sample_idx}"}
    if save_backbone_trajectory:
        paths["traj_path"] = write_prot_to_pdb(bb_prot_traj, paths["traj_path"], **kw)
    if save_pred_x0_trajectory:
        paths["x0_traj_path"] = write_prot_to_pdb(x0_traj, paths["x0_traj_path"], **kw)
    return paths


def get_diffused_region_per_chain(diffused_mask, chain_index):
    """(chains, starts, ends): one entry per contiguous diffused run, chains numbered 0.. in np.unique order, indices local to
    the chain."""
    diffused_mask = np.asarray(diffused_mask).astype(bool)
    chain_index = np.asarray(chain_index)
    number = {c: i for i, c in enumerate(np.unique(chain_index))}
    chains, starts, ends = [], [], []
    for c in np.unique(chain_index[diffused_mask]):
        idx = np.where(diffused_mask[chain_index == c])[0]
        cut = np.where(np.diff(idx) > 1)[0]  # last position of every run but the final one
        for s, e in zip(idx[np.concatenate([[0], cut + 1]).astype(int)], idx[np.concatenate([cut, [-1]]).astype(int)]):
            chains.append(number[c])
            starts.append(s)
            ends.append(e)
    return chains, starts, ends


def save_diffusion_info(output_dir, pdb_name: str, seq: str, diffused_mask, chain_index) -> None:
    """``diffusion_info.csv`` (tab separated: pdb_name, seq, chain, start, end); residues of type X are dropped before the runs
    are located, as the evaluation expects."""
    diffused_mask, chain_index = np.asarray(diffused_mask), np.asarray(chain_index)
    if len(diffused_mask) != len(chain_index):
        raise ValueError(f"Length of diffused_mask and chain_index should be the same, got {len(diffused_mask)} != {len(chain_index)}.")
    standard = np.array([c != "X" for c in seq])
    chains, starts, ends = get_diffused_region_per_chain(diffused_mask[standard], chain_index[standard])
    row = [pdb_name, seq, ",".join(chr(ord("A") + c) for c in chains), ",".join(str(s) for s in starts), ",".join(str(e) for e in ends)]
    with open(pathlib.Path(output_dir) / "diffusion_info.csv", "w", encoding="utf-8") as f:
        f.write("\t".join(["pdb_name", "seq", "chain", "start", "end"]) + "\n" + "\t".join(row) + "\n")


def save_confidence(diffusion_info_path, sample_dir, sample_i: int, log_p: float, log_probs, diffused_region_len) -> None:
    """What ``run_conditional_sampling`` does with an EigenFold score (experiments/inference.py:357-372): three ``log_p_sample_<i>*``
    columns appended to ``diffusion_info.csv`` (re-written through pandas, which adds its index column, as there) and
    ``<sample_dir>/log_probs.csv``."""
    import pandas as pd
    cols = {f"log_p_sample_{sample_i}": log_p,
            f"log_p_sample_{sample_i}_per_residue": log_p / diffused_region_len,
            f"log_p_sample_{sample_i}_per_residue_norm": log_p / (6 * diffused_region_len - 1)}
    pd.read_csv(diffusion_info_path, sep="\t").assign(**cols).to_csv(diffusion_info_path, sep="\t")
    pd.DataFrame({"log_probs": list(log_probs)}).to_csv(pathlib.Path(sample_dir) / "log_probs.csv")


def aatype_to_seq(aatype) -> str:
    """framedipt/data/utils.py:74-83."""
    return "".join((RESTYPES + "X")[int(a)] for a in aatype)
