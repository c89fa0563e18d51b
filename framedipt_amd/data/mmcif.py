"""mmCIF -> processed structure features (the pickles ``process_csv_row`` reads), without Biopython.

Stand-in for the inference-time part of ``framedipt/data/process_pdb_dataset.py`` (``extract_features_from_mmcif`` :83-217,
``process_mmcif`` :464-564), which walks a Bio.PDB structure: chains by author chain id, residues in file order keyed by
(hetero flag, author sequence number, insertion code), the highest-occupancy alternate location of an atom, the first model;
``parsers.process_chain`` (:16-69) then keeps the 37 AlphaFold atom types and maps non-standard residues to X.  This module
reads the ``_atom_site`` loop directly and applies the same rules.  PARITY UNPINNED against Bio.PDB (not installed in this image):
what is pinned (tests/test_host_cpu.py) is everything downstream of the feature dict this module returns.
"""
from __future__ import annotations

import pathlib
import pickle

import numpy as np

from . import features as F


def _tokens(line: str):
    """mmCIF whitespace-separated values; '...' / "..." quote a value (a quote ends only before whitespace)."""
    out, i, n = [], 0, len(line)
    while i < n:
        c = line[i]
        if c.isspace():
            i += 1
        elif c in "'\"":
            j = i + 1
            while j < n and not (line[j] == c and (j + 1 == n or line[j + 1].isspace())):
                j += 1
            out.append(line[i + 1:j])
            i = j + 1
        else:
            j = i
            while j < n and not line[j].isspace():
                j += 1
            out.append(line[i:j])
            i = j
    return out


def read_atom_site(path):
    """-> (list of column dicts of the ``_atom_site`` loop (strings), {tag: value} of the single-value items that were seen)."""
    rows, items, cols, in_loop, in_atom = [], {}, [], False, False
    with open(path, encoding="utf-8") as f:
        for line in f:
            s = line.strip()
            if not s or s.startswith("#"):
                in_loop = in_atom = False
                continue
            if s == "loop_":
                in_loop, in_atom, cols = True, False, []
                continue
            if s.startswith("_"):
                if in_loop:
                    if s.startswith("_atom_site."):
                        in_atom = True
                        cols.append(s.split(".", 1)[1].split()[0])
                    else:
                        in_atom = False
                else:
                    t = _tokens(s)
                    if len(t) == 2:
                        items[t[0]] = t[1]
                continue
            if in_loop and in_atom:
                t = _tokens(s)
                if len(t) == len(cols):
                    rows.append(dict(zip(cols, t)))
    return rows, items


def chain_features(rows):
    """Per author chain (file order, first model): the arrays ``parsers.process_chain`` builds, chain ids still strings."""
    first_model = rows[0].get("pdbx_PDB_model_num", "1") if rows else "1"
    chains: dict = {}
    for r in rows:
        if r.get("pdbx_PDB_model_num", first_model) != first_model:
            continue
        cid = r.get("auth_asym_id", r.get("label_asym_id"))
        het = r["group_PDB"] == "HETATM"
        resname = r["label_comp_id"]
        hetflag = " " if not het else ("W" if resname in ("HOH", "WAT") else "H_" + resname)
        icode = r.get("pdbx_PDB_ins_code", "?")
        key = (hetflag, int(r.get("auth_seq_id", r.get("label_seq_id"))), " " if icode in ("?", ".") else icode)
        ch = chains.setdefault(cid.upper(), {})
        res = ch.setdefault(key, {"resname": resname, "atoms": {}})
        name = r["label_atom_id"]
        occ = float(r.get("occupancy", "1.0"))
        prev = res["atoms"].get(name)
        if prev is None or occ > prev[0]:  # DisorderedAtom keeps the alternate location with the highest occupancy (first on ties)
            res["atoms"][name] = (occ, (float(r["Cartn_x"]), float(r["Cartn_y"]), float(r["Cartn_z"])), float(r.get("B_iso_or_equiv", "0")))
    out = {}
    for cid, residues in chains.items():
        n = len(residues)
        pos, mask, bfac = np.zeros((n, 37, 3)), np.zeros((n, 37)), np.zeros((n, 37))
        aatype, resid = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        for i, (key, res) in enumerate(residues.items()):
            aatype[i] = F.RESTYPE_3_TO_INDEX.get(res["resname"], 20)
            resid[i] = key[1]
            for name, (_, xyz, b) in res["atoms"].items():
                a = F.ATOM_ORDER.get(name)
                if a is not None:
                    # Bio.PDB stores coordinates and B factors as float32
                    pos[i, a], mask[i, a], bfac[i, a] = np.asarray(xyz, dtype=np.float32), 1.0, np.float32(b)
        out[cid] = {"atom_positions": pos, "atom_mask": mask, "aatype": aatype, "residue_index": resid, "b_factors": bfac}
    return out


def get_modeled_chain_len(aatype, chain_max_len=None, chain_min_len=None):
    """process_pdb_dataset.py:222-255: unknown residues at the two ends of a chain are not modelled."""
    idx = np.where(aatype != 20)[0]
    if len(idx) == 0:
        raise ValueError("No modeled residues.")
    lo, hi = int(idx.min()), int(idx.max())
    n = hi - lo + 1
    if chain_max_len is not None and n > chain_max_len:
        raise ValueError(f"Too long {n}.")
    if chain_min_len is not None and n < chain_min_len:
        raise ValueError(f"Too short {n}.")
    return len(aatype), n, lo, hi


def extract_features_from_mmcif(mmcif_path, chains=None, chain_max_len=None, chain_min_len=None, max_num_chains=None):
    """process_pdb_dataset.py:83-217 -> (number of chains in the file, chain lengths, modelled chain lengths, complex features with
    ``chain_index`` / ``bb_mask`` / ``bb_positions`` / ``min_modeled_idxs`` / ``max_modeled_idxs``)."""
    rows, _ = read_atom_site(mmcif_path)
    per_chain = chain_features(rows)
    num_chains = len(per_chain)
    chains = list(per_chain) if chains is None else [c.upper() for c in chains]
    for c in chains:
        if c not in per_chain:
            raise ValueError(f"The input list of chains should be in mmcif file, got {c} not in {list(per_chain)}.")
    feats, lens, mlens, los, his, k = [], [], [], [], [], 0
    for c in chains:
        d = dict(per_chain[c])
        try:
            n, m, lo, hi = get_modeled_chain_len(d["aatype"], chain_max_len, chain_min_len)
        except ValueError:
            continue  # the chain is filtered from the structure
        d["chain_index"] = np.full(len(d["aatype"]), F.chain_str_to_int(F.map_to_new_str_name(k)))
        feats.append(d)
        lens.append(n); mlens.append(m); los.append(lo); his.append(hi)
        k += 1
        if max_num_chains is not None and k > max_num_chains:
            raise ValueError(f"Too many modeled chains (more than {max_num_chains}), overall {num_chains} chains.")
    if not feats:
        raise ValueError("No chain is modeled.")
    cf = F.parse_chain_feats(F.concat_np_features(feats, False))
    cf["min_modeled_idxs"], cf["max_modeled_idxs"] = np.array(los), np.array(his)
    return num_chains, lens, mlens, cf


def process_mmcif(mmcif_path, write_dir, chains=None, max_len=None, min_len=None, chain_max_len=None, chain_min_len=None,
                  max_num_chains=None) -> dict:
    """process_pdb_dataset.py:464-564 without the resolution / secondary-structure columns: writes
    ``<write_dir>/<name[1:3]>/<name>.pkl`` and returns the metadata row the samplers read."""
    mmcif_path, write_dir = pathlib.Path(mmcif_path), pathlib.Path(write_dir)
    name = mmcif_path.stem
    sub = write_dir / name[1:3].lower()
    sub.mkdir(parents=True, exist_ok=True)
    out = (sub / f"{name}.pkl").resolve()
    num_chains, lens, mlens, cf = extract_features_from_mmcif(mmcif_path, chains, chain_max_len, chain_min_len, max_num_chains)
    n = int(np.sum(mlens))
    if max_len is not None and n > max_len:
        raise ValueError(f"Too long {n}.")
    if min_len is not None and n < min_len:
        raise ValueError(f"Too short {n}.")
    with open(out, "wb") as f:
        pickle.dump(cf, f, protocol=pickle.HIGHEST_PROTOCOL)
    return {"pdb_name": name, "raw_path": str(mmcif_path), "processed_path": str(out), "num_chains": num_chains,
            "seq_len": int(np.sum(lens)), "modeled_seq_len": n}


def process_serially(all_mmcif_paths, write_dir, all_chains_to_process=None, **filters) -> list:
    """process_pdb_dataset.py:567-640: structures that hit a filtering rule are skipped."""
    rows = []
    for i, p in enumerate(all_mmcif_paths):
        chains = None if all_chains_to_process is None else all_chains_to_process[i]
        try:
            rows.append(process_mmcif(p, write_dir, chains=chains, **filters))
        except ValueError:
            continue
    return rows
