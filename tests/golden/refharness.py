"""Import harness for the *reference* FrameDiPT sources (this container only).

Used only by ``make_goldens.py`` to generate the committed fixtures under
``tests/golden/``.  It never runs on the GPU box (``/root/reference`` does not
exist there) and nothing in the product or in the tests imports it.

Recipe = SURVEY.md section 9.F: stub the packages the reference imports but this
image lacks, put ``/root/reference`` on ``sys.path``, build the config from
``config/base.yaml`` by hand.
"""
from __future__ import annotations

import sys
import types
from unittest import mock

import numpy as np
import yaml

REF = "/root/reference"


class AttrDict(dict):
    """Minimal stand-in for omegaconf.DictConfig (attribute access on a dict)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    if isinstance(d, list):
        return [to_attr(v) for v in d]
    return d


def install_stubs() -> None:
    if REF not in sys.path:
        sys.path.insert(0, REF)
    oc = types.ModuleType("omegaconf")
    oc.DictConfig = AttrDict
    oc.OmegaConf = mock.MagicMock()
    oc.ListConfig = list
    sys.modules.setdefault("omegaconf", oc)

    def map_structure(fn, *structs):
        s0 = structs[0]
        if isinstance(s0, dict):
            return {k: map_structure(fn, *[s[k] for s in structs]) for k in s0}
        if isinstance(s0, (list, tuple)):
            return type(s0)(map_structure(fn, *xs) for xs in zip(*structs))
        return fn(*structs)

    tree = types.ModuleType("tree")
    tree.map_structure = map_structure
    sys.modules.setdefault("tree", tree)
    for name in [
        "absl", "absl.logging", "Bio", "Bio.PDB", "Bio.PDB.Chain", "Bio.PDB.Model",
        "Bio.PDB.Structure", "Bio.PDB.PDBIO", "Bio.PDB.Polypeptide", "Bio.Data",
        "Bio.Data.SCOPData", "Bio.Data.PDBData", "Bio.PDB.MMCIFParser", "Bio.PDB.PDBParser",
        "Bio.PDB.MMCIF2Dict", "Bio.PDB.Residue", "Bio.PDB.Atom", "Bio.PDB.mmcifio",
        "Bio.SeqUtils", "Bio.Seq", "Bio.SeqRecord", "Bio.SeqIO", "Bio.Align",
        "hydra", "hydra.core", "hydra.core.hydra_config", "GPUtil", "esm",
        "biotite", "biotite.sequence", "biotite.sequence.io", "biotite.sequence.io.fasta",
        "mdtraj", "tmtools", "anarci", "ml_collections", "neptune",
    ]:
        sys.modules.setdefault(name, mock.MagicMock())


def load_cfg(inpainting: bool = False, input_aatype: bool = False, cache_dir: str = "/tmp/fdipt_ref_cache/"):
    with open(f"{REF}/config/base.yaml") as f:
        cfg = yaml.safe_load(f)
    cfg["model"]["ipa"]["c_s"] = cfg["model"]["node_embed_size"]
    cfg["model"]["ipa"]["c_z"] = cfg["model"]["edge_embed_size"]
    cfg["model"]["ipa"]["coordinate_scaling"] = cfg["diffuser"]["r3"]["coordinate_scaling"]
    cfg["model"]["embed"]["min_bin"] = float(cfg["model"]["embed"]["min_bin"])
    cfg["model"]["input_aatype"] = input_aatype
    cfg["model"]["inpainting"] = inpainting
    cfg["diffuser"]["so3"]["cache_dir"] = cache_dir
    cfg["diffuser"]["so3"]["seed"] = 123
    cfg["diffuser"]["r3"]["seed"] = 123
    return to_attr(cfg)


def small_model_cfg(cfg):
    """Small-config score network for unit goldens (SURVEY 8c 'Weights')."""
    m = cfg.model
    m.node_embed_size = 64
    m.edge_embed_size = 32
    m.ipa.c_s = 64
    m.ipa.c_z = 32
    m.ipa.c_hidden = 16
    m.ipa.c_skip = 16
    m.ipa.no_heads = 4
    m.ipa.no_qk_points = 4
    m.ipa.no_v_points = 6
    m.ipa.seq_tfmr_num_heads = 2
    m.ipa.seq_tfmr_num_layers = 1
    m.ipa.num_blocks = 2
    return cfg
