"""Configuration values that define the hot path.

Restates the parameter source of the reference (``config/base.yaml:33-79`` and
``config/inference.yaml:166,183-189``) as plain attribute dictionaries, so that
``SE3Diffuser(conf.diffuser)`` / ``ScoreNetwork(conf.model, ...)`` are built
with the same field names a hydra/omegaconf config would carry.
"""
from __future__ import annotations

import copy


class Conf(dict):
    """Attribute-style dict; duck-types the ``omegaconf.DictConfig`` reads used on the path."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Conf({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_conf(d):
    if isinstance(d, dict):
        return Conf({k: to_conf(v) for k, v in d.items()})
    return d


def base_config(inpainting: bool = False, input_aatype: bool = False) -> Conf:
    """Reference defaults (config/base.yaml:33-79, config/inference.yaml:160-189)."""
    return to_conf(
        {
            "diffuser": {
                "diffuse_trans": True,
                "diffuse_rot": True,
                "r3": {"min_b": 0.1, "max_b": 20.0, "coordinate_scaling": 0.1, "seed": 123},
                "so3": {
                    "num_omega": 1000,
                    "num_sigma": 1000,
                    "min_sigma": 0.1,
                    "max_sigma": 1.5,
                    "schedule": "logarithmic",
                    "cache_dir": ".cache/",
                    "use_cached_score": False,
                    "seed": 123,
                },
            },
            "model": {
                "input_aatype": input_aatype,
                "inpainting": inpainting,
                "node_embed_size": 256,
                "edge_embed_size": 128,
                "dropout": 0.0,
                "embed": {
                    "index_embed_size": 32,
                    "aatype_embed_size": 64,
                    "embed_self_conditioning": True,
                    "num_bins": 22,
                    "min_bin": 1e-5,
                    "max_bin": 20.0,
                },
                "ipa": {
                    "c_s": 256,
                    "c_z": 128,
                    "c_hidden": 256,
                    "c_skip": 64,
                    "no_heads": 8,
                    "no_qk_points": 8,
                    "no_v_points": 12,
                    "seq_tfmr_num_heads": 4,
                    "seq_tfmr_num_layers": 2,
                    "num_blocks": 4,
                    "coordinate_scaling": 0.1,
                },
            },
            "inference": {
                "seed": 123,
                "diffusion": {"num_t": 100, "noise_scale": 0.1, "min_t": 0.01},
            },
        }
    )


def small_config(inpainting: bool = False, input_aatype: bool = False) -> Conf:
    """Small score network used by unit goldens (same topology, reduced widths)."""
    c = base_config(inpainting, input_aatype)
    m = c.model
    m.node_embed_size = 64
    m.edge_embed_size = 32
    m.ipa.update(
        c_s=64, c_z=32, c_hidden=16, c_skip=16, no_heads=4, no_qk_points=4, no_v_points=6,
        seq_tfmr_num_heads=2, seq_tfmr_num_layers=1, num_blocks=2,
    )
    return c
