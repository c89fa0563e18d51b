"""Host-side (NumPy float32) positional / timestep embeddings.

These are inputs of the device forward, constant per trajectory (index embeddings) or known ahead of time
(one timestep embedding per reverse step), so they are evaluated once on the host exactly as the reference
evaluates them in float32 (``framedipt/model/score_network.py:17-64``) and uploaded.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32

# exp(-k ln(1e4)/15), k = 0..15, as torch evaluates it in float32 (score_network.py:49-53).  t*1e4*freq reaches
# 1e4 rad in float32, so one ulp in a frequency moves sin/cos by 1e-4: the constants are reference behaviour.
_TIMESTEP_FREQS = np.array([float.fromhex(h) for h in (
    "0x1.0000000000000p+0", "0x1.15142c0000000p-1", "0x1.2be4aa0000000p-2", "0x1.44960e0000000p-3",
    "0x1.5f4ff00000000p-4", "0x1.7c3d2e0000000p-5", "0x1.9b8c2e0000000p-6", "0x1.bd6f1a0000000p-7",
    "0x1.e21c500000000p-8", "0x1.04e74e0000000p-8", "0x1.1a62d60000000p-9", "0x1.31a3320000000p-10",
    "0x1.4acdb40000000p-11", "0x1.660aa40000000p-12", "0x1.8385b80000000p-13", "0x1.a36e2c0000000p-14")], dtype=F32)


def get_index_embedding(indices, embed_size: int = 32, max_len: int = 2056) -> np.ndarray:
    """score_network.py:17-38: [sin(i*pi/max_len^(2k/E)), cos(...)], k < E/2, float32."""
    k = np.arange(embed_size // 2)
    denom = np.power(float(max_len), 2 * k / embed_size).astype(F32)
    arg = ((np.asarray(indices).astype(F32)[..., None] * F32(math.pi)).astype(F32) / denom).astype(F32)
    return np.concatenate([np.sin(arg), np.cos(arg)], axis=-1).astype(F32)


def get_timestep_embedding(timesteps, embedding_dim: int = 32, max_positions: int = 10000) -> np.ndarray:
    """score_network.py:41-64 (1-D ``timesteps``), float32."""
    timesteps = np.asarray(timesteps, dtype=F32)
    if timesteps.ndim != 1:
        raise ValueError(f"timesteps should have 1D shape, got {timesteps.shape}.")
    half = embedding_dim // 2
    if embedding_dim == 32 and max_positions == 10000:
        freqs = _TIMESTEP_FREQS
    else:
        freqs = np.exp(np.arange(half, dtype=F32) * F32(-math.log(max_positions) / (half - 1))).astype(F32)
    emb = ((timesteps * F32(max_positions))[:, None] * freqs[None]).astype(F32)
    emb = np.concatenate([np.sin(emb), np.cos(emb)], axis=1).astype(F32)
    if embedding_dim % 2 == 1:
        emb = np.pad(emb, ((0, 0), (0, 1)))
    return emb
