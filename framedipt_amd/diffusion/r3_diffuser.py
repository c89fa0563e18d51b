"""Host side of the R^3 VP-SDE diffuser (``framedipt/diffusion/r3_diffuser.py``): schedule scalars and the
stationary sample for x_T.  The reverse step and the score run on the device."""
from __future__ import annotations

import numpy as np


class R3Diffuser:
    def __init__(self, r3_conf) -> None:
        self._r3_conf = r3_conf
        self.min_b = r3_conf.min_b
        self.max_b = r3_conf.max_b
        np.random.seed(r3_conf.seed)  # r3_diffuser.py:24

    def _scale(self, x):
        return x * self._r3_conf.coordinate_scaling

    def _unscale(self, x):
        return x / self._r3_conf.coordinate_scaling

    def b_t(self, t):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f"Invalid t={t}")
        return self.min_b + t * (self.max_b - self.min_b)

    def diffusion_coef(self, t):
        return np.sqrt(self.b_t(t))

    def drift_coef(self, x, t):
        return -1 / 2 * self.b_t(t) * x

    def marginal_b_t(self, t):
        return t * self.min_b + (1 / 2) * (t**2) * (self.max_b - self.min_b)

    def conditional_var(self, t):
        return 1 - np.exp(-self.marginal_b_t(t))

    def score_scaling(self, t):
        return 1 / np.sqrt(self.conditional_var(t))

    def sample_stationary_distribution(self, x_reference, diffuse_mask, chain_indices=None):
        """r3_diffuser.py:294-331."""
        x_reference_scaled = self._scale(x_reference)
        if diffuse_mask is not None:
            bool_mask = diffuse_mask.astype(bool)
        else:
            bool_mask = np.ones(x_reference.shape[:-1], dtype=np.bool_)
        loc = np.zeros_like(x_reference[bool_mask])
        scale = np.ones_like(x_reference[bool_mask])
        inpaint_region = np.random.normal(loc=loc, scale=scale)
        x_out_scaled = x_reference_scaled.copy()
        x_out_scaled[bool_mask] = inpaint_region
        return self._unscale(x_out_scaled)
