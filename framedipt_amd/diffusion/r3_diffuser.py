"""Host side of the translation (R^3, variance-preserving) diffuser.

Only the scalars of the linear noise schedule and the x_T draw live on the host; the reverse step and the score are device
kernels (``fdipt_se3_reverse_step``, ``fdipt_r3_trans_score``).  The public names are the ones the callers of
``framedipt/diffusion/r3_diffuser.py`` use (``b_t`` :48, ``diffusion_coef`` :62, ``drift_coef`` :75, ``marginal_b_t`` :87,
``conditional_var`` :387, ``score_scaling`` :333, ``sample_stationary_distribution`` :294).

Schedule: beta(t) = b0 + t (b1 - b0) on t in [0, 1]; its integral B(t) = b0 t + (b1 - b0) t^2 / 2; the forward kernel
from x_0 is N(exp(-B/2) x_0, 1 - exp(-B)) in coordinates multiplied by ``coordinate_scaling`` (0.1: Angstrom -> nm).
"""
from __future__ import annotations

import numpy as np


def _check_unit_interval(t) -> None:
    if np.any(t < 0) or np.any(t > 1):
        raise ValueError(f"Invalid t={t}")


class R3Diffuser:
    def __init__(self, r3_conf) -> None:
        self._r3_conf = r3_conf
        self.min_b, self.max_b = r3_conf.min_b, r3_conf.max_b
        # the reference restarts the global legacy stream when a diffuser is built (r3_diffuser.py:24); x_T and the noise
        # tape are drawn from that stream, so the restart is part of the reproducible behaviour
        np.random.seed(r3_conf.seed)

    # -- unit handling ---------------------------------------------------------------------------------------------
    def _scale(self, x):
        return x * self._r3_conf.coordinate_scaling

    def _unscale(self, x):
        return x / self._r3_conf.coordinate_scaling

    # -- schedule --------------------------------------------------------------------------------------------------
    def b_t(self, t):
        _check_unit_interval(t)
        return self.min_b + t * (self.max_b - self.min_b)

    def marginal_b_t(self, t):
        return t * self.min_b + (1 / 2) * (t**2) * (self.max_b - self.min_b)

    def diffusion_coef(self, t):
        """g(t) = sqrt(beta(t))."""
        return np.sqrt(self.b_t(t))

    def drift_coef(self, x, t):
        """f(x, t) = -beta(t) x / 2."""
        return -1 / 2 * self.b_t(t) * x

    def conditional_var(self, t):
        """Variance of x_t given x_0."""
        return 1 - np.exp(-self.marginal_b_t(t))

    def score_scaling(self, t):
        """1 / std of x_t given x_0: the factor the loss (and the `_set_t_feats` scalars) divide the score by."""
        return 1 / np.sqrt(self.conditional_var(t))

    # -- x_T -------------------------------------------------------------------------------------------------------
    def sample_stationary_distribution(self, x_reference, diffuse_mask, chain_indices=None):
        """x_T for the translations: rows selected by ``diffuse_mask`` (all rows when it is None) are replaced by unit
        normals in scaled coordinates, every other row keeps its reference position.  One ``np.random.normal`` call of
        shape [n_selected, 3] — the draw the reference makes at this point of the global stream.  ``chain_indices`` is
        accepted for signature compatibility and not used (the reference ignores it here as well)."""
        ref = np.asarray(x_reference)
        if diffuse_mask is None:
            rows = np.ones(ref.shape[:-1], dtype=bool)
        else:
            rows = np.asarray(diffuse_mask).astype(bool)
        n_rows = int(rows.sum())
        noise = np.random.normal(loc=np.zeros((n_rows, ref.shape[-1]), dtype=ref.dtype),
                                 scale=np.ones((n_rows, ref.shape[-1]), dtype=ref.dtype))
        out = self._scale(ref).copy()
        out[rows] = noise
        return self._unscale(out)
