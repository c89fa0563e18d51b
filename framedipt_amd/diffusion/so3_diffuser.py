"""Host side of the SO(3) diffuser: schedules, IGSO(3) table rows, x_T sampling.

Mirrors ``framedipt/diffusion/so3_diffuser.py`` (SO3Diffuser :194-602) for the members the sampler path reads.
The reference builds 1000x1000 pdf/cdf/score-norm tables at start-up (:235-283, 46 s cold); here rows are
evaluated on demand with the same formulas (only row ``t_to_idx(1.0)`` is needed to sample x_T, plus one row per
``score_scaling(t)`` call).  The per-step rotation score itself is a HIP kernel (``fdipt_igso3_rot_score``).
"""
from __future__ import annotations

import numpy as np


def igso3_expansion(omega, eps, truncation_level: int = 1000):
    """so3_diffuser.py:18-77 (NumPy float64 branch used for the tables)."""
    l = np.arange(truncation_level)[None]
    omega = np.asarray(omega)[..., None]
    p = (2 * l + 1) * np.exp(-l * (l + 1) * eps**2 / 2) * np.sin(omega * (l + 1 / 2)) / np.sin(omega / 2)
    return p.sum(axis=-1)


def density(expansion, omega, marginal: bool = True):
    """so3_diffuser.py:80-96."""
    if marginal:
        return expansion * (1 - np.cos(omega)) / np.pi
    return expansion / 8 / np.pi**2


def score(exp, omega, eps, truncation_level: int = 1000):
    """d/d omega of log f(omega) with f the IGSO(3) series, regularised as the reference does (so3_diffuser.py:122-191,
    NumPy float64 branch used for the tables): every term's ratio sin((l + 1/2) w) / sin(w / 2) is differentiated by the
    quotient rule and the weighted sum is divided by ``f + 1e-4``."""
    order = np.arange(truncation_level)[None]
    w = np.asarray(omega)[..., None]
    half_odd = order + 1 / 2
    weight = (2 * order + 1) * np.exp(-order * (order + 1) * eps**2 / 2)
    num, den = np.sin(w * half_odd), np.sin(w / 2)
    d_num, d_den = half_odd * np.cos(w * half_odd), 1 / 2 * np.cos(w / 2)
    d_series = (weight * (den * d_num - num * d_den) / den**2).sum(axis=-1)
    return d_series / (exp + 1e-4)


class SO3Diffuser:
    def __init__(self, so3_conf) -> None:
        self.schedule = so3_conf.schedule
        if self.schedule != "logarithmic":
            raise ValueError(f"Unrecognize schedule {self.schedule}")
        self.min_sigma = so3_conf.min_sigma
        self.max_sigma = so3_conf.max_sigma
        self.num_sigma = so3_conf.num_sigma
        self.num_omega = so3_conf.num_omega
        # so3_diffuser.py:389-396: with use_cached_score the score norm is looked up in the [num_sigma, num_omega] table at
        # (t_to_idx(t), bucketize(omega, discrete_omega[:-1])) instead of evaluated; the rows of that table are built on demand
        # here (`score_table_rows`), the lookup runs in the rotation-score kernel
        self.use_cached_score = bool(so3_conf.use_cached_score)
        self.discrete_omega = np.linspace(0, np.pi, so3_conf.num_omega + 1)[1:]
        self._rows: dict = {}
        np.random.seed(so3_conf.seed)  # so3_diffuser.py:286

    @property
    def discrete_sigma(self) -> np.ndarray:
        return self.sigma(np.linspace(0.0, 1.0, self.num_sigma))

    def sigma_idx(self, sigma):
        return np.digitize(sigma, self.discrete_sigma) - 1

    def sigma(self, t):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f"Invalid t={t}")
        return np.log(t * np.exp(self.max_sigma) + (1 - t) * np.exp(self.min_sigma))

    def diffusion_coef(self, t):
        return np.sqrt(2 * (np.exp(self.max_sigma) - np.exp(self.min_sigma)) * self.sigma(t) / np.exp(self.sigma(t)))

    def t_to_idx(self, t):
        return self.sigma_idx(self.sigma(t))

    def score_sigma(self, t_f32) -> np.ndarray:
        """sigma snapped to the grid as ``torch_score`` does (so3_diffuser.py:398).  t arrives there as a float32 array
        (``move_to_np(t)``), and under the reference's pinned numpy 1.22.4 (value-based casting) ``sigma(t)`` — the mix of the
        two exponentials and the log — is evaluated in float32 before the digitize against the float64 grid.  (NumPy >= 2
        promotes to float64; the two differ by one grid bin, 0.3 % in sigma, for 3 of the 1000 steps of a num_t = 1000 schedule
        and for none at num_t <= 500: tests/test_host_cpu.py.)"""
        t = np.asarray(t_f32, dtype=np.float32).reshape(-1)
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f"Invalid t={t}")
        sig32 = np.log(t * np.float32(np.exp(self.max_sigma)) + (np.float32(1) - t) * np.float32(np.exp(self.min_sigma)))
        assert sig32.dtype == np.float32
        return self.discrete_sigma[self.sigma_idx(sig32)]

    def _row(self, idx: int):
        idx = int(idx)
        if idx not in self._rows:
            sig = self.discrete_sigma[idx]
            ev = igso3_expansion(self.discrete_omega, sig)
            pdf = density(ev, self.discrete_omega, marginal=True)
            cdf = pdf.cumsum() / self.num_omega * np.pi
            self._rows[idx] = (pdf, cdf, score(ev, self.discrete_omega, sig))
        return self._rows[idx]

    def score_table_rows(self, t_f32) -> np.ndarray:
        """Rows ``_score_norms[t_to_idx(t)]`` ([len(t), num_omega] float64) for ``use_cached_score`` — the index is taken as
        ``torch_score`` takes it (``t`` arrives as float32, see ``score_sigma``)."""
        t = np.asarray(t_f32, dtype=np.float32).reshape(-1)
        sig32 = np.log(t * np.float32(np.exp(self.max_sigma)) + (np.float32(1) - t) * np.float32(np.exp(self.min_sigma)))
        return np.stack([self._row(i)[2] for i in self.sigma_idx(sig32)])

    @property
    def omega_edges(self) -> np.ndarray:
        return np.ascontiguousarray(self.discrete_omega[:-1], dtype=np.float64)

    def score_scaling(self, t):
        """so3_diffuser.py:404-414; scalar or array-valued t."""
        def one(ti):
            pdf, _, sn = self._row(self.t_to_idx(ti))
            return np.sqrt(np.abs(np.sum(sn**2 * pdf, axis=-1) / np.sum(pdf, axis=-1))) / np.sqrt(3)
        if np.ndim(t) == 0:
            return one(t)
        return np.array([one(ti) for ti in np.asarray(t).reshape(-1)]).reshape(np.shape(t))

    def sample_igso3(self, t: float, n_samples: int = 1) -> np.ndarray:
        """Rotation angles by inverse-CDF lookup on the table row of t (so3_diffuser.py:325-340): one ``rand`` draw."""
        cdf_row = self._row(self.t_to_idx(t))[1]
        return np.interp(np.random.rand(n_samples), cdf_row, self.discrete_omega)

    def sample(self, t: float, n_samples: int = 1) -> np.ndarray:
        """Rotation vectors: a uniform axis (normalised ``randn`` draw, made BEFORE the angle draw as in
        so3_diffuser.py:342-354 — the order fixes the position in the global stream) times an IGSO(3) angle."""
        axis = np.random.randn(n_samples, 3)
        axis = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
        angle = self.sample_igso3(t, n_samples=n_samples)
        return axis * angle[:, None]

    def sample_ref(self, n_samples: int = 1) -> np.ndarray:
        return self.sample(1.0, n_samples=n_samples)
