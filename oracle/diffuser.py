"""ORACLE (test infrastructure, not product code): SE(3) diffuser in NumPy.

CPU restatement of ``framedipt/diffusion/{so3,r3,se3}_diffuser.py`` for the
functions on the sampler path.
Randomness is the global legacy ``np.random`` stream, consumed in the
reference's order (SURVEY.md section 0 finding 10).  IGSO(3) table rows are computed on
demand (the reference builds all 1000 rows at start-up, ``so3_diffuser.py:235-283``).
"""
from __future__ import annotations

import numpy as np

from . import frames as fr

F32 = np.float32


def igso3_expansion_np(omega, eps, L=1000):
    """so3_diffuser.py:18-77, numpy float64 branch (table build)."""
    l = np.arange(L)[None]
    omega = np.asarray(omega)[..., None]
    p = (2 * l + 1) * np.exp(-l * (l + 1) * eps**2 / 2) * np.sin(omega * (l + 1 / 2)) / np.sin(omega / 2)
    return p.sum(axis=-1)


def score_np(exp, omega, eps, L=1000):
    """so3_diffuser.py:122-191, numpy float64 branch (table build)."""
    l = np.arange(L)[None]
    omega = np.asarray(omega)[..., None]
    hi = np.sin(omega * (l + 1 / 2))
    dhi = (l + 1 / 2) * np.cos(omega * (l + 1 / 2))
    lo = np.sin(omega / 2)
    dlo = 1 / 2 * np.cos(omega / 2)
    ds = (2 * l + 1) * np.exp(-l * (l + 1) * eps**2 / 2) * (lo * dhi - hi * dlo) / lo**2
    return ds.sum(axis=-1) / (exp + 1e-4)


def torch_score_mixed(vec_f32: np.ndarray, sigma: float, eps: float = 1e-6, L: int = 1000) -> np.ndarray:
    """so3_diffuser.py:373-402 with the dtype flow torch produces on the path.

    ``vec`` is float32 (from float32 quaternions); sin/cos of ``omega*(l+1/2)``
    are evaluated in float32, the Gaussian weights and the sums in float64
    (type promotion of int64*float64*float32 tensors); output float64.
    """
    vec = np.asarray(vec_f32, dtype=F32)
    omega = (np.linalg.norm(vec, axis=-1) + F32(eps)).astype(F32)  # [*]
    l = np.arange(L)
    lh = (l + 0.5).astype(F32)
    w = (2 * l + 1) * np.exp(-l * (l + 1) * sigma**2 / 2)  # f64 [L]
    om = omega[..., None]
    arg = (om * lh).astype(F32)
    hi = np.sin(arg)
    dhi = lh * np.cos(arg)
    lo = np.sin(om / F32(2))
    dlo = F32(0.5) * np.cos(om / F32(2))
    f = (w * hi.astype(np.float64) / lo.astype(np.float64)).sum(-1)
    num = (lo * dhi - hi * dlo).astype(F32)
    den = (lo * lo).astype(F32)
    ds = (w * num.astype(np.float64) / den.astype(np.float64)).sum(-1)
    sc = ds / (f + 1e-4)
    return sc[..., None] * vec.astype(np.float64) / omega.astype(np.float64)[..., None]


class SO3Diffuser:
    """so3_diffuser.py:194-602 (sampler-path subset)."""

    def __init__(self, conf):
        self.schedule = conf.schedule
        self.min_sigma, self.max_sigma = conf.min_sigma, conf.max_sigma
        self.num_sigma, self.num_omega = conf.num_sigma, conf.num_omega
        self.discrete_omega = np.linspace(0, np.pi, conf.num_omega + 1)[1:]
        self._rows: dict = {}
        np.random.seed(conf.seed)  # so3_diffuser.py:286

    @property
    def discrete_sigma(self):
        return self.sigma(np.linspace(0.0, 1.0, self.num_sigma))

    def sigma(self, t):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f"Invalid t={t}")
        return np.log(t * np.exp(self.max_sigma) + (1 - t) * np.exp(self.min_sigma))

    def sigma_idx(self, sigma):
        return np.digitize(sigma, self.discrete_sigma) - 1

    def t_to_idx(self, t):
        return self.sigma_idx(self.sigma(t))

    def diffusion_coef(self, t):
        return np.sqrt(2 * (np.exp(self.max_sigma) - np.exp(self.min_sigma)) * self.sigma(t) / np.exp(self.sigma(t)))

    def _row(self, idx: int):
        """(pdf, cdf, score_norms) rows of the start-up tables, so3_diffuser.py:247-276."""
        idx = int(idx)
        if idx not in self._rows:
            sig = self.discrete_sigma[idx]
            ev = igso3_expansion_np(self.discrete_omega, sig)
            pdf = ev * (1 - np.cos(self.discrete_omega)) / np.pi
            cdf = pdf.cumsum() / self.num_omega * np.pi
            sn = score_np(ev, self.discrete_omega, sig)
            self._rows[idx] = (pdf, cdf, sn)
        return self._rows[idx]

    def score_scaling(self, t):
        pdf, _, sn = self._row(self.t_to_idx(t))
        return np.sqrt(np.abs(np.sum(sn**2 * pdf) / np.sum(pdf))) / np.sqrt(3)

    def sample_igso3(self, t, n_samples=1):
        x = np.random.rand(n_samples)
        return np.interp(x, self._row(self.t_to_idx(t))[1], self.discrete_omega)

    def sample(self, t, n_samples=1):
        x = np.random.randn(n_samples, 3)
        x /= np.linalg.norm(x, axis=-1, keepdims=True)
        return x * self.sample_igso3(t, n_samples=n_samples)[:, None]

    def sample_ref(self, n_samples=1):
        return self.sample(1.0, n_samples=n_samples)

    def torch_score(self, vec_f32, t):
        """vec [B,N,3] f32, t [B] -> [B,N,3] f64."""
        t = np.asarray(t, dtype=np.float64).reshape(-1)
        out = np.empty(vec_f32.shape, dtype=np.float64)
        for b in range(vec_f32.shape[0]):
            sig = self.discrete_sigma[self.t_to_idx(np.float64(F32(t[b])))]
            out[b] = torch_score_mixed(vec_f32[b], sig)
        return out

    def reverse(self, rot_t, score_t, t, dt, noise_scale=1.0, z=None, orthogonalize=False):
        """so3_diffuser.py:569-602."""
        g_t = self.diffusion_coef(t)
        if z is None:
            z = np.random.normal(size=score_t.shape)
        z = noise_scale * z
        perturb = (g_t**2) * score_t * dt + g_t * np.sqrt(dt) * z
        n = int(np.prod(rot_t.shape[:-1]))
        return fr.compose_rotvec(rot_t.reshape(n, 3), perturb.reshape(n, 3), orthogonalize).reshape(rot_t.shape)


class R3Diffuser:
    """r3_diffuser.py:12-440 (sampler-path subset)."""

    def __init__(self, conf):
        self._conf = conf
        self.min_b, self.max_b = conf.min_b, conf.max_b
        np.random.seed(conf.seed)  # r3_diffuser.py:24

    def _scale(self, x):
        return x * self._conf.coordinate_scaling

    def _unscale(self, x):
        return x / self._conf.coordinate_scaling

    def b_t(self, t):
        if np.any(t < 0) or np.any(t > 1):
            raise ValueError(f"Invalid t={t}")
        return self.min_b + t * (self.max_b - self.min_b)

    def marginal_b_t(self, t):
        return t * self.min_b + (1 / 2) * (t**2) * (self.max_b - self.min_b)

    def conditional_var(self, t):
        return 1 - np.exp(-self.marginal_b_t(t))

    def score_scaling(self, t):
        return 1 / np.sqrt(self.conditional_var(t))

    def sample_stationary_distribution(self, x_reference, diffuse_mask, chain_indices=None):
        """r3_diffuser.py:294-331."""
        xs = self._scale(x_reference)
        bm = diffuse_mask.astype(bool) if diffuse_mask is not None else np.ones(x_reference.shape[:-1], dtype=np.bool_)
        loc = np.zeros_like(x_reference[bm])
        inpaint = np.random.normal(loc=loc, scale=np.ones_like(loc))
        out = xs.copy()
        out[bm] = inpaint
        return self._unscale(out)

    def reverse(self, x_t, score_t, t, dt, diffuse_mask=None, center=True, noise_scale=1.0, z=None):
        """r3_diffuser.py:344-385 (COM quirk: sum over all rows / number of diffused rows)."""
        x_t = self._scale(x_t)
        g_t = np.sqrt(self.b_t(t))
        f_t = -1 / 2 * self.b_t(t) * x_t
        if z is None:
            z = np.random.normal(size=score_t.shape)
        z = noise_scale * z
        perturb = (f_t - g_t**2 * score_t) * dt + g_t * np.sqrt(dt) * z
        if diffuse_mask is not None:
            perturb *= diffuse_mask[..., None]
        else:
            diffuse_mask = np.ones(x_t.shape[:-1])
        x = x_t - perturb
        if center:
            com = np.sum(x, axis=-2) / np.sum(diffuse_mask, axis=-1)[..., None]
            x -= com[..., None, :]
        return self._unscale(x)

    def score_f32(self, x_t, x_0, t_f32):
        """r3_diffuser.py:410-440 with use_torch=True, scale=True: float32 throughout."""
        s = F32(self._conf.coordinate_scaling)
        x_t = x_t.astype(F32) * s
        x_0 = x_0.astype(F32) * s
        t = np.asarray(t_f32, dtype=F32)
        mb = (t * F32(self.min_b) + F32(0.5) * (t * t) * F32(self.max_b - self.min_b)).astype(F32)
        e = np.exp(F32(-0.5) * mb).astype(F32)
        cv = (F32(1) - np.exp(-mb)).astype(F32)
        return (-(x_t - e * x_0) / cv).astype(F32)


class SE3Diffuser:
    """se3_diffuser.py:39-529 (sampler-path subset) on (quat f32, trans f32) arrays."""

    def __init__(self, conf):
        self._conf = conf
        self._diffuse_rot, self._diffuse_trans = conf.diffuse_rot, conf.diffuse_trans
        self._so3_diffuser = SO3Diffuser(conf.so3)
        self._r3_diffuser = R3Diffuser(conf.r3)

    def score_scaling(self, t):
        return self._so3_diffuser.score_scaling(t), self._r3_diffuser.score_scaling(t)

    @staticmethod
    def _extract(quat_f32, trans_f32, orthogonalize=False):
        """se3_diffuser.py:16-23: f32 quat -> f32 matrix -> SciPy rotvec f64."""
        rot = fr.quat_to_rot(quat_f32.astype(F32)).astype(F32)
        rv = fr.scipy_from_matrix_as_rotvec(rot.astype(np.float64), orthogonalize)
        return trans_f32.astype(F32), rv

    @staticmethod
    def _assemble(rotvec, trans):
        """se3_diffuser.py:26-36: rotation matrices and translations rounded to float32."""
        rot = fr.scipy_from_rotvec_as_matrix(rotvec).astype(F32)
        return rot, np.asarray(trans)

    def sample_ref(self, n_samples, impute=None, diffuse_mask=None):
        """se3_diffuser.py:455-529; impute = (quat [n,4], trans [n,3]) or None.

        Returns (rot_mats f32 [n,3,3], trans [n,3]) — the Rigid before ``to_tensor_7``.
        """
        if impute is None:
            if diffuse_mask is not None:
                raise ValueError("Must provide imputation values for unmasked regions!")
            trans_imp = np.zeros((n_samples, 3), dtype=F32)
            rot_imp = np.zeros((n_samples, 3))
        else:
            if impute[0].shape[0] != n_samples:
                raise ValueError(f"impute should have shape ({n_samples}, ...)")
            trans_imp, rot_imp = self._extract(impute[0], impute[1])
        rot_ref = self._so3_diffuser.sample_ref(n_samples=n_samples)
        trans_ref = self._r3_diffuser.sample_stationary_distribution(trans_imp, diffuse_mask, None)
        if diffuse_mask is not None:
            m = diffuse_mask[..., None]
            rot_ref = m * rot_ref + (1 - m) * rot_imp
        return self._assemble(rot_ref, trans_ref)

    def reverse(self, quat_t, trans_t, rot_score, trans_score, t, dt, diffuse_mask=None, center=True,
                noise_scale=1.0, z_rot=None, z_trans=None, orthogonalize=False):
        """se3_diffuser.py:346-401. Returns (rot_mats f32, trans f32) of x_{t-1}."""
        tr, rv = self._extract(quat_t, trans_t, orthogonalize)
        rv1 = self._so3_diffuser.reverse(rv, rot_score, t, dt, noise_scale=noise_scale, z=z_rot,
                                         orthogonalize=orthogonalize)
        tr1 = self._r3_diffuser.reverse(tr, trans_score, t, dt, diffuse_mask=diffuse_mask, center=center,
                                        noise_scale=noise_scale, z=z_trans)
        if diffuse_mask is not None:
            m = diffuse_mask[..., None]
            tr1 = m * tr1 + (1 - m) * tr
            rv1 = m * rv1 + (1 - m) * rv
        rot, tr1 = self._assemble(rv1, tr1)
        return rot, tr1.astype(F32)

    def calc_rot_score(self, quats_t, quats_0, t):
        """se3_diffuser.py:281-292 (float32 quaternion algebra, float64 score)."""
        q0inv = fr.invert_quat(quats_0.astype(F32)).astype(F32)
        q0t = fr.quat_multiply(q0inv, quats_t.astype(F32)).astype(F32)
        rv = fr.quat_to_rotvec(q0t)
        return self._so3_diffuser.torch_score(rv, t)

    def calc_trans_score(self, trans_t, trans_0, t):
        return self._r3_diffuser.score_f32(trans_t, trans_0, t)
