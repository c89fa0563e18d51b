"""ORACLE (test infrastructure, not product code): rigid-frame math in NumPy.

CPU restatement of the reference's frame algebra, used only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the
checker for the HIP kernels.  Every function cites the reference lines it
follows (paths relative to the reference repository root).

Third-party arithmetic restated here: ``scipy.spatial.transform.Rotation``
(reference pin scipy 1.7.3, ``environment.yml:241``), call sites
``framedipt/data/transforms.py:42,46`` and ``framedipt/diffusion/se3_diffuser.py:21,31``.
The restatement follows SciPy's published algorithm and is
pinned against SciPy itself in ``tests/test_oracle_frames.py``.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# ---------------------------------------------------------------- quaternions
def quat_to_rot(q: np.ndarray) -> np.ndarray:
    """openfold/utils/rigid_utils.py:173-205 (``_QTR_MAT`` expansion; no normalisation)."""
    a, b, c, d = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    r = np.empty(q.shape[:-1] + (3, 3), dtype=q.dtype)
    r[..., 0, 0] = a * a + b * b - c * c - d * d
    r[..., 0, 1] = 2 * b * c - 2 * a * d
    r[..., 0, 2] = 2 * b * d + 2 * a * c
    r[..., 1, 0] = 2 * b * c + 2 * a * d
    r[..., 1, 1] = a * a - b * b + c * c - d * d
    r[..., 1, 2] = 2 * c * d - 2 * a * b
    r[..., 2, 0] = 2 * b * d - 2 * a * c
    r[..., 2, 1] = 2 * c * d + 2 * a * b
    r[..., 2, 2] = a * a - b * b - c * c + d * d
    return r


def rot_to_quat(rot: np.ndarray) -> np.ndarray:
    """openfold/utils/rigid_utils.py:208-227: top eigenvector of the 4x4 K/3 matrix (sign arbitrary)."""
    xx, xy, xz = rot[..., 0, 0], rot[..., 0, 1], rot[..., 0, 2]
    yx, yy, yz = rot[..., 1, 0], rot[..., 1, 1], rot[..., 1, 2]
    zx, zy, zz = rot[..., 2, 0], rot[..., 2, 1], rot[..., 2, 2]
    k = np.empty(rot.shape[:-2] + (4, 4), dtype=rot.dtype)
    k[..., 0, 0] = xx + yy + zz
    k[..., 0, 1] = k[..., 1, 0] = zy - yz
    k[..., 0, 2] = k[..., 2, 0] = xz - zx
    k[..., 0, 3] = k[..., 3, 0] = yx - xy
    k[..., 1, 1] = xx - yy - zz
    k[..., 1, 2] = k[..., 2, 1] = xy + yx
    k[..., 1, 3] = k[..., 3, 1] = xz + zx
    k[..., 2, 2] = yy - xx - zz
    k[..., 2, 3] = k[..., 3, 2] = yz + zy
    k[..., 3, 3] = zz - xx - yy
    k = k * rot.dtype.type(1.0 / 3.0)
    _, v = np.linalg.eigh(k)
    return v[..., -1]


def quat_multiply(q1: np.ndarray, q2: np.ndarray) -> np.ndarray:
    """openfold/utils/rigid_utils.py:230-263 (Hamilton product, scalar first)."""
    a1, b1, c1, d1 = (q1[..., i] for i in range(4))
    a2, b2, c2, d2 = (q2[..., i] for i in range(4))
    return np.stack(
        [
            a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2,
            a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
            a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2,
            a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2,
        ],
        axis=-1,
    )


def quat_multiply_by_vec(q: np.ndarray, v: np.ndarray) -> np.ndarray:
    """openfold/utils/rigid_utils.py:266-279: q (x) (0, v)."""
    a, b, c, d = (q[..., i] for i in range(4))
    x, y, z = (v[..., i] for i in range(3))
    return np.stack(
        [
            -b * x - c * y - d * z,
            a * x + c * z - d * y,
            a * y - b * z + d * x,
            a * z + b * y - c * x,
        ],
        axis=-1,
    )


def invert_quat(q: np.ndarray) -> np.ndarray:
    """openfold/utils/rigid_utils.py:282-286: conjugate / |q|^2."""
    qp = q.copy()
    qp[..., 1:] *= -1
    return qp / np.sum(q * q, axis=-1, keepdims=True)


def rot_vec_mul(r: np.ndarray, t: np.ndarray) -> np.ndarray:
    """openfold/utils/rigid_utils.py:82-106."""
    x, y, z = t[..., 0], t[..., 1], t[..., 2]
    return np.stack(
        [
            r[..., 0, 0] * x + r[..., 0, 1] * y + r[..., 0, 2] * z,
            r[..., 1, 0] * x + r[..., 1, 1] * y + r[..., 1, 2] * z,
            r[..., 2, 0] * x + r[..., 2, 1] * y + r[..., 2, 2] * z,
        ],
        axis=-1,
    )


def rot_matmul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """openfold/utils/rigid_utils.py:22-79."""
    return np.einsum("...ij,...jk->...ik", a, b)


def rigid_apply(rot: np.ndarray, trans: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """Rigid.apply, openfold/utils/rigid_utils.py:1104-1116."""
    return rot_vec_mul(rot, pts) + trans


def rigid_invert_apply(rot: np.ndarray, trans: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """Rigid.invert_apply, openfold/utils/rigid_utils.py:1118-1130."""
    return rot_vec_mul(np.swapaxes(rot, -1, -2), pts - trans)


def rigid_compose(r1, t1, r2, t2):
    """Rigid.compose, openfold/utils/rigid_utils.py:1065-1079."""
    return rot_matmul(r1, r2), rot_vec_mul(r1, t2) + t1


def rigid_invert(r, t):
    """Rigid.invert, openfold/utils/rigid_utils.py:1132-1143."""
    rt = np.swapaxes(r, -1, -2)
    return rt, -rot_vec_mul(rt, t)


def compose_q_update_vec(quat: np.ndarray, trans: np.ndarray, upd: np.ndarray, mask: np.ndarray):
    """Rigid.compose_q_update_vec with update_mask (fork): rigid_utils.py:587-616,1039-1063.

    quat [*,4] f32, trans [*,3] f32 (scaled units), upd [*,6], mask [*,1].
    The translation update uses the pre-update rotation.
    """
    q_vec, t_vec = upd[..., :3], upd[..., 3:]
    dq = quat_multiply_by_vec(quat, q_vec) * mask
    new_q = quat + dq
    new_q = new_q / np.linalg.norm(new_q, axis=-1, keepdims=True)
    dt = rot_vec_mul(quat_to_rot(quat), t_vec) * mask
    return new_q.astype(F32), (trans + dt).astype(F32)


def quat_to_rotvec(quat: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """framedipt/data/transforms.py:53-69 (torch twin of SciPy as_rotvec, with +eps in the sine)."""
    dt = quat.dtype.type
    flip = (quat[..., :1] < 0).astype(quat.dtype)
    quat = (-1 * quat) * flip + (1 - flip) * quat
    angle = 2 * np.arctan2(np.linalg.norm(quat[..., 1:], axis=-1), quat[..., 0])
    angle2 = angle * angle
    small = 2 + angle2 / 12 + 7 * angle2 * angle2 / 2880
    large = angle / np.sin(angle / 2 + dt(eps))
    sm = (angle <= 1e-3).astype(quat.dtype)
    scale = small * sm + (1 - sm) * large
    return (scale[..., None] * quat[..., 1:]).astype(quat.dtype)


# ------------------------------------------------ SciPy Rotation conventions
def scipy_from_rotvec_as_matrix(rv: np.ndarray) -> np.ndarray:
    """Rotation.from_rotvec(rv).as_matrix() (SciPy _rotation.pyx), float64."""
    rv = np.asarray(rv, dtype=np.float64)
    th = np.linalg.norm(rv, axis=-1)
    th2 = th * th
    small = th <= 1e-3
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(small, 0.5 - th2 / 48 + th2 * th2 / 3840, np.sin(th / 2) / np.where(small, 1.0, th))
    x, y, z = (scale * rv[..., i] for i in range(3))
    w = np.cos(th / 2)
    # SciPy normalises in from_rotvec? No: the quaternion is unit by construction.
    r = np.empty(rv.shape[:-1] + (3, 3))
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    r[..., 0, 0] = x2 - y2 - z2 + w2
    r[..., 1, 0] = 2 * (xy + zw)
    r[..., 2, 0] = 2 * (xz - yw)
    r[..., 0, 1] = 2 * (xy - zw)
    r[..., 1, 1] = -x2 + y2 - z2 + w2
    r[..., 2, 1] = 2 * (yz + xw)
    r[..., 0, 2] = 2 * (xz + yw)
    r[..., 1, 2] = 2 * (yz - xw)
    r[..., 2, 2] = -x2 - y2 + z2 + w2
    return r


def scipy_from_matrix_as_rotvec(m: np.ndarray, orthogonalize: bool = False) -> np.ndarray:
    """Rotation.from_matrix(m).as_rotvec(): Markley quaternion, then log map (float64).

    ``orthogonalize=True`` adds the SVD projection SciPy >= 1.8 applies first
    (the oracle container has 1.15.3; the reference pins 1.7.3 which does not).
    """
    m = np.asarray(m, dtype=np.float64)
    if orthogonalize:
        u, _, vt = np.linalg.svd(m)
        det = np.linalg.det(u @ vt)
        u = u.copy()
        u[..., :, 2] *= det[..., None]
        m = u @ vt
    shp = m.shape[:-2]
    m = m.reshape(-1, 3, 3)
    n = m.shape[0]
    dec = np.empty((n, 4))
    dec[:, 0], dec[:, 1], dec[:, 2] = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    dec[:, 3] = dec[:, 0] + dec[:, 1] + dec[:, 2]
    ch = dec.argmax(axis=1)
    q = np.empty((n, 4))  # scalar-last
    idx = np.nonzero(ch != 3)[0]
    i = ch[idx]
    j = (i + 1) % 3
    k = (j + 1) % 3
    q[idx, i] = 1 - dec[idx, 3] + 2 * m[idx, i, i]
    q[idx, j] = m[idx, j, i] + m[idx, i, j]
    q[idx, k] = m[idx, k, i] + m[idx, i, k]
    q[idx, 3] = m[idx, k, j] - m[idx, j, k]
    idx = np.nonzero(ch == 3)[0]
    q[idx, 0] = m[idx, 2, 1] - m[idx, 1, 2]
    q[idx, 1] = m[idx, 0, 2] - m[idx, 2, 0]
    q[idx, 2] = m[idx, 1, 0] - m[idx, 0, 1]
    q[idx, 3] = 1 + dec[idx, 3]
    q /= np.linalg.norm(q, axis=1)[:, None]
    q = np.where(q[:, 3:4] < 0, -q, q)
    ang = 2 * np.arctan2(np.linalg.norm(q[:, :3], axis=1), q[:, 3])
    a2 = ang * ang
    small = ang <= 1e-3
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(small, 2 + a2 / 12 + 7 * a2 * a2 / 2880, ang / np.sin(np.where(small, 1.0, ang) / 2))
    return (scale[:, None] * q[:, :3]).reshape(shp + (3,))


def compose_rotvec(r1: np.ndarray, r2: np.ndarray, orthogonalize: bool = False) -> np.ndarray:
    """framedipt/data/transforms.py:33-46: log(exp(r1) exp(r2))."""
    c = np.einsum("...ij,...jk->...ik", scipy_from_rotvec_as_matrix(r1), scipy_from_rotvec_as_matrix(r2))
    return scipy_from_matrix_as_rotvec(c, orthogonalize)
