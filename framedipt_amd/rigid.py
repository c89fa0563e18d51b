"""Tensor-like ``Rotation`` / ``Rigid`` objects over the HIP frame kernels.

Mirrors the subset of ``openfold/utils/rigid_utils.py`` (Rotation :289-850, Rigid :853-1448) that the sampler
API surface exchanges: ``SE3Diffuser.sample_ref`` / ``reverse`` return a ``Rigid``, ``inference_fn`` converts with
``from_tensor_7`` / ``to_tensor_7``.  All arithmetic is delegated to ``libfdipt_hip`` (no torch math).
"""
from __future__ import annotations

import torch

from . import _lib


def _flat(t, last):
    return t.reshape(-1, *last).contiguous().float()


def _call(fn, n, *tensors):
    """Launch ``fn`` over ``n`` items on the device (and current stream of the device) the tensors live on."""
    lib = _lib.load()
    dev = None
    for t in tensors:
        if t is not None:
            _lib.require_cuda(t, fn)
            if dev is not None and t.device != dev:
                raise _lib.FdiptError(f"{fn}: tensors on different devices ({dev} and {t.device})")
            dev = t.device
    with torch.cuda.device(dev):
        _lib.check(getattr(lib, fn)(n, *[_lib.ptr(t) for t in tensors], _lib.stream_ptr()), fn)


def quat_to_rot(quat: torch.Tensor) -> torch.Tensor:  # rigid_utils.py:185
    q = _flat(quat, (4,))
    out = torch.empty(q.shape[0], 3, 3, device=q.device)
    _call("fdipt_quat_to_rot", q.shape[0], q, out)
    return out.reshape(*quat.shape[:-1], 3, 3)


def rot_to_quat(rot: torch.Tensor) -> torch.Tensor:  # rigid_utils.py:208 (sign convention: w >= 0)
    r = _flat(rot, (3, 3))
    out = torch.empty(r.shape[0], 4, device=r.device)
    _call("fdipt_rot_to_quat", r.shape[0], r, out)
    return out.reshape(*rot.shape[:-2], 4)


def quat_multiply(q1, q2):  # rigid_utils.py:254
    a, b = _flat(q1, (4,)), _flat(q2, (4,))
    out = torch.empty_like(a)
    _call("fdipt_quat_multiply", a.shape[0], a, b, out)
    return out.reshape(q1.shape)


def quat_multiply_by_vec(q, v):  # rigid_utils.py:266
    a, b = _flat(q, (4,)), _flat(v, (3,))
    out = torch.empty_like(a)
    _call("fdipt_quat_multiply_by_vec", a.shape[0], a, b, out)
    return out.reshape(q.shape)


def invert_quat(q):  # rigid_utils.py:282
    a = _flat(q, (4,))
    out = torch.empty_like(a)
    _call("fdipt_invert_quat", a.shape[0], a, out)
    return out.reshape(q.shape)


class Rotation:
    """Holds either quaternions [*,4] or rotation matrices [*,3,3] (float32), like the reference class."""

    def __init__(self, rot_mats=None, quats=None, normalize_quats: bool = True):
        if (rot_mats is None) == (quats is None):
            raise ValueError("Exactly one input argument must be specified")
        if (rot_mats is not None and rot_mats.shape[-2:] != (3, 3)) or (quats is not None and quats.shape[-1] != 4):
            raise ValueError("Incorrectly shaped rotation matrix or quaternion")
        if quats is not None:
            quats = quats.float()
            if normalize_quats:
                quats = quats / torch.linalg.norm(quats, dim=-1, keepdim=True)
        self._rot_mats = rot_mats.float() if rot_mats is not None else None
        self._quats = quats

    @property
    def shape(self):
        return self._rot_mats.shape[:-2] if self._rot_mats is not None else self._quats.shape[:-1]

    @property
    def device(self):
        return (self._rot_mats if self._rot_mats is not None else self._quats).device

    def get_rot_mats(self):
        return self._rot_mats if self._rot_mats is not None else quat_to_rot(self._quats)

    def get_quats(self):
        return self._quats if self._quats is not None else rot_to_quat(self._rot_mats)

    def invert(self):
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.transpose(-1, -2).contiguous())
        return Rotation(quats=invert_quat(self._quats), normalize_quats=False)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats[idx + (slice(None), slice(None))])
        return Rotation(quats=self._quats[idx + (slice(None),)], normalize_quats=False)

    def to(self, device):
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.to(device))
        return Rotation(quats=self._quats.to(device), normalize_quats=False)


class Rigid:
    def __init__(self, rots: Rotation, trans: torch.Tensor):
        if rots.shape != trans.shape[:-1]:
            raise ValueError("Rots and trans incompatible")
        self._rots = rots
        self._trans = trans.float()  # forced full precision, rigid_utils.py:898-899

    @property
    def shape(self):
        return self._trans.shape[:-1]

    @property
    def device(self):
        return self._trans.device

    def get_rots(self):
        return self._rots

    def get_trans(self):
        return self._trans

    @staticmethod
    def identity(shape, device=None, requires_grad: bool = False):
        q = torch.zeros(*shape, 4, device=device)
        q[..., 0] = 1
        return Rigid(Rotation(quats=q, normalize_quats=False), torch.zeros(*shape, 3, device=device))

    @staticmethod
    def from_tensor_7(t: torch.Tensor, normalize_quats: bool = False):  # rigid_utils.py:1215-1230
        if t.shape[-1] != 7:
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(quats=t[..., :4], normalize_quats=normalize_quats), t[..., 4:])

    @staticmethod
    def from_tensor_4x4(t: torch.Tensor):  # rigid_utils.py:1180-1198
        if t.shape[-2:] != (4, 4):
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(rot_mats=t[..., :3, :3].contiguous()), t[..., :3, 3].contiguous())

    def to_tensor_4x4(self) -> torch.Tensor:  # rigid_utils.py:1165-1178
        out = self._trans.new_zeros((*self.shape, 4, 4))
        out[..., :3, :3] = self._rots.get_rot_mats()
        out[..., :3, 3] = self._trans
        out[..., 3, 3] = 1
        return out

    @staticmethod
    def from_3_points(p_neg_x_axis, origin, p_xy_plane, eps: float = 1e-8):  # rigid_utils.py:1233-1275 (Gram-Schmidt frame)
        a, o, c = (_flat(x, (3,)) for x in (p_neg_x_axis, origin, p_xy_plane))
        rot = torch.empty(o.shape[0], 3, 3, device=o.device)
        lib = _lib.load()
        _lib.require_cuda(o, "from_3_points")
        with torch.cuda.device(o.device):
            _lib.check(lib.fdipt_rigid_from_3_points(o.shape[0], _lib.ptr(a), _lib.ptr(o), _lib.ptr(c), float(eps), _lib.ptr(rot),
                                                     _lib.stream_ptr()), "fdipt_rigid_from_3_points")
        return Rigid(Rotation(rot_mats=rot.reshape(*origin.shape[:-1], 3, 3)), origin.float())

    @staticmethod
    def cat(rigids, dim: int):  # rigid_utils.py:1333-1352 (data movement only)
        d = dim if dim >= 0 else dim + len(rigids[0].shape)
        return Rigid(Rotation(rot_mats=torch.cat([r.get_rots().get_rot_mats() for r in rigids], dim=d)),
                     torch.cat([r.get_trans() for r in rigids], dim=d))

    def unsqueeze(self, dim: int):  # rigid_utils.py:1309-1331
        if dim >= len(self.shape) + 1 or dim < -len(self.shape) - 1:
            raise ValueError("Invalid dimension")
        d = dim if dim >= 0 else dim + len(self.shape) + 1
        rots = self._rots
        rots = (Rotation(rot_mats=rots._rot_mats.unsqueeze(d)) if rots._rot_mats is not None
                else Rotation(quats=rots._quats.unsqueeze(d), normalize_quats=False))
        return Rigid(rots, self._trans.unsqueeze(d))

    def apply_trans_fn(self, fn):  # rigid_utils.py:1371-1385
        return Rigid(self._rots, fn(self._trans))

    def scale_translation(self, trans_scale_factor: float):  # rigid_utils.py:1387-1399
        return self.apply_trans_fn(lambda t: t * trans_scale_factor)

    def map_tensor_fn(self, fn):  # rigid_utils.py:1149-1163
        r = self._rots
        if r._rot_mats is not None:
            m = r._rot_mats.reshape(*r._rot_mats.shape[:-2], 9)
            rots = Rotation(rot_mats=torch.stack(list(map(fn, torch.unbind(m, dim=-1))), dim=-1).reshape(*m.shape[:-1], 3, 3))
        else:
            rots = Rotation(quats=torch.stack(list(map(fn, torch.unbind(r._quats, dim=-1))), dim=-1), normalize_quats=False)
        return Rigid(rots, torch.stack(list(map(fn, torch.unbind(self._trans, dim=-1))), dim=-1))

    def stop_rot_gradient(self):  # rigid_utils.py:1401 (inference only: nothing to detach)
        return self

    def to_tensor_7(self) -> torch.Tensor:  # rigid_utils.py:1200-1212
        return torch.cat([self._rots.get_quats(), self._trans], dim=-1)

    def _t7(self):
        return _flat(self.to_tensor_7(), (7,))

    def apply(self, pts):  # rigid_utils.py:1104
        p = _flat(pts.expand(*self.shape, 3), (3,))
        out = torch.empty_like(p)
        _call("fdipt_rigid_apply", p.shape[0], self._t7(), p, out)
        return out.reshape(*self.shape, 3)

    def invert_apply(self, pts):  # rigid_utils.py:1118
        p = _flat(pts.expand(*self.shape, 3), (3,))
        out = torch.empty_like(p)
        _call("fdipt_rigid_invert_apply", p.shape[0], self._t7(), p, out)
        return out.reshape(*self.shape, 3)

    def compose(self, r):  # rigid_utils.py:1065
        a, b = self._t7(), r._t7()
        rot = torch.empty(a.shape[0], 3, 3, device=a.device)
        tr = torch.empty(a.shape[0], 3, device=a.device)
        _call("fdipt_rigid_compose", a.shape[0], a, b, rot, tr)
        return Rigid(Rotation(rot_mats=rot.reshape(*self.shape, 3, 3)), tr.reshape(*self.shape, 3))

    def invert(self):  # rigid_utils.py:1132
        a = self._t7()
        rot = torch.empty(a.shape[0], 3, 3, device=a.device)
        tr = torch.empty(a.shape[0], 3, device=a.device)
        _call("fdipt_rigid_invert", a.shape[0], a, rot, tr)
        return Rigid(Rotation(rot_mats=rot.reshape(*self.shape, 3, 3)), tr.reshape(*self.shape, 3))

    def compose_q_update_vec(self, q_update_vec, update_mask=None):  # rigid_utils.py:1039 (fork: update_mask)
        a = self._t7()
        u = _flat(q_update_vec, (6,))
        m = None if update_mask is None else update_mask.reshape(-1).contiguous().float()
        out = torch.empty_like(a)
        _call("fdipt_rigid_compose_q_update", a.shape[0], a, u, m, out)
        return Rigid.from_tensor_7(out.reshape(*self.shape, 7))

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        return Rigid(self._rots[idx], self._trans[idx + (slice(None),)])

    def to(self, device):
        return Rigid(self._rots.to(device), self._trans.to(device))
