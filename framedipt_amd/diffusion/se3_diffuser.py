"""SE(3) diffuser with the API surface of ``framedipt.diffusion.se3_diffuser.SE3Diffuser``.

Host code owns configuration, the legacy ``np.random`` noise stream (SURVEY.md section 0 finding 10) and the
IGSO(3) table rows; every frame computation (SciPy-convention exp/log, the reverse step, the scores) is a HIP
kernel behind ``libfdipt_hip``.  There is no CPU fallback: calls raise ``FdiptError`` without the library / a GPU.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from ..rigid import Rigid, Rotation
from . import r3_diffuser, so3_diffuser


def _dev(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise _lib.FdiptError("SE3Diffuser needs an MI355X device: the frame kernels have no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def so3_log(rot: torch.Tensor) -> torch.Tensor:
    """Rotation.from_matrix(.).as_rotvec() on the device (float64)."""
    lib = _lib.load()
    r = rot.reshape(-1, 3, 3).double().contiguous()
    _lib.require_cuda(r, "so3_log")
    out = torch.empty(r.shape[0], 3, dtype=torch.float64, device=r.device)
    with torch.cuda.device(r.device):
        _lib.check(lib.fdipt_so3_log(r.shape[0], _lib.ptr(r), _lib.ptr(out), _lib.stream_ptr()), "so3_log")
    return out.reshape(*rot.shape[:-2], 3)


def so3_exp(rotvec: torch.Tensor) -> torch.Tensor:
    """Rotation.from_rotvec(.).as_matrix() on the device (float64)."""
    lib = _lib.load()
    v = rotvec.reshape(-1, 3).double().contiguous()
    _lib.require_cuda(v, "so3_exp")
    out = torch.empty(v.shape[0], 3, 3, dtype=torch.float64, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(lib.fdipt_so3_exp(v.shape[0], _lib.ptr(v), _lib.ptr(out), _lib.stream_ptr()), "so3_exp")
    return out.reshape(*rotvec.shape[:-1], 3, 3)


def _extract_trans_rots(rigid: Rigid):
    """se3_diffuser.py:16-23: (trans float32, rotvec float64) as NumPy arrays."""
    rot = rigid.get_rots().get_rot_mats()
    rv = so3_log(rot).cpu().numpy()
    return rigid.get_trans().cpu().numpy(), rv


def _assemble_rigid(rotvec: np.ndarray, trans: np.ndarray, device) -> Rigid:
    """se3_diffuser.py:26-36: Rigid(rot_mats float32, trans float32)."""
    rot = so3_exp(torch.as_tensor(np.asarray(rotvec, dtype=np.float64), device=device)).float()
    return Rigid(Rotation(rot_mats=rot), torch.as_tensor(np.asarray(trans), device=device).float())


class SE3Diffuser:
    def __init__(self, se3_conf, device=None) -> None:
        self._se3_conf = se3_conf
        self._device = device
        self._diffuse_rot = se3_conf.diffuse_rot
        self._so3_diffuser = so3_diffuser.SO3Diffuser(se3_conf.so3)
        self._diffuse_trans = se3_conf.diffuse_trans
        self._r3_diffuser = r3_diffuser.R3Diffuser(se3_conf.r3)

    # ------------------------------------------------------------------ scalars
    def score_scaling(self, t: float):
        return self._so3_diffuser.score_scaling(t), self._r3_diffuser.score_scaling(t)

    # ------------------------------------------------------------------ scores
    def calc_trans_score(self, trans_t, trans_0, t, use_torch: bool = False, scale: bool = True):
        """se3_diffuser.py:269-279 -> r3_diffuser.py:410-440 (device float32 path); ``scale`` multiplies both translations by
        the coordinate scaling first (r3_diffuser.py:436-438)."""
        lib = _lib.load()
        tt = torch.as_tensor(trans_t).float()
        t0 = torch.as_tensor(trans_0).float().to(tt.device)
        _lib.require_cuda(tt, "calc_trans_score")
        shp = tt.shape
        B = shp[0] if tt.dim() == 3 else 1
        N = shp[-2]
        tv = torch.as_tensor(t, dtype=torch.float32, device=tt.device).reshape(-1).expand(B).contiguous()
        out = torch.empty(B, N, 3, device=tt.device)
        r3 = self._r3_diffuser
        with torch.cuda.device(tt.device):
            _lib.check(lib.fdipt_r3_trans_score(B, N, _lib.ptr(tt.reshape(B, N, 3).contiguous()),
                                                _lib.ptr(t0.reshape(B, N, 3).contiguous()), _lib.ptr(tv), r3.min_b, r3.max_b,
                                                r3._r3_conf.coordinate_scaling if scale else 1.0, None, _lib.ptr(out),
                                                _lib.stream_ptr()), "r3_trans_score")
        out = out.reshape(shp)
        return out if use_torch else out.cpu().numpy()

    def calc_rot_score(self, rots_t: Rotation, rots_0: Rotation, t: torch.Tensor) -> torch.Tensor:
        """se3_diffuser.py:281-292: float64 [B,N,3] IGSO(3) score of log(R_0^-1 R_t)."""
        lib = _lib.load()
        qt, q0 = rots_t.get_quats(), rots_0.get_quats()
        _lib.require_cuda(qt, "calc_rot_score")
        shp = qt.shape[:-1]
        B = shp[0] if len(shp) == 2 else 1
        N = shp[-1]
        t_np = torch.as_tensor(t).detach().cpu().numpy().reshape(-1)
        out = torch.empty(B, N, 3, dtype=torch.float64, device=qt.device)
        so3 = self._so3_diffuser
        if so3.use_cached_score:  # so3_diffuser.py:389-396: table lookup instead of the series
            rows = np.broadcast_to(so3.score_table_rows(t_np), (B, so3.num_omega)).copy()
            tab, edges = torch.as_tensor(rows, device=qt.device), torch.as_tensor(so3.omega_edges, device=qt.device)
            with torch.cuda.device(qt.device):
                _lib.check(lib.fdipt_igso3_rot_score_cached(B, N, _lib.ptr(qt.reshape(B, N, 4).contiguous()),
                                                            _lib.ptr(q0.reshape(B, N, 4).to(qt.device).contiguous()), _lib.ptr(tab),
                                                            _lib.ptr(edges), so3.num_omega, None, _lib.ptr(out), _lib.stream_ptr()),
                           "igso3_rot_score_cached")
            return out.reshape(*shp, 3)
        sig = so3.score_sigma(t_np)
        sig = torch.as_tensor(np.broadcast_to(sig, (B,)).copy(), device=qt.device)
        with torch.cuda.device(qt.device):
            _lib.check(lib.fdipt_igso3_rot_score(B, N, _lib.ptr(qt.reshape(B, N, 4).contiguous()),
                                                 _lib.ptr(q0.reshape(B, N, 4).to(qt.device).contiguous()), _lib.ptr(sig), None,
                                                 _lib.ptr(out), _lib.stream_ptr()), "igso3_rot_score")
        return out.reshape(*shp, 3)

    def _apply_mask(self, x_diff, x_fixed, diff_mask):
        return diff_mask * x_diff + (1 - diff_mask) * x_fixed

    # ------------------------------------------------------------------ reverse step
    def reverse_device(self, rigids_t, rot_score, trans_score, diffuse_mask, z_rot, z_trans, t, dt, center=True,
                       noise_scale=1.0, rigids_out=None, rot_out=None, atoms=None, traj=None):
        """Device-resident reverse step on tensor_7 frames [B,N,7]; noise given (N(0,1), float64).
        ``atoms=(psi, aatype, bb_tables, atom37)``: also compute_backbone of x_{t-1} in the same launch.
        ``traj=(pred_rigids, fixed_mask * res_mask, trans_traj_row)``: also the step's trans_traj row (utils.py:390-400)."""
        lib = _lib.load()
        _lib.require_cuda(rigids_t, "reverse")
        B, N = rigids_t.shape[0], rigids_t.shape[1]
        if rigids_out is None:
            rigids_out = torch.empty_like(rigids_t)
        so3, r3 = self._so3_diffuser, self._r3_diffuser
        psi, aatype, tables, atom37 = atoms if atoms is not None else (None, None, None, None)
        pred, tfix, ttraj = traj if traj is not None else (None, None, None)
        for x in (rot_score, trans_score, diffuse_mask, z_rot, z_trans, rigids_out, rot_out, psi, aatype, tables, atom37, pred, tfix, ttraj):
            if x is not None and x.device != rigids_t.device:
                raise _lib.FdiptError(f"reverse: tensors on different devices ({rigids_t.device} and {x.device})")
        with torch.cuda.device(rigids_t.device):
            self._reverse_launch(lib, B, N, rigids_t, rot_score, trans_score, diffuse_mask, z_rot, z_trans, t, dt, noise_scale,
                                 center, rigids_out, rot_out, psi, aatype, tables, atom37, pred, tfix, ttraj)
        return rigids_out

    def _reverse_launch(self, lib, B, N, rigids_t, rot_score, trans_score, diffuse_mask, z_rot, z_trans, t, dt, noise_scale, center,
                        rigids_out, rot_out, psi, aatype, tables, atom37, pred, tfix, ttraj):
        so3, r3 = self._so3_diffuser, self._r3_diffuser
        _lib.check(lib.fdipt_se3_reverse_step_traj(
            B, N, _lib.ptr(rigids_t), _lib.ptr(rot_score), _lib.ptr(trans_score), _lib.ptr(diffuse_mask),
            _lib.ptr(z_rot), _lib.ptr(z_trans), float(t), float(dt), float(noise_scale), int(bool(center)),
            int(bool(self._diffuse_rot)), int(bool(self._diffuse_trans)), so3.min_sigma, so3.max_sigma, r3.min_b, r3.max_b,
            r3._r3_conf.coordinate_scaling, _lib.ptr(rigids_out), _lib.ptr(rot_out), _lib.ptr(psi), _lib.ptr(aatype),
            _lib.ptr(tables), _lib.ptr(atom37), _lib.ptr(pred), _lib.ptr(tfix), _lib.ptr(ttraj), _lib.stream_ptr()), "se3_reverse_step")
        return rigids_out

    def reverse(self, rigid_t: Rigid, rot_score, trans_score, t: float, dt: float, diffuse_mask=None,
                chain_indices=None, center: bool = True, noise_scale: float = 1.0) -> Rigid:
        """se3_diffuser.py:346-401.  Noise comes from the global ``np.random`` stream in the reference order
        (SO(3) draw, then R^3 draw, each of ``score.shape``)."""
        dev = rigid_t.device
        t7 = rigid_t.to_tensor_7().float()
        lead = t7.shape[:-2]
        N = t7.shape[-2]
        B = int(np.prod(lead)) if len(lead) else 1
        rs = torch.as_tensor(np.asarray(rot_score.detach().cpu() if torch.is_tensor(rot_score) else rot_score),
                             dtype=torch.float64)
        ts = torch.as_tensor(np.asarray(trans_score.detach().cpu() if torch.is_tensor(trans_score) else trans_score),
                             dtype=torch.float32)
        z_rot = np.random.normal(size=tuple(rs.shape)) if self._diffuse_rot else np.zeros(tuple(rs.shape))
        z_trans = np.random.normal(size=tuple(ts.shape)) if self._diffuse_trans else np.zeros(tuple(ts.shape))
        dm = None
        if diffuse_mask is not None:
            dm = torch.as_tensor(np.asarray(diffuse_mask.detach().cpu() if torch.is_tensor(diffuse_mask) else diffuse_mask),
                                 dtype=torch.float32).reshape(B, N).to(dev).contiguous()
        rot_out = torch.empty(B, N, 3, 3, device=dev)
        out = self.reverse_device(
            t7.reshape(B, N, 7).contiguous(), rs.reshape(B, N, 3).to(dev).contiguous(),
            ts.reshape(B, N, 3).to(dev).contiguous(), dm,
            torch.as_tensor(z_rot, device=dev).reshape(B, N, 3).contiguous(),
            torch.as_tensor(z_trans, device=dev).reshape(B, N, 3).contiguous(), t, dt, center, noise_scale, rot_out=rot_out)
        return Rigid(Rotation(rot_mats=rot_out.reshape(*lead, N, 3, 3)), out[..., 4:].reshape(*lead, N, 3))

    # ------------------------------------------------------------------ forward noising + log-probabilities (EigenFold confidence)
    def _consts(self):
        so3, r3 = self._so3_diffuser, self._r3_diffuser
        return so3.min_sigma, so3.max_sigma, r3.min_b, r3.max_b, r3._r3_conf.coordinate_scaling

    @staticmethod
    def _state_of(rigid: Rigid):
        """(rot float32 [B,N,3,3], trans float32 [B,N,3], lead shape) of a Rigid shaped [N] or [B,N]."""
        rot, trans = rigid.get_rots().get_rot_mats().float(), rigid.get_trans().float()
        _lib.require_cuda(rot, "SE3Diffuser")
        lead, N = rot.shape[:-3], rot.shape[-3]
        B = int(np.prod(lead)) if len(lead) else 1
        return rot.reshape(B, N, 3, 3).contiguous(), trans.reshape(B, N, 3).contiguous(), lead

    @staticmethod
    def _mask_of(diffuse_mask, B, N, dev):
        if diffuse_mask is None:
            return None
        m = diffuse_mask.detach().cpu().numpy() if torch.is_tensor(diffuse_mask) else np.asarray(diffuse_mask)
        return torch.as_tensor(np.broadcast_to(m.astype(np.float32).reshape(-1, N), (B, N)).copy(), device=dev)

    def forward_device(self, rot_1, trans_1, diffuse_mask, z_rot, z_trans, t_1, dt, noise_scale=1.0, rot_out=None, trans_out=None,
                       rigids_out=None):
        """One-step forward noising on device state (rot [B,N,3,3], trans [B,N,3] float32; noise N(0,1) float64)."""
        lib = _lib.load()
        _lib.require_cuda(rot_1, "forward")
        B, N = rot_1.shape[0], rot_1.shape[1]
        rot_out = torch.empty_like(rot_1) if rot_out is None else rot_out
        trans_out = torch.empty_like(trans_1) if trans_out is None else trans_out
        with torch.cuda.device(rot_1.device):
            _lib.check(lib.fdipt_se3_forward_step(B, N, _lib.ptr(rot_1), _lib.ptr(trans_1), _lib.ptr(diffuse_mask), _lib.ptr(z_rot),
                                                  _lib.ptr(z_trans), float(t_1), float(dt), float(noise_scale), *self._consts(),
                                                  _lib.ptr(rot_out), _lib.ptr(trans_out), _lib.ptr(rigids_out), _lib.stream_ptr()),
                       "se3_forward_step")
        return rot_out, trans_out

    def forward(self, rigids_t_1: Rigid, t_1: float, dt: float, diffuse_mask=None, chain_indices=None) -> Rigid:
        """se3_diffuser.py:50-95: samples q(x_t | x_{t-1}).  Noise from the global ``np.random`` stream in the reference order
        (R^3 draw, then SO(3) draw)."""
        rot, trans, lead = self._state_of(rigids_t_1)
        B, N = rot.shape[:2]
        z_trans = np.random.normal(size=(*lead, N, 3))
        z_rot = np.random.normal(size=(*lead, N, 3))
        dev = rot.device
        ro, to = self.forward_device(rot, trans, self._mask_of(diffuse_mask, B, N, dev),
                                     torch.as_tensor(z_rot, device=dev).reshape(B, N, 3), torch.as_tensor(z_trans, device=dev).reshape(B, N, 3),
                                     t_1, dt)
        return Rigid(Rotation(rot_mats=ro.reshape(*lead, N, 3, 3)), to.reshape(*lead, N, 3))

    def step_log_prob_device(self, rot_t, trans_t, rot_1, trans_1, rot_score, trans_score, diffuse_mask, t, t_1, dt, out=None):
        """[B,4] float64: log p_trans, log p_rot (backward, needs the scores at t), log q_trans, log q_rot (forward) of one step."""
        lib = _lib.load()
        B, N = rot_t.shape[0], rot_t.shape[1]
        out = torch.zeros(B, 4, dtype=torch.float64, device=rot_t.device) if out is None else out
        with torch.cuda.device(rot_t.device):
            _lib.check(lib.fdipt_se3_step_log_prob(B, N, _lib.ptr(rot_t), _lib.ptr(trans_t), _lib.ptr(rot_1), _lib.ptr(trans_1),
                                                   _lib.ptr(rot_score), _lib.ptr(trans_score), _lib.ptr(diffuse_mask), float(t), float(t_1),
                                                   float(dt), *self._consts(), _lib.ptr(out), _lib.stream_ptr()), "se3_step_log_prob")
        return out

    def _log_prob(self, rigids_t, rigids_t_1, rot_score, trans_score, t, t_1, dt, diffuse_mask, cols):
        rot_t, trans_t, lead = self._state_of(rigids_t)
        rot_1, trans_1, _ = self._state_of(rigids_t_1)
        B, N = rot_t.shape[:2]
        dev = rot_t.device
        if rot_score is not None:
            rot_score = torch.as_tensor(np.asarray(rot_score.detach().cpu() if torch.is_tensor(rot_score) else rot_score),
                                        dtype=torch.float64).reshape(B, N, 3).to(dev).contiguous()
            trans_score = torch.as_tensor(np.asarray(trans_score.detach().cpu() if torch.is_tensor(trans_score) else trans_score),
                                          dtype=torch.float32).reshape(B, N, 3).to(dev).contiguous()
        out = self.step_log_prob_device(rot_t, trans_t, rot_1, trans_1, rot_score, trans_score, self._mask_of(diffuse_mask, B, N, dev),
                                        t, t_1, dt).cpu().numpy()
        lp = out[:, cols[0]] + out[:, cols[1]]
        return float(lp[0]) if not len(lead) else lp.reshape(lead)

    def log_prob_forward(self, rigids_t: Rigid, rigids_t_1: Rigid, t_1: float, dt: float, diffuse_mask=None, chain_indices=None):
        """se3_diffuser.py:97-144: log q(x_t | x_{t-1}) summed over the diffused residues (per sample for batched input)."""
        return self._log_prob(rigids_t, rigids_t_1, None, None, t_1, t_1, dt, diffuse_mask, (2, 3))

    def log_prob_backward(self, rigids_t: Rigid, rigids_t_1: Rigid, trans_score_t, rot_score_t, t: float, dt: float, diffuse_mask,
                          chain_indices=None):
        """se3_diffuser.py:146-196: log p(x_{t-1} | x_t) under the scores at time t."""
        return self._log_prob(rigids_t, rigids_t_1, rot_score_t, trans_score_t, t, t, dt, diffuse_mask, (0, 1))

    # ------------------------------------------------------------------ x_T
    def sample_ref(self, n_samples: int, chain_index=None, impute: Rigid | None = None, diffuse_mask=None,
                   as_tensor_7: bool = False):
        """se3_diffuser.py:455-529."""
        dev = _dev(self._device if impute is None else impute.device)
        if impute is None:
            if not self._diffuse_rot:
                raise ValueError("Must provide impute values as we're not diffusing rotations!")
            if not self._diffuse_trans:
                raise ValueError("Must provide impute values as we're not diffusing translations!")
            if diffuse_mask is not None:
                raise ValueError("Must provide imputation values for unmasked regions!")
            trans_impute = np.zeros((n_samples, 3), dtype=np.float32)
            rot_impute = np.zeros((n_samples, 3))
        else:
            if impute.shape[0] != n_samples:
                raise ValueError(f"impute should have shape ({n_samples}, ...), got {impute.shape}.")
            trans_impute, rot_impute = _extract_trans_rots(impute)
            trans_impute = trans_impute.reshape((n_samples, 3))
            rot_impute = rot_impute.reshape((n_samples, 3))
        rot_ref = self._so3_diffuser.sample_ref(n_samples=n_samples) if self._diffuse_rot else rot_impute
        if self._diffuse_trans:
            trans_ref = self._r3_diffuser.sample_stationary_distribution(trans_impute, diffuse_mask, chain_index)
        else:
            trans_ref = trans_impute
        if diffuse_mask is not None:
            rot_ref = self._apply_mask(rot_ref, rot_impute, np.asarray(diffuse_mask)[..., None])
        rigids_t = _assemble_rigid(rot_ref, trans_ref, dev)
        if as_tensor_7:
            rigids_t = rigids_t.to_tensor_7()
        return {"rigids_t": rigids_t}
