"""EigenFold confidence score: ``experiments/utils.py:752-869 logp_confidence_score``.

The walk x_0 -> x_T by one-step forward noising with log p(x_{t-1} | x_t) - log q(x_t | x_{t-1}) summed along the way is the
second consumer of the sampler's kernels: every step is the forward-noising launch (``fdipt_se3_forward_step``), two
score-network forwards (self-conditioning + the scoring one, as ``one_step_inference_score`` :251-289 does) and the log-probability
launch (``fdipt_se3_step_log_prob``).  Nothing crosses to the host inside the loop: the per-step sums land in a [T,B,4] float64
device buffer that is read once at the end (the reference converts frames and scores to NumPy at every step).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .model.score_network import preprocess_aatype
from .rigid import Rigid, quat_to_rot


def draw_forward_noise_tape(n_steps: int, B: int, N: int):
    """Noise of the forward steps from the global ``np.random`` stream in the reference's order (per step: R^3 draw
    ``r3_diffuser.py:146`` then SO(3) draw ``so3_diffuser.py:431``)."""
    z_rot = np.zeros((n_steps, B, N, 3))
    z_trans = np.zeros((n_steps, B, N, 3))
    for s in range(n_steps):
        z_trans[s] = np.random.normal(size=(B, N, 3))
        z_rot[s] = np.random.normal(size=(B, N, 3))
    return z_rot, z_trans


class ConfidenceLoop:
    """Device-resident state of the forward walk for a batch of B structures of equal length."""

    def __init__(self, model, diffuser, rigids_0, sample_feats, diffuse_mask, num_t, min_t, self_condition=True, noise_tape=None,
                 state=None):
        self.model, self.diffuser = model, diffuser
        dev = self.dev = model.device
        t7 = rigids_0.to_tensor_7() if isinstance(rigids_0, Rigid) else torch.as_tensor(rigids_0)
        self.batched = t7.dim() == 3
        t7 = t7.reshape(-1, t7.shape[-2], 7).to(device=dev, dtype=torch.float32).contiguous()
        B, N = self.B, self.N = t7.shape[0], t7.shape[1]
        f32 = lambda x: x.to(device=dev, dtype=torch.float32).reshape(B, N, *x.shape[2:]).contiguous().clone()  # noqa: E731
        self.res_mask, self.fixed = f32(sample_feats["res_mask"]), f32(sample_feats["fixed_mask"])
        self.mask = diffuser._mask_of(diffuse_mask, B, N, dev)
        aat = sample_feats.get("aatype")
        aat = None if aat is None else aat.to(dev).reshape(B, N)
        self.net_aatype = preprocess_aatype(aat, self.fixed, model.inpainting, model._model_conf.input_aatype)
        if self.net_aatype is not None:
            self.net_aatype = self.net_aatype.to(device=dev, dtype=torch.int32).contiguous()
        self.gt_psi = f32(sample_feats["torsion_angles_sin_cos"][..., 2, :])
        self.sc_ca = f32(sample_feats["sc_ca_t"])
        self.st = state if state is not None else model.batch_state(sample_feats["seq_idx"])
        self.num_t, self.min_t, self.dt, self.self_condition = num_t, min_t, 1 / num_t, self_condition
        self.forward_steps = np.linspace(min_t, 1.0, num_t)[:-1]                      # t_1 of every step
        self.t_model = np.append(self.forward_steps[1:], 1.0)                         # time of the scores (last one exactly 1.0)
        n = len(self.forward_steps)
        t32, temb, sig = model.step_scalars(self.t_model)
        with torch.cuda.device(dev):
            self.t_all = torch.as_tensor(np.repeat(t32[:, None], B, 1), device=dev)
            self.temb_all = torch.as_tensor(np.repeat(temb[:, None, :], B, 1), device=dev)
            self.sig_all = torch.as_tensor(np.repeat(sig[:, None], B, 1), device=dev)
            if noise_tape is None:
                noise_tape = draw_forward_noise_tape(n, B, N)
            self.z_rot = torch.as_tensor(np.ascontiguousarray(noise_tape[0], dtype=np.float64), device=dev)
            self.z_trans = torch.as_tensor(np.ascontiguousarray(noise_tape[1], dtype=np.float64), device=dev)
            # the reference's state: float32 rotation matrices of Rigid.from_tensor_7(rigids_0) + float32 translations
            self.rot = [quat_to_rot(t7[..., :4]).contiguous(), torch.empty(B, N, 3, 3, device=dev)]
            self.trans = [t7[..., 4:].contiguous(), torch.empty(B, N, 3, device=dev)]
            self.rigids = torch.empty(B, N, 7, device=dev)
            self.terms = torch.zeros(n, B, 4, dtype=torch.float64, device=dev)
            self.prior = torch.zeros(B, 2, dtype=torch.float64, device=dev)

    def step(self, i):
        st, d = self.st, self.diffuser
        cur, nxt = i & 1, (i + 1) & 1
        with torch.cuda.device(self.dev):
            d.forward_device(self.rot[cur], self.trans[cur], self.mask, self.z_rot[i], self.z_trans[i], self.forward_steps[i], self.dt,
                             rot_out=self.rot[nxt], trans_out=self.trans[nxt], rigids_out=self.rigids)
            args = (self.rigids, self.res_mask, self.fixed, self.sc_ca, self.net_aatype, self.gt_psi, self.t_all[i], self.temb_all[i],
                    self.sig_all[i])
            if self.self_condition:  # self_conditioning(): sc_ca_t <- CA of the prediction, then the scoring forward
                st.forward(*args, False, ca_out=self.sc_ca)
            st.forward(*args, False)
            d.step_log_prob_device(self.rot[nxt], self.trans[nxt], self.rot[cur], self.trans[cur], st.rot_score, st.trans_score,
                                   self.mask, self.t_model[i], self.forward_steps[i], self.dt, out=self.terms[i])

    def finish(self):
        """-> (log_prob [B], log_probs [T,B]) as NumPy float64."""
        lib = _lib.load()
        last = len(self.forward_steps) & 1
        with torch.cuda.device(self.dev):
            _lib.check(lib.fdipt_se3_prior_log_prob(self.B, self.N, _lib.ptr(self.trans[last]), _lib.ptr(self.mask),
                                                    float(self.diffuser._r3_diffuser._r3_conf.coordinate_scaling), _lib.ptr(self.prior),
                                                    _lib.stream_ptr()), "se3_prior_log_prob")
        terms, prior = self.terms.cpu().numpy(), self.prior.cpu().numpy()
        # per step: log_prob += backward (trans + rot); log_prob -= forward (trans + rot)   (utils.py:826-844)
        run = np.zeros(self.B)
        log_probs = []
        for i in range(terms.shape[0]):
            run = run + (terms[i, :, 0] + terms[i, :, 1])
            run = run - (terms[i, :, 2] + terms[i, :, 3])
            log_probs.append(run.copy())
        run = run + (prior[:, 0] + prior[:, 1])
        log_probs.append(run.copy())
        return run, np.stack(log_probs)


def logp_confidence_score(model, diffuser, rigids_t, sample_feats, diffuse_mask, num_t, min_t, device=None, self_condition=True,
                          noise_tape=None):
    """Same arguments / return as the reference: ``(log_prob, log_probs)`` - a float and a list of ``num_t`` floats for one
    structure (``rigids_t`` shaped [N]); arrays [B] / [num_t, B] when ``rigids_t`` carries a batch dimension.  ``noise_tape=(z_rot,
    z_trans)`` ([num_t-1,B,N,3] N(0,1)) replaces the draws from the global ``np.random`` stream."""
    loop = ConfidenceLoop(model, diffuser, rigids_t, sample_feats, diffuse_mask, num_t, min_t, self_condition, noise_tape)
    for i in range(len(loop.forward_steps)):
        loop.step(i)
    log_prob, log_probs = loop.finish()
    if loop.batched:
        return log_prob, log_probs
    return float(log_prob[0]), [float(x) for x in log_probs[:, 0]]
