"""Reverse-diffusion driver with the signature of ``experiments/utils.py:inference_fn`` (:511-626).

Python drives the loop; every step is two C-ABI calls (score-network forward, fused SE(3) reverse step) plus the
backbone-atom kernel, all enqueued on one HIP stream.  Nothing crosses to the host inside the loop: the state
(x_t, self-conditioning CA, trajectories) stays in HBM and the noise tape / per-step scalars are uploaded up front
(the reference does >= 5 device->host syncs per step, SURVEY.md section 0 finding 3).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .model.score_network import preprocess_aatype


def draw_noise_tape(diffuser, n_steps: int, B: int, N: int):
    """Noise of the reverse steps from the global legacy ``np.random`` stream in the reference's order
    (per step: SO(3) draw ``so3_diffuser.py:590`` then R^3 draw ``r3_diffuser.py:371``, each [B,N,3])."""
    z_rot = np.zeros((n_steps, B, N, 3))
    z_trans = np.zeros((n_steps, B, N, 3))
    for s in range(n_steps):
        if diffuser._diffuse_rot:
            z_rot[s] = np.random.normal(size=(B, N, 3))
        if diffuser._diffuse_trans:
            z_trans[s] = np.random.normal(size=(B, N, 3))
    return z_rot, z_trans


def _backbone(net, n, t7, rot, trans, psi, aatype, atom37):
    lib = _lib.load()
    _lib.check(lib.fdipt_backbone_atoms(n, _lib.ptr(t7), _lib.ptr(rot), _lib.ptr(trans), _lib.ptr(psi), _lib.ptr(aatype),
                                        _lib.ptr(net.bb_tables), _lib.ptr(atom37), None, _lib.stream_ptr()),
               "backbone_atoms")


def inference_fn(model, diffuser, data_init, num_t, min_t, center=True, aux_traj=False, self_condition=True,
                 noise_scale=1.0, embed_self_conditioning=True, inpainting=False, input_aatype=False, noise_tape=None,
                 return_device=False):
    """Same arguments / returned keys as the reference.  ``noise_tape=(z_rot, z_trans)`` ([num_t-1,B,N,3] float64
    N(0,1) draws) overrides the global ``np.random`` stream (sample-sharded runs).  ``data_init`` tensors carry a
    leading batch dimension B >= 1 (the reference always passes B = 1)."""
    dev = model.device
    rig0 = data_init["rigids_t"]
    _lib.require_cuda(rig0, "inference_fn")
    if rig0.dim() == 2:
        raise ValueError("rigids_t needs a leading batch dimension")
    B, N = rig0.shape[0], rig0.shape[1]
    f32 = lambda x: x.to(device=dev, dtype=torch.float32).contiguous().clone()  # noqa: E731
    res_mask, fixed = f32(data_init["res_mask"]), f32(data_init["fixed_mask"])
    fixed_mask = fixed * res_mask
    diffuse_mask = ((1 - fixed) * res_mask).contiguous()
    aatype = preprocess_aatype(data_init.get("aatype"), fixed, inpainting, input_aatype)
    aatype_dev = None if aatype is None else aatype.to(device=dev, dtype=torch.int32).contiguous()
    gt_tors = data_init["torsion_angles_sin_cos"]
    gt_psi = f32(gt_tors[..., 2, :])
    st = model.batch_state(data_init["seq_idx"])

    reverse_steps = np.linspace(min_t, 1.0, num_t)[::-1]
    dt = 1 / num_t
    n_noisy = int(np.sum(reverse_steps > min_t))
    t32, temb, sig = model.step_scalars(reverse_steps)
    with torch.cuda.device(dev):
        t_all = torch.as_tensor(np.repeat(t32[:, None], B, 1), device=dev)
        temb_all = torch.as_tensor(np.repeat(temb[:, None, :], B, 1), device=dev)
        sig_all = torch.as_tensor(np.repeat(sig[:, None], B, 1), device=dev)
        if noise_tape is None:
            noise_tape = draw_noise_tape(diffuser, n_noisy, B, N)
        z_rot = torch.as_tensor(np.ascontiguousarray(noise_tape[0], dtype=np.float64), device=dev)
        z_trans = torch.as_tensor(np.ascontiguousarray(noise_tape[1], dtype=np.float64), device=dev)
        rigids_t = f32(rig0)
        sc_ca = f32(data_init["sc_ca_t"])
        rigid_traj = torch.empty(num_t + 1, B, N, 7, device=dev)
        rigid_traj[0] = rigids_t
        prot_traj = torch.empty(num_t, B, N, 37, 3, device=dev)
        bb0_traj = torch.empty(num_t, B, N, 37, 3, device=dev) if aux_traj else None
        trans_traj = torch.empty(num_t, B, N, 3, device=dev) if aux_traj else None
        rot_out = torch.empty(B, N, 3, 3, device=dev)
        trans_c = torch.empty(B, N, 3, device=dev)

        def fwd(k, want_atoms):
            st.forward(rigids_t, res_mask, fixed, sc_ca, aatype_dev, gt_psi, t_all[k], temb_all[k], sig_all[k], want_atoms)

        if embed_self_conditioning and self_condition:  # priming call, utils.py:571-578
            fwd(0, False)
            sc_ca.copy_(st.rigids[..., 4:])
        noisy = 0
        for k, t in enumerate(reverse_steps):
            fwd(k, aux_traj)
            if t > min_t:
                if embed_self_conditioning:
                    sc_ca.copy_(st.rigids[..., 4:])
                nxt = rigid_traj[k + 1]
                diffuser.reverse_device(rigids_t, st.rot_score, st.trans_score, diffuse_mask, z_rot[noisy], z_trans[noisy],
                                        t, dt, center, noise_scale, rigids_out=nxt, rot_out=rot_out)
                noisy += 1
                rigids_t.copy_(nxt)
                trans_c.copy_(nxt[..., 4:])
                _backbone(model, B * N, None, rot_out, trans_c, st.psi, aatype_dev, prot_traj[k])
            else:  # last step: take the x_0 prediction, utils.py:373-374
                rigid_traj[k + 1] = st.rigids
                rigids_t.copy_(st.rigids)
                _backbone(model, B * N, rigids_t, None, None, st.psi, aatype_dev, prot_traj[k])
            if aux_traj:
                bb0_traj[k] = st.atom37
                trans_traj[k] = diffuse_mask[..., None] * st.rigids[..., 4:] + fixed_mask[..., None] * rigids_t[..., 4:]
        psi_pred = st.psi.to(gt_tors.dtype).clone() if gt_tors.dtype == torch.float64 else st.psi.clone()
        conv = (lambda x: torch.flip(x, (0,))) if return_device else (lambda x: np.flip(x.cpu().numpy(), (0,)))
        ret = {"prot_traj": conv(prot_traj)}
        if aux_traj:
            ret["rigid_traj"] = conv(rigid_traj)
            ret["trans_traj"] = conv(trans_traj)
            ret["psi_pred"] = psi_pred[None]
            ret["rigid_0_traj"] = conv(bb0_traj)
        else:
            ret["rigid_traj_final"] = rigid_traj[-1] if return_device else rigid_traj[-1].cpu().numpy()
    return ret
