"""Generate the committed golden fixtures by importing the reference (this container only).

    python tests/golden/make_goldens.py            # writes tests/golden/*.npz

The reference ships no checkpoint and no numeric tests for this path
(SURVEY.md section 0 finding 6, section 8c), so parity is pinned by the vectors captured
here from the reference *source* running on torch 2.10 / numpy 2.2 / scipy 1.15.
Weights are never stored: they are regenerated from
``framedipt_amd.weights.synth_state_dict`` (seed recorded in each fixture).
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refharness as rh  # noqa: E402

rh.install_stubs()

import torch  # noqa: E402

from framedipt_amd import weights as W  # noqa: E402

torch.set_num_threads(8)

from experiments import utils as exp_utils  # noqa: E402
from framedipt.data import transforms  # noqa: E402
from framedipt.diffusion import se3_diffuser  # noqa: E402
from framedipt.model import score_network  # noqa: E402
from framedipt.protein import all_atom  # noqa: E402
from openfold.utils import rigid_utils as ru  # noqa: E402

WEIGHT_SEED = 7


def np32(x):
    return x.detach().cpu().numpy()


def build(cfg, inpainting=False, bb_gain=W.BB_GAIN):
    diff = se3_diffuser.SE3Diffuser(cfg.diffuser)
    model = score_network.ScoreNetwork(cfg.model, diff, inpainting=inpainting)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = W.synth_state_dict(shapes, WEIGHT_SEED, bb_gain)
    model.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    model.eval()
    return diff, model, shapes


def residue_tables():
    out = {
        "default_frames": all_atom.DEFAULT_FRAMES.numpy().astype(np.float32),
        "group_idx": all_atom.GROUP_IDX.numpy().astype(np.int32),
        "atom_mask": all_atom.ATOM_MASK.numpy().astype(np.float32),
        "ideal_pos": all_atom.IDEALIZED_POS.numpy().astype(np.float32),
    }
    os.makedirs(os.path.join(ROOT, "framedipt_amd", "data"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz"), **out)


def rand_quats(rng, n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    # edge cases: identity, angle ~ 0, angle ~ pi, negative w
    q[0] = [1, 0, 0, 0]
    ax = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    for i, ang in enumerate([1e-7, 1e-4, 2e-3, np.pi - 1e-4, np.pi - 1e-7, np.pi]):
        q[1 + i] = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    q[7] = -q[7]
    q[8, 0] = -abs(q[8, 0])
    return q.astype(np.float32)


def ops_goldens(diff):
    rng = np.random.default_rng(11)
    g = {}
    n = 40
    q1, q2 = rand_quats(rng, n), rand_quats(rng, n)[::-1].copy()
    t1, t2 = rng.standard_normal((n, 3)).astype(np.float32) * 5, rng.standard_normal((n, 3)).astype(np.float32) * 5
    pts = rng.standard_normal((n, 3)).astype(np.float32) * 3
    vec = rng.standard_normal((n, 3)).astype(np.float32)
    upd = (rng.standard_normal((n, 6)) * 0.5).astype(np.float32)
    mask = (rng.random((n, 1)) > 0.3).astype(np.float32)
    tq1, tq2, tt1, tt2 = map(torch.tensor, (q1, q2, t1, t2))
    g.update(q1=q1, q2=q2, t1=t1, t2=t2, pts=pts, vec=vec, upd=upd, mask=mask)
    g["quat_to_rot"] = np32(ru.quat_to_rot(tq1))
    g["rot_to_quat_rot"] = np32(ru.quat_to_rot(ru.rot_to_quat(ru.quat_to_rot(tq1))))
    g["quat_multiply"] = np32(ru.quat_multiply(tq1, tq2))
    g["quat_multiply_by_vec"] = np32(ru.quat_multiply_by_vec(tq1, torch.tensor(vec)))
    g["invert_quat"] = np32(ru.invert_quat(tq1))
    r1 = ru.Rigid.from_tensor_7(torch.cat([tq1, tt1], -1))
    r2 = ru.Rigid.from_tensor_7(torch.cat([tq2, tt2], -1))
    g["apply"] = np32(r1.apply(torch.tensor(pts)))
    g["invert_apply"] = np32(r1.invert_apply(torch.tensor(pts)))
    c = r1.compose(r2)
    g["compose_rot"], g["compose_trans"] = np32(c.get_rots().get_rot_mats()), np32(c.get_trans())
    iv = r1.invert()
    g["invert_rot"], g["invert_trans"] = np32(iv.get_rots().get_rot_mats()), np32(iv.get_trans())
    cu = r1.compose_q_update_vec(torch.tensor(upd), torch.tensor(mask))
    g["cqu_quat"], g["cqu_trans"] = np32(cu.get_rots().get_quats()), np32(cu.get_trans())
    g["quat_to_rotvec"] = np32(transforms.quat_to_rotvec(tq1))
    # SciPy conventions on the path
    rv1 = (rng.standard_normal((n, 3)) * 1.2)
    rv2 = (rng.standard_normal((n, 3)) * 0.3)
    rv1[0], rv1[1], rv2[2] = 0, [1e-5, 0, 0], [0, 1e-9, 0]
    rv1[3] = np.array([0.0, 0.0, 1.0]) * (np.pi - 1e-6)
    g["rv1"], g["rv2"] = rv1, rv2
    g["compose_rotvec"] = transforms.compose_rotvec(rv1, rv2)
    g["rotvec_to_matrix"] = transforms.rotvec_to_matrix(rv1)
    tr, rv = se3_diffuser._extract_trans_rots(r1)
    g["extract_rotvec"], g["extract_trans"] = rv, tr
    asm = se3_diffuser._assemble_rigid(rv1, t1.astype(np.float64))
    g["assemble_rot"], g["assemble_t7_rot"] = np32(asm.get_rots().get_rot_mats()), np32(
        ru.quat_to_rot(asm.to_tensor_7()[..., :4]))
    # schedules, scores
    ts = np.array([0.01, 0.0298, 0.5, 0.77, 1.0])
    so3, r3 = diff._so3_diffuser, diff._r3_diffuser
    g["ts"] = ts
    g["so3_sigma"] = np.array([so3.sigma(t) for t in ts])
    g["so3_g"] = np.array([so3.diffusion_coef(t) for t in ts])
    g["so3_idx"] = np.array([so3.t_to_idx(t) for t in ts])
    g["so3_score_scaling"] = np.array([so3.score_scaling(t) for t in ts])
    g["r3_score_scaling"] = np.array([r3.score_scaling(t) for t in ts])
    g["cdf_t1"] = so3._cdf[so3.t_to_idx(1.0)]
    q0 = q2.copy()  # first 24: small relative rotations (conditioned IGSO3 regime), rest: arbitrary
    q0[:24] = q1[:24] + (rng.standard_normal((24, 4)) * np.linspace(0.005, 0.2, 24)[:, None]).astype(np.float32)
    rq0 = ru.Rotation(quats=torch.tensor(q0)[None], normalize_quats=True)
    rqt = ru.Rotation(quats=tq1[None], normalize_quats=False)
    g["rot_score_q0"] = np32(rq0.get_quats())[0]
    for i, t in enumerate([0.01, 0.5, 1.0]):
        tt = torch.tensor([t], dtype=torch.float32)
        g[f"rot_score_{i}"] = np32(diff.calc_rot_score(rqt, rq0, tt))[0]
        g[f"trans_score_{i}"] = np32(diff.calc_trans_score(tt1[None], tt2[None], tt[:, None, None], use_torch=True))[0]
    # embedding constants as torch evaluates them (score_network.py:17-64)
    import math
    g["timestep_freqs"] = torch.exp(torch.arange(16, dtype=torch.float32) * -(math.log(10000) / 15)).numpy()
    g["index_denoms"] = (2056 ** (2 * torch.arange(16)[None] / 32)).numpy()[0]
    tt = torch.tensor([0.01, 0.123, 0.7, 1.0], dtype=torch.float32)
    g["temb_t"], g["temb"] = tt.numpy(), score_network.get_timestep_embedding(tt, 32).numpy()
    ii = torch.tensor([[-499, -3, 0, 1, 17, 300, 1234]])
    g["iemb_i"], g["iemb"] = ii.numpy(), score_network.get_index_embedding(ii, 32).numpy()
    # backbone with aatype
    aatype = torch.tensor(rng.integers(0, 21, size=(1, n)))
    psi = rng.standard_normal((1, n, 2)).astype(np.float32)
    psi /= np.linalg.norm(psi, axis=-1, keepdims=True)
    a37, m37, _, a14 = all_atom.compute_backbone(r1[None], torch.tensor(psi), aatype=aatype)
    g.update(bb_aatype=aatype.numpy(), bb_psi=psi, bb_atom37=np32(a37), bb_atom14=np32(a14))
    a37, _, _, a14 = all_atom.compute_backbone(r1[None], torch.tensor(psi), aatype=None)
    g.update(bb_atom37_none=np32(a37), bb_atom14_none=np32(a14))
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **g)


def xT_goldens(cfg):
    g = {}
    diff = se3_diffuser.SE3Diffuser(cfg.diffuser)  # reseeds np.random with 123 (twice)
    with torch.no_grad():
        g["denovo_t7"] = np32(diff.sample_ref(n_samples=50, as_tensor_7=True)["rigids_t"])
        rng = np.random.default_rng(5)
        q = rand_quats(rng, 30)
        tr = (rng.standard_normal((30, 3)) * 8).astype(np.float32)
        imp = ru.Rigid.from_tensor_7(torch.tensor(np.concatenate([q, tr], -1)))
        dm = np.zeros(30)
        dm[5:12] = 1
        dm[20:24] = 1
        g.update(imp_t7=np.concatenate([q, tr], -1), imp_mask=dm)
        r = diff.sample_ref(n_samples=30, impute=imp, diffuse_mask=dm, as_tensor_7=False)["rigids_t"]
        g["inpaint_rot"], g["inpaint_trans"] = np32(r.get_rots().get_rot_mats()), np32(r.get_trans())
    np.savez_compressed(os.path.join(HERE, "xT.npz"), **g)


def make_feats(n, rng, inpainting, diff):
    if not inpainting:
        with torch.no_grad():
            ref = diff.sample_ref(n_samples=n, as_tensor_7=True)
        return {
            "res_mask": torch.ones(1, n, dtype=torch.float64),
            "seq_idx": torch.arange(1, n + 1)[None],
            "fixed_mask": torch.zeros(1, n, dtype=torch.float64),
            "torsion_angles_sin_cos": torch.zeros(1, n, 7, 2, dtype=torch.float64),
            "sc_ca_t": torch.zeros(1, n, 3, dtype=torch.float64),
            "rigids_t": ref["rigids_t"][None],
        }
    # synthetic 2-chain complex with a diffused window per chain
    q = rand_quats(rng, n)
    tr = np.cumsum(rng.standard_normal((n, 3)) * 2.2, axis=0).astype(np.float32)
    tr -= tr.mean(0)
    gt = ru.Rigid.from_tensor_7(torch.tensor(np.concatenate([q, tr], -1)))
    dm = np.zeros(n)
    dm[3:8] = 1
    dm[n - 7:n - 3] = 1
    n1 = n // 2
    seq_idx = np.concatenate([np.arange(n1), np.arange(n - n1) + n1 + 200])
    chain_idx = np.concatenate([np.zeros(n1), np.ones(n - n1)]).astype(np.int64)
    with torch.no_grad():
        ref = diff.sample_ref(n_samples=n, impute=gt, diffuse_mask=dm, chain_index=chain_idx, as_tensor_7=True)
    tors = rng.standard_normal((1, n, 7, 2))
    tors /= np.linalg.norm(tors, axis=-1, keepdims=True)
    return {
        "res_mask": torch.ones(1, n, dtype=torch.float64),
        "seq_idx": torch.tensor(seq_idx)[None],
        "fixed_mask": torch.tensor(1 - dm)[None],
        "torsion_angles_sin_cos": torch.tensor(tors),
        "sc_ca_t": torch.zeros(1, n, 3, dtype=torch.float64),
        "rigids_t": ref["rigids_t"][None],
        "aatype": torch.tensor(rng.integers(0, 20, size=(1, n))),
        "chain_idx": torch.tensor(chain_idx)[None],
    }


def feats_np(f):
    return {"in_" + k: v.numpy() for k, v in f.items() if torch.is_tensor(v)}


def forward_golden(name, cfg, n, inpainting, t, sc_scale=0.0, trace_rows=(0, 3), bb_gain=W.BB_GAIN):
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    diff, model, shapes = build(cfg, inpainting, bb_gain)
    f = make_feats(n, rng, inpainting, diff)
    if sc_scale:
        f["sc_ca_t"] = torch.tensor(f["rigids_t"][..., 4:].numpy() + rng.standard_normal((1, n, 3)).astype(np.float32) * sc_scale)
    f["t"] = torch.tensor([t], dtype=torch.float32)
    g = feats_np(f)
    trace = {}
    trunk = model.score_model.trunk
    hooks = []

    def hk(key, fn=lambda o: o):
        def _h(_m, _i, o):
            trace[key] = np32(fn(o))
        return _h

    nb = cfg.model.ipa.num_blocks
    for b in range(nb):
        hooks.append(trunk[f"ipa_{b}"].register_forward_hook(hk(f"tr_ipa_{b}")))
        hooks.append(trunk[f"ipa_ln_{b}"].register_forward_hook(hk(f"tr_ipa_ln_{b}")))
        hooks.append(trunk[f"seq_tfmr_{b}"].register_forward_hook(hk(f"tr_tfmr_{b}")))
        hooks.append(trunk[f"node_transition_{b}"].register_forward_hook(hk(f"tr_node_{b}")))
        hooks.append(trunk[f"bb_update_{b}"].register_forward_hook(hk(f"tr_bbupd_{b}")))
        if b < nb - 1:
            rows = list(trace_rows)
            hooks.append(trunk[f"edge_transition_{b}"].register_forward_hook(hk(f"tr_edge_{b}", lambda o: o[:, rows])))
    hooks.append(model.embedding_layer.register_forward_hook(
        lambda _m, _i, o: trace.update(tr_node_init=np32(o[0]), tr_edge_init=np32(o[1][:, list(trace_rows)]))))
    with torch.no_grad():
        out = model(f)
    for h in hooks:
        h.remove()
    g.update({"out_" + k: np32(v) for k, v in out.items()})
    g.update(trace)
    g["trace_rows"] = np.array(trace_rows)
    g["weight_seed"] = WEIGHT_SEED
    g["bb_gain"] = bb_gain
    g["param_names"] = np.array(list(shapes.keys()))
    g["param_shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
    np.savez_compressed(os.path.join(HERE, f"fwd_{name}.npz"), **g)
    print(name, {k: float(np.abs(v).max()) for k, v in out.items()})


def traj_golden(name, cfg, n, inpainting, num_t, noise_scale=0.1, min_t=0.01):
    """Free-running reference trajectory with every per-step input captured (teacher forcing)."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000 + 1)
    diff, model, _ = build(cfg, inpainting)
    np.random.seed(123)  # model construction consumed the global stream (truncnorm init): restart it here
    f = make_feats(n, rng, inpainting, diff)
    g = feats_np(f)
    tape, steps = [], []
    orig_normal = np.random.normal

    def rec_normal(*a, **k):
        z = orig_normal(*a, **k)
        tape.append(z.copy())
        return z

    orig_reverse = diff.reverse

    def rec_reverse(rigid_t, rot_score, trans_score, t, dt, **kw):
        out = orig_reverse(rigid_t=rigid_t, rot_score=rot_score, trans_score=trans_score, t=t, dt=dt, **kw)
        steps.append(dict(t=t, rigids_t=np32(rigid_t.to_tensor_7()), rot_score=rot_score.copy(),
                          trans_score=trans_score.copy(), out_rot=np32(out.get_rots().get_rot_mats()),
                          out_trans=np32(out.get_trans())))
        return out

    sc_in = []
    orig_fwd = model.forward

    def rec_fwd(feats):
        sc_in.append(np32(feats["sc_ca_t"]).astype(np.float32))
        return orig_fwd(feats)

    model.forward = rec_fwd
    diff.reverse = rec_reverse
    np.random.normal = rec_normal
    try:
        res = exp_utils.inference_fn(model, diff, f, num_t=num_t, min_t=min_t, aux_traj=True, noise_scale=noise_scale,
                                     inpainting=inpainting, input_aatype=False)
    finally:
        np.random.normal = orig_normal
    g.update({"res_" + k: np.asarray(v).astype(np.float32) if k != "psi_pred" else np32(v) for k, v in res.items()})
    g["noise_tape"] = np.stack(tape)  # [2*(T-1), 1, N, 3] order: so3, r3 per step
    g["step_t"] = np.array([s["t"] for s in steps])
    for k in ("rigids_t", "rot_score", "trans_score", "out_rot", "out_trans"):
        g["step_" + k] = np.stack([s[k] for s in steps])
    g["sc_in"] = np.stack(sc_in)  # [T+1, 1, N, 3] sc_ca_t seen by each forward (priming first)
    g.update(num_t=num_t, min_t=min_t, noise_scale=noise_scale, weight_seed=WEIGHT_SEED, bb_gain=W.BB_GAIN)
    np.savez_compressed(os.path.join(HERE, f"traj_{name}.npz"), **g)
    print(name, "final CA span", np.abs(res["prot_traj"][0]).max())


def main():
    residue_tables()
    cfg = rh.load_cfg()
    diff = se3_diffuser.SE3Diffuser(cfg.diffuser)
    ops_goldens(diff)
    xT_goldens(cfg)
    small = rh.small_model_cfg(rh.load_cfg())
    small_inp = rh.small_model_cfg(rh.load_cfg(inpainting=True))
    forward_golden("small_denovo_n16", small, 16, False, t=0.7, sc_scale=1.0)
    forward_golden("small_inpaint_n24", small_inp, 24, True, t=0.31, sc_scale=1.5)
    forward_golden("small_denovo_n16_stress", small, 16, False, t=0.02, sc_scale=1.0, bb_gain=0.3)
    forward_golden("full_denovo_n64", rh.load_cfg(), 64, False, t=0.5, sc_scale=1.0)
    forward_golden("full_inpaint_n40", rh.load_cfg(inpainting=True), 40, True, t=0.05, sc_scale=0.5)
    traj_golden("small_denovo_n16_T10", small, 16, False, 10)
    traj_golden("small_inpaint_n24_T10", small_inp, 24, True, 10)
    traj_golden("full_denovo_n64_T20", rh.load_cfg(), 64, False, 20)


if __name__ == "__main__":
    main()
