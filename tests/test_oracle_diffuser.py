"""Oracle pinning (CPU): SE(3) diffuser restatement vs reference goldens."""
import numpy as np

from conftest import load_golden
from framedipt_amd import config
from oracle import diffuser as od
from oracle import frames as fr

G = load_golden("ops.npz")
X = load_golden("xT.npz")


def _diff():
    return od.SE3Diffuser(config.base_config().diffuser)


def test_schedules_and_scalings():
    d = _diff()
    so3, r3 = d._so3_diffuser, d._r3_diffuser
    ts = G["ts"]
    np.testing.assert_allclose([so3.sigma(t) for t in ts], G["so3_sigma"], rtol=1e-14)
    np.testing.assert_allclose([so3.diffusion_coef(t) for t in ts], G["so3_g"], rtol=1e-14)
    np.testing.assert_array_equal([so3.t_to_idx(t) for t in ts], G["so3_idx"])
    np.testing.assert_allclose([so3.score_scaling(t) for t in ts], G["so3_score_scaling"], rtol=1e-10)
    np.testing.assert_allclose([r3.score_scaling(t) for t in ts], G["r3_score_scaling"], rtol=1e-14)
    np.testing.assert_allclose(so3._row(so3.t_to_idx(1.0))[1], G["cdf_t1"], rtol=1e-12)


def test_scores():
    d = _diff()
    q1, q0 = G["q1"][None], G["rot_score_q0"][None]
    for i, t in enumerate([0.01, 0.5, 1.0]):
        rs = d.calc_rot_score(q1, q0, np.array([t], dtype=np.float32))[0]
        ref = G[f"rot_score_{i}"]
        # The reference evaluates sin((l+1/2)w) in float32 (so3_diffuser.py:68-77,180-191): where the true
        # series value f underflows (w >> sigma) its score is float32 round-off divided by the 1e-4
        # regulariser, i.e. implementation-defined noise.  Parity is asserted where f is conditioned.
        rv = fr.quat_to_rotvec(fr.quat_multiply(fr.invert_quat(q0), q1).astype(np.float32))[0]
        sig = d._so3_diffuser.discrete_sigma[d._so3_diffuser.t_to_idx(np.float64(np.float32(t)))]
        f = od.igso3_expansion_np(np.linalg.norm(rv, axis=-1).astype(np.float64), sig)
        ok = f > 1e-2
        assert ok.sum() >= 3
        err = np.abs(rs - ref).max(-1)
        mag = np.abs(ref).max(-1)
        assert (err[ok] <= 2e-3 * mag[ok] + 1e-5).all(), (i, err[ok], mag[ok])
        assert np.abs(rs[~ok]).max(initial=0) < 50 and np.abs(ref[~ok]).max(initial=0) < 50
        ts = d.calc_trans_score(G["t1"][None], G["t2"][None], np.array([t], dtype=np.float32)[:, None, None])[0]
        np.testing.assert_allclose(ts, G[f"trans_score_{i}"], rtol=2e-6, atol=1e-6)


def test_sample_ref_stream():
    d = _diff()  # seeds np.random (so3 then r3), like the reference constructor
    rot, trans = d.sample_ref(50)
    t7 = X["denovo_t7"]
    np.testing.assert_allclose(fr.quat_to_rot(fr.rot_to_quat(rot)), fr.quat_to_rot(t7[:, :4]), atol=3e-6)
    np.testing.assert_allclose(trans.astype(np.float32), t7[:, 4:], atol=1e-6)
    imp = X["imp_t7"].astype(np.float32)
    rot, trans = d.sample_ref(30, impute=(imp[:, :4], imp[:, 4:]), diffuse_mask=X["imp_mask"])
    np.testing.assert_allclose(rot, X["inpaint_rot"], atol=1e-6)
    np.testing.assert_allclose(trans.astype(np.float32), X["inpaint_trans"], atol=1e-5)


def test_reverse_steps_teacher_forced():
    d = _diff()
    for name in ("traj_small_denovo_n16_T10.npz", "traj_small_inpaint_n24_T10.npz", "traj_full_denovo_n64_T20.npz"):
        T = load_golden(name)
        dm = (1 - T["in_fixed_mask"]) * T["in_res_mask"]
        dt = 1.0 / int(T["num_t"])
        for s in range(len(T["step_t"])):
            rig = T["step_rigids_t"][s]
            rot, tr = d.reverse(rig[..., :4], rig[..., 4:], T["step_rot_score"][s], T["step_trans_score"][s],
                                float(T["step_t"][s]), dt, diffuse_mask=dm, noise_scale=float(T["noise_scale"]),
                                z_rot=T["noise_tape"][2 * s], z_trans=T["noise_tape"][2 * s + 1], orthogonalize=True)
            np.testing.assert_allclose(rot, T["step_out_rot"][s], atol=2e-7)
            np.testing.assert_allclose(tr, T["step_out_trans"][s], atol=2e-5)
