"""Oracle pinning (CPU): frame algebra + SciPy conventions vs reference goldens and SciPy itself."""
import numpy as np
from scipy.spatial.transform import Rotation

from conftest import load_golden
from oracle import frames as fr
from oracle.score_network import compute_backbone

G = load_golden("ops.npz")


def test_quat_ops_match_reference():
    q1, q2 = G["q1"], G["q2"]
    np.testing.assert_allclose(fr.quat_to_rot(q1), G["quat_to_rot"], atol=1e-6)
    np.testing.assert_allclose(fr.quat_multiply(q1, q2), G["quat_multiply"], atol=1e-6)
    np.testing.assert_allclose(fr.quat_multiply_by_vec(q1, G["vec"]), G["quat_multiply_by_vec"], atol=1e-6)
    np.testing.assert_allclose(fr.invert_quat(q1), G["invert_quat"], atol=1e-6)
    # rot_to_quat: eigenvector sign is arbitrary -> compare through the rotation matrix
    r = fr.quat_to_rot(fr.rot_to_quat(fr.quat_to_rot(q1)))
    np.testing.assert_allclose(r, G["rot_to_quat_rot"], atol=2e-6)
    np.testing.assert_allclose(fr.quat_to_rotvec(q1), G["quat_to_rotvec"], atol=2e-6)


def test_rigid_ops_match_reference():
    q1, q2, t1, t2, pts = G["q1"], G["q2"], G["t1"], G["t2"], G["pts"]
    r1, r2 = fr.quat_to_rot(q1), fr.quat_to_rot(q2)
    np.testing.assert_allclose(fr.rigid_apply(r1, t1, pts), G["apply"], atol=1e-5)
    np.testing.assert_allclose(fr.rigid_invert_apply(r1, t1, pts), G["invert_apply"], atol=1e-5)
    cr, ct = fr.rigid_compose(r1, t1, r2, t2)
    np.testing.assert_allclose(cr, G["compose_rot"], atol=1e-6)
    np.testing.assert_allclose(ct, G["compose_trans"], atol=1e-5)
    ir, it = fr.rigid_invert(r1, t1)
    np.testing.assert_allclose(ir, G["invert_rot"], atol=1e-6)
    np.testing.assert_allclose(it, G["invert_trans"], atol=1e-5)
    nq, nt = fr.compose_q_update_vec(q1, t1, G["upd"], G["mask"])
    np.testing.assert_allclose(nq, G["cqu_quat"], atol=1e-6)
    np.testing.assert_allclose(nt, G["cqu_trans"], atol=1e-5)


def test_scipy_conventions():
    rv1, rv2 = G["rv1"], G["rv2"]
    np.testing.assert_allclose(fr.scipy_from_rotvec_as_matrix(rv1), G["rotvec_to_matrix"], atol=1e-14)
    np.testing.assert_allclose(fr.scipy_from_rotvec_as_matrix(rv1), Rotation.from_rotvec(rv1).as_matrix(), atol=1e-14)
    m = Rotation.from_rotvec(rv1).as_matrix()
    np.testing.assert_allclose(fr.scipy_from_matrix_as_rotvec(m), Rotation.from_matrix(m).as_rotvec(), atol=1e-9)
    np.testing.assert_allclose(fr.compose_rotvec(rv1, rv2), G["compose_rotvec"], atol=1e-9)
    # f32-rounded matrices: SVD projection (SciPy >= 1.8) vs plain Markley (pinned 1.7.3) differ by < 2e-7 rad
    rot32 = fr.quat_to_rot(G["q1"]).astype(np.float64)
    a = fr.scipy_from_matrix_as_rotvec(rot32, orthogonalize=True)
    np.testing.assert_allclose(a, G["extract_rotvec"], atol=1e-9)
    b = fr.scipy_from_matrix_as_rotvec(rot32, orthogonalize=False)
    ang = np.linalg.norm(a, axis=-1)
    ok = ang < np.pi - 1e-2  # near pi the axis sign is ill-conditioned
    assert np.abs(a - b)[ok].max() < 5e-7


def test_assemble_matches_reference():
    np.testing.assert_allclose(fr.scipy_from_rotvec_as_matrix(G["rv1"]).astype(np.float32), G["assemble_rot"], atol=1e-7)
    q = fr.rot_to_quat(G["assemble_rot"])
    np.testing.assert_allclose(fr.quat_to_rot(q), G["assemble_t7_rot"], atol=2e-6)


def test_backbone_matches_reference(tables):
    q, t = G["q1"][None], G["t1"][None]
    a37, a14 = compute_backbone(q, t, G["bb_psi"], G["bb_aatype"], tables)
    np.testing.assert_allclose(a37, G["bb_atom37"], atol=2e-5)
    np.testing.assert_allclose(a14, G["bb_atom14"], atol=2e-5)
    a37, a14 = compute_backbone(q, t, G["bb_psi"], None, tables)
    np.testing.assert_allclose(a37, G["bb_atom37_none"], atol=2e-5)
    np.testing.assert_allclose(a14, G["bb_atom14_none"], atol=2e-5)
