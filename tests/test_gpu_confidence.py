"""GPU parity tests (-m gpu) of the EigenFold confidence path against walks captured from the
reference (tests/golden/make_goldens_r2.py: confidence_golden): forward noising, the two log-probabilities per step and the
whole logp_confidence_score."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import _net, dev

pytestmark = pytest.mark.gpu

WALKS = ["small_denovo_n24_T6", "full_inpaint_n40_T5"]


def _tape(G):
    tape = G["noise_tape"]  # [2 (T-1), N, 3]: R^3 draw then SO(3) draw per step
    return tape[1::2][:, None], tape[0::2][:, None]  # z_rot, z_trans  [T-1, 1, N, 3]


def _feats(G):
    return {k[3:]: torch.as_tensor(G[k]) for k in G if k.startswith("in_")}


@pytest.mark.parametrize("name", WALKS)
def test_forward_noising_and_step_log_probs_teacher_forced(name):
    """Every step from the reference's own state: x_t of fdipt_se3_forward_step and the four sums of fdipt_se3_step_log_prob
    under the reference's scores."""
    from framedipt_amd.rigid import quat_to_rot
    G = load_golden(f"conf_{name}.npz")
    _, d, _ = _net(name, G, "fp32")
    z_rot, z_trans = _tape(G)
    mask = dev(G["diffuse_mask"][None].astype(np.float32))
    dt = 1.0 / int(G["num_t"])
    rot = quat_to_rot(dev(G["x0"][None, :, :4]))
    trans = dev(G["x0"][None, :, 4:])
    for i, t_1 in enumerate(G["step_t_1"]):
        t7 = torch.empty(1, rot.shape[1], 7, device="cuda")
        ro, to = d.forward_device(rot, trans, mask, dev(z_rot[i]), dev(z_trans[i]), float(t_1), dt, rigids_out=t7)
        np.testing.assert_allclose(ro.cpu().numpy()[0], G["step_rot"][i], atol=2e-6, err_msg=f"rot step {i}")
        np.testing.assert_allclose(to.cpu().numpy()[0], G["step_trans"][i], rtol=1e-6, atol=2e-6, err_msg=f"trans step {i}")
        np.testing.assert_allclose(quat_to_rot(t7[..., :4]).cpu().numpy()[0], G["step_rot"][i], atol=2e-6)
        # teacher-forced: the reference's frames on both sides of the step and the reference's scores
        rt, tt = dev(G["step_rot"][i][None]), dev(G["step_trans"][i][None])
        out = d.step_log_prob_device(rt, tt, rot, trans, dev(G["step_rot_score"][i][None].astype(np.float64)),
                                     dev(G["step_trans_score"][i][None].astype(np.float32)), mask, float(G["step_t"][i]), float(t_1),
                                     dt).cpu().numpy()[0]
        lpb, lpf = float(G["step_lp_backward"][i]), float(G["step_lp_forward"][i])
        assert abs(out[0] + out[1] - lpb) <= 2e-5 * max(abs(lpb), 1.0), (i, out, lpb)
        assert abs(out[2] + out[3] - lpf) <= 2e-5 * max(abs(lpf), 1.0), (i, out, lpf)
        rot, trans = rt, tt


@pytest.mark.parametrize("name", WALKS)
def test_diffuser_api_forward_and_log_probs(name):
    """SE3Diffuser.forward / log_prob_forward / log_prob_backward with Rigid arguments and the np.random stream."""
    from framedipt_amd.rigid import Rigid, Rotation
    G = load_golden(f"conf_{name}.npz")
    _, d, _ = _net(name, G, "fp32")
    dt = 1.0 / int(G["num_t"])
    x0 = Rigid.from_tensor_7(dev(G["x0"]))
    tape = list(G["noise_tape"])
    orig = np.random.normal
    np.random.normal = lambda size=None, **k: tape.pop(0).reshape(size)
    try:
        x1 = d.forward(rigids_t_1=x0, t_1=float(G["step_t_1"][0]), dt=dt, diffuse_mask=G["diffuse_mask"])
    finally:
        np.random.normal = orig
    np.testing.assert_allclose(x1.get_rots().get_rot_mats().cpu().numpy(), G["step_rot"][0], atol=2e-6)
    np.testing.assert_allclose(x1.get_trans().cpu().numpy(), G["step_trans"][0], rtol=1e-6, atol=2e-6)
    ref1 = Rigid(Rotation(rot_mats=dev(G["step_rot"][0])), dev(G["step_trans"][0]))
    lpf = d.log_prob_forward(rigids_t=ref1, rigids_t_1=x0, t_1=float(G["step_t_1"][0]), dt=dt, diffuse_mask=G["diffuse_mask"])
    lpb = d.log_prob_backward(rigids_t=ref1, rigids_t_1=x0, trans_score_t=G["step_trans_score"][0], rot_score_t=G["step_rot_score"][0],
                              t=float(G["step_t"][0]), dt=dt, diffuse_mask=G["diffuse_mask"])
    assert isinstance(lpf, float) and isinstance(lpb, float)
    assert abs(lpf - float(G["step_lp_forward"][0])) <= 2e-5 * max(abs(float(G["step_lp_forward"][0])), 1.0)
    assert abs(lpb - float(G["step_lp_backward"][0])) <= 2e-5 * max(abs(float(G["step_lp_backward"][0])), 1.0)


@pytest.mark.parametrize("name,prec,rtol", [("small_denovo_n24_T6", "fp32", 2e-4), ("full_inpaint_n40_T5", "fp32", 2e-4),
                                            ("full_inpaint_n40_T5", "fp16", 5e-3)])
def test_logp_confidence_score_vs_reference(name, prec, rtol):
    """The whole walk with this library's score network: log_prob and every partial sum of log_probs.  The scores enter the
    backward term as (x_{t-1} - mu(score))^2 / (2 g^2 dt): the bound is relative to the largest partial sum."""
    from framedipt_amd.confidence import logp_confidence_score
    from framedipt_amd.rigid import Rigid
    G = load_golden(f"conf_{name}.npz")
    net, d, _ = _net(name, G, prec)
    lp, lps = logp_confidence_score(net, d, Rigid.from_tensor_7(dev(G["x0"])), _feats(G), G["diffuse_mask"], int(G["num_t"]),
                                    float(G["min_t"]), "cuda", True, noise_tape=_tape(G))
    assert isinstance(lp, float) and len(lps) == int(G["num_t"])
    scale = np.abs(G["log_probs"]).max()
    np.testing.assert_allclose(np.array(lps), G["log_probs"], atol=rtol * scale)
    assert abs(lp - float(G["log_prob"])) <= rtol * scale


def test_confidence_batch_matches_single():
    """Two structures in one batch (same structure, different noise) == the two single walks."""
    from framedipt_amd.confidence import logp_confidence_score
    G = load_golden("conf_full_inpaint_n40_T5.npz")
    net, d, _ = _net("full_inpaint_n40_T5", G, "fp16")
    z_rot, z_trans = _tape(G)
    rng = np.random.default_rng(5)
    z_rot2, z_trans2 = rng.standard_normal(z_rot.shape), rng.standard_normal(z_trans.shape)
    f1 = _feats(G)
    f2 = {k: torch.cat([v, v], 0) for k, v in f1.items()}
    x0 = dev(G["x0"])
    single = [logp_confidence_score(net, d, x0, f1, G["diffuse_mask"], int(G["num_t"]), float(G["min_t"]), noise_tape=nt)
              for nt in ((z_rot, z_trans), (z_rot2, z_trans2))]
    lp, lps = logp_confidence_score(net, d, torch.stack([x0, x0]), f2, G["diffuse_mask"], int(G["num_t"]), float(G["min_t"]),
                                    noise_tape=(np.concatenate([z_rot, z_rot2], 1), np.concatenate([z_trans, z_trans2], 1)))
    assert lp.shape == (2,) and lps.shape == (int(G["num_t"]), 2)
    for b in range(2):
        assert lp[b] == single[b][0]
        np.testing.assert_array_equal(lps[:, b], np.array(single[b][1]))


def test_conditional_sampler_from_processed_pickle_matches_reference(tmp_path):
    """ConditionalSampler(data_conf, diffuser, device) on a processed-structure pickle of the TCR-pMHC complex 1fyt (the reference's
    own test data): metadata.csv -> process_csv_row (row f2) -> redaction -> sample_ref -> padding, vs the item captured from the
    reference sampler (tests/golden/make_goldens_r2.py sampler_goldens): keys, dtypes, shapes, values."""
    import pickle

    import pandas as pd

    from framedipt_amd import config
    from framedipt_amd import rigid as R
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.sampler import ConditionalSampler
    S, F = load_golden("sampler_dicts.npz"), load_golden("features.npz")
    cf = {k[len("1fyt_in_"):]: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in F.items() if k.startswith("1fyt_in_")}
    (tmp_path / "processed").mkdir()
    with open(tmp_path / "processed" / "1fyt.pkl", "wb") as f:
        pickle.dump(cf, f)
    pd.DataFrame([{"pdb_name": str(S["cond_meta"][0]), "processed_path": str(tmp_path / "processed" / "1fyt.pkl"),
                   "modeled_seq_len": int(S["cond_meta"][2])}]).to_csv(tmp_path / "processed" / "metadata.csv", index=False)
    d = SE3Diffuser(config.base_config(inpainting=True).diffuser, device="cuda")
    ds = ConditionalSampler(config.to_conf({"download_dir": str(tmp_path), "samples": 2, "seed": 123,
                                            "redaction": {"redact_min_len": 8, "redact_max_len": 14}}), d, "cuda")
    np.random.seed(77)
    name, sample_id, feats = ds[1]
    assert (name, str(sample_id)) == (str(S["cond_meta"][0]), str(S["cond_meta"][1]))
    keys = sorted(k[5:] for k in S if k.startswith("cond_") and not k.endswith("_dtype") and k != "cond_meta")
    assert sorted(feats) == keys
    for k in keys:
        ref = S[f"cond_{k}"]
        assert str(feats[k].dtype) == "torch." + str(S[f"cond_{k}_dtype"]), (k, feats[k].dtype)
        assert tuple(feats[k].shape) == ref.shape, k
        v = feats[k].cpu().numpy()
        if k in ("rigids_t", "rigids_0"):
            np.testing.assert_allclose(R.quat_to_rot(feats[k][0, :, :4].float()).cpu().numpy(),
                                       R.quat_to_rot(dev(ref[0, :, :4])).cpu().numpy(), atol=3e-6, err_msg=k)
            np.testing.assert_allclose(v[..., 4:], ref[..., 4:], atol=2e-5, err_msg=k)
        elif v.dtype.kind in "iu":
            np.testing.assert_array_equal(v, ref, err_msg=k)
        else:
            np.testing.assert_allclose(v, ref, rtol=1e-6, atol=2e-6, err_msg=k)
    assert int((1 - feats["fixed_mask"]).sum()) == 50  # five chains, one redacted loop each


def test_inpainting_on_real_complex_keeps_the_motif(tmp_path):
    """BASELINE config 1 / 3 in miniature: redesign loops of the TCR-pMHC complex 1fyt (N = 810, 5 chains) for 4 reverse steps in the
    fp16 mode: motif frames stay put, the redesigned residues move and stay finite, per-chain seq_idx gaps reach the network."""
    import pickle

    import pandas as pd

    from framedipt_amd import config
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.inference import inference_fn
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import ConditionalSampler
    F = load_golden("features.npz")
    cf = {k[len("1fyt_in_"):]: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in F.items() if k.startswith("1fyt_in_")}
    (tmp_path / "processed").mkdir()
    with open(tmp_path / "processed" / "1fyt.pkl", "wb") as f:
        pickle.dump(cf, f)
    pd.DataFrame([{"pdb_name": "1fyt-assembly1", "processed_path": str(tmp_path / "processed" / "1fyt.pkl"), "modeled_seq_len": 810}]
                 ).to_csv(tmp_path / "processed" / "metadata.csv", index=False)
    conf = config.base_config(inpainting=True)
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, inpainting=True, precision="fp16")
    net.load_synthetic(7).to("cuda")
    ds = ConditionalSampler(config.to_conf({"download_dir": str(tmp_path), "samples": 1, "seed": 123,
                                            "redaction": {"redact_min_len": 8, "redact_max_len": 14}}), d, "cuda")
    np.random.seed(5)
    _, _, feats = ds[0]
    assert int(feats["seq_idx"].max()) >= 810 + 4 * 200 - 1
    out = inference_fn(net, d, feats, num_t=4, min_t=0.01, aux_traj=True, inpainting=True, embed_self_conditioning=True)
    fixed = feats["fixed_mask"][0].bool().cpu().numpy()
    x0, xT = out["rigid_traj"][0][0], out["rigid_traj"][-1][0]
    assert np.isfinite(out["prot_traj"]).all()
    np.testing.assert_allclose(x0[fixed, 4:], feats["rigids_0"][0, fixed, 4:].cpu().numpy(), atol=1e-4)
    assert np.abs(x0[~fixed, 4:] - xT[~fixed, 4:]).max() > 0.1
    ca = out["prot_traj"][0][0][:, 1]
    np.testing.assert_allclose(ca[fixed], feats["atom37_pos"][0, fixed, 1].cpu().numpy(), atol=2e-3)
