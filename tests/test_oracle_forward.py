"""Oracle pinning (CPU): score-network forward + reverse loop vs reference goldens."""
import numpy as np
import pytest

from conftest import kabsch_free_rmsd, load_golden
from framedipt_amd import config
from framedipt_amd import weights as W
from oracle import diffuser as od
from oracle import inference as oi
from oracle.score_network import ScoreNetwork


def _conf(name):
    inp = "inpaint" in name
    return (config.small_config(inp) if name.startswith("small") else config.base_config(inp)), inp


def _model(name, G, tables):
    conf, inp = _conf(name)
    shapes = W.param_shapes(conf.model, inp)
    assert list(shapes.keys()) == list(G["param_names"]) if "param_names" in G else True
    sd = W.synth_state_dict(shapes, int(G["weight_seed"]), float(G["bb_gain"]))
    diff = od.SE3Diffuser(conf.diffuser)
    return ScoreNetwork(conf.model, diff, sd, inpainting=inp, tables=tables), diff


def _feats(G):
    f = {k[3:]: G[k] for k in G if k.startswith("in_")}
    return f


@pytest.mark.parametrize("name", ["small_denovo_n16", "small_inpaint_n24", "small_denovo_n16_stress",
                                  "full_denovo_n64", "full_inpaint_n40"])
def test_forward_matches_reference(name, tables):
    G = load_golden(f"fwd_{name}.npz")
    conf, inp = _conf(name)
    shapes = W.param_shapes(conf.model, inp)
    assert list(shapes.keys()) == [str(s) for s in G["param_names"]]
    assert [",".join(map(str, s)) for s in shapes.values()] == [str(s) for s in G["param_shapes"]]
    model, _ = _model(name, G, tables)
    model.trace = {}
    out = model(_feats(G))
    rows = list(G["trace_rows"])
    tr = model.trace
    np.testing.assert_allclose(tr["node_init"], G["tr_node_init"], atol=5e-5)
    np.testing.assert_allclose(tr["edge_init"][:, rows], G["tr_edge_init"], atol=5e-5)
    nb = conf.model.ipa.num_blocks
    for b in range(nb):
        np.testing.assert_allclose(tr[f"node_{b}"], G[f"tr_node_{b}"] * G["in_res_mask"][..., None], atol=5e-5)
        if b < nb - 1:
            np.testing.assert_allclose(tr[f"edge_{b}"][:, rows], G[f"tr_edge_{b}"], atol=5e-5)
    stress = "stress" in name or name == "full_inpaint_n40"
    np.testing.assert_allclose(out["rigids"][..., 4:], G["out_rigids"][..., 4:], atol=5e-5)
    np.testing.assert_allclose(np.abs(out["rigids"][..., :4]), np.abs(G["out_rigids"][..., :4]), atol=1e-5)
    np.testing.assert_allclose(out["psi"], G["out_psi"], atol=1e-4)
    np.testing.assert_allclose(out["atom37"], G["out_atom37"], atol=3e-4)
    np.testing.assert_allclose(out["atom14"], G["out_atom14"], atol=3e-4)
    ts = np.abs(G["out_trans_score"]).max()
    np.testing.assert_allclose(out["trans_score"], G["out_trans_score"], atol=2e-4 * max(ts, 1.0))
    if not stress:
        rs = np.abs(G["out_rot_score"]).max()
        np.testing.assert_allclose(out["rot_score"], G["out_rot_score"], atol=2e-3 * max(rs, 1.0))


@pytest.mark.parametrize("name", ["small_denovo_n16_T10", "small_inpaint_n24_T10", "full_denovo_n64_T20"])
def test_teacher_forced_steps(name, tables):
    """Per-step parity: reference state in, one oracle step, compare x_{t-1} backbone (< 1e-3 A)."""
    G = load_golden(f"traj_{name}.npz")
    model, diff = _model(name, G, tables)
    base = _feats(G)
    num_t, min_t = int(G["num_t"]), float(G["min_t"])
    steps = np.linspace(min_t, 1.0, num_t)[::-1]
    aatype = model.preprocess_aatype(base.get("aatype"), base["fixed_mask"])
    tp = np.ones((1,), dtype=np.float32)
    rigid_traj = G["res_rigid_traj"][::-1]  # forward order: x_T, x_{T-1}, ...
    prot = G["res_prot_traj"][::-1]
    worst = 0.0
    for i, t in enumerate(steps):
        f = dict(base)
        f["rigids_t"] = rigid_traj[i]
        f["sc_ca_t"] = G["sc_in"][i + 1]
        nz = (G["noise_tape"][2 * i], G["noise_tape"][2 * i + 1]) if i < num_t - 1 else None
        _, _, bb, _, _, _ = oi.one_step(model, diff, f, t, min_t, 1 / num_t, tp, noise_scale=float(G["noise_scale"]),
                                        aatype=aatype, noise=nz, orthogonalize=True)
        worst = max(worst, kabsch_free_rmsd(bb, prot[i]))
    assert worst < 1e-3, worst


def test_free_running_small(tables):
    G = load_golden("traj_small_denovo_n16_T10.npz")
    model, diff = _model("small_denovo_n16_T10", G, tables)
    n = len(G["noise_tape"]) // 2
    tape = [(G["noise_tape"][2 * i], G["noise_tape"][2 * i + 1]) for i in range(n)]
    res = oi.inference_fn(model, diff, _feats(G), int(G["num_t"]), float(G["min_t"]),
                          noise_scale=float(G["noise_scale"]), noise_tape=tape, orthogonalize=True)
    for k in ("prot_traj", "rigid_0_traj", "trans_traj"):
        assert res[k].shape == G["res_" + k].shape
    # free-running: reported against the reference's own 8-vs-1-thread divergence floor (5e-2 A at T=10)
    assert kabsch_free_rmsd(res["prot_traj"][0], G["res_prot_traj"][0]) < 5e-2


def test_embedding_constants_pinned():
    from oracle import score_network as osn
    O = load_golden("ops.npz")
    np.testing.assert_array_equal(osn.TIMESTEP_FREQS, O["timestep_freqs"])
    np.testing.assert_array_equal(np.power(2056.0, 2 * np.arange(16) / 32).astype(np.float32), O["index_denoms"])
    np.testing.assert_allclose(osn.timestep_embedding(O["temb_t"], 32), O["temb"], atol=2e-6)
    np.testing.assert_allclose(osn.index_embedding(O["iemb_i"], 32), O["iemb"], atol=2e-6)


def test_torch_cpu_port_matches_the_numpy_oracle(tables):
    """oracle/torch_port.py (bench.py's cpu_baseline: the same forward on multithreaded torch-CPU ops) against the NumPy oracle
    and, through it, the reference golden: N = 64, full network."""
    from oracle.torch_port import TorchScoreNetwork
    G = load_golden("fwd_full_denovo_n64.npz")
    conf, inp = _conf("full_denovo_n64")
    ref, diff = _model("full_denovo_n64", G, tables)
    net = TorchScoreNetwork(conf.model, diff, ref.sd, inpainting=inp, tables=tables)
    a, b = ref(_feats(G)), net(_feats(G))
    for k in ("rigids", "psi", "trans_score", "atom37"):
        np.testing.assert_allclose(b[k], a[k], atol=2e-4, err_msg=k)
    np.testing.assert_allclose(b["atom37"], G["out_atom37"], atol=5e-4)
