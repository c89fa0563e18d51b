"""GPU parity tests (-m gpu) added in round 4: the benchmarked batch (B = 8, N = 300) in the parity suite, the fp16 margin at a second
weight seed and at bb_gain 0.5 (tests/golden/make_goldens_r4.py), per-step numbers of the literal bf16 build, the attention's
fp16 hi / lo point logits against an fp32 evaluation of the reference formula (the HIP-graph tests moved to test_gpu_round5.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import kabsch_free_rmsd, load_golden
from test_gpu_parity import _feats, _net, _teacher_forced_steps

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batch_of_eight_n300():
    """bench.py's batch in the parity suite: 8 DIFFERENT x_t at N = 300 in the fp16 mode — every sample of the batch is torch.equal to its
    own B = 1 run (a sample's bits do not depend on its batch mates), and sample 0 is the reference golden fwd_full_denovo_n300_t50."""
    G = load_golden("fwd_full_denovo_n300_t50.npz")
    net, d, conf = _net("full_denovo_n300_t50", G, "fp16")
    f1 = _feats(G)
    gen = torch.Generator().manual_seed(21)
    B, N = 8, f1["rigids_t"].shape[1]
    rig = [f1["rigids_t"].float()]
    sc = [f1["sc_ca_t"].float()]
    for k in range(1, B):  # other noisy frames / self-conditioning inputs of the same length
        q = torch.nn.functional.normalize(torch.randn(1, N, 4, generator=gen), dim=-1)
        rig.append(torch.cat([q, 12.0 * torch.randn(1, N, 3, generator=gen)], -1).cuda())
        sc.append((8.0 * torch.randn(1, N, 3, generator=gen)).cuda())
    keys = ("rigids", "psi", "rot_score", "trans_score", "atom37")
    singles = []
    for k in range(B):
        f = dict(f1)
        f["rigids_t"], f["sc_ca_t"] = rig[k].to(f1["rigids_t"].dtype), sc[k].to(f1["sc_ca_t"].dtype)
        singles.append({kk: v.clone() for kk, v in net(f).items() if kk in keys})
    fb = {k: torch.cat([v] * B, 0) for k, v in f1.items()}
    fb["rigids_t"] = torch.cat(rig, 0).to(f1["rigids_t"].dtype)
    fb["sc_ca_t"] = torch.cat(sc, 0).to(f1["sc_ca_t"].dtype)
    out = net(fb)
    for k in range(B):
        for kk in keys:
            assert torch.equal(out[kk][k], singles[k][kk][0]), (k, kk, float((out[kk][k] - singles[k][kk][0]).abs().max()))
    # the batch did hold eight different samples
    assert len({float(out["rigids"][k, 5, 4]) for k in range(B)}) == B
    o0 = {kk: out[kk][:1].cpu().numpy() for kk in keys}
    assert np.abs(o0["rigids"][..., 4:] - G["out_rigids"][..., 4:]).max() < 5e-4
    assert kabsch_free_rmsd(o0["atom37"], G["out_atom37"]) < 3e-4


# Where does the fp16 mode's per-step margin break (round-3 review)?  Measured on the MI355X (teacher-forced, N = 300, T = 5; worst step of
# x_{t-1} / of the x_0 prediction; round 6, `python tools/step_margins.py fp16`):
#   bb_gain 0.3, weight seed 7 (round 3's fixture): 0.76e-3 / 0.52e-3 A        bb_gain 0.3, weight seed 11: 0.37e-3 / 0.39e-3 A
#   bb_gain 0.5, weight seed 7: 1.43e-3 / 0.97e-3 A — the bar breaks between gain 0.3 and 0.5 for this seed (one step, t = 0.2575, exceeds it)
# (fp32 mode on the same fixtures: <= 5e-5 A.)  The bound of the mode is therefore stated per fixture: < 1e-3 A up to trained-weight scale 0.3;
# at 0.5 the fp16 mode MISSES the north-star bar.  That miss is a documented number, not a loosened bound (round-5 review): the test asserts
# the measured worst step within +-10 %, so that a regression AND an improvement (which should then move DESIGN.md section 5 and the bench
# line's precision_mode) both trip it.
FP16_BOUND = {"full_denovo_n300_T5_gain03_seed11": 1.0e-3}
FP16_DOCUMENTED_MISS = {"full_denovo_n300_T5_gain05": (1.43e-3, 1.1e-3)}  # (measured worst x_(t-1) step, bound that the x_0 prediction does hold)


@pytest.mark.parametrize("name", sorted(FP16_BOUND) + sorted(FP16_DOCUMENTED_MISS))
@pytest.mark.parametrize("prec", ["fp32", "fp16"])
def test_teacher_forced_margin_second_seed_and_gain05(name, prec):
    r = _teacher_forced_steps(name, prec)
    fmt = lambda v: " ".join(f"{x:.2e}" for x in v)  # noqa: E731
    print(f"{prec} {name}: x_(t-1) per step [{fmt(r[:, 1])}] A, x_0 prediction [{fmt(r[:, 2])}] A")
    if prec == "fp32":
        assert r[:, 1].max() < 1e-4 and r[:, 2].max() < 1e-4
    elif name in FP16_BOUND:
        assert r[:, 1].max() < FP16_BOUND[name] and r[:, 2].max() < FP16_BOUND[name]
    else:  # the documented miss of the 1e-3 A bar: the measured value, both ways
        miss, x0_bound = FP16_DOCUMENTED_MISS[name]
        assert 0.9 * miss < r[:, 1].max() < 1.1 * miss, (r[:, 1].max(), "the documented miss moved: update DESIGN.md section 5 and bench.py PREC_MODE")
        assert (r[:, 1] > 1e-3).sum() == 1  # one step of the five (t = 0.2575) is outside the bar
        assert r[:, 2].max() < x0_bound


def test_bf16_build_per_step_numbers():
    """The literal bf16 build (BASELINE configs[1]; lib/libfdipt_hip_bf16.so) per STEP, teacher-forced on traj_full_denovo_n64_T20: the numbers
    bench.py quotes for `--bf16` (round 1's 5e-3 ... 1.2e-2 A predate the split operands).  A subprocess: a process binds one library."""
    lib = os.path.join(ROOT, "framedipt_amd", "lib", "libfdipt_hip_bf16.so")
    assert os.path.exists(lib), "run __graft_entry__.build() first"
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_parity import _teacher_forced_steps
r = _teacher_forced_steps("full_denovo_n64_T20", "bf16")
print("BF16STEP", float(r[:, 1].max()), float(r[:, 2].max()), float(np.median(r[:, 1])))
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FDIPT_LIB=lib), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    worst, worst0, med = (float(v) for v in next(l for l in r.stdout.splitlines() if l.startswith("BF16STEP")).split()[1:])
    print(f"bf16 build, N = 64, T = 20 teacher-forced: x_(t-1) worst step {worst:.2e} A (median {med:.2e}), x_0 prediction worst {worst0:.2e} A")
    assert worst < 5e-3 and worst0 < 5e-3  # bf16's own, looser bound (8 significand bits); the fp16 build holds 1e-3 on the same fixture


def test_attention_point_logits_hi_lo_against_fp32_formula():
    """The attention's point term on fp16 hi / lo MFMAs (csrc/attention3.hip; -gamma |q - k|^2 / 2 with the row-constant -gamma |q|^2 / 2
    dropped) against the fp32 mode of the same module (LDS-score kernel: squared point distances summed directly in fp32, the
    reference's own arithmetic, ipa_pytorch.py:255-283): the IPA module's output, block 0, N = 300, on a chain walk ~100 A wide (the
    cancellation in q.k - |k|^2 / 2 grows with the coordinates), within the rounding of the fp16 Q K^T / P V operands."""
    import ctypes as C

    from framedipt_amd import _lib
    from framedipt_amd.model.score_network import BatchState
    G = load_golden("fwd_full_denovo_n300_t50.npz")
    net, d, conf = _net("full_denovo_n300_t50", G, "fp16")
    lib = _lib.load()
    B, N, H = 1, 300, 8
    gen = torch.Generator().manual_seed(3)
    node = torch.randn(B, N, 256, generator=gen).cuda()
    z = (0.5 * torch.randn(B, N, N, 128, generator=gen)).cuda().half().contiguous()
    q = torch.nn.functional.normalize(torch.randn(B, N, 4, generator=gen), dim=-1)
    walk = torch.cumsum(3.8 * torch.nn.functional.normalize(torch.randn(B, N, 3, generator=gen), dim=-1), 1)  # a 3.8 A chain walk
    rig = torch.cat([q, walk - walk.mean(1, keepdim=True)], -1).cuda().contiguous()
    mask = torch.ones(B, N).cuda()
    out = torch.empty(B, N, 256).cuda()
    st = BatchState(net, torch.arange(N)[None].cuda())
    P = _lib.ptr
    dm, pr, dr = C.byref(net.dims), P(net.params), P(net.derived)
    _lib.check(lib.fdipt_ipa_attention_fwd(dm, pr, dr, 0, B, N, P(node), P(z), P(rig), P(mask), P(out), P(st.ws), st.ws_bytes, _lib.stream_ptr()))
    out2 = torch.empty_like(out)
    _lib.check(lib.fdipt_ipa_attention_fwd(dm, pr, dr, 0, B, N, P(node), P(z), P(rig), P(mask), P(out2), P(st.ws), st.ws_bytes, _lib.stream_ptr()))
    assert torch.equal(out, out2) and bool(torch.isfinite(out).all())
    # the same module in the fp32 mode (LDS-score kernel, VALU point distances in fp32: the reference's own arithmetic)
    net32, _, _ = _net("full_denovo_n300_t50", G, "fp32")
    st32 = BatchState(net32, torch.arange(N)[None].cuda())
    z32 = z.float().contiguous()
    out32 = torch.empty_like(out)
    _lib.check(lib.fdipt_ipa_attention_fwd(C.byref(net32.dims), P(net32.params), P(net32.derived), 0, B, N, P(node), P(z32), P(rig), P(mask), P(out32),
                                           P(st32.ws), st32.ws_bytes, _lib.stream_ptr()))
    rel = float((out - out32).norm() / out32.norm())
    print(f"IPA module, block 0, N = 300, 100 A wide structure: fp16 mode (hi / lo point logits) vs fp32 mode, relative {rel:.2e}")
    assert rel < 1e-3


def test_run_sharded_inpainting_entry_two_ranks(tmp_path):
    """BASELINE configs[2] at the file level (experiments/inference.py:244-389,480-556): ``run_sharded --download-dir`` builds a
    ``ConditionalSampler`` over processed structures (the three TCR-pMHC complexes of the reference's own test data, ~810 residues, 5 chains),
    runs mixed-length batches and writes the reference's directory layout through ``framedipt_amd.output``; two ranks (on this box's one GPU:
    FDIPT_ONE_GPU=1) leave a file tree byte-identical to the one-rank run."""
    import json
    import pickle

    import pandas as pd
    F = load_golden("features.npz")
    data = tmp_path / "data"
    (data / "processed").mkdir(parents=True)
    rows = []
    for name in ("1fyt", "5ksa", "7t2d"):
        cf = {k[len(name) + 4:]: (v.astype(np.float64) if v.dtype.kind == "f" else v) for k, v in F.items() if k.startswith(name + "_in_")}
        with open(data / "processed" / f"{name}.pkl", "wb") as f:
            pickle.dump(cf, f)
        n = int(np.sum(np.asarray(cf["max_modeled_idxs"]) - np.asarray(cf["min_modeled_idxs"]) + 1))  # modelled residues of the five chains (810 / 820 / 801)
        rows.append({"pdb_name": f"{name}-assembly1", "processed_path": str(data / "processed" / f"{name}.pkl"), "modeled_seq_len": n})
    pd.DataFrame(rows).to_csv(data / "processed" / "metadata.csv", index=False)
    trees = {}
    for world, port in ((2, "29651"), (1, "29652")):
        out_dir = tmp_path / f"w{world}"
        env = dict(os.environ, FDIPT_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", port, "-m", "framedipt_amd.run_sharded", "--out-dir", str(out_dir), "--download-dir", str(data),
               "--samples-per-structure", "2", "--num-t", "3", "--max-batch", "4", "--precision", "fp16"]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-3000:]
        with open(out_dir / "manifest.json") as f:
            man = json.load(f)
        assert man["n_items"] == 6 and man["world_size"] == world
        if world == 2:
            assert {s["rank"] for s in man["samples"]} == {0, 1}
        files = sorted(str(p.relative_to(out_dir)) for p in out_dir.rglob("*") if p.is_file() and p.suffix in (".pdb", ".csv"))
        trees[world] = {f: (out_dir / f).read_bytes() for f in files}
    assert sorted(trees[1]) == sorted(trees[2])
    # per structure: ground truth + diffusion_info.csv + two samples
    assert len(trees[1]) == 3 * 4, sorted(trees[1])
    for f in trees[1]:
        assert trees[1][f] == trees[2][f], f
    some = next(f for f in trees[1] if f.endswith("sample_1_1.pdb"))
    text = trees[1][some].decode()
    assert text.startswith("MODEL     1") and text.rstrip().endswith("END") and "100.00" in text


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["full_denovo_n64", "full_inpaint_n40"])
def test_score_launch_atoms_equal_the_backbone_kernel(name):
    """The forward's atom37 / atom14 (built on the score launch, one atom per lane: frames.hip d_backbone_atom) against the
    stand-alone backbone kernel (`fdipt_backbone_atoms`, one residue per thread) on the forward's own frames and psi: the same
    expressions in the same order, so the same bits; and the unfolded forward (FDIPT_KF_UNFOLDED: backbone_kernel launch)
    agrees with both up to what its differently tiled EdgeTransition rows move the frames."""
    from framedipt_amd import _lib
    from framedipt_amd.inference import get_atom_positions_from_rigids
    from framedipt_amd.model.score_network import preprocess_aatype
    G = load_golden(f"fwd_{name}.npz")
    net, _, conf = _net(name, G, "fp16")
    feats = _feats(G)
    out = net(feats)
    aatype = preprocess_aatype(feats.get("aatype"), feats["fixed_mask"].to(torch.float32), net.inpainting, net._model_conf.input_aatype)
    a37 = get_atom_positions_from_rigids(net, out["rigids"], out["psi"], aatype)
    assert np.array_equal(a37, out["atom37"].cpu().numpy())
    net_u, _, _ = _net(name, G, "fp16", _lib.KF_UNFOLDED)
    out_u = net_u(feats)
    np.testing.assert_allclose(out_u["atom37"].cpu().numpy(), out["atom37"].cpu().numpy(), atol=3e-3)
