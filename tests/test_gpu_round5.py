"""GPU parity tests (-m gpu) added in round 5: the step-cursor launches and the step graph of the reverse loop (the default product
path) against the launch-by-launch loop, configs 3 and 5 at their own batch sizes, the N > 1024 attention fallback, the shared-GPU
guard and the verify mode."""
import numpy as np
import pytest
import torch

from conftest import kabsch_free_rmsd, load_golden
from test_gpu_parity import _feats, _net

pytestmark = pytest.mark.gpu


def _host(v):
    return v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


def _assert_same(ref, got, what=""):
    assert sorted(ref) == sorted(got)
    for k in ref:
        np.testing.assert_array_equal(_host(ref[k]), _host(got[k]), err_msg=f"{what}: {k}")


def _denovo_batch(N, B, T, precision, seed=5, cached_score=False):
    from framedipt_amd import config, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    conf = config.base_config()
    if cached_score:
        conf.diffuser.so3.use_cached_score = True
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision=precision).load_synthetic(7).to("cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
    feats, tape = sharding.stack_items([sharding.seeded_item(ds, i, seed, d, T, 0.01) for i in range(B)])
    return net, d, feats, tape


@pytest.mark.parametrize("N,B,T,precision", [(64, 2, 12, "fp16"), (24, 1, 5, "fp32"), (300, 2, 11, "fp16")])
def test_step_graph_equals_the_launch_by_launch_loop(N, B, T, precision):
    """inference_fn's default path — first noisy step enqueued eagerly through the device-side step cursor, every later one a replay of
    a captured HIP graph (one-step graph and GRAPH_CHUNK-step graph) — returns every array bit-identical to graph=False."""
    from framedipt_amd.inference import ReverseLoop, inference_fn
    net, d, feats, tape = _denovo_batch(N, B, T, precision)
    kw = dict(num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    ref = inference_fn(net, d, feats, graph=False, **kw)
    got = inference_fn(net, d, feats, **kw)
    _assert_same(ref, got, "run()")
    # chunk graph of 4 steps + single-step replays for the remainder
    old = ReverseLoop.GRAPH_CHUNK
    try:
        ReverseLoop.GRAPH_CHUNK = 4
        loop = ReverseLoop(net, d, feats, T, 0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape).run()
        assert loop._g1 is not None or loop._gn is not None
        _assert_same(ref, loop.results(), "chunk of 4")
    finally:
        ReverseLoop.GRAPH_CHUNK = old
    # aux_traj=False (no rigid_0_traj / trans_traj rows)
    ref2 = inference_fn(net, d, feats, graph=False, num_t=T, min_t=0.01, noise_scale=0.1, noise_tape=tape)
    got2 = inference_fn(net, d, feats, num_t=T, min_t=0.01, noise_scale=0.1, noise_tape=tape)
    _assert_same(ref2, got2, "aux_traj=False")


def test_step_graph_steps_out_of_order_and_twice():
    """step(k) through the graph for k out of sequence (bench.py's K < T: timed steps spread over the schedule): the cursor is set on
    the stream first; every step reads only row k and writes rows k / k + 1, so re-running a step reproduces its rows."""
    from framedipt_amd.inference import ReverseLoop
    T = 9
    net, d, feats, tape = _denovo_batch(48, 2, T, "fp16")
    kw = dict(aux_traj=True, noise_scale=0.1, noise_tape=tape)
    ref = ReverseLoop(net, d, feats, T, 0.01, graph=False, **kw).run()
    loop = ReverseLoop(net, d, feats, T, 0.01, **kw)
    loop.prime()
    for k in range(T):
        loop.step(k)
    # re-run steps 5 and 2 from the rows they read (x_t rows are still there); self-conditioning input = the previous step's CA prediction
    for k in (5, 2):
        loop.sc_ca.copy_(ref_sc(ref, k))
        loop.step(k)
        for name in ("rigid_traj", "prot_traj", "bb0_traj", "trans_traj"):
            a, b = getattr(ref, name), getattr(loop, name)
            assert torch.equal(a[k], b[k]), (k, name)
        assert torch.equal(ref.rigid_traj[k + 1], loop.rigid_traj[k + 1])
    assert int(loop.cursor[0]) == 3 and int(loop.cursor[1]) == 0


def ref_sc(ref, k):
    """Self-conditioning CA input of step k of a finished loop: the x_0 prediction of step k - 1 (its trans_traj row), or the priming
    forward's for k = 0 (not reproduced here: k >= 1 only)."""
    assert k >= 1
    return ref.trans_traj[k - 1]  # (de novo: diffuse_mask = 1, so the row is the predicted translation itself)


def test_step_graph_inpainting_rows_from_the_backbone_launch():
    """The reference's default inpainting configuration (inference.input_aatype=True, model.input_aatype=False): rigid_0_traj rows come
    from a backbone launch with the caller's residue types — in the step graph through fdipt_backbone_atoms_indexed."""
    from framedipt_amd import config
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.inference import ReverseLoop, draw_noise_tape
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import ConditionalSampler
    import bench
    conf = config.base_config(inpainting=True)
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, inpainting=True, precision="fp16").load_synthetic(7).to("cuda")
    ds = ConditionalSampler.from_features([("synthetic", bench.synthetic_complex((30, 26), ((10, 22),)))], d, "cuda", samples=2)
    from framedipt_amd import sharding
    T = 8
    feats, tape = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, T, 0.01) for i in range(2)])
    kw = dict(aux_traj=True, noise_scale=0.1, noise_tape=tape, inpainting=True, input_aatype=True)
    ref = ReverseLoop(net, d, feats, T, 0.01, graph=False, **kw)
    assert not ref.bb0_from_forward
    got = ReverseLoop(net, d, feats, T, 0.01, **kw)
    _assert_same(ref.run().results(), got.run().results(), "inpainting")


def test_step_graph_with_the_cached_rotation_score_table():
    """so3.use_cached_score: the per-step table rows are addressed through the cursor as well."""
    from framedipt_amd.inference import inference_fn
    T = 7
    net, d, feats, tape = _denovo_batch(32, 2, T, "fp32", cached_score=True)
    kw = dict(num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    _assert_same(inference_fn(net, d, feats, graph=False, **kw), inference_fn(net, d, feats, **kw), "cached score")


def test_weight_reload_between_trajectories():
    """A graph holds weight pointers; graphs live in the ReverseLoop of one trajectory, so a reload between two inference_fn calls is
    picked up (the round-4 whole-trajectory cache on the model is gone)."""
    from framedipt_amd.inference import inference_fn
    T = 6
    net, d, feats, tape = _denovo_batch(32, 1, T, "fp16")
    kw = dict(num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    a = inference_fn(net, d, feats, **kw)
    net.load_synthetic(11)
    b = inference_fn(net, d, feats, **kw)
    b_ref = inference_fn(net, d, feats, graph=False, **kw)
    _assert_same(b_ref, b, "after reload")
    assert not np.array_equal(a["prot_traj"], b["prot_traj"])


def _other_inputs(f1, B, seed, masked_tail=0):
    """B - 1 more samples of the golden's shape: other noisy frames / self-conditioning inputs; with ``masked_tail`` sample k also loses
    its last 4 k residues (res_mask = 0 rows, as sharding.stack_items_padded pads a shorter member of a mixed-length batch)."""
    gen = torch.Generator().manual_seed(seed)
    N = f1["rigids_t"].shape[1]
    feats = [dict(f1)]
    for k in range(1, B):
        f = dict(f1)
        q = torch.nn.functional.normalize(torch.randn(1, N, 4, generator=gen), dim=-1)
        f["rigids_t"] = torch.cat([q, 15.0 * torch.randn(1, N, 3, generator=gen)], -1).cuda().to(f1["rigids_t"].dtype)
        f["sc_ca_t"] = (10.0 * torch.randn(1, N, 3, generator=gen)).cuda().to(f1["sc_ca_t"].dtype)
        if masked_tail:
            m = f1["res_mask"].clone()
            m[:, N - masked_tail * k:] = 0
            f["res_mask"] = m
        feats.append(f)
    return feats


def _batch_vs_singles(net, feats, keys=("rigids", "psi", "rot_score", "trans_score", "atom37")):
    singles = [{kk: v.clone() for kk, v in net(f).items() if kk in keys} for f in feats]
    fb = {k: torch.cat([f[k] for f in feats], 0) for k in feats[0]}
    out = net(fb)
    for k, one in enumerate(singles):
        for kk in keys:
            assert torch.equal(out[kk][k], one[kk][0]), (k, kk, float((out[kk][k] - one[kk][0]).abs().max()))
    assert len({float(out["rigids"][k, 5, 4]) for k in range(len(feats))}) == len(feats)  # (the batch did hold different samples)
    return out


def test_batch_n1000_b4_fp32():
    """BASELINE configs[4] at its own batch: N = 1000, 4 samples, fp32 mode — four different x_t, each torch.equal to its B = 1 run,
    sample 0 against the reference golden fwd_full_inpaint_n1000."""
    G = load_golden("fwd_full_inpaint_n1000.npz")
    net, d, conf = _net("full_inpaint_n1000", G, "fp32")
    out = _batch_vs_singles(net, _other_inputs(_feats(G), 4, 31))
    o0 = {k: out[k][:1].cpu().numpy() for k in ("rigids", "atom37")}
    np.testing.assert_allclose(o0["rigids"][..., 4:], G["out_rigids"][..., 4:], atol=3e-4)
    assert kabsch_free_rmsd(o0["atom37"], G["out_atom37"]) < 1e-4


def test_batch_n724_b8_fp16_padded_mixed():
    """BASELINE configs[2]'s shape at the benchmarked batch: eight members of a padded mixed-length batch at N = 724 (sample k's last
    4 k residues are res_mask = 0 padding), fp16 mode — each torch.equal to its B = 1 run, sample 0 (the full-length member) against the
    reference golden fwd_full_inpaint_n724_4chain at the fp16 mode's stated bounds."""
    from test_gpu_sizes import FP16_BOUND
    G = load_golden("fwd_full_inpaint_n724_4chain.npz")
    net, d, conf = _net("full_inpaint_n724_4chain", G, "fp16")
    out = _batch_vs_singles(net, _other_inputs(_feats(G), 8, 32, masked_tail=4))
    o0 = {k: out[k][:1].cpu().numpy() for k in ("rigids", "atom37")}
    assert np.abs(o0["rigids"][..., 4:] - G["out_rigids"][..., 4:]).max() < FP16_BOUND["ca"]
    assert kabsch_free_rmsd(o0["atom37"], G["out_atom37"]) < FP16_BOUND["bb_rmsd"]


def test_verify_mode_reruns_forwards_and_detects_a_difference(monkeypatch):
    """inference_fn(verify=k): the forward of every k-th step runs twice and its outputs are compared bit for bit; same results as without;
    a forward that does not reproduce (here: the first pass's rotation score is disturbed) raises FdiptError."""
    from framedipt_amd import _lib
    from framedipt_amd.inference import ReverseLoop, inference_fn
    T = 7
    net, d, feats, tape = _denovo_batch(32, 2, T, "fp16")
    kw = dict(num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    ref = inference_fn(net, d, feats, **kw)
    loop = ReverseLoop(net, d, feats, T, 0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, verify=3).run()
    assert loop.verified == 3  # steps 0, 3, 6
    _assert_same(ref, loop.results(), "verify=3")
    plain = ReverseLoop._verify_forward

    def disturbed(self, k):
        out = plain(self, k)
        out[2][0, 5, 1] += 1e-9
        return out
    monkeypatch.setattr(ReverseLoop, "_verify_forward", disturbed)
    with pytest.raises(_lib.FdiptError, match="verify"):
        inference_fn(net, d, feats, verify=2, **kw)


def test_shared_gpu_guard_on_this_box():
    """gpu_guard.check: alone on the GPU -> 0 foreign processes (or None where the KFD process list is not readable); with a second
    process holding a queue on the device -> a RuntimeWarning, SharedGpuError under the `refuse` policy."""
    import subprocess
    import sys
    import time
    import warnings

    from framedipt_amd import gpu_guard
    n = gpu_guard.check("cuda:0", policy="warn", once=False)
    if n is None:
        pytest.skip("KFD process list not readable in this container")
    base = n  # (0 on an exclusive box; a monitoring agent with a queue on the device would show up here and is not this test's business)
    child = subprocess.Popen([sys.executable, "-c", "import torch,time,sys; x=torch.zeros(8,device='cuda:0')+1; torch.cuda.synchronize(); "
                              "print('up',flush=True); time.sleep(60)"], stdout=subprocess.PIPE, text=True)
    try:
        assert child.stdout.readline().strip() == "up"
        time.sleep(0.5)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert gpu_guard.check("cuda:0", policy="warn", once=False) == base + 1
        assert any("other compute process" in str(x.message) for x in w)
        with pytest.raises(gpu_guard.SharedGpuError):
            gpu_guard.check("cuda:0", policy="refuse", once=False)
        assert gpu_guard.check("cuda:0", policy="allow", once=False) is None
    finally:
        child.kill()
        child.wait()


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_length_limit_is_refused_loudly(prec):
    """N <= 1024 is the compiled limit of the attention tilings (both the register kernels and the LDS-score kernels of attention.hip):
    N = 1024 runs in the fp16 mode (N = 1000, config 5's length, in the fp32 mode), N = 1100 raises FdiptError (FDIPT_ESIZE) instead of
    computing something else."""
    from framedipt_amd import _lib, config
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision=prec).load_synthetic(5).to("cuda")
    for n, ok in ((1024 if prec == "fp16" else 1000, True), (1100, False)):
        ds = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": 1}), d, "cuda")
        feats = dict(ds[0][2])
        feats["t"] = torch.ones(1, device="cuda")
        if ok:
            out = net(feats)
            assert torch.isfinite(out["rigids"]).all() and torch.isfinite(out["rot_score"]).all()
        else:
            with pytest.raises(_lib.FdiptError, match="FDIPT_ESIZE"):
                net(feats)
        del ds, feats
        torch.cuda.empty_cache()


def test_bench_line_of_the_driver_command_shape():
    """bench.py with the driver's flag shape (--gpus 1 --steps K --warmup W; here on config c2 and without the slow sub-records' defaults
    changed): ONE JSON line carrying the contract keys, the roofline and loop records, the flat copies of the sub-records' values — and the
    timed region went through graph replays (the default product path), K consecutive steps."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "3", "--config", "c2", "--no-cpu-baseline",
           "--reference-steps", "2"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "loop", "whole_forward_frac", "fp32_value", "fp32_whole_forward_frac"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["dtype"] == "fp16" and d["vs_baseline"] is None
    assert d["loop"]["graph"] is True and d["loop"]["host_enqueue_ms_per_step"] < d["ms_per_step"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and r["launches_timed"] == 3 * 4 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "12 consecutive steps" in d["config"]["workload"]
    assert abs(d["value"] - 8 * 128 * 12 / (d["ms_per_step"] * 12e-3)) / d["value"] < 1e-6


def test_batch_of_64_n300_keeps_a_samples_bits():
    """bench.py's all_samples_one_gpu batch (64 samples at N = 300, fp16 mode): samples 0, 31 and 63 of the batch are torch.equal to their B = 1
    runs, sample 0 is the reference golden — the 64-sample sub-record times the kernels the parity suite pins."""
    from test_gpu_sizes import FP16_BOUND
    G = load_golden("fwd_full_denovo_n300_t50.npz")
    net, d, conf = _net("full_denovo_n300_t50", G, "fp16")
    feats = _other_inputs(_feats(G), 64, 77)
    keys = ("rigids", "psi", "rot_score", "trans_score", "atom37")
    fb = {k: torch.cat([f[k] for f in feats], 0) for k in feats[0]}
    out = {k: v.clone() for k, v in net(fb).items() if k in keys}
    for s in (0, 31, 63):
        one = net(feats[s])
        for kk in keys:
            assert torch.equal(out[kk][s], one[kk][0]), (s, kk, float((out[kk][s] - one[kk][0]).abs().max()))
    assert kabsch_free_rmsd(out["atom37"][:1].cpu().numpy(), G["out_atom37"]) < FP16_BOUND["bb_rmsd"]


@pytest.mark.parametrize("n,n_pad", [(300, 320), (324, 384), (388, 512), (516, 640), (644, 768), (772, 960), (964, 1024)])
def test_padding_inside_a_kernel_class_keeps_a_samples_bits(n, n_pad):
    """sharding.kernel_class / fdipt_kernel_class_bounds: a sample padded (res_mask = 0 rows, identity frames) to any length of its own
    kernel-selection class gives, on its real residues, the BITS of its unpadded run — one forward in the fp16 mode at the two ends of every
    class (the dispatch boundaries 320 / 384 / 512 / 640 / 768 / 960 / 1024 of o_pair, attention, sequence attention and the 16-row
    node path).  This is what lets run_sharded batch different lengths without a sample's result depending on its batch mates."""
    from framedipt_amd import config, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    assert sharding.kernel_class(n) == sharding.kernel_class(n_pad)
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(5).to("cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": 1}), d, "cuda")
    item = sharding.seeded_item(ds, 0, 13, d, 3, 0.01)
    feats = dict(item[2])
    feats["t"] = torch.full((1,), 0.4, device="cuda")
    gen = torch.Generator().manual_seed(n)
    feats["sc_ca_t"] = (9.0 * torch.randn(1, n, 3, generator=gen)).cuda().to(feats["sc_ca_t"].dtype)
    padded, _ = sharding.pad_item(feats, item[3], n_pad)
    padded["t"] = feats["t"]
    keys = ("rigids", "psi", "rot_score", "trans_score", "atom37")
    one = {k: v.clone() for k, v in net(feats).items() if k in keys}
    two = net(padded)
    for k in keys:
        assert torch.equal(two[k][:, :n], one[k]), (k, float((two[k][:, :n] - one[k]).abs().max()))
    assert float(two["rigids"][:, n:, 4:].abs().max()) < 1e30  # (padded rows: finite, never read back)
