"""GPU robustness tests (-m gpu) of the score-network forward: no write outside the caller's buffers, no dependence on bytes of
the workspace the forward did not write itself, bit-reproducible results."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n, b, prec):
    from framedipt_amd import config, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.model.score_network import BatchState
    from framedipt_amd.sampler import UnconditionalSampler
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision=prec).load_synthetic(7).to("cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": b}), d, "cuda")
    feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(b)])
    st = BatchState(net, feats["seq_idx"])
    t32, temb, sig = net.step_scalars(np.full(b, 0.5))
    f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()  # noqa: E731
    args = (f32(feats["rigids_t"]), f32(feats["res_mask"]), f32(feats["fixed_mask"]), f32(feats["sc_ca_t"]) + 1.0, None,
            f32(feats["torsion_angles_sin_cos"][..., 2, :]), torch.as_tensor(t32, device="cuda"), torch.as_tensor(temb, device="cuda"),
            torch.as_tensor(sig, device="cuda"))
    return net, st, args


OUT = ("rigids", "psi", "rot_score", "trans_score", "atom37", "atom14")


@pytest.mark.parametrize("n,b,prec", [(100, 2, "fp16"), (301, 2, "fp16"), (300, 3, "fp16"), (64, 1, "fp32")])
def test_forward_writes_inside_its_buffers_only(n, b, prec):
    """Workspace and every output buffer between 1 MiB guard bands: the bands are intact after two forwards."""
    net, st, args = _setup(n, b, prec)
    G, guards = 1 << 20, {}

    def guard(name, t):
        nb = t.numel() * t.element_size()
        big = torch.full((G + nb + G,), 0xAB, dtype=torch.uint8, device="cuda")
        guards[name] = (big, nb)
        return big[G:G + nb].view(t.dtype).view(t.shape)

    setup = st.setup.clone()
    st.ws = guard("ws", st.ws)
    for nm in OUT + ("setup",):
        setattr(st, nm, guard(nm, getattr(st, nm)))
    st.setup.copy_(setup)
    ca = guard("ca_out", torch.empty(b, n, 3, device="cuda"))
    for _ in range(2):
        st.forward(*args, ca_out=ca)
    torch.cuda.synchronize()
    for name, (big, nb) in guards.items():
        assert bool((big[:G] == 0xAB).all()) and bool((big[G + nb:] == 0xAB).all()), f"write outside {name}"


@pytest.mark.parametrize("n,b,prec", [(100, 2, "fp16"), (301, 2, "fp16"), (300, 3, "fp16"), (300, 2, "fp16"), (300, 8, "fp16"), (128, 3, "fp16"),
                                      (724, 2, "fp16"), (64, 1, "fp32")])
def test_forward_independent_of_workspace_contents(n, b, prec):
    """Same forward with the workspace pre-filled with zeros, 0xFF (NaN patterns) and random bytes: bit-identical outputs - the
    forward reads nothing it did not write (pads and never-written slots are zeroed by the forward itself)."""
    net, st, args = _setup(n, b, prec)
    res = {}
    for tag in ("zeros", "ff", "random", "zeros_again"):
        if tag.startswith("zeros"):
            st.ws.zero_()
        elif tag == "ff":
            st.ws.fill_(0xFF)
        else:
            st.ws.copy_(torch.randint(0, 256, (st.ws.numel(),), dtype=torch.uint8, device="cuda"))
        st.forward(*args)
        torch.cuda.synchronize()
        res[tag] = {k: getattr(st, k).clone() for k in OUT}
    for tag in ("ff", "random", "zeros_again"):
        for k in OUT:
            assert torch.equal(res[tag][k], res["zeros"][k]), (tag, k)


def test_concurrent_forwards_are_bit_identical():
    """Two forwards of this library in flight on two HIP streams == the same forwards one after the other, bit for bit.  Regression
    test of the round-2 finding: a merged `global_store_dwordx3` in points16_kernel (three floats of a point) read its data registers
    after the next point's arithmetic had overwritten the first of them whenever another kernel contended the CU's memory pipeline -
    ~28 % of the forwards that ran next to another forward differed (one residue's point x coordinates, then the sample's whole
    node representation through the attention).  DESIGN.md section 5; tools/check_store_hazard.py, tools/conc_*.py."""
    from framedipt_amd import config, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.model.score_network import BatchState
    from framedipt_amd.sampler import UnconditionalSampler
    N, B = 128, 8
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
    feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(B)])
    t32, temb, sig = net.step_scalars(np.full(B // 2, 0.5))
    f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()  # noqa: E731
    halves = []
    for lo, hi in ((0, B // 2), (B // 2, B)):
        st = BatchState(net, feats["seq_idx"][lo:hi])
        args = (f32(feats["rigids_t"][lo:hi]), f32(feats["res_mask"][lo:hi]), f32(feats["fixed_mask"][lo:hi]), f32(feats["sc_ca_t"][lo:hi]) + 1.0,
                None, f32(feats["torsion_angles_sin_cos"][lo:hi][..., 2, :]), torch.as_tensor(t32, device="cuda"),
                torch.as_tensor(temb, device="cuda"), torch.as_tensor(sig, device="cuda"))
        halves.append((st, args))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()

    def run(conc):
        for _ in range(3):
            for (st, args), s in zip(halves, streams):
                with torch.cuda.stream(s if conc else streams[0]):
                    st.forward(*args)
        torch.cuda.synchronize()
        return [(st.rigids.cpu().numpy().copy(), st.psi.cpu().numpy().copy(), st.rot_score.cpu().numpy().copy()) for st, _ in halves]

    ref = run(False)
    for rep in range(40):
        got = run(True)
        for h in range(2):
            for a, b in zip(got[h], ref[h]):
                np.testing.assert_array_equal(a, b, err_msg=f"repetition {rep}, sub-batch {h}")


def test_streamed_sub_batches_match_the_single_stream_trajectory():
    """inference_fn(streams=2): sub-batches on two HIP streams give the trajectory of the single-stream batch, bit for bit."""
    from framedipt_amd import config, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.inference import draw_noise_tape, inference_fn
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    N, B, T = 64, 6, 8
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
    feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, T, 0.01) for i in range(B)])
    np.random.seed(11)
    tape = draw_noise_tape(d, T - 1, B, N)
    one = inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    with pytest.raises(ValueError, match="opt in"):  # more than one stream is never silently allowed
        inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, streams=2)
    two = inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, streams=2, experimental_streams=True)
    assert sorted(one) == sorted(two)
    for k in one:
        host = lambda v: v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)  # noqa: E731
        assert host(one[k]).shape == host(two[k]).shape, k
        np.testing.assert_array_equal(host(one[k]), host(two[k]), err_msg=k)


def test_sub_batch_streams_are_refused_beyond_the_verified_range():
    """More than two streams, or two streams at N > 384, are refused (soak results in inference.StreamedLoops / DESIGN.md section 5)."""
    from framedipt_amd import config, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.inference import inference_fn
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
    for n, streams in ((64, 3), (388, 2)):
        ds = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": 4}), d, "cuda")
        feats, tape = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 3, 0.01) for i in range(4)])
        with pytest.raises(ValueError, match="verified bit-identical"):
            inference_fn(net, d, feats, num_t=3, min_t=0.01, noise_tape=tape, streams=streams, experimental_streams=True)
    # the limits apply to the streams that would actually run: a 2-sample batch asked for 3 streams runs on 2 (refused only for its length) ...
    one = {k: (v[:1] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 4 else v) for k, v in feats.items()}
    from framedipt_amd.inference import StreamedLoops
    lp = StreamedLoops(net, d, one, 2, 3, 0.01, noise_tape=tuple(z[:, :1] for z in tape))  # ... and B = 1 at N = 388 is a single stream: allowed
    assert len(lp.loops) == 1


def test_edge_transition_clock_probe():
    """FdiptForwardArgs.clock_out (bench.py's ``roofline.clock_ghz``): opt-in and caller-owned.  With the buffer set, a forward at
    N % 4 == 0 counts the blocks of the num_blocks - 1 EdgeTransition launches and the cycle / tick ratio is a plausible shader
    clock, in both precision modes; the outputs are bit-identical with and without the probe (no state outside the caller's buffers)."""
    for prec in ("fp16", "fp32"):
        net, st, args = _setup(100, 2, prec)
        st.forward(*args)
        torch.cuda.synchronize()
        ref = {k: getattr(st, k).clone() for k in ("rigids", "psi", "rot_score", "trans_score")}
        st.clock_out = torch.zeros(3, dtype=torch.int64, device="cuda")
        st.forward(*args)
        torch.cuda.synchronize()
        cycles, ticks, blocks = (int(v) for v in st.clock_out.cpu())
        st.clock_out = None
        n_launch = net.dims.num_blocks - 1
        assert blocks > 0 and blocks % n_launch == 0 and blocks // n_launch <= 256, (prec, blocks)
        assert 0.3 < cycles / ticks / 10 < 2.6  # GHz (ticks are 10 ns)
        for k, v in ref.items():
            assert torch.equal(getattr(st, k), v), (prec, k)


@pytest.mark.parametrize("n,b", [(12, 3), (44, 2), (132, 3), (260, 2)])
def test_half_precision_forward_tracks_the_fp32_mode_at_odd_sizes(n, b):
    """Sizes between the golden fixtures (N % 8 == 4: EdgeTransition patches straddle samples; N = 12 / 44: the small-N kernel
    selections): the half-precision forward (default pair kernels, and the edge_transition3 fallback via FDIPT_KF_ET3) stays within
    5e-4 A backbone RMSD of the fp32 mode on the same inputs (measured 1.0e-4 ... 1.8e-4: tools/size_sweep.py)."""
    from framedipt_amd import _lib
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.model.score_network import BatchState
    net32, st32, args = _setup(n, b, "fp32")
    st32.forward(*args)
    torch.cuda.synchronize()
    ref = st32.atom37.double().cpu().numpy()[:, :, [0, 1, 2, 4]]
    for kf in (0, _lib.KF_ET3):
        net = ScoreNetwork(net32._model_conf, net32.diffuser, precision="fp16", kernel_flags=kf).load_synthetic(7).to("cuda")
        st = BatchState(net, st32.seq_idx)
        st.forward(*args)
        torch.cuda.synchronize()
        got = st.atom37.double().cpu().numpy()[:, :, [0, 1, 2, 4]]
        assert np.isfinite(got).all()
        rmsd = np.sqrt(((got - ref) ** 2).sum(-1).mean(axis=(1, 2))).max()
        assert rmsd < 5e-4, (n, b, kf, rmsd)


def test_bench_two_ranks_on_one_gpu():
    """bench.py's multi-rank path (the driver launches it with torch.distributed.run on an 8-GPU node): two ranks on this box's one GPU
    (FDIPT_BENCH_ONE_GPU=1: gloo rendezvous, both on cuda:0).  Rank 0 prints ONE JSON line with n_gpus = 2 and the whole-job value."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDIPT_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert "x2" in d["config"]["parallelism"]
