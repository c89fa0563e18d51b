"""GPU parity tests (-m gpu) added in round 6: o_pair from the pair_z image that the producers of z emit (edge embedder / EdgeTransition
epilogues -> opair_pz_kernel), the last EdgeTransition launch that no longer stores z', and the size boundary of that path."""
import numpy as np
import pytest
import torch

from conftest import kabsch_free_rmsd, load_golden
from test_gpu_parity import _feats
from test_gpu_robustness import _setup

pytestmark = pytest.mark.gpu


def _host(v):
    return v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


def _forward(G, kernel_flags, trace=False):
    from framedipt_amd import config
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision="fp16", kernel_flags=kernel_flags).load_synthetic(int(G["weight_seed"]), float(G["bb_gain"])).to("cuda")
    out = net(_feats(G), trace=trace)
    return {k: v.cpu().numpy().copy() for k, v in out.items()}


@pytest.mark.parametrize("name", ["fwd_full_denovo_n64", "fwd_full_denovo_n128", "fwd_full_denovo_n300_t50"])
def test_pair_z_path_against_the_pass_over_z(name):
    """Default forward (pair bias and pair_z = down_z(z') + b from the epilogues that produce z, o_pair = sum_j a pair_z on 32 channels,
    csrc/attention.hip: opair_pz_kernel; the last EdgeTransition launch stores no z') against FDIPT_KF_PASS_Z (o_pair as round 5 ran it:
    its own pass over the 128 channels of z, sum_j a z first and down_z on split operands afterwards, ipa_pytorch.py:317-322; everything
    else unchanged): the same algebra — down_z commutes with the key sum — up to the fp16 rounding of pair_z, which the attention
    averages over keys.  Both within the fp16 bounds of the reference golden.  (Against FDIPT_KF_UNFOLDED, which also moves other folds,
    the two differ by 1.5e-4 in the node rows and 3.3e-4 A in the frames with AND without the pair_z path.)"""
    from framedipt_amd import _lib
    G = load_golden(name + ".npz")
    a, b = _forward(G, 0, trace=True), _forward(G, _lib.KF_PASS_Z, trace=True)
    assert not np.array_equal(a["trace_node"][1], b["trace_node"][1])  # (two different sets of kernels did run)
    rels = []
    for blk in range(1, 5):
        rel = np.linalg.norm(a["trace_node"][blk] - b["trace_node"][blk]) / np.linalg.norm(b["trace_node"][blk])
        rels.append(float(rel))
        assert rel < 5e-5, (blk, rel)  # measured 2e-6 ... 2e-5 (N = 300 ... 64)
    # the pair representation itself (fp32 traces of every EdgeTransition epilogue, incl. the last one whose z' is not stored by default;
    # with a trace requested it is) feels the other o_pair only through the node rows of the previous block
    for blk in range(1, 4):
        rel = np.linalg.norm(a["trace_edge"][blk] - b["trace_edge"][blk]) / np.linalg.norm(b["trace_edge"][blk])
        rels.append(float(rel))
        assert rel < 1e-3, (blk, rel)
    dca = float(np.abs(a["rigids"][..., 4:] - b["rigids"][..., 4:]).max())
    print(f"{name}: pair_z path vs pass over z: node rel per block / edge rel per EdgeTransition [{' '.join(f'{r:.1e}' for r in rels)}], frames max {dca:.2e} A")
    assert dca < 1e-4  # measured 1.5e-5 ... 2.6e-5 A
    for o in (a, b):
        assert kabsch_free_rmsd(o["atom37"], G["out_atom37"]) < 5e-4


@pytest.mark.parametrize("name", ["fwd_full_denovo_n64", "fwd_full_denovo_n300_t50"])
def test_last_edge_transition_without_its_z_store_changes_no_output(name):
    """With a trace the last EdgeTransition keeps its z' store, without one it drops it (nothing reads z' behind it: the last block's attention
    has its pair bias and pair_z from the epilogue): every output of the forward is bit-identical between the two."""
    G = load_golden(name + ".npz")
    a, b = _forward(G, 0, trace=False), _forward(G, 0, trace=True)
    for k in ("rigids", "psi", "rot_score", "trans_score", "atom37"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


@pytest.mark.parametrize("n,b", [(16, 2), (20, 3), (12, 2), (32, 1)])
def test_pair_z_path_at_its_size_boundary(n, b):
    """N = 16 is the smallest length on the pair_z path (below it attention3 hands its weights over as fp32 rows and o_pair stays the pass
    over z: N = 12; the EdgeTransition launch then keeps its z' store); N = 20: key groups of four that do not fill the 16-key step of
    opair_pz_kernel.  Half-precision forward within 5e-4 A backbone RMSD of the fp32 mode on the same inputs, twice with the same bits."""
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.model.score_network import BatchState
    net32, st32, args = _setup(n, b, "fp32")
    st32.forward(*args)
    torch.cuda.synchronize()
    ref = st32.atom37.double().cpu().numpy()[:, :, [0, 1, 2, 4]]
    net = ScoreNetwork(net32._model_conf, net32.diffuser, precision="fp16").load_synthetic(7).to("cuda")
    st = BatchState(net, st32.seq_idx)
    runs = []
    for _ in range(2):
        st.forward(*args)
        torch.cuda.synchronize()
        runs.append(st.atom37.double().cpu().numpy()[:, :, [0, 1, 2, 4]])
    assert np.isfinite(runs[0]).all()
    np.testing.assert_array_equal(runs[0], runs[1])
    rmsd = np.sqrt(((runs[0] - ref) ** 2).sum(-1).mean(axis=(1, 2))).max()
    assert rmsd < 5e-4, (n, b, rmsd)


@pytest.mark.parametrize("n,b", [(30, 2), (45, 1)])
def test_inference_fn_pads_lengths_that_are_no_multiple_of_four(n, b):
    """inference_fn(pad_to_four=True, the default) runs a sample of N % 4 != 0 residues with masked pad rows on the fast pair kernels
    (edge_transition4 + the pair_z path) and returns arrays of the ORIGINAL length: bit-identical to the same sample padded by hand
    (sharding.pad_item: what run_sharded does), and within the half-precision bounds of the un-padded run on the fall-back kernels."""
    from framedipt_amd import config, sharding
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.inference import draw_noise_tape, inference_fn
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": b}), d, "cuda")
    T = 4
    feats, tape = sharding.stack_items([sharding.seeded_item(ds, i, 5, d, T, 0.01) for i in range(b)])
    kw = dict(num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1)
    auto = inference_fn(net, d, feats, noise_tape=tape, **kw)
    for k, v in auto.items():
        assert v.shape[2] == n, (k, v.shape)
    n_pad = -(-n // 4) * 4
    fp, tp = sharding.pad_item(feats, tape, n_pad)
    hand = inference_fn(net, d, fp, noise_tape=tp, **kw)
    for k in auto:
        np.testing.assert_array_equal(_host(auto[k]), _host(hand[k])[:, :, :n], err_msg=k)
    plain = inference_fn(net, d, feats, noise_tape=tape, pad_to_four=False, **kw)
    assert kabsch_free_rmsd(np.asarray(auto["prot_traj"][-1]), np.asarray(plain["prot_traj"][-1])) < 1e-3
    # without a tape the draws follow the reference's np.random order for the REAL residues (the pad rows draw nothing)
    np.random.seed(11)
    t_ref = draw_noise_tape(d, T - 1, b, n)
    np.random.seed(11)
    a2 = inference_fn(net, d, feats, **kw)
    a3 = inference_fn(net, d, feats, noise_tape=t_ref, **kw)
    np.testing.assert_array_equal(np.asarray(a2["prot_traj"]), np.asarray(a3["prot_traj"]))


@pytest.mark.parametrize("n,b", [(37, 3), (5, 7), (100, 1), (44, 2)])
@pytest.mark.parametrize("prec,tol", [("fp32", 3e-4), ("fp16", 4e-3)])
def test_edge_transition_entry_at_ragged_sizes(prec, tol, n, b):
    """fdipt_edge_transition_fwd (ipa_pytorch.py:84-102 times the pair mask, :549) against the NumPy oracle's EdgeTransition on random
    node rows, pair rows and a residue mask with holes, at sizes where the row tiles of the kernels are ragged: B N^2 no multiple of 32 (the
    fp32 kernel's last 32-pair row tile; its movers' stepped (i, j) arithmetic and its LDS-DMA weight stream run here exactly as in the
    sampler), N < 8 (several sample rows per tile), N % 4 != 0 (edge_transition3 in half precision) and N % 4 == 0 (edge_transition4)."""
    import ctypes as C
    from framedipt_amd import _lib
    from test_gpu_parity import _net, dev
    from test_oracle_forward import _model as omodel
    lib = _lib.load()
    name = "full_denovo_n64"
    G = load_golden(f"fwd_{name}.npz")
    net, d, conf = _net(name, G, prec)
    onet, _ = omodel(name, G, None)
    rng = np.random.default_rng(100 * n + b)
    node = rng.standard_normal((b, n, 256)).astype(np.float32)
    z = rng.standard_normal((b, n, n, 128)).astype(np.float32)
    mask = (rng.random((b, n)) > 0.2).astype(np.float32)
    blk = 1
    zt = torch.float32 if prec == "fp32" else torch.float16
    z_d = dev(z).to(zt).contiguous()
    ref = onet.edge_transition(blk, node, z_d.float().cpu().numpy()) * (mask[:, :, None] * mask[:, None, :])[..., None]
    st = net.batch_state(dev(np.tile(np.arange(n, dtype=np.int64), (b, 1))))
    z2 = torch.full_like(z_d, float("nan"))
    dm, pr, dr = C.byref(net.dims), _lib.ptr(net.params), _lib.ptr(net.derived)
    node_d, mask_d = dev(node), dev(mask)
    _lib.check(lib.fdipt_edge_transition_fwd(dm, pr, dr, blk, b, n, _lib.ptr(node_d), _lib.ptr(mask_d), _lib.ptr(z_d), _lib.ptr(z2),
                                             _lib.ptr(st.ws), st.ws_bytes, _lib.stream_ptr()))
    torch.cuda.synchronize()
    got = z2.float().cpu().numpy()
    assert np.isfinite(got).all()
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert err < tol, (prec, n, b, err)
    # in place (z_out may alias z_in) gives the same bits
    _lib.check(lib.fdipt_edge_transition_fwd(dm, pr, dr, blk, b, n, _lib.ptr(node_d), _lib.ptr(mask_d), _lib.ptr(z_d), _lib.ptr(z_d),
                                             _lib.ptr(st.ws), st.ws_bytes, _lib.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(z_d.float().cpu().numpy(), got)
