"""Free-running (not teacher-forced) sampling against the NumPy oracle over a whole schedule: the same x_T, weights and noise tape, every
step feeding on the previous step's own output.  The per-step parity tests bound the error of ONE step; this one shows that the errors do
not compound over the reverse steps of inference_fn (small-config network T = 100, full network T = 40; N = 24)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("size,prec,num_t,bar", [("small", "fp32", 100, 1e-4), ("small", "fp16", 100, 5e-3), ("full", "fp32", 40, 1e-4),
                                                 ("full", "fp16", 40, 5e-3)])
def test_free_running_trajectory_tracks_the_oracle(size, prec, num_t, bar):
    from framedipt_amd import config, inference
    from framedipt_amd import weights as W
    from framedipt_amd.diffusion import SE3Diffuser
    from framedipt_amd.model import ScoreNetwork
    from framedipt_amd.sampler import UnconditionalSampler
    from oracle import diffuser as od
    from oracle import inference as oi
    from oracle.score_network import ScoreNetwork as OracleNet

    conf = config.small_config() if size == "small" else config.base_config()  # full: the 17.4 M-parameter network (split operands in fp16)
    n = 24
    diff = SE3Diffuser(conf.diffuser, device="cuda:0")
    net = ScoreNetwork(conf.model, diff, precision=prec).load_synthetic(3).to("cuda:0")
    sampler = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": 1}), diff,
                                   "cuda:0")
    np.random.seed(11)
    _, _, feats = sampler[0]
    tape = inference.draw_noise_tape(diff, num_t - 1, 1, n)
    res = inference.inference_fn(net, diff, feats, num_t=num_t, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    torch.cuda.synchronize()
    tables = dict(np.load(os.path.join(ROOT, "framedipt_amd", "data", "residue_tables.npz")))
    odiff = od.SE3Diffuser(conf.diffuser)
    onet = OracleNet(conf.model, odiff, W.synth_state_dict(W.param_shapes(conf.model), 3), tables=tables)
    ref = oi.inference_fn(onet, odiff, {k: v.cpu().numpy() for k, v in feats.items()}, num_t, 0.01, noise_scale=0.1,
                          noise_tape=[(tape[0][i], tape[1][i]) for i in range(num_t - 1)])
    d = res["prot_traj"][..., :5, :] - ref["prot_traj"][..., :5, :]          # [T, 1, N, 5 backbone atoms, 3], reversed time order
    per_step = np.sqrt((d ** 2).sum(-1).mean(axis=(1, 2, 3)))
    print(f"{size} {prec} T={num_t}: backbone RMSD vs oracle: final structure {per_step[0]:.2e} A, worst step {per_step.max():.2e} A")
    assert per_step[0] < bar and per_step.max() < bar
