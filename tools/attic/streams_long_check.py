import os
os.environ["FDIPT_EXPERIMENTAL_STREAMS"] = "1"  # (investigation tool)
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.inference import inference_fn
from framedipt_amd import inference as _inf
_inf.StreamedLoops.MAX_LENGTH = 1 << 30  # (investigation tool: the product refuses N > 384 on sub-batch streams)
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.sampler import UnconditionalSampler
N, B, T = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (300, 8, 100)
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
items = [sharding.seeded_item(ds, i, 3, d, T, 0.01) for i in range(B)]
feats, tape = sharding.stack_items(items)
outs = []
for streams in (1, 2, 2, 1):
    o = inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, streams=streams)
    outs.append(o)
    print("streams", streams, "final CA span", float(np.abs(o["prot_traj"][0]).max()))
for k in outs[0]:
    h = lambda v: v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    for i in (1, 2, 3):
        assert np.array_equal(h(outs[0][k]), h(outs[i][k])), (k, i)
print(f"T={T} N={N} B={B} trajectories bit-identical for 1 / 2 streams and on repetition")
