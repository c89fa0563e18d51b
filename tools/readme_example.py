import sys; sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sampler
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.inference import inference_fn
from framedipt_amd.output import save_traj
import os
cfg = config.base_config()
diffuser = SE3Diffuser(cfg.diffuser, device="cuda")
model = ScoreNetwork(cfg.model, diffuser, precision="fp16").load_synthetic(7).to("cuda")
ds = sampler.UnconditionalSampler(config.to_conf({"min_length": 64, "max_length": 64, "length_step": 1, "samples_per_length": 8}), diffuser, "cuda")
length, sample_id, feats = ds[0]
out = inference_fn(model, diffuser, feats, num_t=10, min_t=0.01, noise_scale=0.1, aux_traj=True)
os.makedirs("/tmp/samples", exist_ok=True)
paths = save_traj(out["prot_traj"][:, 0], out["rigid_0_traj"][:, 0], (1 - feats["fixed_mask"][0]).cpu().numpy(), "/tmp/samples/", 0)
print(paths, out["prot_traj"].shape)
