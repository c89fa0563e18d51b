import time, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from framedipt_amd import config, weights as W
from oracle import diffuser as od, inference as oi
from oracle.torch_port import TorchScoreNetwork
conf=config.base_config()
tables = dict(np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "framedipt_amd/data/residue_tables.npz")))
odiff=od.SE3Diffuser(conf.diffuser)
net=TorchScoreNetwork(conf.model, odiff, W.synth_state_dict(W.param_shapes(conf.model), 7), tables=tables)
feats=oi.unconditional_feats(odiff,300)
tp=np.ones((1,),dtype=np.float32)
feats=oi.set_t_feats(feats,1.0,tp,odiff)
print("logical cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for k in (1, 4, 8, 16, 32, 64, 128):
    torch.set_num_threads(k)
    net(feats); t0=time.perf_counter(); net(feats); net(feats); print(k, "threads: forward", (time.perf_counter()-t0)/2, flush=True)
