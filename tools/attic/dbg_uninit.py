"""Does the forward's result depend on workspace bytes it did not write itself?  Same forward with the workspace pre-filled with
zeros / 0xFF (NaN patterns) / random bytes (developer aid)."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.model.score_network import BatchState
from framedipt_amd.sampler import UnconditionalSampler
N, B, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision=prec).load_synthetic(7).to("cuda")
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(B)])
st = BatchState(net, feats["seq_idx"])
t32, temb, sig = net.step_scalars(np.full(B, 0.5))
f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()
args = (f32(feats["rigids_t"]), f32(feats["res_mask"]), f32(feats["fixed_mask"]), f32(feats["sc_ca_t"]) + 1.0, None,
        f32(feats["torsion_angles_sin_cos"][..., 2, :]), torch.as_tensor(t32, device="cuda"), torch.as_tensor(temb, device="cuda"),
        torch.as_tensor(sig, device="cuda"))
res = {}
for tag in ("zeros", "ff", "random", "zeros2"):
    if tag.startswith("zeros"): st.ws.zero_()
    elif tag == "ff": st.ws.fill_(0xFF)
    else: st.ws.copy_(torch.randint(0, 256, (st.ws.numel(),), dtype=torch.uint8, device="cuda"))
    st.forward(*args)
    torch.cuda.synchronize()
    res[tag] = {k: getattr(st, k).cpu().numpy().copy() for k in ("rigids", "psi", "rot_score", "trans_score", "atom37")}
for tag in ("ff", "random", "zeros2"):
    print(N, B, prec, tag, {k: float(np.nan_to_num(np.abs(res[tag][k] - res["zeros"][k]), nan=1e9).max()) for k in res[tag]})
