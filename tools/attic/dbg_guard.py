"""Out-of-bounds check of the forward: the workspace and every output buffer sit between guard bands (developer aid)."""
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
G = 1 << 20
guards = {}
def guard(name, t):
    nb = t.numel() * t.element_size()
    big = torch.full((G + nb + G,), 0xAB, dtype=torch.uint8, device="cuda")
    guards[name] = (big, nb)
    return big[G:G + nb].view(t.dtype).view(t.shape)
st.ws = guard("ws", st.ws)
for nm in ("psi", "rot_score", "trans_score", "rigids", "atom37", "atom14", "setup"):
    setattr(st, nm, guard(nm, getattr(st, nm)))
# (the setup table was computed into the old buffer: recompute into the guarded one)
st2 = BatchState(net, feats["seq_idx"]); st.setup.copy_(st2.setup)
t32, temb, sig = net.step_scalars(np.full(B, 0.5))
f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()
ca = guard("ca_out", torch.empty(B, N, 3, device="cuda"))
args = (f32(feats["rigids_t"]), f32(feats["res_mask"]), f32(feats["fixed_mask"]), f32(feats["sc_ca_t"]) + 1.0, None,
        f32(feats["torsion_angles_sin_cos"][..., 2, :]), torch.as_tensor(t32, device="cuda"), torch.as_tensor(temb, device="cuda"),
        torch.as_tensor(sig, device="cuda"))
for rep in range(2):
    st.forward(*args, ca_out=ca)
torch.cuda.synchronize()
ok = True
for name, (big, nb) in guards.items():
    lo, hi = big[:G].cpu().numpy(), big[G + nb:].cpu().numpy()
    for tag, arr in (("before", lo), ("after", hi)):
        badidx = np.nonzero(arr != 0xAB)[0]
        if len(badidx):
            ok = False
            print(f"OOB write {tag} {name}: {len(badidx)} bytes, offsets {badidx.min()}..{badidx.max()} (relative to the {'start of the guard' if tag == 'after' else 'guard start; buffer begins at ' + str(G)})")
print("N", N, "B", B, prec, "guards intact" if ok else "OUT-OF-BOUNDS WRITES FOUND")
