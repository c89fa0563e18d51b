"""Which sub-module, run in a loop on another stream (own model state, own buffers), disturbs a full forward?  (developer aid)"""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding, _lib, embedding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.model.score_network import BatchState
from framedipt_amd.sampler import UnconditionalSampler
N, B, what = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lib = _lib.load()
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(B)])
st = BatchState(net, feats["seq_idx"], trace=True)
t32, temb, sig = net.step_scalars(np.full(B, 0.5))
f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()
args = (f32(feats["rigids_t"]), f32(feats["res_mask"]), f32(feats["fixed_mask"]), f32(feats["sc_ca_t"]) + 1.0, None,
        f32(feats["torsion_angles_sin_cos"][..., 2, :]), torch.as_tensor(t32, device="cuda"), torch.as_tensor(temb, device="cuda"),
        torch.as_tensor(sig, device="cuda"))
# aggressor state
g = torch.Generator().manual_seed(1)
ast = BatchState(net, feats["seq_idx"])
node = torch.randn(B, N, 256, generator=g).cuda(); z = torch.randn(B, N, N, 128, generator=g).cuda().half().contiguous()
rig = f32(feats["rigids_t"]).clone(); mask = torch.ones(B, N).cuda(); out = torch.empty(B, N, 256).cuda(); z2 = torch.empty_like(z)
qp, kp, vp = torch.empty(B, N, 8, 8, 3).cuda(), torch.empty(B, N, 8, 8, 3).cuda(), torch.empty(B, N, 8, 12, 3).cuda()
fa = _lib.ForwardArgs(); fa.B, fa.N, fa.n_rel, fa.rel_off = B, N, ast.n_rel, ast.rel_off
keep = [mask, torch.zeros(B, N).cuda(), torch.zeros(B, N, 3).cuda(), torch.as_tensor(temb, device="cuda")]
for nm, tn in (("res_mask", keep[0]), ("fixed_mask", keep[1]), ("sc_ca_t", keep[2]), ("seq_idx", ast.seq_idx), ("idx_emb", ast.idx_emb), ("t_emb", keep[3]), ("t_emb_eps", ast.t_emb_eps)):
    setattr(fa, nm, _lib.ptr(tn))
dm, pr, dr = C.byref(net.dims), _lib.ptr(net.params), _lib.ptr(net.derived)
P = _lib.ptr
def aggress():
    sp = _lib.stream_ptr()
    if what == "embed": _lib.check(lib.fdipt_edge_embed_fwd(dm, pr, dr, P(ast.setup), C.byref(fa), P(out), P(z2), P(ast.ws), ast.ws_bytes, sp))
    elif what == "ipa": _lib.check(lib.fdipt_ipa_attention_fwd(dm, pr, dr, 1, B, N, P(node), P(z), P(rig), P(mask), P(out), P(ast.ws), ast.ws_bytes, sp))
    elif what == "points": _lib.check(lib.fdipt_ipa_project_points(dm, pr, dr, 1, B, N, P(node), P(rig), P(mask), P(qp), P(kp), P(vp), P(ast.ws), ast.ws_bytes, sp))
    elif what == "et": _lib.check(lib.fdipt_edge_transition_fwd(dm, pr, dr, 1, B, N, P(node), P(mask), P(z), P(z2), P(ast.ws), ast.ws_bytes, sp))
    elif what == "fwd": ast.forward(*args)
s_fwd, s_load = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
def fwd(loaded):
    if loaded:
        with torch.cuda.stream(s_load):
            for _ in range(12): aggress()
    with torch.cuda.stream(s_fwd):
        for _ in range(3): st.forward(*args)
    torch.cuda.synchronize()
    return st.trace_node.cpu().numpy().copy(), st.rigids.cpu().numpy().copy()
ref = fwd(False)
bad = 0
for rep in range(12):
    got = fwd(True)
    dn = [float(np.abs(got[0][i] - ref[0][i]).max()) for i in range(5)]
    if max(dn) > 0 or np.abs(got[1] - ref[1]).max() > 0:
        bad += 1; print("rep", rep, "node diffs", [f"{x:.1e}" for x in dn])
print("aggressor", what, "N", N, "B", B, "bad", bad, "of 12")
