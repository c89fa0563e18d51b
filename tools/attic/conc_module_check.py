"""Concurrent per-module calls (IPA attention / EdgeTransition / points) of two sub-batches on two HIP streams vs sequential."""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding, _lib
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.model.score_network import BatchState
N, B, what = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
lib = _lib.load()
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
g = torch.Generator().manual_seed(1)
halves = []
for h in range(2):
    b = B // 2
    seq = torch.arange(N)[None].repeat(b, 1).cuda()
    st = BatchState(net, seq)
    node = torch.randn(b, N, 256, generator=g).cuda()
    z = torch.randn(b, N, N, 128, generator=g).cuda().half().contiguous()
    q = torch.nn.functional.normalize(torch.randn(b, N, 4, generator=g), dim=-1)
    rig = torch.cat([q, 10 * torch.randn(b, N, 3, generator=g)], -1).cuda().contiguous()
    mask = torch.ones(b, N).cuda()
    out = torch.empty(b, N, 256).cuda()
    z2 = torch.empty_like(z)
    qp, kp, vp = torch.empty(b, N, 8, 8, 3).cuda(), torch.empty(b, N, 8, 8, 3).cuda(), torch.empty(b, N, 8, 12, 3).cuda()
    halves.append(dict(st=st, node=node, z=z, rig=rig, mask=mask, out=out, z2=z2, qp=qp, kp=kp, vp=vp, b=b))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
dm, pr, dr = C.byref(net.dims), _lib.ptr(net.params), _lib.ptr(net.derived)
def call(hv, blk, what=what):
    st = hv["st"]; sp = _lib.stream_ptr(); P = _lib.ptr
    if what == "mix":  # EdgeTransition of one half next to the IPA block of the other
        return call(hv, blk, "et" if hv is halves[0] else "ipa")
    if what == "ipa":
        _lib.check(lib.fdipt_ipa_attention_fwd(dm, pr, dr, blk, hv["b"], N, P(hv["node"]), P(hv["z"]), P(hv["rig"]), P(hv["mask"]), P(hv["out"]), P(st.ws), st.ws_bytes, sp))
    elif what == "points":
        _lib.check(lib.fdipt_ipa_project_points(dm, pr, dr, blk, hv["b"], N, P(hv["node"]), P(hv["rig"]), P(hv["mask"]), P(hv["qp"]), P(hv["kp"]), P(hv["vp"]), P(st.ws), st.ws_bytes, sp))
    else:
        _lib.check(lib.fdipt_edge_transition_fwd(dm, pr, dr, blk, hv["b"], N, P(hv["node"]), P(hv["mask"]), P(hv["z"]), P(hv["z2"]), P(st.ws), st.ws_bytes, sp))
def outs(hv):
    w = what if what != "mix" else ("et" if hv is halves[0] else "ipa")
    return [hv[k].float().cpu().numpy().copy() for k in (("out",) if w == "ipa" else ("qp", "kp", "vp") if w == "points" else ("z2",))]
def run(conc):
    for rep in range(4):
        for hv, s in zip(halves, streams):
            with torch.cuda.stream(s if conc else streams[0]):
                call(hv, rep % 3)
    torch.cuda.synchronize()
    return [outs(hv) for hv in halves]
ref = run(False)
bad = 0
for rep in range(20):
    got = run(True)
    for h in range(2):
        dmax = max(float(np.abs(a - b).max()) for a, b in zip(got[h], ref[h]))
        if dmax > 0:
            bad += 1; print("rep", rep, "half", h, "max diff", dmax)
print(what, "N", N, "B", B, "bad", bad, "of 40")
