"""One module (victim: ipa / et / points) repeated on one HIP stream while FULL forwards of another batch run on a second stream:
which module's result is disturbed by a concurrent forward?   python tools/conc_victim_check.py N B what [reps]"""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding, _lib
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.model.score_network import BatchState
from framedipt_amd.sampler import UnconditionalSampler
N, B, what = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lib = _lib.load()
conf = config.base_config()
import os
if os.environ.get("CONC_BLOCKS"):
    conf.model.ipa.num_blocks = int(os.environ["CONC_BLOCKS"])
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
b = B // 2
g = torch.Generator().manual_seed(1)
seq = torch.arange(N)[None].repeat(b, 1).cuda()
vst = BatchState(net, seq)
node = torch.randn(b, N, 256, generator=g).cuda()
z = torch.randn(b, N, N, 128, generator=g).cuda().half().contiguous()
q = torch.nn.functional.normalize(torch.randn(b, N, 4, generator=g), dim=-1)
rig = torch.cat([q, 10 * torch.randn(b, N, 3, generator=g)], -1).cuda().contiguous()
mask = torch.ones(b, N).cuda()
out = torch.empty(b, N, 256).cuda()
z2 = torch.empty_like(z)
qp, kp, vp = torch.empty(b, N, 8, 8, 3).cuda(), torch.empty(b, N, 8, 8, 3).cuda(), torch.empty(b, N, 8, 12, 3).cuda()
dm, pr, dr = C.byref(net.dims), _lib.ptr(net.params), _lib.ptr(net.derived)
P = _lib.ptr
def victim(blk):
    sp = _lib.stream_ptr()
    if what == "ipa":
        _lib.check(lib.fdipt_ipa_attention_fwd(dm, pr, dr, blk, b, N, P(node), P(z), P(rig), P(mask), P(out), P(vst.ws), vst.ws_bytes, sp))
    elif what == "points":
        _lib.check(lib.fdipt_ipa_project_points(dm, pr, dr, blk, b, N, P(node), P(rig), P(mask), P(qp), P(kp), P(vp), P(vst.ws), vst.ws_bytes, sp))
    else:
        _lib.check(lib.fdipt_edge_transition_fwd(dm, pr, dr, blk, b, N, P(node), P(mask), P(z), P(z2), P(vst.ws), vst.ws_bytes, sp))
def outs():
    return [t.float().cpu().numpy().copy() for t in ((out,) if what == "ipa" else (qp, kp, vp) if what == "points" else (z2,))]
# aggressor: full forwards of another batch
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": b}), d, "cuda")
feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(b)])
t32, temb, sig = net.step_scalars(np.full(b, 0.5))
f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()
ast = BatchState(net, feats["seq_idx"])
aargs = (f32(feats["rigids_t"]), f32(feats["res_mask"]), f32(feats["fixed_mask"]), f32(feats["sc_ca_t"]) + 1.0, None,
         f32(feats["torsion_angles_sin_cos"][..., 2, :]), torch.as_tensor(t32, device="cuda"), torch.as_tensor(temb, device="cuda"),
         torch.as_tensor(sig, device="cuda"))
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
if what.startswith("fwd-"):  # roles exchanged: the full forward is the victim, the module after the dash the aggressor
    what = what[4:]
    ast2 = BatchState(net, feats["seq_idx"], trace=True)
    tiny = torch.zeros(256, device="cuda")
    biga, bigb = torch.zeros(64 << 20, device="cuda"), torch.zeros(64 << 20, device="cuda")
    def fouts():
        return [ast2.trace_node.cpu().numpy().copy(), ast2.rigids.cpu().numpy().copy(), ast2.psi.cpu().numpy().copy()]
    with torch.cuda.stream(s0):
        ast2.forward(*aargs)
    torch.cuda.synchronize()
    ref = fouts()
    ws_ref = ast2.ws.cpu().numpy().copy().view(np.uint8)
    layout = []
    if os.environ.get("CONC_LAYOUT"):  # "name offset" lines captured from FDIPT_DUMP_LAYOUT
        for line in open(os.environ["CONC_LAYOUT"]):
            if line.startswith("FDIPT_LAYOUT"):
                _, nm, off = line.split()
                layout.append((nm, int(off)))
        layout = sorted(set(layout), key=lambda t: t[1])
    bad = 0
    for rep in range(REPS):
        with torch.cuda.stream(s1):
            if what == "tiny":      # many short launches of an unrelated kernel: only kernel boundaries on the second queue
                for _ in range(1500):
                    tiny.add_(1.0)
            elif what == "big":     # one unrelated kernel stream with heavy memory traffic
                for _ in range(60):
                    bigb.copy_(biga)
            else:
                for _ in range(40 if what != "fwd" else 2):
                    ast.forward(*aargs) if what == "fwd" else victim(0)
        with torch.cuda.stream(s0):
            for _ in range(2):
                ast2.forward(*aargs)
        torch.cuda.synchronize()
        got = fouts()
        dmax = [float(np.abs(a - c).max()) for a, c in zip(got, ref)]
        if max(dmax) > 0:
            bad += 1
            first = next(k for k in range(5) if np.abs(got[0][k] - ref[0][k]).max() > 0)
            if layout:
                ws_now = ast2.ws.cpu().numpy().view(np.uint8)
                diff = ws_now != ws_ref
                rep_l = []
                for (nm, off), (_, nxt) in zip(layout[:-1], layout[1:]):
                    nd = int(diff[off:nxt].sum())
                    if nd:
                        first_b = int(np.argmax(diff[off:nxt]))
                        rep_l.append(f"{nm}:{nd}B(first@{first_b})")
                print("   workspace buffers differing:", " ".join(rep_l))
                lay = dict(layout)
                for nm in ("qp", "kp", "vp", "pts", "rot"):
                    if nm in lay:
                        nxt = min(o for _, o in layout if o > lay[nm])
                        a_ = ws_now[lay[nm]:nxt].view(np.float32); r_ = ws_ref[lay[nm]:nxt].view(np.float32)
                        idx = np.nonzero(a_ != r_)[0]
                        if len(idx):
                            print(f"   {nm}: {len(idx)} floats differ; idx {idx[:12].tolist()} got {a_[idx[:6]].tolist()} ref {r_[idx[:6]].tolist()}")
            dd = np.abs(got[0][first] - ref[0][first])  # [b, n, c]
            rows = dd.max(-1)
            bs, ns = np.nonzero(rows > 0.2 * rows.max())
            print("rep", rep, "first differing node trace", first, "max diffs", [f"{x:.1e}" for x in dmax], "elements differing", f"{float((dd > 0).mean()):.3f}",
                  "strong rows (sample, row):", list(zip(bs.tolist(), ns.tolist()))[:24])
    print("victim forward, aggressor", what, "bad", bad, "of", REPS)
    sys.exit(0)
with torch.cuda.stream(s0):
    victim(0)
torch.cuda.synchronize()
ref = outs()
bad = 0
for rep in range(REPS):
    with torch.cuda.stream(s1):
        for _ in range(2):
            ast.forward(*aargs)
    with torch.cuda.stream(s0):
        for _ in range(6):
            victim(0)
    torch.cuda.synchronize()
    got = outs()
    dmax = max(float(np.abs(a - c).max()) for a, c in zip(got, ref))
    if dmax > 0:
        bad += 1
        print("rep", rep, "max diff", dmax)
print("victim", what, "N", N, "B", B, "bad", bad, "of", REPS)
