"""Concurrent forwards of two sub-batches on two HIP streams vs the same forwards one after the other: first diverging trace."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.model.score_network import BatchState
from framedipt_amd.sampler import UnconditionalSampler
N, B = int(sys.argv[1]), int(sys.argv[2])
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(B)])
t32, temb, sig = net.step_scalars(np.full(B // 2, 0.5))
f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()
halves = []
import os
SAME = os.environ.get("CONC_SAME")  # both streams run the SAME samples (separate buffers): data cross-talk would then be invisible
for lo, hi in (((0, B // 2), (0, B // 2)) if SAME else ((0, B // 2), (B // 2, B))):
    st = BatchState(net, feats["seq_idx"][lo:hi], trace=True)
    args = (f32(feats["rigids_t"][lo:hi]), f32(feats["res_mask"][lo:hi]), f32(feats["fixed_mask"][lo:hi]), f32(feats["sc_ca_t"][lo:hi]) + 1.0,
            None, f32(feats["torsion_angles_sin_cos"][lo:hi][..., 2, :]), torch.as_tensor(t32, device="cuda"), torch.as_tensor(temb, device="cuda"),
            torch.as_tensor(sig, device="cuda"))
    halves.append((st, args))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
def run(conc, reps=3):
    for _ in range(reps):
        for (st, args), s in zip(halves, streams):
            with torch.cuda.stream(s if conc else streams[0]):
                st.forward(*args)
    torch.cuda.synchronize()
    return [(st.trace_node.cpu().numpy().copy(), st.trace_edge.cpu().numpy().copy(), st.rigids.cpu().numpy().copy(), st.psi.cpu().numpy().copy()) for st, _ in halves]
ref = run(False)
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for rep in range(REPS):
    got = run(True)
    for h in range(2):
        tn = [float(np.abs(got[h][0][b] - ref[h][0][b]).max()) for b in range(5)]
        te = [float(np.abs(got[h][1][b] - ref[h][1][b]).max()) for b in range(4)]
        print(f"rep {rep} half {h}: node {tn} edge {te} rigids {np.abs(got[h][2]-ref[h][2]).max():.1e} psi {np.abs(got[h][3]-ref[h][3]).max():.1e}")
        dn = np.abs(got[h][0][1] - ref[h][0][1]).max(-1)  # [b, n]
        if dn.max() > 0:
            bad = np.argwhere(dn > 0)
            for sb in sorted(set(bad[:, 0].tolist()))[:2]:
                dd = np.abs(got[h][0][1][sb] - ref[h][0][1][sb])  # [n, c]
                rows = dd.max(-1)
                print(f"   sample {sb}: elements differing {float((dd > 0).mean()):.3f}; row max diff median {np.median(rows):.2e} max {rows.max():.2e} "
                      f"argmax row {int(rows.argmax())}; per 64-column block max {[float(f'{dd[:, c:c+64].max():.1e}') for c in range(0, dd.shape[1], 64)]}; "
                      f"top rows {np.argsort(-rows)[:6].tolist()}")
            print("   node[1] rows differing:", len(bad), "samples", sorted(set(bad[:, 0].tolist())), "rows min/max", bad[:, 1].min(), bad[:, 1].max(),
                  "first", bad[:12].tolist())
