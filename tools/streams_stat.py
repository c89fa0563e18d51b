"""Which sub-batch stream counts reproduce the single-stream trajectory bit for bit, how often, and where they first differ.
    python tools/streams_stat.py N B T reps [kernel_flags]"""
import os
os.environ["FDIPT_EXPERIMENTAL_STREAMS"] = "1"  # (investigation tool)
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd import inference
from framedipt_amd.inference import inference_fn
inference.StreamedLoops.MAX_STREAMS = 8  # (investigation tool: the product refuses more than two streams ...)
inference.StreamedLoops.MAX_LENGTH = 1 << 30  # (... and N > 384)
if len(sys.argv) > 7:  # reserve_cus override (argv[7])
    _init = inference.StreamedLoops.__init__
    def _forced(self, *a, **k):
        k["reserve_cus"] = int(sys.argv[7])
        _init(self, *a, **k)
    inference.StreamedLoops.__init__ = _forced
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.sampler import UnconditionalSampler
N, B, T, reps = (int(v) for v in sys.argv[1:5])
kf = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
STREAMS = tuple(int(c) for c in sys.argv[6].split(',')) if len(sys.argv) > 6 else (1, 2, 3)
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16", kernel_flags=kf).load_synthetic(7).to("cuda")
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
items = [sharding.seeded_item(ds, i, 3, d, T, 0.01) for i in range(B)]
feats, tape = sharding.stack_items(items)
REC = []
if len(sys.argv) > 8:  # argv[8]: record the rotation score handed to every reverse step (a copy kernel per step on the loop's stream)
    from framedipt_amd.diffusion import se3_diffuser as _sd
    _rd = _sd.SE3Diffuser.reverse_device
    def _rec(self, rigids_t, rot_score, *a, **k):
        REC.append(rot_score.clone())
        return _rd(self, rigids_t, rot_score, *a, **k)
    _sd.SE3Diffuser.reverse_device = _rec
def scores(n_streams):
    """[steps][B, N, 3] from the recorded clones (n_streams consecutive calls per step = the sub-batches in order)"""
    torch.cuda.synchronize()
    out = [torch.cat(REC[i:i + n_streams], 0).cpu().numpy() for i in range(0, len(REC), n_streams)]
    REC.clear()
    return out
REF = inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, streams=1)
ref = REF["rigid_traj"]
REF_SC = scores(1) if len(sys.argv) > 8 else None
bad = {1: 0, 2: 0, 3: 0, 4: 0}
for r in range(reps):
    for streams in STREAMS:
        O = inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, streams=streams)
        o = O["rigid_traj"]
        SC = scores(min(streams, B)) if len(sys.argv) > 8 else None
        if SC is not None and not np.array_equal(o, ref):
            bad_steps = [k for k in range(len(SC)) if not np.array_equal(SC[k], REF_SC[k])]
            if bad_steps:
                k0 = bad_steps[0]
                bb, nn = np.nonzero(np.abs(SC[k0] - REF_SC[k0]).max(-1))
                print(f"   rotation score first differs at the reverse step {k0}: residues {list(zip(bb.tolist(), nn.tolist()))[:8]}, got {SC[k0][bb[0], nn[0]]} ref {REF_SC[k0][bb[0], nn[0]]}")
            else:
                print("   rotation scores identical at every step")
        if not np.array_equal(o, ref):
            # which output of which step differs first?  (arrays are reversed in time: index T = x_T, index 0 = final)
            firsts = {}
            for key in ("rigid_traj", "rigid_0_traj", "prot_traj", "trans_traj"):
                dd = np.abs(np.asarray(O[key], dtype=np.float64) - np.asarray(REF[key], dtype=np.float64))
                dd = dd.reshape(dd.shape[0], -1).max(-1)
                nz = np.nonzero(dd)[0]
                firsts[key] = (dd.shape[0] - 1 - int(nz.max())) if len(nz) else None   # forward-time index of the first difference
            print("   first differing forward-time index per output:", firsts)
            j = firsts["rigid_traj"]
            a_, r_ = np.asarray(O["rigid_traj"])[T - j], np.asarray(REF["rigid_traj"])[T - j]      # [B, N, 7]
            dq, dt_ = np.abs(a_[..., :4] - r_[..., :4]).max(-1), np.abs(a_[..., 4:] - r_[..., 4:]).max(-1)
            bq, nq = np.nonzero(dq)
            bt, nt_ = np.nonzero(dt_)
            print(f"   x at that index: {len(nq)} residues differ in the quaternion (max {dq.max():.2e}), {len(nt_)} in the translation (max {dt_.max():.2e});"
                  f" residues (sample, index) quat {list(zip(bq.tolist(), nq.tolist()))[:10]} trans {list(zip(bt.tolist(), nt_.tolist()))[:6]}")
            bad[streams] += 1
            diff = np.abs(o - ref).reshape(T + 1, B, -1).max(-1)   # [T+1 (reversed time), B]
            first = max(s for s in range(T + 1) if diff[s].max() > 0)  # earliest step (largest reversed index) that differs
            print(f"  rep {r} streams {streams}: first difference at step {T - first} of {T}, samples {np.nonzero(diff[first])[0].tolist()}, max |d| there {diff[first].max():.3g}")
print(f"N={N} B={B} T={T} kf={kf}: mismatching runs of {reps}: 1 stream {bad[1]}, 2 streams {bad[2]}, 3 streams {bad[3]}, 4 streams {bad[4]}")
