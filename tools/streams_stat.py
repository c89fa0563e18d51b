"""Which sub-batch stream counts reproduce the single-stream trajectory bit for bit, how often, and where they first differ.
    python tools/streams_stat.py N B T reps [kernel_flags]"""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd import inference
from framedipt_amd.inference import inference_fn
inference.StreamedLoops.MAX_STREAMS = 8  # (investigation tool: the product refuses more than two streams)
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
ref = inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, streams=1)["rigid_traj"]
bad = {1: 0, 2: 0, 3: 0, 4: 0}
for r in range(reps):
    for streams in STREAMS:
        o = inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape, streams=streams)["rigid_traj"]
        if not np.array_equal(o, ref):
            bad[streams] += 1
            diff = np.abs(o - ref).reshape(T + 1, B, -1).max(-1)   # [T+1 (reversed time), B]
            first = max(s for s in range(T + 1) if diff[s].max() > 0)  # earliest step (largest reversed index) that differs
            print(f"  rep {r} streams {streams}: first difference at step {T - first} of {T}, samples {np.nonzero(diff[first])[0].tolist()}, max |d| there {diff[first].max():.3g}")
print(f"N={N} B={B} T={T} kf={kf}: mismatching runs of {reps}: 1 stream {bad[1]}, 2 streams {bad[2]}, 3 streams {bad[3]}, 4 streams {bad[4]}")
