"""One reverse step captured as a HIP graph vs the same step launched eagerly (is the step launch-bound?).
    python tools/graph_step.py [c2|c4]
The step's ~70 kernel launches are captured once (torch.cuda.CUDAGraph around ReverseLoop.step: every C-ABI call only enqueues work on
the current stream) and replayed; the eager loop issues the same launches from Python / ctypes.  Same trajectory slot both ways."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, inference, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.sampler import UnconditionalSampler
cfgs = {"c2": (128, 8), "c4": (300, 8)}
for name in (sys.argv[1:] or ["c2", "c4"]):
    N, B = cfgs[name]
    T = 50
    conf = config.base_config()
    d = SE3Diffuser(conf.diffuser, device="cuda")
    net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
    ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
    items = [sharding.seeded_item(ds, i, 3, d, T, 0.01) for i in range(B)]
    feats, tape = sharding.stack_items(items)
    loop = inference.ReverseLoop(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
    loop.prime()
    for k in range(5):
        loop.step(k)
    torch.cuda.synchronize()
    k, reps = 5, 100
    state = (loop.rigids_t, loop.noisy)
    sc0 = loop.sc_ca.clone()  # (the forward hands the predicted CA positions to the next step IN PLACE: restored for the identity check)

    def eager():
        loop.rigids_t, loop.noisy = state
        loop.step(k)

    for _ in range(5):
        eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eager()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        eager()
    t_host = (time.perf_counter() - t0) / reps   # host time to ENQUEUE a step (no sync inside)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eager()
    torch.cuda.current_stream().wait_stream(s)
    loop.rigids_t, loop.noisy = state
    with torch.cuda.graph(g):
        loop.step(k)
    torch.cuda.synchronize()
    loop.sc_ca.copy_(sc0)
    eager()
    ref = loop.rigid_traj[k + 1].clone()
    loop.sc_ca.copy_(sc0)
    g.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(ref, loop.rigid_traj[k + 1]))
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / reps
    print(f"{name} (N={N}, B={B}): eager step {t_eager * 1e3:.3f} ms (host enqueue {t_host * 1e3:.3f} ms), HIP-graph replay {t_graph * 1e3:.3f} ms "
          f"({(t_eager / t_graph - 1) * 100:+.1f} %), results identical: {same}")
