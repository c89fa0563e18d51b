"""Do the library's launches from two HIP streams overlap?  (developer aid)"""
import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from framedipt_amd import config, inference, sharding, _lib
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.sampler import UnconditionalSampler
N, B, T = 300, 4, 500
conf = config.base_config(); dev = "cuda:0"
diff = SE3Diffuser(conf.diffuser, device=dev)
net = ScoreNetwork(conf.model, diff, precision="fp16").load_synthetic(7).to(dev)
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": 2 * B}), diff, dev)
items = [sharding.seeded_item(ds, i, 1, diff, T, 0.01) for i in range(2 * B)]
loops = []
for h in range(2):
    feats, tape = sharding.stack_items(items[h * B:(h + 1) * B])
    loops.append(inference.ReverseLoop(net, diff, feats, num_t=T, min_t=0.01, aux_traj=(sys.argv[1] == "aux"), noise_scale=0.1,
                                       noise_tape=tape, state=net.new_batch_state(feats["seq_idx"])))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
mode = sys.argv[2]
for lp in loops: lp.st.reserve_cus = int(sys.argv[3]) if len(sys.argv) > 3 else 0
def sub(lp, k):
    if mode == "fwd": lp._fwd(k, False, False)
    else: lp.step(k)
for rep in range(2):
    for two in (False, True):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(40):
            for lp, s in zip(loops, streams):
                with torch.cuda.stream(s if two else streams[0]):
                    sub(lp, k)
        torch.cuda.synchronize()
        print(f"{mode} aux={sys.argv[1]} {'two streams' if two else 'one stream '}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per step pair")
