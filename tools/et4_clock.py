"""Shader clock sustained inside the EdgeTransition kernel during real sampling steps :
  python tools/et4_clock.py [N B T]
Prints core-clock cycles / 100 MHz ticks summed over the blocks of all launches = the average clock the matrix cores ran at."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from framedipt_amd import _lib, config, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.inference import inference_fn
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.sampler import UnconditionalSampler
N, B, T = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (300, 8, 100)
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
items = [sharding.seeded_item(ds, i, 3, d, T, 0.01) for i in range(B)]
feats, tape = sharding.stack_items(items)
inference_fn(net, d, feats, num_t=10, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tuple(z[:10] for z in tape))
torch.cuda.synchronize()
st = net.batch_state(feats["seq_idx"])
st.clock_out = torch.zeros(3, dtype=torch.int64, device="cuda")  # FdiptForwardArgs.clock_out: opt-in, caller-owned
inference_fn(net, d, feats, num_t=T, min_t=0.01, aux_traj=True, noise_scale=0.1, noise_tape=tape)
torch.cuda.synchronize()
cyc, ticks, blocks = (int(v) for v in st.clock_out.cpu())
print(f"N={N} B={B}: {blocks} blocks, {cyc / blocks:.0f} cycles and {ticks / blocks / 100:.1f} us per block -> {cyc / ticks / 10:.3f} GHz sustained in the kernel")
