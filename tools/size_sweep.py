"""Forward at a sweep of lengths / batch sizes in the half-precision mode: outputs finite, and the default pair kernels
(edge_transition4 for N % 4 == 0) against the same forward with the edge_transition3 fallback (FDIPT_KF_ET3): two independent
implementations of the EdgeTransition, expected to agree to half-precision rounding of the pair representation."""
import sys
import numpy as np
import torch
sys.path.insert(0, '/root/repo')
from framedipt_amd import _lib, config, sharding
from framedipt_amd.diffusion import SE3Diffuser
from framedipt_amd.model import ScoreNetwork
from framedipt_amd.model.score_network import BatchState
from framedipt_amd.sampler import UnconditionalSampler

conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
nets = {kf: ScoreNetwork(conf.model, d, precision="fp16", kernel_flags=kf).load_synthetic(7).to("cuda") for kf in (0, _lib.KF_ET3)}
nets["fp32"] = ScoreNetwork(conf.model, d, precision="fp32").load_synthetic(7).to("cuda")  # the reference arithmetic
sizes = [int(x) for x in sys.argv[1:]] or [8, 12, 44, 48, 100, 132, 260, 388, 516, 644, 772, 900, 1024]
worst = 0.0
for n in sizes:
    for b in (1, 3):
        if n >= 900 and b > 1:
            continue
        ds = UnconditionalSampler(config.to_conf({"min_length": n, "max_length": n, "length_step": 1, "samples_per_length": b}), d, "cuda")
        feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(b)])
        f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()  # noqa: E731
        out = {}
        for kf, net in nets.items():
            st = BatchState(net, feats["seq_idx"])
            t32, temb, sig = net.step_scalars(np.full(b, 0.5))
            st.forward(f32(feats["rigids_t"]), f32(feats["res_mask"]), f32(feats["fixed_mask"]), f32(feats["sc_ca_t"]) + 1.0, None,
                       f32(feats["torsion_angles_sin_cos"][..., 2, :]), torch.as_tensor(t32, device="cuda"),
                       torch.as_tensor(temb, device="cuda"), torch.as_tensor(sig, device="cuda"))
            torch.cuda.synchronize()
            out[kf] = {k: getattr(st, k).double().cpu().numpy() for k in ("rigids", "psi", "rot_score", "trans_score", "atom37")}
            for k, v in out[kf].items():
                assert np.isfinite(v).all(), (n, b, kf, k)
        dr = float(np.abs(out[0]["atom37"] - out[_lib.KF_ET3]["atom37"]).max())
        worst = max(worst, dr)
        bb = lambda o: o["atom37"][:, :, [0, 1, 2, 4]]  # noqa: E731  (N, CA, C, O)
        rmsd = lambda x, y: float(np.sqrt(((bb(x) - bb(y)) ** 2).sum(-1).mean(axis=(1, 2))).max())  # noqa: E731  worst sample
        print(f"N={n:5d} B={b}: finite; max |atom37(ET4 path) - atom37(ET3 path)| = {dr:.2e} A; backbone RMSD vs the fp32 mode: "
              f"ET4 path {rmsd(out[0], out['fp32']):.2e}, ET3 path {rmsd(out[_lib.KF_ET3], out['fp32']):.2e} A")
print("worst", worst)
