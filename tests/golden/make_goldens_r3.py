"""Round-3 golden fixtures (same harness as make_goldens.py / make_goldens_r2.py: the reference is imported in the build
container, only inputs and outputs are stored):

  traj_full_denovo_n300_T5_gain03   the benchmarked size with BackboneUpdate weights at trained-weight scale (bb_gain 0.3)
  traj_full_inpaint_n40_T4_aatype   inpainting trajectory with inference.input_aatype=True / model.input_aatype=False (the
                                    reference's default inpainting configuration: the network sees 20 = unknown on diffused
                                    residues, the x_0 / x_t backbone atoms are built with the true residue types)

  cached_score                      SE3Diffuser.calc_rot_score with so3.use_cached_score = True (table lookup instead of the series)

    python tests/golden/make_goldens_r3.py [job ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refharness as rh  # noqa: E402  (stubs + sys.path for /root/reference)
import make_goldens as mg  # noqa: E402
import make_goldens_r2 as r2  # noqa: E402


def traj_inpaint_aatype(name, n, num_t):
    """mg.traj_golden with input_aatype=True on the inference side and model.input_aatype=False (experiments/utils.py:397-402,
    score_network.py:231-238, framedipt/data/utils.py:565-610)."""
    cfg = rh.load_cfg(inpainting=True)
    assert not cfg.model.input_aatype
    orig = mg.exp_utils.inference_fn

    def with_aatype(*a, **k):
        k["input_aatype"] = True
        return orig(*a, **k)

    mg.exp_utils.inference_fn = with_aatype
    try:
        mg.traj_golden(name, cfg, n, True, num_t)
    finally:
        mg.exp_utils.inference_fn = orig
    p = os.path.join(HERE, f"traj_{name}.npz")
    g = dict(np.load(p))
    g["input_aatype"] = np.array(1)
    np.savez_compressed(p, **g)


def cached_score_golden():
    """so3.use_cached_score = True (so3_diffuser.py:389-396): calc_rot_score through the bucketised score-norm table, at three
    noise levels, and one small-network forward with the flag set (its rot_score output)."""
    import torch
    from framedipt.diffusion import se3_diffuser
    from openfold.utils import rigid_utils as ru
    cfg = rh.load_cfg()
    cfg.diffuser.so3.use_cached_score = True
    diff = se3_diffuser.SE3Diffuser(cfg.diffuser)
    rng = np.random.default_rng(5)
    g = {}
    n = 48
    for i, (t, spread) in enumerate(((0.02, 0.15), (0.5, 0.9), (1.0, 2.5))):
        q0 = mg.rand_quats(rng, n)
        rv = rng.standard_normal((n, 3)) * spread
        dq = np.concatenate([np.cos(np.linalg.norm(rv, axis=-1, keepdims=True) / 2),
                             rv / np.linalg.norm(rv, axis=-1, keepdims=True) * np.sin(np.linalg.norm(rv, axis=-1, keepdims=True) / 2)], -1)
        # q_t = q_0 * dq  (so that log(q_0^-1 q_t) = rv)
        a, b = q0, dq
        qt = np.stack([a[:, 0] * b[:, 0] - (a[:, 1:] * b[:, 1:]).sum(-1),
                       a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0] + a[:, 2] * b[:, 3] - a[:, 3] * b[:, 2],
                       a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3] + a[:, 2] * b[:, 0] + a[:, 3] * b[:, 1],
                       a[:, 0] * b[:, 3] + a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1] + a[:, 3] * b[:, 0]], -1).astype(np.float32)
        out = diff.calc_rot_score(ru.Rotation(quats=torch.tensor(qt)[None]), ru.Rotation(quats=torch.tensor(q0.astype(np.float32))[None]),
                                  torch.tensor([t], dtype=torch.float32))
        g[f"t_{i}"], g[f"qt_{i}"], g[f"q0_{i}"], g[f"score_{i}"] = np.float32(t), qt, q0.astype(np.float32), out.numpy()
        g[f"dtype_{i}"] = np.array(str(out.dtype))
    np.savez_compressed(os.path.join(HERE, "cached_score.npz"), **g)
    print("cached_score", {k: (v.shape, float(np.abs(v).max())) for k, v in g.items() if k.startswith("score")})


JOBS = {
    "traj_full_denovo_n300_T5_gain03": lambda: r2.traj_golden_gain("full_denovo_n300_T5_gain03", rh.load_cfg(), 300, 5, 0.3),
    "traj_full_inpaint_n40_T4_aatype": lambda: traj_inpaint_aatype("full_inpaint_n40_T4_aatype", 40, 4),
    "cached_score": cached_score_golden,
}

if __name__ == "__main__":
    for job in (sys.argv[1:] or list(JOBS)):
        JOBS[job]()
