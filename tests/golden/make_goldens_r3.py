"""Round-3 golden fixtures (same harness as make_goldens.py / make_goldens_r2.py: the reference is imported in the build
container, only inputs and outputs are stored):

  traj_full_denovo_n300_T5_gain03   the benchmarked size with BackboneUpdate weights at trained-weight scale (bb_gain 0.3)
  traj_full_inpaint_n40_T4_aatype   inpainting trajectory with inference.input_aatype=True / model.input_aatype=False (the
                                    reference's default inpainting configuration: the network sees 20 = unknown on diffused
                                    residues, the x_0 / x_t backbone atoms are built with the true residue types)

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


JOBS = {
    "traj_full_denovo_n300_T5_gain03": lambda: r2.traj_golden_gain("full_denovo_n300_T5_gain03", rh.load_cfg(), 300, 5, 0.3),
    "traj_full_inpaint_n40_T4_aatype": lambda: traj_inpaint_aatype("full_inpaint_n40_T4_aatype", 40, 4),
}

if __name__ == "__main__":
    for job in (sys.argv[1:] or list(JOBS)):
        JOBS[job]()
