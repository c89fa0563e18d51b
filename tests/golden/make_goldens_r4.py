"""Round-4 golden fixtures (same harness as make_goldens.py / _r2 / _r3: the reference is imported in the build container, only
inputs and outputs are stored).  The round-3 review asked where the fp16 mode's < 1e-3 A per-step margin breaks:

  traj_full_denovo_n300_T5_gain03_seed11   the benchmarked size at trained-weight scale (bb_gain 0.3) with a SECOND weight seed
  traj_full_denovo_n300_T5_gain05          ... and the first seed with BackboneUpdate weights at bb_gain 0.5 (frames move ~5 A per block)

    python tests/golden/make_goldens_r4.py [job ...]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refharness as rh  # noqa: E402  (stubs + sys.path for /root/reference)
import make_goldens as mg  # noqa: E402
import make_goldens_r2 as r2  # noqa: E402


def traj_gain_seed(name, n, num_t, bb_gain, weight_seed):
    """make_goldens_r2.traj_golden_gain with another seed of the synthetic weights (framedipt_amd/weights.py: synth_state_dict)."""
    saved = mg.WEIGHT_SEED
    mg.WEIGHT_SEED = weight_seed
    try:
        r2.traj_golden_gain(name, rh.load_cfg(), n, num_t, bb_gain)
    finally:
        mg.WEIGHT_SEED = saved


JOBS = {
    "traj_full_denovo_n300_T5_gain03_seed11": lambda: traj_gain_seed("full_denovo_n300_T5_gain03_seed11", 300, 5, 0.3, 11),
    "traj_full_denovo_n300_T5_gain05": lambda: traj_gain_seed("full_denovo_n300_T5_gain05", 300, 5, 0.5, 7),
}

if __name__ == "__main__":
    for job in (sys.argv[1:] or list(JOBS)):
        JOBS[job]()
