"""MI355X-native SE(3) frame-diffusion sampler behind the FrameDiPT sampler API.

Hot path only: ``experiments/sampler.py`` samplers,
``framedipt.diffusion.SE3Diffuser`` and the score-network forward driven by
``experiments/utils.py:inference_fn`` of the reference.  All arithmetic on the
path runs in hand-written HIP kernels for gfx950 behind the C ABI declared in
``include/fdipt.h`` (``framedipt_amd/csrc``); Python here is host plumbing.
"""

RESIDUE_GAP = 200  # reference framedipt/__init__.py:3

__all__ = ["RESIDUE_GAP"]
