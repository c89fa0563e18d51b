from .se3_diffuser import SE3Diffuser  # noqa: F401
