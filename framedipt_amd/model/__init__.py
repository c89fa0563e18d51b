from .score_network import ScoreNetwork  # noqa: F401
