"""ORACLE — test infrastructure only.

NumPy restatement of the reference algorithm for the sampler hot path
(SURVEY.md section 8).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package, and only as the
checker.  The product (``framedipt_amd``) never imports it.

Parity pinning: the reference holds no golden vectors for this path
(SURVEY.md section 4); the oracle is pinned against vectors captured from the
reference source imported in the build container
(``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``).
"""
