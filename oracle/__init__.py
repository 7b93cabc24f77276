"""CPU oracle for the coupling-flow hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (plain PyTorch CPU ops, fp32 or fp64) of the
reference's bijector arithmetic: the rational-quadratic-spline transform, the
coupling layers around it, affine coupling, masked affine flows, permutations,
mask builders, the diagonal-Gaussian end caps and the log_prob / sample loops.
Every function cites the reference file:line it follows (paths relative to
``/root/reference``).

Status of the pin: **parity pinned**.  The reference is pure Python and was
imported in the build container (see ``tests/golden/make_golden.py``); the
``.npz`` fixtures under ``tests/golden/`` are its outputs in fp32 and fp64 on
seeded inputs, and ``tests/test_oracle_golden.py`` checks every function here
against them.  The reference's own tests hold no golden vectors (properties
only: round trip <= 1e-4, shapes, finiteness, identity pass-through); those
properties are restated in ``tests/``.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``, and there only as the checker / reported
baseline.  Nothing under ``vcnf_amd/`` imports it; the product path has no CPU
fallback and raises when the HIP library is missing.
"""

from . import rqs, masks, nets, layers  # noqa: F401
