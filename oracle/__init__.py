"""CPU oracle for the curvature hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Nothing under ``laplace_amd/`` imports this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may, and only as the
checker / the timed CPU baseline — never as the thing shipped.

Parity status:
* Jacobians, GGN/EF full + diag, likelihood Hessian, Kron algebra (decompose, logdet, bmm,
  inv_square_form, diag), functional variances, fit/predictive glue: **pinned** against the
  reference's own in-tree code run in this container (``oracle/make_golden.py`` →
  ``tests/golden/*.npz``).
* Raw KFAC factor values: **parity unpinned** against curvlinops 2.0.0 itself (the library is
  not installable here); pinned only through the relations R1–R10 the reference's tests hold
   and the canonical product ``G ⊗ A``.
"""
