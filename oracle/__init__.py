"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the R3D token-fusion training path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline -- never as the thing shipped.
The product package (``r3d_amd``) never imports this package and fails loudly when
its HIP library is missing.

Pinning: the reference (olivesgatech/R3D) holds no tests, fixtures or golden vectors
for this path.  The oracle is therefore pinned against the
reference ITSELF, imported on CPU in the build container by
``tests/golden/make_golden.py`` (committed), whose outputs are the small fixtures
under ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks the oracle
against those fixtures wherever the tests run (the reference does not travel).
"""
