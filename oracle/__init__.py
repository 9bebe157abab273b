"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the gated-fusion caption decoder.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path
(``controllable_xgating_amd``) never imports this package and fails loudly if
its HIP library is missing.

Parity status: PINNED.  The reference holds no tests or golden vectors of its
own, so the oracle is pinned against outputs of the
reference itself, imported on CPU in the build container by
``tools/gen_golden.py`` (fixtures under ``tests/golden/``; checked by
``tests/test_oracle_golden.py``).
"""
