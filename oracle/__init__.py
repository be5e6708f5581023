"""CPU oracle for the udaiic / partial train-step hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32/fp64)
restatement of the reference algorithm.  It is the *checker* for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  Nothing under the product package
(``mi-based-regularized-semi-supervised-segmentation_amd/``) imports it, and the
product never falls back to it.

Parity pinning: every function here is checked against golden vectors that were
produced by importing and running the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.  The reference ships no tests or fixtures of
its own, so those vectors are the pin.

``oracle/augment.py`` (input pipeline) sits on the real Pillow and is pinned by
``tests/golden/augment.npz`` (reference transform objects run by ``tests/golden/make_golden_augment.py``).

Citations ``path:line`` are relative to the reference checkout; ``whl:`` means
inside its vendored ``deepclustering2`` wheel.
"""
