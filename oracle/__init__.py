"""CPU oracle for the LinTransUNet hot path (MaskTransUnet fwd+bwd training step).

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32) restatement
of the reference algorithm; it exists so that tests, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` have something to check / time the
HIP path against.  Nothing under ``lintransunet_amd/`` imports it and the product
path never routes through it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the reference
(`/root/reference`, model/ and loss/ packages) in the build container, checked
every function here against it (max |diff| <= 1e-5, see the script) and wrote
the input/output vectors committed under ``tests/golden/``; ``tests/test_oracle_golden.py``
re-checks the oracle against those vectors without the reference present.

Each function cites the reference file:line it restates (paths relative to the
reference root).
"""
