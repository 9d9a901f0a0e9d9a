"""Oracle = TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, link or execute anything in this package, and only as the *checker*
(never as the thing measured or shipped).  The product (``cleanrl_amd``) never
imports it and fails loudly when the HIP library is missing.

Contents
--------
``ref_extract.py``   runs the reference's own source lines (AST/line-range
                     ``exec`` of ``/root/reference/cleanrl/ppo*.py``).  Works
                     only where ``/root/reference`` exists (the build
                     container); it cannot travel to the GPU box.
``mint_goldens.py``  uses ``ref_extract`` to mint ``tests/golden/*.npz``.
``torch_oracle.py``  CPU/fp32 restatement of the hot path in plain torch ops,
                     each function citing the reference lines it follows.  This
                     is what travels to the GPU box.
``numpy_oracle.py``  independent float64 numpy derivation (closed-form
                     gradients) used as a second opinion.
``c/ppo_oracle.c``   scalar C restatement (gcc) of GAE / Categorical / loss,
                     used for the ``cpu_baseline`` "port" timing and as a third
                     implementation in the parity tests.

Parity pinning
--------------
The reference's own tests pin NO numbers on this path (all PPO tests are
exit-code smoke tests; the single numeric test is JAX-only).  The oracle is
therefore pinned against *outputs of the reference itself run in the build
container*: ``tests/golden/*.npz`` are produced by ``mint_goldens.py`` executing
the reference's verbatim line ranges, and ``tests/test_oracle_golden.py`` checks
every oracle implementation against them.
"""
