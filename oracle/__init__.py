"""CPU oracle for the dominant-eigenpair hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp64 tensor ops, the arithmetic
type the reference itself uses) of the reference algorithms on the hot path:

    Lanczos tridiagonalisation + Ritz extraction   reference Lanczos.py:3-105
    conjugate-gradient solve                        reference CG.py:3-41
    projected (rank n-1) CG autograd primitives     reference CG.py:43-140
    dominant symmetric eigen primitives             reference symeig.py:4-88
    TFIM / 1-D Schroedinger matrix-free operators   reference examples/TFIM/TFIM.py:39-101,
                                                    examples/schrodinger1D.py:18-34

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the *checker*.  Nothing under
``dominantsparseeigenad_amd/`` imports this package; the product path runs on
the HIP library and raises when that library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the real
reference (read-only, from /root/reference, with the two in-process
compatibility shims described in that script) in the build container, ran it
with prescribed start vectors, and stored inputs + outputs under
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks this
restatement against those vectors (and against the reference's own
``examples/TFIM/datas/E0_N_10.npz`` curve values that were copied in as data).
"""
from .solvers import lanczos_tridiag, ritz_extreme, symeig_lanczos, cg_solve  # noqa: F401
from .operators import TFIMTables, Stencil3  # noqa: F401
from .adjoint import (  # noqa: F401
    DenseDominantSymeig,
    make_sparse_dominant_symeig,
    DenseProjectedCG,
    make_sparse_projected_cg,
)
