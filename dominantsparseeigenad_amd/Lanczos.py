"""Lanczos tridiagonalisation and extreme Ritz pairs -- API of reference DominantSparseEigenAD/Lanczos.py.

    Lanczos(A, k, device, *, sparse, dim)                     reference Lanczos.py:3-77
    symeigLanczos(A, k, device, extreme, *, sparse, dim)      reference Lanczos.py:79-105

Same names, argument meaning and return values.  On a CUDA (= MI355X / ROCm) device the loop runs in
hand-written HIP kernels (csrc/dsea_kernels.hip) behind the C ABI of include/dsea.h:

  * ``A`` a native operator (operators.TFIMOperator / CSROperator / Stencil3Operator): the whole k-step
    loop, mat-vec included, is one library call with no host synchronisation;
  * ``A`` any Python callable (``sparse=True``) or a dense tensor: the mat-vec is the caller's torch code,
    every other vector operation of the loop is a library call.

Keyword-only extensions (defaults reproduce the reference): ``q0`` -- start vector instead of the
``torch.randn`` draw of Lanczos.py:52 (the unused second draw of Lanczos.py:59 is still consumed so the
global RNG advances exactly as in the reference).
"""
from __future__ import annotations

import torch

from . import engine
from ._cpu_plumbing import lanczos_host


def _resolve(A, device, sparse, dim):
    if sparse:
        n, dtype, amap = int(dim), torch.float64, A          # Lanczos.py:42-45 (fp64 forced)
    else:
        n, dtype = A.shape[0], A.dtype                         # Lanczos.py:46-48
        amap = lambda v: torch.matmul(A, v)                    # noqa: E731
    return n, dtype, amap


def _lanczos_core(A, k, device, sparse, dim, q0, arena=False):
    global WARM_START
    device = torch.device(device)
    warm = None
    if q0 is None and WARM_START is not None:
        warm, WARM_START = WARM_START, None
    part = engine.native_of(A) if sparse else None
    if part is not None and getattr(part, "partitioned", False):
        # row-partitioned operator (partitioned.py): ``dim`` is the global dimension, every vector is this rank's
        # slab; the two draws of Lanczos.py:52,59 are made per slab (seed the ranks differently)
        nloc, dev = part.nloc, part.device
        if q0 is None:
            q0 = torch.randn(nloc, dtype=torch.float64, device=dev)
            if warm is not None:
                q0 = warm.detach().to(device=dev, dtype=torch.float64)
        torch.randn(nloc, dtype=torch.float64, device=dev)
        Q, ldq, alphas, betas = part.lanczos(k, q0, arena=arena)
        return (part, Q, ldq, nloc, alphas, betas, torch.float64)
    n, dtype, amap = _resolve(A, device, sparse, dim)
    if q0 is None:
        q0 = torch.randn(n, dtype=dtype, device=device)        # Lanczos.py:52
        if warm is not None:
            q0 = warm.detach().to(device=device, dtype=dtype)
    torch.randn(n, dtype=dtype, device=device)                 # Lanczos.py:59 (value multiplies beta = 0)
    if device.type == "cuda":
        if dtype not in (torch.float64, torch.float32):
            raise NotImplementedError("the HIP Lanczos kernels are fp64 (fp32 tensors are promoted); got %s" % dtype)
        if dtype == torch.float32:
            # dense fp32 input (reference Lanczos.py:47: the dense path follows A.dtype): the kernels are fp64, so
            # the loop runs in fp64 on promoted operands and the outputs are rounded back to fp32 by the callers --
            # at least as accurate as fp32 arithmetic, same dtypes in and out
            # The MATRIX itself is not promoted when it goes to the native symmetric operand (read as fp32).
            if not engine.DENSE_SYMMETRIC_KERNEL:
                A64 = A.to(torch.float64)
                amap = lambda v: torch.matmul(A64, v)          # noqa: E731
            q0 = q0.to(torch.float64)
        native = engine.native_of(A) if sparse else None
        if not sparse and engine.DENSE_SYMMETRIC_KERNEL:
            # dense symmetric tensor (Lanczos.py:46-49): the hand-written upper-triangle mat-vec as a native operand,
            # so the whole loop runs inside the library
            from .operators import dense_symmetric_operand
            native = dense_symmetric_operand(A)
        if native is not None:
            try:
                Q, ldq, alphas, betas = engine.lanczos(A, k, n, device, q0, native=native, arena=arena)
            except engine.PartialNeedsPhases:
                if not sparse and dtype == torch.float32:      # (the phase calls are fp64: promote the matrix once)
                    A64 = A.to(torch.float64)
                    amap = lambda v: torch.matmul(A64, v)      # noqa: E731
                Q, ldq, alphas, betas = engine.lanczos(A, k, n, device, q0, callable_A=amap, arena=arena)
        else:
            Q, ldq, alphas, betas = engine.lanczos(A, k, n, device, q0, callable_A=amap, arena=arena)
        return ("cuda", Q, ldq, n, alphas, betas, dtype)
    Qk, alphas, betas = lanczos_host(amap, k, n, dtype, q0, device)
    return ("cpu", Qk, None, n, alphas, betas, dtype)


def Lanczos(A, k, device=torch.device("cpu"), *, sparse=False, dim=None, q0=None):
    """Returns (Qk, T): Qk (n,k) with orthonormal columns, T the k x k tridiagonal (Lanczos.py:76-77).

    On the GPU the basis is stored vector-contiguous, so ``Qk`` is the transposed view of a (k, ldq) buffer.
    """
    where, Q, ldq, n, alphas, betas, dtype = _lanczos_core(A, k, device, sparse, dim, q0)
    Qk = Q if where == "cpu" else Q[:, :n].T
    T = torch.diag(alphas) + torch.diag(betas, diagonal=1) + torch.diag(betas, diagonal=-1)
    return Qk.to(dtype), T.to(dtype)


# Warm start (an extension the reference lacks): when set to a tensor, the NEXT Lanczos run takes it as
# its start vector instead of the ``torch.randn`` draw of Lanczos.py:52 (the draw is still made, so the RNG advances
# as in the reference) and the attribute is cleared.  In a parameter sweep the previous point's eigenvector is an
# excellent start: far fewer Lanczos vectors reach the same residual (examples/TFIM/sweep.py --warm).
WARM_START = None

# Module-level default of the ``reorth`` extension below (the autograd primitives' ``apply`` signatures are fixed by
# the reference API, so they read it here): "full" = the reference's full re-orthogonalisation with a stored basis.
REORTH_DEFAULT = "full"


def symeigLanczos(A, k, device=torch.device("cpu"), extreme="both", *, sparse=False, dim=None, q0=None, reorth=None):
    """Extreme eigenvalue(s)/eigenvector(s); outputs as in Lanczos.py:88-105 (all torch tensors).

    Keyword-only extension ``reorth``: "full" (default, reference Lanczos.py:66), "twice" (the same pass applied twice per
    step: CGS2), "partial" (Simon's partial re-orthogonalisation for native device operators: the basis is
    re-orthogonalised only when the omega recurrence estimates a loss of orthogonality beyond 1e-10 (engine.PARTIAL_REORTH
    sets the threshold) -- typically one step in four to ten; Ritz values at full accuracy, Ritz vector to the threshold; ``engine.last_reorth_steps`` counts) or
    "none" -- basis-free two-pass
    Lanczos for native device operators: three rotating vectors instead of the k-vector basis (so k = 200 at
    n = 2^28 fits ONE GPU) and no k^2 n re-orthogonalisation traffic; the extreme Ritz pair is the same to rounding,
    interior Ritz values may appear more than once (not returned here)."""
    if extreme not in ("both", "min", "max"):
        raise ValueError("extreme must be 'both', 'min' or 'max'")
    reorth = REORTH_DEFAULT if reorth is None else reorth
    if reorth not in ("full", "none", "twice", "partial"):
        raise ValueError("reorth must be 'full', 'twice', 'partial' or 'none'")
    if reorth == "partial":
        if torch.device(device).type != "cuda":
            raise NotImplementedError("reorth='partial' runs on the GPU (native operators, row-partitioned operators on "
                                      "the library driver, callables)")
        cur = engine.partial_reorth()                    # (a threshold set process-wide stays the threshold)
        with engine.reorth_options(partial=0.0 if cur is None else cur):   # this thread only
            return symeigLanczos(A, k, device, extreme, sparse=sparse, dim=dim, q0=q0, reorth="full")
    if reorth == "twice":
        # CGS2: the Gram-Schmidt pass of Lanczos.py:66 applied twice per step (an option the reference lacks; device
        # operators and callables on the GPU) -- same Krylov process, orthogonality of the basis at rounding level even
        # next to an invariant subspace
        part = engine.native_of(A) if sparse else None
        if torch.device(device).type != "cuda":
            raise NotImplementedError("reorth='twice' runs on the GPU (native operators and callables); the host path "
                                      "re-orthogonalises once per step like the reference")
        if part is not None and getattr(part, "partitioned", False) and \
                (part.world > 1 or part.force_driver or part._local_native() is None):
            raise NotImplementedError("reorth='twice' is not implemented by the row-partitioned step drivers "
                                      "(they re-orthogonalise once per step)")
        with engine.reorth_options(passes=2):            # this thread only
            return symeigLanczos(A, k, device, extreme, sparse=sparse, dim=dim, q0=q0, reorth="full")
    if reorth == "none":
        native = engine.native_of(A) if sparse else None
        if native is None or getattr(native, "partitioned", False) or torch.device(device).type != "cuda":
            raise NotImplementedError("reorth='none' (basis-free two-pass Lanczos) needs a native single-GPU operator")
        global WARM_START
        n = int(dim)
        dev = torch.device(device)
        if q0 is None:
            q0 = torch.randn(n, dtype=torch.float64, device=dev)           # Lanczos.py:52
            if WARM_START is not None:
                q0, WARM_START = WARM_START.detach().to(device=dev, dtype=torch.float64), None
        torch.randn(n, dtype=torch.float64, device=dev)                    # Lanczos.py:59: same RNG consumption
        out = []
        for val, vec in engine.lanczos_basisfree(native, k, n, dev, q0, extreme):
            out += [torch.tensor(val, dtype=torch.float64, device=dev), vec]
        return tuple(out)
    # the basis is transient here (only Ritz vectors are returned): it lives in the persistent arena
    where, Q, ldq, n, alphas, betas, dtype = _lanczos_core(A, k, device, sparse, dim, q0, arena=True)
    pairs = engine.tridiag_extreme(alphas, betas, extreme)
    out = []
    for val, s in pairs:
        if where == "cuda":
            vec = engine.ritz_vector(Q, ldq, n, k, s, Q.device)
        elif where != "cpu":
            vec = where.ritz_vector(Q, ldq, k, s)           # row-partitioned: this rank's slab of the Ritz vector
        else:
            vec = torch.matmul(Q[:, :s.shape[0]], torch.from_numpy(s).to(Q.dtype))
        out += [torch.tensor(val, dtype=dtype, device=alphas.device), vec.to(dtype)]
    return tuple(out)
