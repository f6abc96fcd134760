"""Conjugate gradients and the projected (rank n-1) solve primitives -- API of reference
DominantSparseEigenAD/CG.py.

    CG_torch(A, b, initialx, sparse=False)            reference CG.py:3-41
    CGSubspace                                        reference CG.py:43-71
    setCGSubspaceSparse(A, Aadjoint_to_gadjoint) -> module attribute ``CGSubspaceSparse``
                                                      reference CG.py:73-140

On CUDA tensors the iteration runs in HIP kernels (fused x/r update with ||r||^2, direction update,
operator mat-vec with the E0 shift and d.Ad fused) with the CG scalars and the convergence flag kept on
the device; the host only polls the flag every few iterations instead of the per-iteration ``.item()``
of CG.py:28,35.  ``A d`` is evaluated once per iteration (the reference evaluates it twice, CG.py:34 and
:31/:40; the values are identical).

Keyword-only extensions (defaults = the reference's hard-coded constants): ``eps`` (CG.py:25),
``maxiter`` (CG.py:32).  Start vectors are drawn with ``torch.randn`` exactly where the reference
draws them (CG.py:58,121), so patching / seeding the global RNG pins a run the same way.
"""
from __future__ import annotations

import torch

from . import engine
from ._cpu_plumbing import cg_host
from ._space import LOCAL, space_of

EPS_DEFAULT = 1e-7   # CG.py:25; module-level so that the autograd primitives (whose ``apply`` signature is
                     # fixed by the reference API) can be run at a tighter tolerance: ``CG.EPS_DEFAULT = 1e-13``


def _solve(A, b, initialx, sparse, shift=None, eps=None, maxiter=None):
    n = b.shape[0]
    eps = EPS_DEFAULT if eps is None else eps   # read at call time: tests / users may tighten it
    part = engine.native_of(A) if sparse else None
    if part is not None and getattr(part, "partitioned", False):
        # row-partitioned operator: b, initialx are slabs; iteration cap = the GLOBAL dimension (CG.py:32)
        x0 = initialx.detach().to(torch.float64).contiguous().clone()
        return part.solve_shifted(shift, b.detach().to(torch.float64).contiguous(), x0, eps=eps, maxiter=maxiter)
    cap = n if maxiter is None else int(maxiter)
    if b.is_cuda:
        if b.dtype == torch.float32 and not sparse:
            # dense fp32 system: vectors promoted to fp64 for the fp64 kernels, result rounded back (see Lanczos.py);
            # the matrix stays fp32 when it goes to the native symmetric operand
            A_use = A if engine.DENSE_SYMMETRIC_KERNEL else A.to(torch.float64)
            x = _solve(A_use, b.to(torch.float64), initialx.to(torch.float64), False,
                       None if shift is None else shift.to(torch.float64), eps, maxiter)
            return x.to(torch.float32)
        if b.dtype != torch.float64:
            raise NotImplementedError("the HIP CG kernels are fp64; got %s" % b.dtype)
        native = engine.native_of(A) if sparse else None
        if not sparse and engine.DENSE_SYMMETRIC_KERNEL:
            from .operators import dense_symmetric_operand
            native = dense_symmetric_operand(A)        # upper-triangle mat-vec, loop inside the library
        if native is not None:
            return engine.cg(b, initialx, native=native, shift=shift, eps=eps, maxiter=cap)
        amap = A if sparse else (lambda v: torch.matmul(A.to(v.dtype), v))
        return engine.cg(b, initialx, callable_A=amap, shift=shift, eps=eps, maxiter=cap)
    base = A if sparse else (lambda v: torch.matmul(A, v))
    amap = base if shift is None else (lambda v: base(v) - shift * v)   # CG.py:120
    return cg_host(amap, b, initialx, eps, cap, engine.last_cg)


def CG_torch(A, b, initialx, sparse=False, *, eps=None, maxiter=None):
    """Solve A x = b (A symmetric positive (semi-)definite) by CG from ``initialx``; returns x."""
    return _solve(A, b.detach(), initialx.detach(), sparse, eps=eps, maxiter=maxiter)


def _project(v, unit, sp=LOCAL):
    return v - sp.scale(sp.dot(unit, v), unit)


class CGSubspace(torch.autograd.Function):
    """A x = b for a dense symmetric A of rank n-1 with null vector alpha, and alpha.x = 0
    (reference CG.py:43-71).  backward re-enters the primitive, so higher derivatives work."""

    @staticmethod
    def forward(ctx, A, b, alpha):
        initialx = torch.randn(b.shape[0], device=b.device, dtype=b.dtype)        # CG.py:58
        initialx = _project(initialx, alpha)                                     # CG.py:59
        x = _solve(A.detach(), b.detach(), initialx, False)
        ctx.save_for_backward(A, alpha, x)
        return x

    @staticmethod
    def backward(ctx, grad_x):
        A, alpha, x = ctx.saved_tensors
        b = _project(grad_x, alpha)                                              # CG.py:67
        grad_b = CGSubspace.apply(A, b, alpha)                                   # CG.py:68
        grad_A = -grad_b[:, None] * x                                            # CG.py:69
        grad_alpha = -x * torch.matmul(alpha, grad_x)                            # CG.py:70
        return grad_A, grad_b, grad_alpha


class CGSubspaceShifted(torch.autograd.Function):
    """(A - E0 I) x = b, alpha.x = 0 for a DENSE symmetric tensor A without materialising A - E0 I (the
    reference builds it with ``torch.eye``, symeig.py:25: two extra n x n tensors).  Same structure as the
    matrix-free primitive (CG.py:119-138) with the outer product as the adjoint map:
    backward returns (-bbar x^T, bbar.x, bbar, -x (alpha.xbar))."""

    @staticmethod
    def forward(ctx, A, E0, b, alpha):
        initialx = torch.randn(b.shape[0], device=b.device, dtype=b.dtype)        # CG.py:58
        initialx = _project(initialx, alpha.detach())
        x = _solve(A.detach(), b.detach(), initialx, False, shift=E0.detach())
        ctx.save_for_backward(A, E0, alpha, x)
        return x

    @staticmethod
    def backward(ctx, grad_x):
        A, E0, alpha, x = ctx.saved_tensors
        b = _project(grad_x, alpha)
        grad_b = CGSubspaceShifted.apply(A, E0, b, alpha)
        grad_A = -grad_b[:, None] * x                                            # CG.py:69
        grad_E0 = torch.matmul(grad_b, x)                                        # CG.py:136 with v1 = -grad_b
        grad_alpha = -x * torch.matmul(alpha, grad_x)
        return grad_A, grad_E0, grad_b, grad_alpha


def _make_sparse_cg(A, Aadjoint_to_gadjoint):
    sp = space_of(A)   # one device: plain torch expressions; row-partitioned operator: global inner products

    class CGSubspaceSparse(torch.autograd.Function):
        """(A - E0 I) x = b, alpha.x = 0 with A matrix-free; inputs (g, E0, b, alpha) (CG.py:119-138)."""

        @staticmethod
        def forward(ctx, g, E0, b, alpha):
            initialx = torch.randn(b.shape[0], device=b.device, dtype=b.dtype)    # CG.py:121
            initialx = _project(initialx, alpha.detach(), sp)                    # CG.py:122
            x = _solve(A, b.detach(), initialx, True, shift=E0.detach())          # CG.py:120,123
            ctx.g = g
            ctx.save_for_backward(E0, alpha, x)
            return x

        @staticmethod
        def backward(ctx, grad_x):
            g = ctx.g
            E0, alpha, x = ctx.saved_tensors
            b = _project(grad_x, alpha, sp)                                      # CG.py:132
            grad_b = CGSubspaceSparse.apply(g, E0, b, alpha)                     # CG.py:133
            v1, v2 = -grad_b, x
            grad_alpha = -sp.scale(sp.dot(alpha, grad_x), x)                     # CG.py:135
            grad_E0 = -sp.dot(v1, v2)                                            # CG.py:136
            grad_g = Aadjoint_to_gadjoint(v1, v2)                                # CG.py:137
            return grad_g, grad_E0, grad_b, grad_alpha

    return CGSubspaceSparse


def setCGSubspaceSparse(A, Aadjoint_to_gadjoint):
    """Create the matrix-free projected-CG primitive and publish it as the module attribute
    ``CGSubspaceSparse`` (the reference's protocol, CG.py:116,139).  The class is also returned; unlike
    the reference its backward is bound to the class it belongs to, not looked up by name at
    backward time, so calling ``set...`` again does not rebind an existing graph."""
    global CGSubspaceSparse
    CGSubspaceSparse = _make_sparse_cg(A, Aadjoint_to_gadjoint)
    return CGSubspaceSparse
