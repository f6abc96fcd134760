"""Oracle autograd primitives (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

CPU restatement of the differentiable wrappers around the two loops:

  * DenseProjectedCG / make_sparse_projected_cg   reference CG.py:43-71 / :73-140
  * DenseDominantSymeig / make_sparse_dominant_symeig   reference symeig.py:4-31 / :33-88

Differences from the reference that do NOT change arithmetic:
  * the matrix-free classes are returned by a factory instead of being planted
    as module globals (reference CG.py:116,139; symeig.py:66,87); the backward
    re-enters the class it belongs to, so an old graph is not silently rebound;
  * random start vectors come from an injectable ``draw`` so runs can be pinned;
  * CG tolerance / iteration cap are parameters (defaults = reference constants).
"""
from __future__ import annotations

import torch

from .solvers import cg_solve, symeig_lanczos, _default_draw


def _project_out(v, unit):
    """v - (unit . v) unit   (reference CG.py:59,67,122,132; symeig.py:27,80)."""
    return v - torch.matmul(unit, v) * unit


def _make_dense_projected_cg(draw, eps, maxiter):
    class _DenseProjectedCG(torch.autograd.Function):
        """Solve A x = b with A of rank n-1, null vector ``alpha``, subject to alpha.x = 0.

        forward  reference CG.py:57-62 ; backward reference CG.py:63-71
        """

        @staticmethod
        def forward(ctx, A, b, alpha):
            x0 = _project_out(draw(b.shape[0], b.dtype), alpha)
            x = cg_solve(A, b, x0, eps=eps, maxiter=maxiter)
            ctx.save_for_backward(A, alpha, x)
            return x

        @staticmethod
        def backward(ctx, xbar):
            A, alpha, x = ctx.saved_tensors
            rhs = _project_out(xbar, alpha)
            bbar = _DenseProjectedCG.apply(A, rhs, alpha)
            Abar = -bbar[:, None] * x
            alphabar = -x * torch.matmul(alpha, xbar)
            return Abar, bbar, alphabar

    return _DenseProjectedCG


DenseProjectedCG = _make_dense_projected_cg(_default_draw, 1e-7, None)


def make_sparse_projected_cg(A, adjoint_hook, *, draw=_default_draw, eps=1e-7, maxiter=None, stats=None):
    """Matrix-free projected CG primitive, reference CG.py:73-140.

    Inputs of ``apply``: (g, E0, b, alpha); solves (A - E0 I) x = b, alpha.x = 0.
    backward returns (hook(-bbar, x), bbar.x, bbar, -x (alpha.xbar))  (CG.py:134-138).
    ``stats`` (optional list) gets one dict per solve with iteration counts.
    """

    class _SparseProjectedCG(torch.autograd.Function):
        @staticmethod
        def forward(ctx, g, E0, b, alpha):
            shifted = lambda v: A(v) - E0 * v  # noqa: E731  (CG.py:120)
            x0 = _project_out(draw(b.shape[0], b.dtype), alpha)
            st = {} if stats is not None else None
            x = cg_solve(shifted, b, x0, sparse=True, eps=eps, maxiter=maxiter, stats=st)
            if stats is not None:
                stats.append(st)
            ctx.g = g
            ctx.save_for_backward(E0, alpha, x)
            return x

        @staticmethod
        def backward(ctx, xbar):
            g = ctx.g
            E0, alpha, x = ctx.saved_tensors
            rhs = _project_out(xbar, alpha)
            bbar = _SparseProjectedCG.apply(g, E0, rhs, alpha)
            v1, v2 = -bbar, x
            alphabar = -x * torch.matmul(alpha, xbar)
            E0bar = -torch.matmul(v1, v2)
            gbar = adjoint_hook(v1, v2)
            return gbar, E0bar, bbar, alphabar

    return _SparseProjectedCG


def _make_dense_dominant_symeig(draw, eps, maxiter):
    cg_cls = _make_dense_projected_cg(draw, eps, maxiter)

    class _DenseDominantSymeig(torch.autograd.Function):
        """Smallest eigenpair of a dense symmetric tensor; reference symeig.py:15-31."""

        @staticmethod
        def forward(ctx, A, k):
            lam, psi = symeig_lanczos(A, k, extreme="min", draw=draw)
            ctx.save_for_backward(A, lam, psi)
            return lam, psi

        @staticmethod
        def backward(ctx, lambar, psibar):
            A, lam, psi = ctx.saved_tensors
            shifted = A - lam * torch.eye(A.shape[0], dtype=A.dtype)   # symeig.py:25
            rhs = _project_out(psibar, psi)                              # symeig.py:27
            lam0 = cg_cls.apply(shifted, rhs, psi)                       # symeig.py:28
            Abar = (lambar * psi - lam0)[:, None] * psi                  # symeig.py:29
            return Abar, None

    return _DenseDominantSymeig


DenseDominantSymeig = _make_dense_dominant_symeig(_default_draw, 1e-7, None)


def make_dense_dominant_symeig(*, draw=_default_draw, eps=1e-7, maxiter=None):
    return _make_dense_dominant_symeig(draw, eps, maxiter)


def make_sparse_dominant_symeig(A, adjoint_hook, *, draw=_default_draw, eps=1e-7, maxiter=None, stats=None):
    """Matrix-free smallest-eigenpair primitive; reference symeig.py:33-88.

    ``apply(g, k, dim)`` -> (lambda_min (0-dim), psi (n,)).
    backward (symeig.py:77-86): b = psibar - (psi.psibar) psi ; lam0 = projected CG ;
    gbar = hook(lambar psi - lam0, psi).
    """
    cg_cls = make_sparse_projected_cg(A, adjoint_hook, draw=draw, eps=eps, maxiter=maxiter, stats=stats)

    class _SparseDominantSymeig(torch.autograd.Function):
        @staticmethod
        def forward(ctx, g, k, dim):
            lam, psi = symeig_lanczos(A, k, extreme="min", sparse=True, dim=dim, draw=draw)
            ctx.save_for_backward(g, lam, psi)
            return lam, psi

        @staticmethod
        def backward(ctx, lambar, psibar):
            g, lam, psi = ctx.saved_tensors
            rhs = _project_out(psibar, psi)
            lam0 = cg_cls.apply(g, lam, rhs, psi)
            v1, v2 = lambar * psi - lam0, psi
            gbar = adjoint_hook(v1, v2)
            return gbar, None, None

    return _SparseDominantSymeig
