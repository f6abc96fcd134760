"""Dominant symmetric eigen primitives -- API of reference DominantSparseEigenAD/symeig.py.

    DominantSymeig.apply(A, k[, device])                              reference symeig.py:4-31
    setDominantSparseSymeig(A, Aadjoint_to_gadjoint)                  reference symeig.py:33-88
        -> module attribute ``DominantSparseSymeig``; ``.apply(g, k, dim[, device])``

forward  = Lanczos with full re-orthogonalisation (HIP on CUDA devices, see Lanczos.py)
backward = projected CG solve of (A - lambda I) x = b (HIP on CUDA devices, see CG.py) followed by the
           user's ``Aadjoint_to_gadjoint(v1, v2)`` hook; the backward is built from differentiable
           pieces (torch glue + the re-entrant CG primitive) so second derivatives work as in the
           reference (examples/TFIM/E0.py:63-64, chiF.py:49-52).
"""
from __future__ import annotations

import torch

from .Lanczos import symeigLanczos
from . import CG as _CG
from ._space import space_of


# A loss that does not depend on the eigenvector (dE0/dg of examples/TFIM/E0.py:62-63, the first backward of every
# second-derivative evaluation) reaches the reference's backward with a ZERO grad_eigvector, and the reference then runs
# a complete CG solve of (A - E0) x = 0 from a random start vector (87-180 mat-vecs) whose result is rounding noise of
# size eps/gap added to the gradient (FIX-OK).  With SKIP_ZERO_RHS the primitives ask autograd
# not to materialise unused output gradients, recognise the case without looking at any data and take x = 0 -- the
# exact solution.  OFF by default: the skipped solve is also absent from the graph, so a SECOND derivative then makes
# one solve (and one random draw) fewer than the reference, and the default keeps the reference's behaviour draw for
# draw; sweeps that want the time back switch it on (examples/TFIM/sweep.py).
SKIP_ZERO_RHS = False


def _draw_and_discard(like):
    torch.randn(like.shape[0], device=like.device, dtype=like.dtype)        # CG.py:58 / :121 (RNG parity)


class DominantSymeig(torch.autograd.Function):
    """Smallest eigenvalue / eigenvector of a real symmetric matrix given as a torch.Tensor."""

    @staticmethod
    def forward(ctx, A, k, device=torch.device("cpu")):
        device = A.device if A.is_cuda else torch.device(device)
        eigval, eigvector = symeigLanczos(A.detach(), k, device=device, extreme="min")   # symeig.py:16
        ctx.save_for_backward(A, eigval, eigvector)
        ctx.device = device
        ctx.set_materialize_grads(False)
        return eigval, eigvector

    @staticmethod
    def backward(ctx, grad_eigval, grad_eigvector):
        A, eigval, eigvector = ctx.saved_tensors
        if grad_eigval is None:
            grad_eigval = torch.zeros_like(eigval)
        if grad_eigvector is None:
            if SKIP_ZERO_RHS:
                _draw_and_discard(eigvector)
                return (grad_eigval * eigvector)[:, None] * eigvector, None, None        # symeig.py:29 with lambda0 = 0
            grad_eigvector = torch.zeros_like(eigvector)
        b = grad_eigvector - torch.matmul(eigvector, grad_eigvector) * eigvector         # symeig.py:27
        if A.is_cuda:
            # the shift is applied inside the CG kernels: no A - lambda*I copy, no n x n identity (symeig.py:25)
            lambda0 = _CG.CGSubspaceShifted.apply(A, eigval, b, eigvector)
        else:
            Aprime = A - eigval * torch.eye(A.shape[0], device=A.device, dtype=A.dtype)  # symeig.py:25
            lambda0 = _CG.CGSubspace.apply(Aprime, b, eigvector)                         # symeig.py:28
        grad_A = (grad_eigval * eigvector - lambda0)[:, None] * eigvector                # symeig.py:29
        return grad_A, None, None


def _make_sparse_symeig(A, Aadjoint_to_gadjoint, cg_cls):
    sp = space_of(A)   # one device: plain torch expressions; row-partitioned operator: global inner products

    class DominantSparseSymeig(torch.autograd.Function):
        """Smallest eigenpair of a matrix-free real symmetric operator depending on parameters g."""

        @staticmethod
        def forward(ctx, g, k, dim, device=torch.device("cpu")):
            device = g.device if g.is_cuda else torch.device(device)
            eigval, eigvector = symeigLanczos(A, k, device=device, extreme="min", sparse=True, dim=dim)  # symeig.py:72-73
            ctx.save_for_backward(g, eigval, eigvector)
            ctx.set_materialize_grads(False)
            return eigval, eigvector

        @staticmethod
        def backward(ctx, grad_eigval, grad_eigvector):
            g, eigval, eigvector = ctx.saved_tensors
            if grad_eigval is None:
                grad_eigval = torch.zeros_like(eigval)
            if grad_eigvector is None:
                if SKIP_ZERO_RHS:
                    _draw_and_discard(eigvector)
                    return Aadjoint_to_gadjoint(sp.scale(grad_eigval, eigvector), eigvector), None, None, None
                grad_eigvector = torch.zeros_like(eigvector)
            b = grad_eigvector - sp.scale(sp.dot(eigvector, grad_eigvector), eigvector)  # symeig.py:80
            lambda0 = cg_cls.apply(g, eigval, b, eigvector)                              # symeig.py:81
            v1, v2 = sp.scale(grad_eigval, eigvector) - lambda0, eigvector               # symeig.py:82-83
            grad_g = Aadjoint_to_gadjoint(v1, v2)                                        # symeig.py:84
            return grad_g, None, None, None

    return DominantSparseSymeig


def setDominantSparseSymeig(A, Aadjoint_to_gadjoint):
    """Publish ``DominantSparseSymeig`` as a module attribute (the reference's protocol, symeig.py:66,87).

    ``A`` is the operator as a callable v -> A v -- either plain torch code or one of the native operators
    of ``dominantsparseeigenad_amd.operators`` (then both loops run without any Python in them);
    ``Aadjoint_to_gadjoint(v1, v2)`` maps the adjoint  A-bar = v1 v2^T  to the adjoint of g."""
    global DominantSparseSymeig
    cg_cls = _CG.setCGSubspaceSparse(A, Aadjoint_to_gadjoint)                            # symeig.py:67-69
    DominantSparseSymeig = _make_sparse_symeig(A, Aadjoint_to_gadjoint, cg_cls)
    return DominantSparseSymeig
