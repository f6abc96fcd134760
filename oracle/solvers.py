"""Oracle solvers (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

CPU restatement, in torch fp64 tensor ops, of the two numerical loops on the
hot path.  Every function names the reference lines it follows.  Random start
vectors are never drawn implicitly: the caller passes ``draw`` (a callable
``draw(n, dtype) -> tensor``) that is consulted in exactly the order in which
the reference consults ``torch.randn`` so that a run can be pinned.
"""
from __future__ import annotations

import torch


def _default_draw(n, dtype):
    return torch.randn(n, dtype=dtype)


def _as_map(A, sparse):
    """Reference Lanczos.py:42-48 / CG.py:20-23: a tensor acts by matmul, a callable acts as is."""
    if sparse:
        return A
    return lambda v: torch.matmul(A, v)


def lanczos_tridiag(A, k, *, sparse=False, dim=None, draw=_default_draw, with_history=False):
    """k-step Lanczos with single-pass classical Gram-Schmidt full re-orthogonalisation.

    Follows reference Lanczos.py:42-77.

      * sparse path forces fp64 (Lanczos.py:43); dense path follows A.dtype (Lanczos.py:47)
      * q_0 = draw()/||.||            (Lanczos.py:52-53)
      * a second vector is drawn and multiplied by beta = 0 in the first
        three-term update           (Lanczos.py:58-61)
      * r = u - alpha q - beta q_prev (Lanczos.py:61)
      * r -= Q_i (Q_i^T r)           (Lanczos.py:66)
      * beta = ||r||, q = r / beta, u = A q, alpha = q.u (Lanczos.py:68-72)

    Returns (Q (n,k), alphas (k,), betas (k-1,)).
    """
    if sparse:
        n, dtype = int(dim), torch.float64
    else:
        n, dtype = A.shape[0], A.dtype
    apply_A = _as_map(A, sparse)

    Q = torch.zeros((n, k), dtype=dtype)
    alphas = torch.zeros(k, dtype=dtype)
    betas = torch.zeros(max(k - 1, 0), dtype=dtype)

    q = draw(n, dtype)
    q = q / torch.norm(q)
    u = apply_A(q)
    alpha = torch.matmul(q, u)
    Q[:, 0] = q
    alphas[0] = alpha
    beta = 0
    q_prev = draw(n, dtype)  # consumed only to stay draw-for-draw with the reference
    for i in range(1, k):
        r = u - alpha * q - beta * q_prev
        basis = Q[:, :i]
        r = r - torch.matmul(basis, torch.matmul(basis.T, r))
        q_prev = q
        beta = torch.norm(r)
        q = r / beta
        u = apply_A(q)
        alpha = torch.matmul(q, u)
        alphas[i] = alpha
        betas[i - 1] = beta
        Q[:, i] = q
    return Q, alphas, betas


def tridiag_matrix(alphas, betas):
    """Dense T of reference Lanczos.py:76."""
    return torch.diag(alphas) + torch.diag(betas, diagonal=1) + torch.diag(betas, diagonal=-1)


def ritz_extreme(Q, alphas, betas, extreme="both"):
    """Ritz extraction, reference Lanczos.py:98-105.

    The reference calls ``torch.symeig(T, eigenvectors=True)`` (removed from
    current torch); ``torch.linalg.eigh`` is the same LAPACK driver family and
    also returns ascending eigenvalues.  All k Ritz vectors are formed exactly
    as the reference does (``Qk @ S``) and the extreme columns returned.
    """
    T = tridiag_matrix(alphas, betas)
    evals, S = torch.linalg.eigh(T)
    Y = torch.matmul(Q, S)
    if extreme == "both":
        return evals[0], Y[:, 0], evals[-1], Y[:, -1]
    if extreme == "min":
        return evals[0], Y[:, 0]
    if extreme == "max":
        return evals[-1], Y[:, -1]
    raise ValueError("extreme must be 'both', 'min' or 'max'")


def symeig_lanczos(A, k, extreme="both", *, sparse=False, dim=None, draw=_default_draw):
    """reference Lanczos.py:79-105."""
    Q, alphas, betas = lanczos_tridiag(A, k, sparse=sparse, dim=dim, draw=draw)
    return ritz_extreme(Q, alphas, betas, extreme)


def cg_solve(A, b, x0, *, sparse=False, eps=1e-7, maxiter=None, stats=None):
    """Plain conjugate gradients, reference CG.py:20-41.

      * stop on ABSOLUTE ||r|| < eps, eps = 1e-7 hard-coded there (CG.py:25,28,35)
      * at most n iterations (CG.py:32)
      * x is updated before the residual test, so the returned x includes the
        last step (CG.py:33-36)
      * the reference evaluates A(d) twice per iteration on the same d
        (CG.py:34 and :31/:40); evaluating it once is bitwise identical and is
        what is done here.  ``stats['matvecs']`` still reports the reference's
        count (2*iters+1 on the break path) so timings can be related.

    ``stats`` (optional dict) receives ``iters``, ``matvecs``, ``resnorm``.
    """
    apply_A = _as_map(A, sparse)
    n = b.shape[0]
    cap = n if maxiter is None else int(maxiter)
    x = x0
    r = b - apply_A(x)
    ref_matvecs = 1
    rn = torch.norm(r).item()
    iters = 0
    if rn >= eps:
        d = r
        Ad = apply_A(d)
        ref_matvecs += 1
        step = torch.matmul(r, r) / torch.matmul(Ad, d)
        for _ in range(cap):
            iters += 1
            x = x + step * d
            r_next = r - step * Ad
            ref_matvecs += 1
            rn = torch.norm(r_next).item()
            if rn < eps:
                break
            ratio = torch.matmul(r_next, r_next) / torch.matmul(r, r)
            r = r_next
            d = r + ratio * d
            Ad = apply_A(d)
            ref_matvecs += 1
            step = torch.matmul(r, r) / torch.matmul(Ad, d)
    if stats is not None:
        stats["iters"] = iters
        stats["matvecs"] = ref_matvecs
        stats["resnorm"] = rn
    return x
