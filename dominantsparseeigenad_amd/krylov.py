"""Device-side Krylov solvers for the NON-symmetric primitives: what the
reference delegates to SciPy on the host -- ARPACK ``eigs(A, k=1, which, ncv=k)`` (reference eig.py:29-30,
116-117) and ``gmres(A - lambda I, b, tol=1e-12, atol=1e-12)`` (eig.py:52-57,137-144) -- restated on GPU
vectors with the same HIP phase kernels the Lanczos path uses for its orthogonalisation:

    arnoldi_dominant   explicitly restarted Arnoldi, ncv basis vectors, classical Gram-Schmidt on the dots /
                       correction kernel pair with a second round when the DGKS test asks for it (ARPACK's
                       rule), Hessenberg eigen-solve on the host (ncv x ncv), restart from the wanted Ritz
                       vector until the residual estimate |h_{m+1,m} e_m^T y| is at rounding level
    gmres              restarted GMRES(20) with the same orthogonalisation, Givens rotations on the host

ARPACK's implicitly restarted method and this explicitly restarted one converge to the same eigenpair
(ARPACK is called with tol = 0 = machine precision); eigenvectors are compared up to the gauge the
primitives fix afterwards (l.r = 1, r.r = 1, sign of r free).
"""
from __future__ import annotations

import numpy as np
import torch

from . import engine

F64 = torch.float64


class ArnoldiNoConvergence(RuntimeError):
    """restart budget exhausted; carries the last Ritz pair and its residual estimate (like ARPACK's exception)"""

    def __init__(self, msg, eigenvalue, eigenvector, residual):
        super().__init__(msg)
        self.eigenvalue, self.eigenvector, self.residual = eigenvalue, eigenvector, residual


def _select(evals, which):
    if which == "LM":
        return int(np.argmax(np.abs(evals)))
    if which == "SM":
        return int(np.argmin(np.abs(evals)))
    if which == "LR":
        return int(np.argmax(evals.real))
    if which == "SR":
        return int(np.argmin(evals.real))
    raise ValueError("which must be one of LM, SM, LR, SR")


def _cgs2(ph, V, ldq, n, j, w, zero, bufs):
    """orthogonalise w against V[0..j] twice; returns (h (j+1,) device, w_orth, ||w_orth||^2 (1,) device)"""
    w1, w2, h1, h2, nrm2 = bufs
    i = j + 1
    ph.rdots(V, ldq, n, i, w, zero, None, w1, h1)        # w1 = w, h1 = V^T w    (alpha = 0: no three-term part)
    ph.axpy_norm(V, ldq, n, i, h1, w1, nrm2)             # w1 -= V h1
    ph.rdots(V, ldq, n, i, w1, zero, None, w2, h2)       # second round on the corrected vector
    ph.axpy_norm(V, ldq, n, i, h2, w2, nrm2)
    return h1[:i] + h2[:i], w2, nrm2


def _cgs_dgks(ph, V, ldq, n, j, w, zero, bufs):
    """Classical Gram-Schmidt with the second round only when the DGKS test asks for it (what ARPACK does):
    re-orthogonalise iff ||w - V V^T w||^2 < 1/2 ||w||^2.  The test needs two scalars on the host (one small
    D2H copy per step); for long bases it saves a full dots + correction pass on almost every step."""
    w1, w2, h1, h2, nrm2 = bufs
    i = j + 1
    ph.rdots(V, ldq, n, i, w, zero, None, w1, h1)        # w1 = w, h1 = V^T w, h1[i] = w.w
    ph.axpy_norm(V, ldq, n, i, h1, w1, nrm2)             # w1 -= V h1, nrm2 = ||w1||^2
    before, after = torch.stack((h1[i], nrm2[0])).tolist()
    if after >= 0.5 * before:
        return h1[:i], w1, nrm2
    ph.rdots(V, ldq, n, i, w1, zero, None, w2, h2)
    ph.axpy_norm(V, ldq, n, i, h2, w2, nrm2)
    return h1[:i] + h2[:i], w2, nrm2


def arnoldi_dominant(matvec, n, ncv, device, which="LM", v0=None, tol=1e-13, max_restarts=60):
    """Wanted eigenvalue (real, asserted as in eig.py:31-32) and unit-norm eigenvector of a real matrix
    given by ``matvec`` (torch device vector -> torch device vector)."""
    device = torch.device(device)
    ncv = int(min(ncv, n))
    ph = engine.Phases(n, device, kmax=ncv + 1)
    ldq = engine.round_up(n, 32)
    V = ph.empty(ncv + 1, ldq)
    H = ph.zeros(ncv + 1, ncv)
    zero = ph.zeros(1)
    bufs = (ph.empty(n), ph.empty(n), ph.zeros(ncv + 2), ph.zeros(ncv + 2), ph.zeros(1))
    v = torch.randn(n, dtype=F64, device=device) if v0 is None else engine.as_vector(v0, n).clone()
    nrm2 = ph.zeros(1)
    last_res, theta, x, res, converged = None, None, None, float("inf"), False
    for _ in range(max_restarts):
        ph.dot(v, v, nrm2)
        ph.scale_store(v, nrm2, V[0], None)
        H.zero_()
        for j in range(ncv):
            w = engine.as_vector(matvec(V[j, :n]), n)
            h, w_orth, wn2 = _cgs_dgks(ph, V, ldq, n, j, w, zero, bufs)
            H[: j + 1, j] = h
            ph.scale_store(w_orth, wn2, V[j + 1], H[j + 1, j: j + 1])
        Hh = H.cpu().numpy()
        # invariant subspace reached (n <= ncv or lucky breakdown): use the leading block only
        m = ncv
        scale = np.abs(Hh[:ncv, :ncv]).max()
        sub = np.array([abs(Hh[j + 1, j]) for j in range(ncv)])
        bad = np.where(~np.isfinite(sub) | (sub <= 1e-13 * scale))[0]
        if bad.size:
            m = int(bad[0]) + 1
        evals, evecs = np.linalg.eig(Hh[:m, :m])
        idx = _select(evals, which)
        theta, y = evals[idx], evecs[:, idx]
        if abs(theta.imag) > 1e-9 * max(abs(theta), 1e-300):
            raise ValueError("The desired eigenvalue of the matrix must be real")      # eig.py:31-32
        y = (y / y[np.argmax(np.abs(y))]).real
        y = y / np.linalg.norm(y)
        sub_m = 0.0 if m < ncv or not np.isfinite(sub[m - 1]) else sub[m - 1]
        res = abs(sub_m * y[-1]) if m == ncv else 0.0
        x = ph.empty(n)
        ph.ritz(V, ldq, n, m, torch.from_numpy(np.ascontiguousarray(y)).to(device), x)
        x = x / x.norm()
        if res <= tol * abs(theta.real) or (last_res is not None and res >= 0.5 * last_res and res <= 1e-10 * abs(theta.real)):
            converged = True
            break
        last_res = res
        v = x
    if not converged:
        # ARPACK raises ArpackNoConvergence here (eig.py:29-30 would propagate it); an unconverged pair must not
        # reach the backward pass, whose solves of (A - theta I) assume an exact eigenvalue
        raise ArnoldiNoConvergence("Arnoldi did not converge in %d restarts of %d vectors: residual estimate %.3e "
                                   "relative to |theta| = %.3e" % (max_restarts, ncv, res, abs(theta.real)),
                                   float(theta.real), x, res)
    return float(theta.real), x


def gmres(matvec, b, rtol=1e-12, atol=1e-12, restart=20, maxiter=None, check_every=5):
    """Restarted GMRES from x0 = 0; stops when ||b - A x|| <= max(rtol ||b||, atol) (scipy's rule).

    The Hessenberg columns stay on the device; the host pulls them (one small D2H copy = one sync) only every
    ``check_every`` inner steps to advance the Givens rotations and test the residual, instead of after every
    step -- the vectors here are small (2 MB at D = 512) and each step is latency-bound."""
    device, n = b.device, b.numel()
    b = engine.as_vector(b, n)
    restart = int(min(restart, n))
    ph = engine.Phases(n, device, kmax=restart + 1)
    ldq = engine.round_up(n, 32)
    V = ph.empty(restart + 1, ldq)
    zero, nrm2 = ph.zeros(1), ph.zeros(1)
    bufs = (ph.empty(n), ph.empty(n), ph.zeros(restart + 2), ph.zeros(restart + 2), ph.zeros(1))
    Hdev = ph.zeros(restart, restart + 1)          # row j = column j of the Hessenberg matrix
    x = ph.zeros(n)
    target = max(rtol * float(b.norm()), atol)
    cycles = 10 * n if maxiter is None else int(maxiter)
    first = True
    for _ in range(cycles):
        r = b.clone() if first else b - engine.as_vector(matvec(x), n)
        first = False
        beta = float(r.norm())
        if beta <= target:
            break
        ph.dot(r, r, nrm2)
        ph.scale_store(r, nrm2, V[0], None)
        Hm = np.zeros((restart + 1, restart))
        cs, sn = np.zeros(restart), np.zeros(restart)
        gvec = np.zeros(restart + 1)
        gvec[0] = beta
        m, done_cols, converged = 0, 0, False
        Hdev.zero_()
        for j in range(restart):
            w = engine.as_vector(matvec(V[j, :n]), n)
            h, w_orth, wn2 = _cgs2(ph, V, ldq, n, j, w, zero, bufs)
            Hdev[j, : j + 1] = h
            ph.scale_store(w_orth, wn2, V[j + 1], Hdev[j, j + 1: j + 2])
            m = j + 1
            if m % check_every == 0 or m == restart:
                cols = Hdev[done_cols:m].cpu().numpy()                  # the host round trip
                for jj in range(done_cols, m):
                    col = cols[jj - done_cols, : jj + 2].copy()
                    for t in range(jj):                                  # previous rotations
                        a, c2 = col[t], col[t + 1]
                        col[t], col[t + 1] = cs[t] * a + sn[t] * c2, -sn[t] * a + cs[t] * c2
                    rho = np.hypot(col[jj], col[jj + 1])
                    cs[jj], sn[jj] = (1.0, 0.0) if rho == 0.0 else (col[jj] / rho, col[jj + 1] / rho)
                    col[jj], col[jj + 1] = rho, 0.0
                    Hm[: jj + 2, jj] = col
                    gvec[jj + 1] = -sn[jj] * gvec[jj]
                    gvec[jj] = cs[jj] * gvec[jj]
                    if abs(gvec[jj + 1]) <= target:
                        m, converged = jj + 1, True
                        break
                done_cols = m
                if converged:
                    break
        yv = np.linalg.solve(np.triu(Hm[:m, :m]), gvec[:m])
        dx = ph.empty(n)
        ph.ritz(V, ldq, n, m, torch.from_numpy(np.ascontiguousarray(yv)).to(device), dx)
        x = x + dx
    return x


class TorchLinearOperator:
    """Device counterpart of the scipy ``LinearOperator`` the reference's DominantSparseEig takes
    (eig.py:96-100): ``shape`` and a ``matvec`` acting on torch vectors of ``device``."""

    def __init__(self, shape, matvec, device):
        self.shape = tuple(shape)
        self.matvec = matvec
        self.device = torch.device(device)

    def __call__(self, v):
        return self.matvec(v)
