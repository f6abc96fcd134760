"""Dominant eigen primitives for general (non-symmetric) real matrices -- API of reference
DominantSparseEigenAD/eig.py.

    DominantEig.apply(A, k[, which])                                   reference eig.py:5-62
    setDominantSparseEig(A, AT, Aadjoint_to_gadjoint)                  reference eig.py:64-152
        -> module attribute ``DominantSparseEig``; ``.apply(g, k)``

Scope note: in the reference the arithmetic of this path is not its
own code but SciPy's ARPACK ``eigs`` and ``gmres`` on host NumPy arrays (eig.py:28-30,54-57), first order
only.  Host tensors / scipy ``LinearOperator``s keep exactly those third-party calls (identical numerics).
CUDA tensors (``DominantEig``) and ``krylov.TorchLinearOperator`` operands (``DominantSparseEig``) -- which the
reference cannot take at all (eig.py:28 ``.numpy()``) -- run the device Arnoldi / GMRES of ``krylov.py`` on the
HIP orthogonalisation kernels; the adjoint hook then receives torch device vectors instead of numpy arrays.
"""
from __future__ import annotations

import inspect

import numpy as np
import torch
from scipy.sparse import linalg as sla

_GMRES_TOL = 1e-12
_gmres_kw = "rtol" if "rtol" in inspect.signature(sla.gmres).parameters else "tol"


def _gmres(op, rhs):
    sol, _info = sla.gmres(op, rhs, atol=_GMRES_TOL, **{_gmres_kw: _GMRES_TOL})   # eig.py:54,57,140,144
    return sol


def _dominant_pair(A, AT, k, which):
    """right / left eigenvectors of the wanted eigenvalue, normalised l.r = 1, r.r = 1 (eig.py:29-36)."""
    wr, vr = sla.eigs(A, k=1, which=which, ncv=k)
    wl, vl = sla.eigs(AT, k=1, which=which, ncv=k)
    if not np.allclose(wr.imag, 0.0):
        raise AssertionError("The desired eigenvalue of the matrix must be real")      # eig.py:31-32
    lam = wr.real
    r = vr[:, 0].real
    l = vl[:, 0].real
    l = l / np.dot(l, r)
    return lam, l, r


def _dominant_pair_device(opA, opAT, n, k, which, device):
    """opA / opAT: native operators (``.handle``) or mat-vec callables on device vectors"""
    from . import krylov
    # both start vectors are drawn HERE, on the caller's thread and stream, in a fixed order: the two solves run
    # concurrently and must not race for the global RNG (runs stay repeatable under torch.manual_seed)
    v0r = torch.randn(n, dtype=torch.float64, device=device)
    v0l = torch.randn(n, dtype=torch.float64, device=device)
    lam, r, lam_l, l = _two_sides(lambda: krylov.arnoldi_dominant(opA, n, k, device, which, v0=v0r),
                                  lambda: krylov.arnoldi_dominant(opAT, n, k, device, which, v0=v0l), device)
    if abs(lam - lam_l) > 1e-8 * max(abs(lam), 1e-300):
        raise RuntimeError("left / right eigenvalues disagree: %.15e vs %.15e" % (lam, lam_l))
    l = l / torch.dot(l, r)
    return torch.tensor([lam], dtype=torch.float64, device=device), l, r


def _adjoint_solves_device(opA, opAT, lam, l, r, g_l, g_r):
    from . import krylov
    rhs_l = g_l - r * torch.dot(l, g_l)                                          # eig.py:53
    rhs_r = g_r - l * torch.dot(r, g_r)                                          # eig.py:56
    lam_l, lam_r = _two_sides(
        lambda: (krylov.gmres(opA, rhs_l, shift=lam, rtol=_GMRES_TOL, atol=_GMRES_TOL),),    # eig.py:54
        lambda: (krylov.gmres(opAT, rhs_r, shift=lam, rtol=_GMRES_TOL, atol=_GMRES_TOL),), rhs_l.device)
    return lam_l, lam_r


CONCURRENT_SIDES = True
_SIDE_STREAMS = {}


def _two_sides(right, left, device):
    """The right- and left-eigenvector problems (eig.py:29-30; the two adjoint solves, eig.py:54,57) are independent:
    the second one runs on a worker thread with its own HIP stream, so that one side's host work -- the small
    Hessenberg eigen-solve of a Krylov-Schur cycle, the per-cycle state read of GMRES -- overlaps the other side's
    device loop.  Workspaces and the basis arena are per stream (engine.Workspace / BasisArena), ctypes and LAPACK
    release the GIL.  Returns the concatenated results (right..., left...)."""
    if not CONCURRENT_SIDES or torch.device(device).type != "cuda":
        return tuple(right()) + tuple(left())
    import threading
    main = torch.cuda.current_stream(device)
    side = _SIDE_STREAMS.get(str(device))       # one side stream per device: its workspace / arena entries are reused
    if side is None:
        side = _SIDE_STREAMS[str(device)] = torch.cuda.Stream(device=device)
    ready = torch.cuda.Event()
    ready.record(main)
    box = {}
    grad_mode = torch.is_grad_enabled()         # grad mode is thread-local and defaults to ON in a new thread: the
                                                # worker must run under the caller's mode (forward / backward of an
                                                # autograd Function run under no_grad), or a torch-coded mat-vec that
                                                # closes over a requires_grad tensor would record a graph per step

    def work():
        try:
            with torch.set_grad_enabled(grad_mode), torch.cuda.device(device), torch.cuda.stream(side):
                side.wait_event(ready)          # the operands were produced on the caller's stream
                box["out"] = tuple(left())
                done = torch.cuda.Event()
                done.record(side)
                box["done"] = done
        except BaseException as exc:            # noqa: BLE001 -- re-raised on the caller's thread
            box["exc"] = exc

    th = threading.Thread(target=work)
    th.start()
    try:
        out_r = tuple(right())
    finally:
        th.join()
    if "exc" in box:
        raise box["exc"]
    main.wait_event(box["done"])
    for t in box["out"]:
        if torch.is_tensor(t):
            t.record_stream(main)
    return out_r + box["out"]


class DominantEig(torch.autograd.Function):
    """(eigval (1,), left eigenvector, right eigenvector) of a real diagonalisable matrix tensor."""

    @staticmethod
    def forward(ctx, A, k, which="LM"):
        if A.is_cuda:
            from .operators import DenseOperator
            Ad = A.detach().to(torch.float64).contiguous()
            ops = (DenseOperator(Ad), DenseOperator(Ad, transpose=True))     # rocBLAS GEMV inside the library loops
            lam, l, r = _dominant_pair_device(ops[0], ops[1], Ad.shape[0], k, which, A.device)
            ctx.device_path, ctx.ops, ctx.trip = True, ops, (lam, l, r)
            return lam, l, r
        ctx.device_path = False
        M = A.detach().cpu().numpy()
        lam, l, r = _dominant_pair(M, M.T, k, which)
        ctx.M, ctx.lam, ctx.l, ctx.r = M, lam, l, r
        return torch.from_numpy(lam), torch.from_numpy(l), torch.from_numpy(r)

    @staticmethod
    def backward(ctx, g_lam, g_l, g_r):
        if ctx.device_path:
            ops, (lam, l, r) = ctx.ops, ctx.trip
            lam_l, lam_r = _adjoint_solves_device(ops[0], ops[1], lam, l, r, g_l, g_r)
            gA = g_lam * l[:, None] * r - l[:, None] * lam_l - lam_r[:, None] * r        # eig.py:58-60
            return gA, None, None
        M, lam, l, r = ctx.M, ctx.lam, ctx.l, ctx.r
        g_l, g_r = g_l.numpy(), g_r.numpy()
        eye = np.eye(M.shape[0])
        rhs = g_l - r * np.dot(l, g_l)                                           # eig.py:53
        lam_l = _gmres(M - lam * eye, rhs)
        rhs = g_r - l * np.dot(r, g_r)                                           # eig.py:56
        lam_r = _gmres(M.T - lam * eye, rhs)
        gA = g_lam.numpy() * l[:, None] * r - l[:, None] * lam_l - lam_r[:, None] * r   # eig.py:58-60
        return torch.from_numpy(gA), None, None


def _on_device(op):
    """device operand: a ``krylov.TorchLinearOperator`` or a native operator (``operators.TransferOperator`` ...)"""
    from .krylov import TorchLinearOperator
    dev = getattr(op, "device", None)
    return (isinstance(op, TorchLinearOperator) or hasattr(op, "handle")) and dev is not None and dev.type == "cuda"


def _make_sparse_eig(A, AT, Aadjoint_to_gadjoint):
    class DominantSparseEig(torch.autograd.Function):
        """As DominantEig with A, A^T given as scipy LinearOperators; inputs (g, k) (eig.py:115-149)."""

        @staticmethod
        def forward(ctx, g, k):
            ctx.device_path = _on_device(A)
            if ctx.device_path:
                lam, l, r = _dominant_pair_device(A, AT, A.shape[0], k, "LM", A.device)
                ctx.trip = (lam, l, r)
                return lam, l, r
            lam, l, r = _dominant_pair(A, AT, k, "LM")
            ctx.lam, ctx.l, ctx.r = lam, l, r
            return torch.from_numpy(lam), torch.from_numpy(l), torch.from_numpy(r)

        @staticmethod
        def backward(ctx, g_lam, g_l, g_r):
            if ctx.device_path:
                lam, l, r = ctx.trip
                lam_l, lam_r = _adjoint_solves_device(A, AT, lam, l, r, g_l, g_r)
                pieces = ((g_lam * l, r), (-l, lam_l), (-lam_r, r))                      # eig.py:145-147
                return Aadjoint_to_gadjoint(pieces), None
            lam, l, r = ctx.lam, ctx.l, ctx.r
            g_lam, g_l, g_r = g_lam.numpy(), g_l.numpy(), g_r.numpy()
            shifted = sla.LinearOperator(A.shape, matvec=lambda v: A.matvec(v) - lam * v)
            lam_l = _gmres(shifted, g_l - r * np.dot(l, g_l))
            shifted_T = sla.LinearOperator(AT.shape, matvec=lambda v: AT.matvec(v) - lam * v)
            lam_r = _gmres(shifted_T, g_r - l * np.dot(r, g_r))
            pieces = ((g_lam * l, r), (-l, lam_l), (-lam_r, r))                  # eig.py:145-147
            return Aadjoint_to_gadjoint(pieces), None

    return DominantSparseEig


def setDominantSparseEig(A, AT, Aadjoint_to_gadjoint):
    """Publish ``DominantSparseEig`` as a module attribute (reference protocol, eig.py:112,151)."""
    global DominantSparseEig
    DominantSparseEig = _make_sparse_eig(A, AT, Aadjoint_to_gadjoint)
    return DominantSparseEig
