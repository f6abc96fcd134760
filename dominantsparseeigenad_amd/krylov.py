"""Device-side Krylov solvers for the NON-symmetric primitives: what the
reference delegates to SciPy on the host -- ARPACK ``eigs(A, k=1, which, ncv=k)`` (reference eig.py:29-30,
116-117) and ``gmres(A - lambda I, b, tol=1e-12, atol=1e-12)`` (eig.py:52-57,137-144).

The loops run in libdsea (include/dsea.h "non-symmetric Krylov loops"):

    arnoldi_dominant   Krylov-Schur (thick-restart Arnoldi) with ncv basis vectors.  One cycle = ONE library call
                       (``dsea_arnoldi_extend``) for native operands -- classical Gram-Schmidt on the basis-streaming
                       kernels of the Lanczos path, second pass only when the DGKS test asks for it (ARPACK's rule),
                       decided on the device, no host sync per step.  The host sees the small Hessenberg matrix once per
                       cycle: ordered real Schur form, residual test |h_{m+1,m} e_m^T y| <= tol |theta|, and -- if not
                       converged -- a restart that KEEPS the leading Schur vectors.
    gmres              restarted GMRES(20): residual, Arnoldi steps, Givens rotations, back-substitution and the
                       update on the device; the host reads 8 doubles per cycle.

An operand is either a native operator (``operators.DenseOperator`` / ``TransferOperator`` / any object exposing
``.handle``) or a Python callable / ``TorchLinearOperator`` (then the mat-vec is the caller's torch code and every
other stage is a library call).  ARPACK's implicitly restarted method and Krylov-Schur converge to the same
eigenpair (ARPACK is called with tol = 0 = machine precision); eigenvectors are compared up to the gauge the
primitives fix afterwards (l.r = 1, r.r = 1, sign of r free).
"""
from __future__ import annotations

import ctypes
import time
from ctypes import byref, c_void_p

import numpy as np
import torch

from . import _lib, engine
from ._lib import check

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


def _rank_key(evals, which):
    """larger = more wanted"""
    return {"LM": np.abs(evals), "SM": -np.abs(evals), "LR": evals.real, "SR": -evals.real}[which]


class _one_thread:
    """Small dense LAPACK calls (a 200 x 200 Hessenberg matrix) are faster on ONE thread than on a thread pool
    (measured: eig 19 ms vs 41-190 ms, LU-based inverse iteration 1.5 vs 27 ms).  The limit is process-wide state and
    the right / left solves of eig._two_sides enter it from two threads at once: it is reference-counted under a lock
    (first entrant sets it, last one restores it) -- nested per-thread contexts would restore each other's value and
    could leave the whole process single-threaded."""

    _lock = None
    _depth = 0
    _ctx = None
    _controller = None

    def __enter__(self):
        import threading
        cls = _one_thread
        if cls._lock is None:
            cls._lock = threading.Lock()
        with cls._lock:
            if cls._depth == 0:
                try:
                    # ONE controller for the life of the process: constructing it scans the loaded libraries (0.6 ms per
                    # ``threadpool_limits(...)`` -- as long as the 32 x 32 eigen-solve it was protecting); ``limit`` on a
                    # cached controller costs 15 us
                    # (the controller only knows the BLAS libraries loaded when it was built: scipy.linalg's bundled OpenBLAS
                    # must be in the process first, or its pool is never limited -- import it before the scan)
                    if cls._controller is None:
                        import scipy.linalg  # noqa: F401
                        from threadpoolctl import ThreadpoolController
                        cls._controller = ThreadpoolController()
                    # BLAS pools only (numpy's and scipy's OpenBLAS: process-wide settings, hence the reference count).
                    # The OpenMP runtime torch uses keeps its thread count PER THREAD: limiting it here and restoring it on
                    # whichever thread leaves last used to leave the caller's thread at one OpenMP thread for good.
                    cls._ctx = cls._controller.limit(limits=1, user_api="blas")
                    cls._ctx.__enter__()
                except Exception:           # threadpoolctl not installed: run as is
                    cls._ctx = None
            cls._depth += 1
        return self

    def __exit__(self, *exc):
        cls = _one_thread
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._ctx is not None:
                ctx, cls._ctx = cls._ctx, None
                ctx.__exit__(None, None, None)


def _wanted_pair(B, which):
    """wanted eigenvalue of the small matrix B and its unit eigenvector: eigenvalues only (LAPACK hseqr without the
    eigenvector back-transformation, which costs as much again) + two steps of inverse iteration"""
    from scipy.linalg import lu_factor, lu_solve
    with _one_thread():
        evals = np.linalg.eigvals(B)
        idx = _select(evals, which)
        theta = evals[idx]
        if abs(theta.imag) > 1e-9 * max(abs(theta), 1e-300):
            raise ValueError("The desired eigenvalue of the matrix must be real")      # eig.py:31-32
        th = float(theta.real)
        mm = B.shape[0]
        if mm == 1:
            return th, np.ones(1), evals
        lu = lu_factor(B - (th + 1e-10 * max(abs(th), 1e-300)) * np.eye(mm))
        y = np.ones(mm) / np.sqrt(mm)
        for _ in range(3):
            y = lu_solve(lu, y)
            y = y / np.linalg.norm(y)
    return th, y, evals


# Diagnostics of the last call ON THE CALLING THREAD (arnoldi_cycles / arnoldi_columns / arnoldi_stages, gmres_cycles /
# gmres_residual): the left and right problems of the non-symmetric primitives run on two host threads
# (eig._two_sides), so module-level "last_*" attributes would be written from both without synchronisation.
import threading as _threading  # noqa: E402

DIAG = _threading.local()


def last(name, default=None):
    """diagnostic ``name`` of the last Arnoldi / GMRES call made by the calling thread"""
    return getattr(DIAG, name, default)


# Convergence of the wanted Ritz pair is tested after STAGES of the factorisation, not only once all ncv columns exist
# (0 = the latter, ARPACK's schedule).  STAGE_FIRST: columns before the first test; later stage ends are extrapolated
# from the observed residual decay.
STAGE_FIRST = 32
STAGE_MIN = 8
STAGE_MAX = 64
# Second Gram-Schmidt pass of the library's Arnoldi extension (native operands): True = OPTIMISTIC -- the pass is not enqueued
# (6 launches per step instead of 11), the DGKS test still runs on the device in every step, and a step that fails it is handed
# back by dsea_arnoldi_status and repeated with the pass enqueued (include/dsea.h).  H and V are bit-identical either way.
OPTIMISTIC_SECOND_PASS = True
# The GMRES cycles of the adjoint solves can run the same way (a failing step ends its cycle early -- a restart -- and the rest of
# the solve enqueues the pass), but on the near-singular shifted systems of eig.py:54-57 the test fails within the first cycle of
# every solve: measured no gain (config 4 backward 10.4-11.0 ms against 9.1-10.5), so it is off unless asked for.
OPTIMISTIC_SECOND_PASS_GMRES = False
STAGE_LOG = None        # set to a list to record [columns, ms until the stage's H is on the host, host ms, residual]
# The stage test of stage s runs on the host WHILE THE DEVICE ALREADY EXECUTES STAGE s + 1 (native operands, optimistic second
# pass): the H block and the break record of a stage are copied to pinned memory behind its last kernel, the next stage is
# enqueued at once, and only then does the host wait for the copy and solve the small eigen-problem.  The device no longer
# idles for the host's 1 - 4 ms per test (40 % of a config-4 forward); the price is at most one speculative stage of columns
# that a converged test makes unnecessary (they are computed and ignored).  False = test, then enqueue (round-4 behaviour).
PIPELINED_STAGES = __import__("os").environ.get("DSEA_ARNOLDI_PIPELINED", "1") != "0"
# A stage whose end WAS the extrapolated convergence point (margin: 0.3 tol and two columns) is expected to pass its test, so
# nothing is enqueued behind it: the columns of a speculative stage there are computed in vain in the common case, and the Ritz
# combination and the other side's kernels wait behind them.  If the test fails after all, the next stage is enqueued then (one
# host test exposed).  True = always keep a speculative stage in flight.
SPECULATE_PAST_PREDICTION = __import__("os").environ.get("DSEA_ARNOLDI_SPECULATE_PAST_PREDICTION", "0") == "1"


def _speculative_stage_end(j1, p, m, hist, tol):
    """end of the stage enqueued BEHIND the stage ending at j1 whose test is still pending: the convergence history knows
    the stages up to the one before, so the extrapolated target is taken from its last point"""
    if j1 >= m:
        return None
    j2 = j1 + STAGE_FIRST // 2
    if len(hist) >= 2:
        (ja, ra), (jb, rb) = hist[-2], hist[-1]
        if rb < ra and rb > 0.0 and jb > ja:
            rate = np.log(rb / ra) / (jb - ja)
            jstar = jb + int(np.ceil(np.log(0.3 * tol / rb) / rate)) + 2
            if jstar <= j1 and not SPECULATE_PAST_PREDICTION:
                return None                 # the pending stage ends at the predicted convergence point: nothing behind it
            j2 = max(j1 + STAGE_MIN, min(jstar, j1 + STAGE_MAX))
    return m if j2 + STAGE_MIN > m else j2


class _StagePipe:
    """two pinned snapshots (H block + break record) and their events, for the pipelined stage tests of arnoldi_dominant"""

    # pinned buffers are POOLED per (m, ldh) for the life of the process: allocating and freeing pinned memory costs ~1 ms and
    # the free synchronises the device (the left solve of eig._two_sides runs on a fresh thread every call, so a thread-local
    # cache would allocate and free on every forward pass)
    _pool, _pool_lock = {}, _threading.Lock()

    def __init__(self, lp, V, ldv, Hd, ldh, m):
        self.lp, self.V, self.ldv, self.Hd, self.ldh, self.m = lp, V, ldv, Hd, ldh, m
        self.key = (m, ldh)
        with _StagePipe._pool_lock:
            free = _StagePipe._pool.setdefault(self.key, [])
            entry = free.pop() if free else None
        if entry is None:
            entry = ([torch.empty((m, ldh), dtype=F64).pin_memory() for _ in range(2)],
                     [torch.zeros(1, dtype=F64).pin_memory() for _ in range(2)], [None, None])
        for ev in entry[2]:                 # the previous user's last (speculative) copy into these buffers, long finished
            if ev is not None:
                ev.synchronize()
        self.entry = entry
        self.Hpin, self.rec = entry[0], entry[1]
        self.ev = [None, None]
        self.t_issue = [0.0, 0.0]

    def release(self):
        """give the pinned buffers back WITHOUT waiting for the speculative stage's copy: its event travels with them and the
        next user waits for it before writing"""
        if self.entry is not None:
            entry = (self.entry[0], self.entry[1], list(self.ev))
            with _StagePipe._pool_lock:
                _StagePipe._pool[self.key].append(entry)
            self.entry = None

    def extend(self, j0, j1):
        lp = self.lp
        lib, ws = lp.lib, lp.ws
        check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 1), "dsea_ws_set_arnoldi_optimistic")
        try:
            check(lib.dsea_arnoldi_extend(lp.native.handle, ws.handle, None, _ptr(self.V), self.ldv, j0, j1, _ptr(self.Hd),
                                          self.ldh, lp.st()), "dsea_arnoldi_extend")
        finally:
            check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 0), "dsea_ws_set_arnoldi_optimistic")

    def snapshot(self, j1, slot):
        """behind everything enqueued so far: H[:j1] and the break record to pinned memory, and an event"""
        lp = self.lp
        self.Hpin[slot][:j1].copy_(self.Hd[:j1], non_blocking=True)
        check(lp.lib.dsea_arnoldi_status_enqueue(lp.ws.handle, c_void_p(self.rec[slot].data_ptr()), lp.st()),
              "dsea_arnoldi_status_enqueue")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(lp.device))
        self.ev[slot] = ev
        self.t_issue[slot] = time.perf_counter()

    def wait(self, j1, slot):
        self.ev[slot].synchronize()
        return float(self.rec[slot][0]), self.Hpin[slot][:j1, :j1 + 1].numpy()


def _next_stage_end(j, p, m, hist, tol):
    """end column of the next stage of a cycle that has j columns (p of them kept from a restart)"""
    if STAGE_FIRST <= 0 or m <= STAGE_FIRST + STAGE_MIN:
        return m
    if j == p:
        return min(m, max(p + STAGE_MIN, STAGE_FIRST))
    step = STAGE_FIRST // 2
    if len(hist) >= 2:
        (ja, ra), (jb, rb) = hist[-2], hist[-1]
        if rb < ra and rb > 0.0 and jb > ja:
            rate = np.log(rb / ra) / (jb - ja)                    # < 0: decades per column
            step = int(np.ceil(np.log(0.3 * tol / rb) / rate)) + 2
    step = max(STAGE_MIN, min(STAGE_MAX, step))
    j1 = j + step
    return m if j1 + STAGE_MIN > m else j1


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)


def _native(A):
    """operator object with a C-ABI handle behind ``A`` (the object itself or its bound mat-vec), or None"""
    if hasattr(A, "handle") and hasattr(A, "n"):
        return A
    owner = getattr(A, "__self__", None)
    if owner is not None and hasattr(owner, "handle") and hasattr(owner, "n"):
        return owner
    return None


class _Loop:
    """shared plumbing of the two solvers: workspace, stream, the operand in either form"""

    def __init__(self, A, n, device, kmax):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.n = int(n)
        self.native = _native(A)
        self.callable = None if self.native is not None else (A.matvec if hasattr(A, "matvec") else A)
        self.ws = engine.Workspace.get(self.n, kmax, self.device)
        self.ldv = engine.round_up(self.n, 32)

    def st(self):
        return engine._stream(self.device)

    def apply(self, v):
        """A v as a library-ready vector"""
        if self.native is not None:
            return engine.spmv(self.native, v)
        return engine.as_vector(self.callable(v), self.n)


def arnoldi_dominant(A, n, ncv, device, which="LM", v0=None, tol=1e-13, max_restarts=60):
    """Wanted eigenvalue (real, checked as in eig.py:31-32) and unit-norm eigenvector of a real matrix given as a
    native operator or a mat-vec callable."""
    from scipy.linalg import schur

    device = torch.device(device)
    m = int(min(ncv, n))
    lp = _Loop(A, n, device, m + 2)
    lib, ws, ldv, st = lp.lib, lp.ws, lp.ldv, lp.st
    V = torch.zeros((m + 1, ldv), dtype=F64, device=device)
    ldh = m + 1
    Hd = torch.zeros((m, ldh), dtype=F64, device=device)       # row j = column j of H (column-major, ld = m + 1)
    v = torch.randn(n, dtype=F64, device=device) if v0 is None else engine.as_vector(v0, n).clone()
    nrm2 = torch.zeros(1, dtype=F64, device=device)
    check(lib.dsea_nrm2sq(ws.handle, _ptr(v), n, _ptr(nrm2), st()), "dsea_nrm2sq")
    check(lib.dsea_scale_store(ws.handle, _ptr(v), _ptr(nrm2), _ptr(V), None, n, st()), "dsea_scale_store")
    p = 0                      # vectors kept from the previous cycle (Krylov-Schur block)
    theta, x, res = None, None, float("inf")
    stages_run = 0
    DIAG.arnoldi_second_pass_redos = 0
    for cycle in range(max_restarts + 1):
        # ---- extend the factorisation towards m columns in STAGES: one library call per stage (native) / one
        # orthogonalisation call per step (callable); nothing returns to the host inside a stage.  After each stage
        # the host tests the wanted Ritz pair of the leading j x j block (ARPACK tests only once all ncv columns
        # exist, eig.py:29; a converged pair is the same pair, found with fewer mat-vecs) and chooses the end of the
        # next stage from the observed convergence rate.
        j, hist = p, []
        pipelined = PIPELINED_STAGES and lp.native is not None and OPTIMISTIC_SECOND_PASS and STAGE_FIRST > 0
        if pipelined:
            pipe = _StagePipe(lp, V, ldv, Hd, ldh, m)
            slot, j1 = 0, _next_stage_end(j, p, m, hist, tol)
            pipe.extend(j, j1)
            pipe.snapshot(j1, slot)
        # V and Hd must outlive the speculative stage that may still be in flight when the loop leaves (they do: both are
        # locals of this function and the stream is drained by the Ritz-vector combination below before they go)
        try:
            while pipelined:
                # the NEXT stage goes to the device before the host looks at this one
                j2 = _speculative_stage_end(j1, p, m, hist, tol)
                if j2 is not None:
                    pipe.extend(j1, j2)
                    pipe.snapshot(j2, 1 - slot)
                rec, Hh = pipe.wait(j1, slot)
                t_issue = pipe.t_issue[slot]
                stages_run += 1
                if rec < 0.0:
                    # step `redo` of this stage needs its second Gram-Schmidt pass: every launch behind it (the rest of the stage
                    # and the speculative one) was a no-op.  Clear the record, repeat the step in the default mode, re-enqueue.
                    brk_c, redo_c = ctypes.c_int(0), ctypes.c_int(-1)
                    check(lib.dsea_arnoldi_status(ws.handle, byref(brk_c), byref(redo_c), st()), "dsea_arnoldi_status",
                          allow=(_lib.ERR_BREAKDOWN, _lib.ERR_SECOND_PASS))
                    redo = int(-rec) - 1
                    DIAG.arnoldi_second_pass_redos = getattr(DIAG, "arnoldi_second_pass_redos", 0) + 1
                    check(lib.dsea_arnoldi_extend(lp.native.handle, ws.handle, None, _ptr(V), ldv, redo, redo + 1, _ptr(Hd), ldh,
                                                  st()), "dsea_arnoldi_extend")
                    if redo + 1 < j1:
                        pipe.extend(redo + 1, j1)
                    pipe.snapshot(j1, slot)
                    stages_run -= 1
                    continue                        # (the speculative stage is enqueued again at the top of the loop)
                brk_value = int(rec) if rec > 0.0 else 0
                me = j1 if brk_value == 0 else brk_value              # invariant subspace reached at step me
                B = Hh[:me, :me].T.copy()
                coupling = 0.0 if me < j1 or brk_value else float(Hh[j1 - 1, j1])
                j = j1
                t_dev = time.perf_counter()
                if STAGE_LOG is not None:
                    STAGE_LOG.append([j1, (t_dev - t_issue) * 1e3, None])
                try:
                    theta, y, evals = _wanted_pair(B, which)
                except ValueError:
                    if j1 >= m or me < j1:
                        raise                      # the full factorisation says the wanted eigenvalue is complex: eig.py:31-32
                    theta = None
                if theta is not None:
                    res = abs(coupling * y[-1])
                    if STAGE_LOG is not None:
                        STAGE_LOG[-1][2] = (time.perf_counter() - t_dev) * 1e3
                        STAGE_LOG[-1].append(res / max(abs(theta), 1e-300))
                    if res <= tol * abs(theta) or me < j1 or j1 >= m:
                        break
                    hist.append((j1, res / max(abs(theta), 1e-300)))
                if j2 is None:                      # the stage was expected to converge and did not: enqueue the next one now
                    j2 = _next_stage_end(j1, p, m, hist, tol)
                    pipe.extend(j1, j2)
                    pipe.snapshot(j2, 1 - slot)
                j1, slot = j2, 1 - slot
        finally:
            if pipelined:
                # a speculative stage may still be running behind the converged one: whatever it records must not reach a
                # later continuation on this cached workspace; and the pinned buffers go back to the pool on every path
                lib.dsea_arnoldi_clear_record(ws.handle, st())
                pipe.release()
        while not pipelined:
            j1 = _next_stage_end(j, p, m, hist, tol)
            brk = ctypes.c_int(0)
            if lp.native is not None and OPTIMISTIC_SECOND_PASS:
                # the stage without its second Gram-Schmidt passes; a step that turns out to need one comes back through
                # the status call, is repeated in the default mode, and the stage goes on behind it
                redo, jj = ctypes.c_int(-1), j
                t_issue = None
                while True:
                    if jj < j1:
                        check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 1), "dsea_ws_set_arnoldi_optimistic")
                        try:
                            check(lib.dsea_arnoldi_extend(lp.native.handle, ws.handle, None, _ptr(V), ldv, jj, j1, _ptr(Hd),
                                                          ldh, st()), "dsea_arnoldi_extend")
                        finally:
                            check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 0), "dsea_ws_set_arnoldi_optimistic")
                    if t_issue is None:
                        t_issue = time.perf_counter()
                    rc = check(lib.dsea_arnoldi_status(ws.handle, byref(brk), byref(redo), st()), "dsea_arnoldi_status",
                               allow=(_lib.ERR_BREAKDOWN, _lib.ERR_SECOND_PASS))
                    if rc != _lib.ERR_SECOND_PASS:
                        break
                    DIAG.arnoldi_second_pass_redos = getattr(DIAG, "arnoldi_second_pass_redos", 0) + 1
                    check(lib.dsea_arnoldi_extend(lp.native.handle, ws.handle, None, _ptr(V), ldv, redo.value, redo.value + 1,
                                                  _ptr(Hd), ldh, st()), "dsea_arnoldi_extend")
                    jj = redo.value + 1
                stages_run += 1
            else:
                if lp.native is not None:
                    check(lib.dsea_arnoldi_extend(lp.native.handle, ws.handle, None, _ptr(V), ldv, j, j1, _ptr(Hd), ldh,
                                                  st()), "dsea_arnoldi_extend")
                else:
                    for jj in range(j, j1):
                        u = lp.apply(V[jj, :n])
                        check(lib.dsea_arnoldi_orth(ws.handle, _ptr(u), None, _ptr(V), ldv, n, jj, _ptr(Hd), ldh, st()),
                              "dsea_arnoldi_orth")
                stages_run += 1
                t_issue = time.perf_counter()
                check(lib.dsea_lanczos_status(ws.handle, byref(brk), st()), "dsea_lanczos_status",
                      allow=(_lib.ERR_BREAKDOWN,))
            Hh = Hd[:j1, :j1 + 1].cpu().numpy()         # (j1, j1+1): Hh[j, i] = H[i, j]
            me = j1 if brk.value == 0 else int(brk.value)         # invariant subspace reached at step me
            B = Hh[:me, :me].T.copy()
            coupling = 0.0 if me < j1 or brk.value else float(Hh[j1 - 1, j1])
            j = j1
            t_dev = time.perf_counter()
            if STAGE_LOG is not None:
                STAGE_LOG.append([j1, (t_dev - t_issue) * 1e3, None])
            try:
                theta, y, evals = _wanted_pair(B, which)
            except ValueError:
                if j1 >= m or me < j1:
                    raise                      # the full factorisation says the wanted eigenvalue is complex: eig.py:31-32
                continue                       # an early block may well have a complex wanted Ritz value: keep going
            res = abs(coupling * y[-1])
            if STAGE_LOG is not None:
                STAGE_LOG[-1][2] = (time.perf_counter() - t_dev) * 1e3
                STAGE_LOG[-1].append(res / max(abs(theta), 1e-300))
            if res <= tol * abs(theta) or me < j1 or j1 >= m:
                break
            hist.append((j1, res / max(abs(theta), 1e-300)))
        if res <= tol * abs(theta) or me < j:
            break
        if cycle == max_restarts:
            break
        # ---- thick restart: keep the p most wanted Schur vectors (never splitting a complex pair)
        order = np.argsort(-_rank_key(evals, which))
        p_new = max(1, min(me - 2, m // 3))
        cut = _rank_key(evals, which)[order[p_new - 1]]
        keep = lambda re, im: _rank_key(np.array([complex(re, im)]), which)[0] >= cut - 1e-14 * abs(cut)  # noqa: E731
        with _one_thread():
            T, Z, sdim = schur(B, output="real", sort=keep)
            p_new = int(sdim)
            if p_new < 1 or p_new >= me:
                T, Z, sdim = schur(B, output="real",
                                   sort=lambda re, im: abs(complex(re, im) - theta) <= 1e-12 * abs(theta))
                p_new = max(int(sdim), 1)
        Zp = torch.from_numpy(np.ascontiguousarray(Z[:, :p_new].T)).to(device)          # (p, m)
        Vnew = torch.matmul(Zp, V[:me])                                                   # rocBLAS GEMM, once per restart
        last = V[me].clone()
        V[:p_new] = Vnew
        V[p_new] = last
        V[p_new + 1:] = 0.0
        Hn = np.zeros((m, ldh))
        Hn[:p_new, :p_new] = T[:p_new, :p_new].T
        Hn[:p_new, p_new] = coupling * Z[me - 1, :p_new]
        Hd.copy_(torch.from_numpy(Hn))
        p = p_new
    DIAG.arnoldi_cycles = cycle + 1
    DIAG.arnoldi_columns = j              # columns of the last cycle's factorisation when it stopped
    DIAG.arnoldi_stages = stages_run
    xv = torch.empty(n, dtype=F64, device=device)
    ys = torch.from_numpy(np.ascontiguousarray(y)).to(device)
    check(lib.dsea_ritz_combine(ws.handle, _ptr(V), ldv, n, int(ys.numel()), _ptr(ys), _ptr(xv), st()), "dsea_ritz_combine")
    x = xv / xv.norm()
    if not (res <= tol * abs(theta) or me < j):
        # the residual estimate stalls at the rounding level of the factorisation for ill-conditioned eigenvalues:
        # accept a measured residual at that level, otherwise report (ARPACK raises ArpackNoConvergence, eig.py:29)
        true_res = float((lp.apply(x) - theta * x).norm())
        if true_res > 1e-10 * abs(theta):
            raise ArnoldiNoConvergence("Arnoldi did not converge in %d restarts of %d vectors: residual %.3e relative to "
                                       "|theta| = %.3e" % (max_restarts, m, true_res, abs(theta)), theta, x, true_res)
    return theta, x


def gmres(A, b, shift=None, rtol=1e-12, atol=1e-12, restart=20, maxiter=None):
    """Restarted GMRES from x0 = 0 for (A - shift I) x = b; stops when the residual is <= max(rtol ||b||, atol)
    (scipy's rule, eig.py:54).  ``shift``: 0-dim / 1-element device tensor or None."""
    device, n = b.device, b.numel()
    b = engine.as_vector(b, n)
    m = int(min(restart, n, 64))
    lp = _Loop(A, n, device, m + 2)
    lib, ws, ldv, st = lp.lib, lp.ws, lp.ldv, lp.st
    V = torch.zeros((m + 1, ldv), dtype=F64, device=device)
    work = torch.zeros(int(lib.dsea_gmres_work_doubles(m)), dtype=F64, device=device)
    state = torch.zeros(8, dtype=F64, device=device)
    x = torch.zeros(n, dtype=F64, device=device)
    sh = None if shift is None else shift.detach().reshape(-1)[:1].to(device=device, dtype=F64).contiguous()
    bnorm = float(b.norm())
    DIAG.gmres_cycles, DIAG.gmres_residual = 0, 0.0
    if bnorm == 0.0:
        return x                       # scipy returns x0 = 0 for a zero right-hand side
    target = max(rtol * bnorm, atol)
    cycles = 10 * n if maxiter is None else int(maxiter)
    info = None
    optimistic = lp.native is not None and OPTIMISTIC_SECOND_PASS_GMRES
    DIAG.gmres_second_pass_fallbacks = 0
    for c in range(cycles):
        if lp.native is not None:
            # optimistic second pass (see OPTIMISTIC_SECOND_PASS): a step that needs it ends its cycle early (= a restart)
            # and the rest of the solve runs with the pass enqueued
            if optimistic:
                check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 1), "dsea_ws_set_arnoldi_optimistic")
            try:
                check(lib.dsea_gmres_cycle(lp.native.handle, ws.handle, _ptr(sh), _ptr(b), _ptr(x), _ptr(V), ldv, m,
                                           _ptr(work), float(target), _ptr(state), int(c == 0), st()), "dsea_gmres_cycle")
            finally:
                if optimistic:
                    check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 0), "dsea_ws_set_arnoldi_optimistic")
        else:
            Ax = None
            if c > 0:
                Ax = lp.apply(x)
                if sh is not None:
                    Ax = Ax - sh * x
                Ax = engine.as_vector(Ax, n)
            check(lib.dsea_gmres_begin(ws.handle, _ptr(b), _ptr(Ax), _ptr(V), ldv, n, m, _ptr(work), float(target),
                                       _ptr(state), st()), "dsea_gmres_begin")
            for j in range(m):
                u = lp.apply(V[j, :n])
                check(lib.dsea_gmres_step(None, ws.handle, _ptr(sh), _ptr(u), _ptr(V), ldv, n, j, m, _ptr(work),
                                          float(target), _ptr(state), st()), "dsea_gmres_step")
            check(lib.dsea_gmres_end(ws.handle, _ptr(V), ldv, n, m, _ptr(work), _ptr(state), _ptr(x), st()),
                  "dsea_gmres_end")
        info = state.cpu()
        if info[1].item() != 0.0:
            break
        if optimistic and info[5].item() != 0.0:
            optimistic = False
            DIAG.gmres_second_pass_fallbacks += 1
    DIAG.gmres_cycles = c + 1
    DIAG.gmres_residual = float(info[0].item()) if info is not None else float("nan")
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
