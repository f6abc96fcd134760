"""Device-side engine: torch tensors in, libdsea.so (HIP) calls out.

PyTorch is used here for device memory, streams and nothing else: every arithmetic
operation on an n-vector in the two hot loops goes through the C ABI of include/dsea.h.
The engine never falls back to torch arithmetic; a missing library raises (``_lib.load``).
"""
from __future__ import annotations

import ctypes
import threading
from ctypes import byref, c_double, c_int64, c_size_t, c_void_p

import numpy as np
import torch

from . import _lib
from ._lib import check

F64 = torch.float64
_CACHE_LOCK = threading.RLock()

# bf16 shadow of the basis for the correction pass of the native Lanczos loop (include/dsea.h,
# dsea_ws_set_shadow).  SHADOW_TAU is the device-side premise bound max|c_j| <= tau ||r||.
USE_SHADOW = True
SHADOW_TAU = 1e-12
# Dense tensors handed to the SYMMETRIC primitives (DominantSymeig, CGSubspace) are applied by the hand-written
# upper-triangle mat-vec (operators.SymmetricDenseOperator): only the upper triangle of the tensor is read.  Set to
# False to apply them with torch.matmul (rocBLAS GEMV on the full matrix) instead.
DENSE_SYMMETRIC_KERNEL = True
# README-sized problems (full-space TFIMOperator, Stencil3Operator; k <= 512): the whole Lanczos loop as ONE launch
# (csrc/dsea_lanczos_persist.hip).  Same algorithm and expressions, T equal to the multi-launch form to rounding (not bit
# for bit).  True = automatic (n <= 4096, where it is measured to win), "force" = wherever it applies (n <= 8192),
# False keeps every workspace on the multi-launch kernels.  The same switch governs the MID-SIZE single-launch form for
# halo-1 operators (csrc/dsea_lanczos_persist_mid.hip: 3-point stencil, 8192 < n <= 131072 rows -- BASELINE configs[2]):
# on with True / "force", off with False or "small" (= only the README-sized form).
# Gram-Schmidt passes per Lanczos step: 1 = the reference (Lanczos.py:66); 2 = CGS2 option (``reorth="twice"`` of
# Lanczos.symeigLanczos / Lanczos.Lanczos): the pass is repeated on the corrected vector.
REORTH_PASSES = 1
# Partial re-orthogonalisation (``reorth="partial"`` of Lanczos.symeigLanczos / Lanczos.Lanczos; an option the reference
# lacks): None = off (the reference's full re-orthogonalisation on every step), a float = the threshold on the estimated
# loss of orthogonality that triggers a pair of full passes (0.0 = the default 1e-10; Simon's classical value is
# sqrt(eps)).  Device operators with a fused tail (TFIM, SELL, stencil), row-partitioned operators on the library driver,
# and callables / operands without a fused tail (one phase call per step).
PARTIAL_REORTH = None
last_reorth_steps = None       # steps of the last native run that were re-orthogonalised (partial mode), else None
# The two module attributes above are the process-wide defaults.  A call that asks for an option (``reorth=`` of
# Lanczos.symeigLanczos) sets it for ITS thread only (``reorth_options``), so the left / right worker threads of eig.py
# and concurrent callers do not see each other's choice; every driver reads the option through the two accessors.
import contextlib as _contextlib
import threading as _threading
_tls = _threading.local()
_UNSET = object()


def reorth_passes():
    v = getattr(_tls, "reorth_passes", None)
    return int(REORTH_PASSES if v is None else v)


def partial_reorth():
    v = getattr(_tls, "partial_reorth", _UNSET)
    return PARTIAL_REORTH if v is _UNSET else v


@_contextlib.contextmanager
def reorth_options(passes=None, partial=_UNSET):
    """Thread-local override of REORTH_PASSES / PARTIAL_REORTH for the duration of the block."""
    prev = (getattr(_tls, "reorth_passes", None), getattr(_tls, "partial_reorth", _UNSET))
    if passes is not None:
        _tls.reorth_passes = int(passes)
    if partial is not _UNSET:
        _tls.partial_reorth = partial
    try:
        yield
    finally:
        _tls.reorth_passes, _tls.partial_reorth = prev
import os as _os
_NO_PERSIST = _os.environ.get("DSEA_NO_PERSIST", "") == "1"
LANCZOS_PERSIST = not _NO_PERSIST
last_lp_steps = (0, 0)
last_break = 0
last_truncated = 0             # dimension of the leading block the last tridiag_extreme took its pair from, 0 = all of T


def _stream(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)


def as_vector(t, n=None):
    """contiguous, fp64, 16-byte aligned device vector (copies only when it has to)."""
    if not t.is_cuda:
        raise ValueError("the HIP path was selected (device='cuda') but got a %s tensor; operator, parameters and "
                         "vectors must live on the same GPU (as in the reference, Lanczos.py:49-52)" % t.device)
    if t.dtype != F64:
        t = t.to(F64)
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    if n is not None and t.numel() != n:
        raise ValueError("expected a vector of %d elements, got %d" % (n, t.numel()))
    return t


class Workspace:
    """Caller-owned scratch handed to the library (include/dsea.h: dsea_ws_*)."""

    _cache = {}

    def __init__(self, n, kmax, device):
        self.lib = _lib.load()
        self.n, self.kmax, self.device = int(n), int(kmax), torch.device(device)
        nbytes = c_size_t()
        check(self.lib.dsea_ws_bytes(self.n, self.kmax, byref(nbytes)), "dsea_ws_bytes")
        self.buffer = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        handle = c_void_p()
        check(self.lib.dsea_ws_create(_ptr(self.buffer), nbytes.value, self.n, self.kmax, byref(handle)),
              "dsea_ws_create")
        self.handle = handle
        self.state = torch.zeros(_lib.CG_STATE_LEN, dtype=F64, device=self.device)
        self.scal = torch.zeros(16, dtype=F64, device=self.device)
        self.busy = None            # name of the solver that currently owns this workspace (see ``owned_by``)
        self.persist_mode = -1
        self.lanczos_persist_mode = -1
        if _NO_PERSIST:             # DSEA_NO_PERSIST=1: every solve on the multi-launch kernels (A/B measurements)
            self.set_persist(0)
        # A/B measurements only: DSEA_WS_SPLIT=<waves> / DSEA_WS_RPL=<rows per lane> force the geometry of the basis-streaming
        # kernels for every workspace of the process (dsea_ws_set_split / dsea_ws_set_rows_per_lane)
        if _os.environ.get("DSEA_WS_SPLIT"):
            self.set_split(int(_os.environ["DSEA_WS_SPLIT"]))
        if _os.environ.get("DSEA_WS_RPL"):
            self.set_rows_per_lane(int(_os.environ["DSEA_WS_RPL"]))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.dsea_ws_destroy(self.handle)
        except Exception:
            pass

    _CACHE_LIMIT = 8   # workspaces kept alive (one per (n, device, stream)); the oldest is dropped beyond this

    @classmethod
    def get(cls, n, kmax, device):
        """The workspace of (n, device, CURRENT STREAM).  include/dsea.h: one workspace per stream, a workspace is
        not re-entrant -- two streams (or threads driving different streams) solving problems of the same size get
        distinct partial-sum / scalar / CG-state buffers.  A solver invoked from INSIDE a user mat-vec of another
        solver of the same size on the same stream would share them: not supported."""
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = (int(n), str(device), int(torch.cuda.current_stream(device).cuda_stream))
        with _CACHE_LOCK:               # two host threads may drive two streams (eig._two_sides)
            ws = cls._cache.pop(key, None)
            if ws is None or ws.kmax < kmax:
                ws = cls(n, max(int(kmax), 8), device)
            cls._cache[key] = ws            # re-insert: dict order = recency
            while len(cls._cache) > cls._CACHE_LIMIT:
                cls._cache.pop(next(iter(cls._cache)))
        return ws

    @classmethod
    def clear_cache(cls):
        cls._cache.clear()

    def owned_by(self, who):
        """Context manager: a SOLVER (Lanczos / CG loop) owns the workspace for its duration.  A workspace is not
        re-entrant (include/dsea.h): a solver started from inside the user mat-vec of another solver of the same size
        on the same stream would share its scalar / CG-state / partial-sum buffers and corrupt both silently -- that
        nesting raises instead.  (Single phase calls -- dots, projections -- from inside a mat-vec are fine: they keep
        no state in the workspace between calls.)  Remedy for a genuinely nested solve: run it on another stream."""
        return _Owned(self, who)

    def set_rows_per_lane(self, rpl):
        check(self.lib.dsea_ws_set_rows_per_lane(self.handle, int(rpl)), "dsea_ws_set_rows_per_lane")

    def set_persist(self, mode):
        check(self.lib.dsea_ws_set_persist(self.handle, int(mode)), "dsea_ws_set_persist")
        self.persist_mode = int(mode)

    def set_lanczos_persist(self, mode):
        """single-launch Lanczos for README-sized problems (include/dsea.h dsea_ws_set_lanczos_persist): -1 = automatic
        (on where it applies), 0 = off"""
        check(self.lib.dsea_ws_set_lanczos_persist(self.handle, int(mode)), "dsea_ws_set_lanczos_persist")
        self.lanczos_persist_mode = int(mode)

    def set_split(self, waves):
        check(self.lib.dsea_ws_set_split(self.handle, int(waves)), "dsea_ws_set_split")


class _Owned:
    def __init__(self, ws, who):
        self.ws, self.who = ws, who

    def __enter__(self):
        if self.ws.busy is not None:
            raise RuntimeError("%s was started while %s is running on the same workspace (same vector length %d, same "
                               "device, same stream): a workspace is not re-entrant -- a solver nested inside the "
                               "mat-vec of another one must run on a different stream (torch.cuda.stream(...))"
                               % (self.who, self.ws.busy, self.ws.n))
        self.ws.busy = self.who
        return self.ws

    def __exit__(self, *exc):
        self.ws.busy = None


def round_up(v, m):
    return (v + m - 1) // m * m


class BasisArena:
    """Persistent device buffers for the TRANSIENT Krylov basis (and its bf16 shadow) of the eigen-solves.

    The basis is the one large object of the path (8 n k bytes: 1.7 GB at the headline size, 215 GB at L = 28,
    k = 100).  ``symeigLanczos`` only hands the Ritz vector on, so the basis of one call is dead when the next call
    starts: allocating it afresh every time makes the caching allocator carve, split and re-map blocks of that size
    (at L = 28 the second call failed with 200 GiB "reserved but unallocated").  One buffer per (device, stream,
    tag), grown on demand, reused by every call.  ``Lanczos()`` -- which returns the basis to the caller -- does not
    use it.  ``release()`` gives the memory back; the arena also shrinks by itself when a request is far below what it
    holds (SHRINK_RATIO).  Memory cost beyond the basis itself: the placement probe's transient candidates, capped by
    PLACEMENT_BUDGET_BYTES (8 GiB)."""

    _bufs = {}
    # PLACEMENT.  The dots pass runs in one of two modes (256 vs 272 us per pass at n = 2^20, i = 199: 6.67 vs
    # 6.28 TB/s) that are a property of WHERE the basis lives, not of the code: on one MI355X box the first large
    # allocation of a process was slow and every one of seven later allocations of the same shape fast, reproducibly
    # per buffer while all were alive; on another box all nine candidates of a process were slow; bench processes on
    # one box alternate (tools/placement_probe.py, profiles/r02_placement_probe.txt).  Initialised or untouched
    # memory, random or zero data make no difference.  Physical placement is not visible from user space, so it is
    # MEASURED: when the arena has to allocate a large fp64 basis it takes up to PLACEMENT_TRIES candidates that are
    # alive at the same time, times the real dots pass on each (three launches on whatever bytes the memory holds)
    # and keeps the fastest.  Costs a few ms once per arena allocation and helps only where candidates differ; the
    # extra candidates are capped by PLACEMENT_BUDGET_BYTES (none for a basis above 8 GiB).  PLACEMENT_TRIES 0 / 1 = off.
    PLACEMENT_TRIES = 3
    PLACEMENT_MIN_BYTES = 128 << 20
    # The probe's EXTRA candidates (alive beside the one that is kept, for a few ms) may use at most this many bytes
    # in total: the headline basis (1.7 GB) gets its three candidates, a 54 GB slab basis (config 5) gets none --
    # placement probing never multiplies a large allocation.  0 switches the probe off.
    PLACEMENT_BUDGET_BYTES = 8 << 30
    # The arena SHRINKS: a request far below the held capacity (less than a quarter, and at least 1 GiB to give back)
    # replaces the buffer, so one L = 28 solve does not pin 215 GB for the life of the process.
    SHRINK_RATIO = 4
    SHRINK_MIN_BYTES = 1 << 30
    last_placement = None           # [us per probe pass of each candidate], for the curious
    # When the arena REPLACES a buffer (growth, shrink) or drops the probe's losing candidates it hands the freed blocks back
    # to the driver with ``torch.cuda.empty_cache()`` -- at L = 28 the caching allocator otherwise keeps 200 GiB "reserved
    # but unallocated" and the next basis does not fit.  That call also releases every OTHER cached block of the process
    # (a training loop pays re-allocation for them): set False to keep the arena from touching the allocator's cache; the
    # freed buffers then stay cached by torch like any other tensor's memory ("memory").
    RELEASE_TO_DRIVER = True

    @classmethod
    def get(cls, device, tag, nbytes, probe=None):
        device = torch.device(device)
        key = (str(device), int(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else 0, tag)
        nbytes = int(nbytes)
        with _CACHE_LOCK:
            buf = cls._bufs.get(key)
            too_big = buf is not None and buf.numel() > cls.SHRINK_RATIO * max(nbytes, 1) and \
                buf.numel() - nbytes >= cls.SHRINK_MIN_BYTES
            if buf is None or buf.numel() < nbytes or too_big:
                if buf is not None:
                    del cls._bufs[key], buf
                    if device.type == "cuda" and cls.RELEASE_TO_DRIVER:
                        torch.cuda.empty_cache()      # the old buffer goes back to the driver before the new one is taken
                tries = 1
                if probe is not None and device.type == "cuda" and nbytes >= cls.PLACEMENT_MIN_BYTES:
                    free_b, _ = torch.cuda.mem_get_info(device)
                    tries = max(1, min(int(cls.PLACEMENT_TRIES), 1 + int(cls.PLACEMENT_BUDGET_BYTES // max(nbytes, 1)),
                                       int(0.7 * free_b // max(nbytes, 1))))
                buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
                if tries > 1:
                    cands, nxt = [(probe(buf), buf)], None
                    for _ in range(tries - 1):
                        try:
                            nxt = torch.empty(nbytes, dtype=torch.uint8, device=device)
                        except RuntimeError:          # out of memory after all: keep what we have
                            break
                        cands.append((probe(nxt), nxt))
                    cls.last_placement = [t for t, _ in cands]
                    buf = min(cands, key=lambda tb: tb[0])[1]
                    del cands, nxt
                    if cls.RELEASE_TO_DRIVER:
                        torch.cuda.empty_cache()      # the losing candidates go back to the driver
                cls._bufs[key] = buf
        return buf

    @classmethod
    def matrix(cls, device, tag, rows, cols, dtype, n_hint=None):
        """(rows, cols) view of the arena buffer ``tag``.  ``n_hint`` (fp64 bases): the vector length of the solver
        that will stream it -- enables the placement selection above."""
        esz = torch.empty(0, dtype=dtype).element_size()
        probe = None
        if n_hint is not None and dtype == F64 and rows >= 8:
            probe = lambda b: _dots_probe_us(b, int(rows), int(cols), int(n_hint), torch.device(device))  # noqa: E731
        buf = cls.get(device, tag, rows * cols * esz, probe)
        return buf[: rows * cols * esz].view(dtype).view(rows, cols)

    @classmethod
    def release(cls):
        cls._bufs.clear()


def _dots_probe_us(buf, rows, ldq, n, device):
    """microseconds per dots pass (dsea_lanczos_rdots, all rows - 1 vectors) over the candidate basis buffer ``buf``;
    the values in the buffer are whatever the memory holds (the results are scratch)"""
    lib = _lib.load()
    ws = Workspace.get(n, rows, device)
    st = _stream(device)
    i = rows - 1
    u = torch.zeros(n, dtype=F64, device=device)
    r = torch.empty(n, dtype=F64, device=device)
    c = torch.zeros(rows + 2, dtype=F64, device=device)
    ab = torch.zeros(2, dtype=F64, device=device)

    def launch():
        check(lib.dsea_lanczos_rdots(ws.handle, _ptr(buf), int(ldq), int(n), int(i), _ptr(u), _ptr(ab),
                                     c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st), "dsea_lanczos_rdots")

    launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        launch()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 3 * 1e3


def shadow_fits(device, k, ldq, n, arena=False):
    """True if the bf16 shadow of a (k, ldq) basis can be allocated next to it with room for the work vectors of a
    forward + backward pass (8 n-vectors).  At L = 28, k = 100 on one 288 GB GPU the fp64 basis is 215 GB and the shadow
    would be another 54 GB: the solve then runs with the all-fp64 correction pass instead of dying in the allocator."""
    device = torch.device(device)
    if device.type != "cuda":
        return True
    need = 2 * int(k) * int(ldq)
    if arena:
        held = BasisArena._bufs.get((str(device), int(torch.cuda.current_stream(device).cuda_stream), "Qs"))
        if held is not None and held.numel() >= need:
            return True
    free_b, _ = torch.cuda.mem_get_info(device)
    cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    return need + 8 * 8 * int(n) <= free_b + max(int(cached), 0)


def native_of(A):
    """The native operator behind ``A`` or None.  ``A`` may be the operator object itself or its bound
    mat-vec method (``model.H``, the form the reference examples pass: examples/TFIM/E0.py:60)."""
    if hasattr(A, "_native_methods"):
        return A
    owner = getattr(A, "__self__", None)
    if owner is not None and hasattr(owner, "_native_methods") and getattr(A, "__name__", "") in owner._native_methods:
        return owner
    return None


# --------------------------------------------------------------------------- phase kernels as methods
class Phases:
    """The vector-phase entry points of include/dsea.h on n-vectors of one device (thin ctypes wrappers;
    every argument is a torch tensor, scalars are 1-element device tensors)."""

    def __init__(self, n, device, kmax=8):
        self.lib = _lib.load()
        self.n, self.device = int(n), torch.device(device)
        self.ws = Workspace.get(self.n, kmax, self.device)

    def reserve(self, k):
        self.ws = Workspace.get(self.n, k, self.device)

    def _st(self):
        return _stream(self.device)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=F64, device=self.device)

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=F64, device=self.device)

    def axpy(self, a_host, a_dev, x, y):
        check(self.lib.dsea_axpy(self.ws.handle, float(a_host), _ptr(a_dev), _ptr(x), _ptr(y), x.numel(), self._st()),
              "dsea_axpy")

    def dot(self, x, y, out):
        check(self.lib.dsea_dot(self.ws.handle, _ptr(x), _ptr(y), x.numel(), _ptr(out), self._st()), "dsea_dot")

    def scale_store(self, r, nrm2, q_out, beta_out):
        check(self.lib.dsea_scale_store(self.ws.handle, _ptr(r), _ptr(nrm2), _ptr(q_out), _ptr(beta_out), r.numel(),
                                        self._st()), "dsea_scale_store")

    def rdots(self, Q, ldq, n, i, u, alpha, beta, r, c):
        check(self.lib.dsea_lanczos_rdots(self.ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(alpha), _ptr(beta),
                                          _ptr(r), _ptr(c), self._st()), "dsea_lanczos_rdots")

    def axpy_norm(self, Q, ldq, n, i, c, r, nrm2):
        check(self.lib.dsea_lanczos_axpy_norm(self.ws.handle, _ptr(Q), ldq, n, i, _ptr(c), _ptr(r), _ptr(nrm2),
                                              self._st()), "dsea_lanczos_axpy_norm")

    def ritz(self, Q, ldq, n, k, s, out):
        check(self.lib.dsea_ritz_combine(self.ws.handle, _ptr(Q), ldq, n, k, _ptr(s), _ptr(out), self._st()),
              "dsea_ritz_combine")


class PartialNeedsPhases(NotImplementedError):
    """reorth='partial' on a native operand whose whole-loop entry point does not take the option: the caller repeats the
    run through the phase calls (Lanczos._lanczos_core does)"""


# --------------------------------------------------------------------------- Lanczos
def lanczos(A, k, n, device, q0, native=None, callable_A=None, arena=False):
    """k-step Lanczos on the GPU (reference Lanczos.py:49-77).

    native     : operator object exposing ``.handle`` -> whole loop in dsea_lanczos_run
    callable_A : python callable v -> A v (torch tensors on ``device``) -> one C-ABI phase call per
                 stage, the mat-vec itself is the user's code
    Returns (Q (k, ldq) basis, ldq, alphas (k,), betas (k-1,)).  ``last_break`` (module attribute) holds the step
    at which the native loop met beta ~ 0 and stopped itself (0 = none; include/dsea.h dsea_lanczos_status).
    """
    global last_break
    last_break = 0
    lib = _lib.load()
    device = torch.device(device)
    ws = Workspace.get(n, k, device)
    st = _stream(device)
    ldq = round_up(n, 32)
    # arena: the caller does not keep the basis (symeigLanczos) -> persistent buffer instead of a fresh allocation
    Q = BasisArena.matrix(device, "Q", k, ldq, F64, n_hint=n) if arena else torch.empty((k, ldq), dtype=F64, device=device)
    new_shadow = (lambda: BasisArena.matrix(device, "Qs", k, ldq, torch.bfloat16)) if arena else \
        (lambda: torch.empty((k, ldq), dtype=torch.bfloat16, device=device))
    alphas = torch.empty(k, dtype=F64, device=device)
    betas = torch.empty(max(k - 1, 1), dtype=F64, device=device)
    q0 = as_vector(q0, n)
    PARTIAL_REORTH, REORTH_PASSES = partial_reorth(), reorth_passes()      # (this thread's view of the two options)
    partial = PARTIAL_REORTH is not None
    if partial and REORTH_PASSES != 1:
        raise NotImplementedError("reorth='partial' and reorth='twice' exclude each other")
    use_shadow = USE_SHADOW and k > 1 and not partial and shadow_fits(device, k, ldq, n, arena)
    global last_reorth_steps
    last_reorth_steps = None
    if native is not None:
        shadow = None
        with ws.owned_by("Lanczos (native operator)"):
            if use_shadow:
                shadow = new_shadow()
                check(lib.dsea_ws_set_shadow(ws.handle, _ptr(shadow), ldq, int(k), float(SHADOW_TAU)), "dsea_ws_set_shadow")
            try:
                if getattr(ws, "reorth_passes", 1) != int(REORTH_PASSES):
                    check(lib.dsea_ws_set_reorth_passes(ws.handle, int(REORTH_PASSES)), "dsea_ws_set_reorth_passes")
                    ws.reorth_passes = int(REORTH_PASSES)
                if getattr(ws, "partial_reorth", None) != PARTIAL_REORTH:
                    check(lib.dsea_ws_set_partial_reorth(ws.handle, 1 if partial else 0, float(PARTIAL_REORTH or 0.0)),
                          "dsea_ws_set_partial_reorth")
                    ws.partial_reorth = PARTIAL_REORTH
                want = 0 if (not LANCZOS_PERSIST or getattr(ws, "lanczos_persist_lost", False)) else \
                    (1 if LANCZOS_PERSIST == "force" else (2 if LANCZOS_PERSIST == "small" else -1))
                if ws.lanczos_persist_mode != want:
                    ws.set_lanczos_persist(want)
                rc0 = lib.dsea_lanczos_run(native.handle, ws.handle, int(k), _ptr(q0), _ptr(Q), ldq, _ptr(alphas),
                                           _ptr(betas), st)
                if partial and rc0 == _lib.ERR_UNSUPPORTED:
                    # an operand without a fused Lanczos tail (plain CSR, dense): the option runs through the phase calls
                    raise PartialNeedsPhases("operand without a fused tail")
                check(rc0, "dsea_lanczos_run")
                brk = ctypes.c_int(0)
                rc = lib.dsea_lanczos_status(ws.handle, byref(brk), st)
                if rc == _lib.ERR_TIMEOUT:
                    # the single-launch form (README-sized problems) needs its <= 64 workgroups resident together; on a
                    # device shared with other work its bounded spins give up.  Repeat with the multi-launch kernels and
                    # keep this workspace on them (``ws.lanczos_persist_lost = False`` re-enables the single-launch form).
                    import warnings
                    warnings.warn("single-launch Lanczos timed out waiting for a peer workgroup (device shared with "
                                  "other work?): repeating the run with the multi-launch kernels, which this workspace "
                                  "keeps using from now on", RuntimeWarning)
                    ws.lanczos_persist_lost = True
                    ws.set_lanczos_persist(0)
                    check(lib.dsea_lanczos_run(native.handle, ws.handle, int(k), _ptr(q0), _ptr(Q), ldq, _ptr(alphas),
                                               _ptr(betas), st), "dsea_lanczos_run")
                    rc = lib.dsea_lanczos_status(ws.handle, byref(brk), st)
                check(rc, "dsea_lanczos_status", allow=(_lib.ERR_BREAKDOWN,))
                last_break = int(brk.value)
                if partial:
                    cnt, an = ctypes.c_int64(0), c_double(0.0)
                    check(lib.dsea_lanczos_reorth_stats(ws.handle, byref(cnt), byref(an), st), "dsea_lanczos_reorth_stats")
                    last_reorth_steps = int(cnt.value)
            finally:
                if shadow is not None:
                    check(lib.dsea_ws_set_shadow(ws.handle, None, 0, 0, 0.0), "dsea_ws_set_shadow")
        if last_break:
            # The device loop stopped itself at step last_break (beta ~ 0: the Krylov space of q0 is exhausted).  What
            # lies behind was never written: make it recognisable instead of handing out stale memory -- NaN alphas,
            # zero betas, ZERO basis vectors -- and tell the caller (the reference has no test: Lanczos.py:69-70).
            import warnings
            alphas[last_break:] = float("nan")
            betas[last_break - 1:] = 0.0
            Q[last_break:].zero_()
            warnings.warn("Lanczos breakdown at step %d of %d: the Krylov space of the start vector has dimension %d; "
                          "basis vectors %d.. are returned as zeros, T[%d:, %d:] holds NaN on its diagonal"
                          % (last_break, k, last_break, last_break, last_break, last_break), RuntimeWarning)
        return Q, ldq, alphas, betas[: k - 1]

    # generic callable: the mat-vec is the caller's torch code, everything else is one phase call per stage;
    # the bf16 shadow is kept current by dsea_lanczos_store and used by dsea_lanczos_axpy_norm
    nrm2 = ws.scal[0:1]
    r = torch.empty(n, dtype=F64, device=device)
    c = torch.empty(k + 2, dtype=F64, device=device)
    shadow = None
    esz = 8
    zero = torch.zeros(1, dtype=F64, device=device)
    with ws.owned_by("Lanczos (callable operator)"):
        if getattr(ws, "partial_reorth", None) != PARTIAL_REORTH:
            check(lib.dsea_ws_set_partial_reorth(ws.handle, 1 if partial else 0, float(PARTIAL_REORTH or 0.0)),
                  "dsea_ws_set_partial_reorth")
            ws.partial_reorth = PARTIAL_REORTH
        if use_shadow:
            shadow = new_shadow()
            check(lib.dsea_ws_set_shadow(ws.handle, _ptr(shadow), ldq, int(k), float(SHADOW_TAU)), "dsea_ws_set_shadow")
        try:
            check(lib.dsea_nrm2sq(ws.handle, _ptr(q0), n, _ptr(nrm2), st), "dsea_nrm2sq")
            check(lib.dsea_lanczos_store(ws.handle, _ptr(q0), _ptr(nrm2), _ptr(Q), ldq, 0, None, n, st), "dsea_lanczos_store")
            u = as_vector(callable_A(Q[0, :n]), n)
            fused_step = CALLABLE_LANCZOS_FUSED_STEP and not partial and int(REORTH_PASSES) == 1
            if fused_step:
                # two library calls per step (dsea_lanczos_callable_step: the four launches of the native step with a fused
                # normalise-and-store; dsea_lanczos_callable_alpha: q.u as partials its consumer sums) instead of five phase calls
                # and eight launches; the mat-vec stays the caller's code
                check(lib.dsea_lanczos_callable_alpha(ws.handle, _ptr(Q), _ptr(u), n, _ptr(alphas) if k == 1 else None, st),
                      "dsea_lanczos_callable_alpha")
                for i in range(1, k):
                    check(lib.dsea_lanczos_callable_step(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(alphas), _ptr(betas),
                                                         _ptr(r), st), "dsea_lanczos_callable_step")
                    qi = Q[i]
                    u = as_vector(callable_A(qi[:n]), n)
                    last = c_void_p(alphas.data_ptr() + i * esz) if i == k - 1 else None
                    check(lib.dsea_lanczos_callable_alpha(ws.handle, _ptr(qi), _ptr(u), n, last, st), "dsea_lanczos_callable_alpha")
            else:
                check(lib.dsea_dot(ws.handle, _ptr(Q), _ptr(u), n, _ptr(alphas), st), "dsea_dot")
            for i in range(1, k if not fused_step else 1):
                a_ptr = c_void_p(alphas.data_ptr() + (i - 1) * esz)
                b_ptr = c_void_p(betas.data_ptr() + (i - 2) * esz) if i >= 2 else c_void_p(None)
                if partial:
                    # the option's device-side sequence as ONE phase call (three-term update, omega estimates, gated dots
                    # and correction); the mat-vec below stays the caller's code
                    check(lib.dsea_lanczos_partial_step(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(alphas), _ptr(betas),
                                                        _ptr(r), _ptr(nrm2), st), "dsea_lanczos_partial_step")
                else:
                    check(lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), a_ptr, b_ptr, _ptr(r), _ptr(c), st),
                          "dsea_lanczos_rdots")
                    check(lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), ldq, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st),
                          "dsea_lanczos_axpy_norm")
                if int(REORTH_PASSES) == 2:      # CGS2 option: the same pass again on the corrected vector (alpha = 0)
                    r2 = r.clone()
                    check(lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(r2), _ptr(zero), None, _ptr(r),
                                                 _ptr(c), st), "dsea_lanczos_rdots")
                    check(lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), ldq, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st),
                          "dsea_lanczos_axpy_norm")
                check(lib.dsea_lanczos_store(ws.handle, _ptr(r), _ptr(nrm2), _ptr(Q), ldq, i,
                                             c_void_p(betas.data_ptr() + (i - 1) * esz), n, st), "dsea_lanczos_store")
                qi = Q[i]
                u = as_vector(callable_A(qi[:n]), n)
                check(lib.dsea_dot(ws.handle, _ptr(qi), _ptr(u), n, c_void_p(alphas.data_ptr() + i * esz), st), "dsea_dot")
            if partial and k > 1:
                cnt, an = ctypes.c_int64(0), c_double(0.0)
                check(lib.dsea_lanczos_reorth_stats(ws.handle, byref(cnt), byref(an), st), "dsea_lanczos_reorth_stats")
                last_reorth_steps = int(cnt.value)
        finally:
            if shadow is not None:
                check(lib.dsea_ws_set_shadow(ws.handle, None, 0, 0, 0.0), "dsea_ws_set_shadow")
    return Q, ldq, alphas, betas[: k - 1]


def lanczos_basisfree(native, k, n, device, q0, which="min"):
    """Basis-free two-pass Lanczos (include/dsea.h dsea_lanczos_run_basisfree): no stored basis, no
    re-orthogonalisation.  Pass 1 -> tridiagonal; host Ritz solve; pass 2 replays the recurrence and accumulates the
    Ritz vector(s).  Returns [(eigval, unit Ritz vector), ...] for ``which`` in {"min", "max", "both"}.
    Memory: 3 rotating vectors + work vectors instead of k basis vectors (L = 28, k = 200: 15 GB instead of 429 GB)."""
    global last_break
    lib = _lib.load()
    device = torch.device(device)
    ws = Workspace.get(n, 8, device)
    st = _stream(device)
    ldq = round_up(n, 32)
    Qrot = torch.empty((3, ldq), dtype=F64, device=device)
    alphas = torch.empty(k, dtype=F64, device=device)
    betas = torch.zeros(max(k - 1, 1), dtype=F64, device=device)
    q0 = as_vector(q0, n)
    with ws.owned_by("Lanczos (basis-free)"):
        check(lib.dsea_lanczos_run_basisfree(native.handle, ws.handle, int(k), _ptr(q0), _ptr(Qrot), ldq, _ptr(alphas),
                                             _ptr(betas), None, None, st), "dsea_lanczos_run_basisfree")
    brk = ctypes.c_int(0)
    check(lib.dsea_lanczos_status(ws.handle, byref(brk), st), "dsea_lanczos_status", allow=(_lib.ERR_BREAKDOWN,))
    last_break = int(brk.value)
    if last_break:
        alphas[last_break:] = float("nan")
    out = []
    for val, s_host in tridiag_extreme(alphas, betas[: k - 1], which, break_at=last_break or None):
        m = int(s_host.shape[0])
        s = torch.zeros(k, dtype=F64, device=device)
        s[:m] = torch.from_numpy(np.ascontiguousarray(s_host)).to(device)
        psi = torch.empty(n, dtype=F64, device=device)
        a2 = torch.empty(k, dtype=F64, device=device)
        b2 = torch.zeros(max(k - 1, 1), dtype=F64, device=device)
        check(lib.dsea_lanczos_run_basisfree(native.handle, ws.handle, int(m), _ptr(q0), _ptr(Qrot), ldq, _ptr(a2),
                                             _ptr(b2), _ptr(s), _ptr(psi), st), "dsea_lanczos_run_basisfree")
        nrm = torch.zeros(1, dtype=F64, device=device)
        check(lib.dsea_nrm2sq(ws.handle, _ptr(psi), n, _ptr(nrm), st), "dsea_nrm2sq")
        unit = torch.empty(n, dtype=F64, device=device)
        check(lib.dsea_scale_store(ws.handle, _ptr(psi), _ptr(nrm), _ptr(unit), None, n, st), "dsea_scale_store")
        out.append((val, unit))
    return out


def lanczos_lp_stats(n, device):
    """(steps that streamed the bf16 shadow, steps that fell back to the fp64 basis) of the last native run"""
    lib = _lib.load()
    ws = Workspace.get(n, 8, device)
    a, b = c_int64(0), c_int64(0)
    check(lib.dsea_lanczos_lp_stats(ws.handle, byref(a), byref(b), _stream(torch.device(device))), "dsea_lanczos_lp_stats")
    return a.value, b.value


def tridiag_extreme(alphas, betas, which, break_at=None):
    """Extreme eigenpair(s) of the k x k tridiagonal T (reference Lanczos.py:98: dense symeig of T).

    T is tiny (k ~ 200): it is solved on the host with LAPACK through scipy (one 3 kB D2H copy,
    which is also the only host sync of the forward pass).  Returns [(eigval, s (m,) numpy), ...] with m = k, or
    the dimension of the Krylov space if the process broke down earlier.
    """
    from scipy.linalg import eigh_tridiagonal

    global last_truncated
    last_truncated = 0
    d = alphas.detach().cpu().numpy()
    e = betas.detach().cpu().numpy()
    k = d.shape[0]
    out = []
    m = k
    if break_at is not None and 0 < int(break_at) < k:
        m = int(break_at)           # the device loop stopped itself there (dsea_lanczos_status)
    if k > 1:
        # The reference has no breakdown test (Lanczos.py:69-70 divides by beta whatever it is; SURVEY Q8: with
        # k beyond the Krylov dimension it normalises rounding noise and returns spurious Ritz values without
        # any error; with an EXACT breakdown, beta = 0, it produces NaNs from there on).  Here the invariant
        # subspace is recognised: the tridiagonal is cut at the first j where beta_j is zero, non-finite or
        # negligible, or where alpha_{j+1} is non-finite -- its leading block carries exact eigenpairs of A --
        # and the caller is told.  The scale is taken over the finite entries only (a NaN must not poison it).
        fin_d, fin_e = d[np.isfinite(d)], e[np.isfinite(e)]
        scale = max(float(np.abs(fin_d).max()) if fin_d.size else 0.0,
                    float(np.abs(fin_e).max()) if fin_e.size else 0.0, 1e-300)
        bad_e = ~np.isfinite(e[:m - 1]) | (np.abs(e[:m - 1]) <= 1e-13 * scale)
        bad_d = ~np.isfinite(d[1:m])
        bad = np.where(bad_e | bad_d)[0]
        if bad.size:
            m = int(bad[0]) + 1
        if m < k:
            import warnings
            last_truncated = m
            bval = float(np.abs(e[m - 1])) if np.isfinite(e[m - 1]) else float("nan")
            warnings.warn("Lanczos breakdown: beta_%d = %.3e relative to %.3e -- the Krylov space from this start "
                          "vector has dimension %d < k = %d; the Ritz pair is taken from the leading %d x %d block"
                          % (m - 1, bval, scale, m, k, m, m), RuntimeWarning)
    # the coefficient vectors have length m (<= k): only the leading m basis vectors enter the Ritz vector (the
    # ones behind a breakdown are not defined)
    if m == 1:
        return [(float(d[0]), np.ones(1))] * (2 if which == "both" else 1)
    picks = {"min": [0], "max": [m - 1], "both": [0, m - 1]}[which]
    for idx in picks:
        w, v = eigh_tridiagonal(d[:m], e[:m - 1], select="i", select_range=(idx, idx))
        out.append((float(w[0]), np.ascontiguousarray(v[:, 0])))
    return out


def ritz_vector(Q, ldq, n, k, s_host, device):
    """sum_j s[j] Q[j] over the len(s) <= k leading basis vectors"""
    lib = _lib.load()
    ws = Workspace.get(n, k, device)
    s = torch.from_numpy(np.asarray(s_host, dtype=np.float64)).to(device)
    k = int(s.numel())
    out = torch.empty(n, dtype=F64, device=device)
    check(lib.dsea_ritz_combine(ws.handle, _ptr(Q), ldq, n, int(k), _ptr(s), _ptr(out), _stream(device)),
          "dsea_ritz_combine")
    return out


# --------------------------------------------------------------------------- CG
class CGInfo:
    """iteration count / final residual norm of the last solve (diagnostics; reference prints nothing)."""
    iters = 0
    resnorm = float("nan")
    converged = True
    # which form ran (dsea_cg_last_form): "streaming" | "persistent" (one launch, the reference's recurrences, bit-identical
    # to streaming) | "persistent, one exchange" (Chronopoulos-Gear recurrences: same iteration in exact arithmetic, not
    # CG.py:31-40's rounding sequence -- the default for the full-space TFIM operand at 2^11 ... 2^20 rows); a persistent
    # launch that timed out is repeated in the streaming form and says so here
    form = "streaming"
    polls = 0          # host looks at the device-side stop flag (callable operand: one synchronising copy each)


last_cg = CGInfo()
CG_FORMS = ("streaming", "persistent", "persistent, one exchange")


# Latency-bound solves on small halo-1 operators (BASELINE config 3): run the persistent single-launch CG in its
# MERGED-REDUCTION form -- one grid-wide exchange per iteration instead of two (3.3-3.5 instead of 5.4 us per
# iteration at N = 1e5).  The same iteration in exact arithmetic, not the rounding sequence of reference
# CG.py:31-40 (measured: 6e-14 relative after 40 iterations, same iteration counts on converged runs), hence off
# by default; ``cg(..., merged_reductions=True)`` selects it per call.
CG_MERGED_REDUCTIONS = False
# Full-space TFIM operator at 2^11 ... 2^20 rows (the adjoint solve of BASELINE configs[1]): the single-launch CG makes ONE
# grid-wide exchange per iteration by default (Chronopoulos-Gear recurrences; same iteration in exact arithmetic, iterates
# within ~1e-13 of the reference's recurrences, same iteration counts).  True selects the two-exchange form whose iterates
# are BIT-IDENTICAL to the streaming kernels / CG.py:31-40 evaluated in fp64 (23 instead of ~15 us per iteration at L = 20).
CG_TFIM_REFERENCE_RECURRENCES = _os.environ.get("DSEA_CG_REFERENCE_RECURRENCES", "") == "1"


# Lanczos around an opaque callable: two library calls per step (dsea_lanczos_callable_step / _alpha) instead of five phase calls.
# False = the phase calls (A/B, tests); the partial / CGS2 options always take the phase calls.
CALLABLE_LANCZOS_FUSED_STEP = _os.environ.get("DSEA_CALLABLE_LANCZOS_FUSED", "1") != "0"
# CG around an opaque callable (the reference's own calling convention): one library call per iteration (dsea_cg_step: the three
# fused launches of the native streaming form) instead of the four phase calls.  False = the phase calls (A/B, tests).
CALLABLE_CG_FUSED_STEP = _os.environ.get("DSEA_CALLABLE_CG_FUSED", "1") != "0"


def cg(b, x0, *, native=None, callable_A=None, shift=None, eps=1e-7, maxiter=None, poll_every=8,
       merged_reductions=None):
    """Conjugate gradients on the GPU (reference CG.py:24-41) for (A - shift I) x = b.

    ``shift`` is a 0-dim/1-element device tensor (the eigenvalue E0 of CG.py:120) or None.
    Returns x (new tensor).  The loop state lives on the device; the host polls a flag.
    ``merged_reductions``: see CG_MERGED_REDUCTIONS (applies where the persistent form does; ignored elsewhere).
    """
    lib = _lib.load()
    device = b.device
    n = b.numel()
    b = as_vector(b, n)
    x = as_vector(x0, n).clone()
    cap = n if maxiter is None else int(maxiter)
    ws = Workspace.get(n, 8, device)
    st = _stream(device)
    state = ws.state
    shift_t = None
    if shift is not None:
        shift_t = shift.detach().reshape(-1)[:1].to(device=device, dtype=F64).contiguous()
    if native is not None:
        iters, res = c_int64(0), c_double(0.0)
        merged = CG_MERGED_REDUCTIONS if merged_reductions is None else bool(merged_reductions)
        prev_mode = getattr(ws, "persist_mode", -1)
        with ws.owned_by("CG (native operator)"):
            if merged and prev_mode == -1:
                ws.set_persist(100)
            elif CG_TFIM_REFERENCE_RECURRENCES and prev_mode == -1:
                ws.set_persist(200)
            try:
                rc = lib.dsea_cg_run(native.handle, ws.handle, _ptr(shift_t), _ptr(b), _ptr(x), _ptr(state), float(eps),
                                     cap, int(poll_every), byref(iters), byref(res), st)
                if rc == _lib.ERR_TIMEOUT:
                    # The persistent single-launch form needs all of its workgroups resident at the same time; if other
                    # work holds compute units (e.g. a second persistent solve on another stream) its bounded spins
                    # give up and report instead of hanging.  The solve is repeated in the streaming form, and the
                    # fallback is STICKY for this workspace: on a shared device every later solve would otherwise wait
                    # out the same timeout again (``ws.set_persist(-1)`` re-enables the persistent form).
                    import warnings
                    warnings.warn("persistent CG launch timed out waiting for a peer workgroup (device shared with "
                                  "other work?): repeating the solve with the streaming kernels, which this workspace "
                                  "keeps using from now on", RuntimeWarning)
                    x.copy_(as_vector(x0, n))
                    ws.set_persist(0)
                    prev_mode = 0
                    rc = lib.dsea_cg_run(native.handle, ws.handle, _ptr(shift_t), _ptr(b), _ptr(x), _ptr(state),
                                         float(eps), cap, int(poll_every), byref(iters), byref(res), st)
            finally:
                if (merged or CG_TFIM_REFERENCE_RECURRENCES) and prev_mode == -1:
                    ws.set_persist(-1)
        check(rc, "dsea_cg_run", allow=(_lib.ERR_NOT_CONVERGED,))
        last_cg.iters, last_cg.resnorm, last_cg.converged = iters.value, res.value, rc == 0
        form = ctypes.c_int(0)
        check(lib.dsea_cg_last_form(ws.handle, byref(form)), "dsea_cg_last_form")
        last_cg.form = CG_FORMS[form.value]
        return x

    r = torch.empty(n, dtype=F64, device=device)
    d = torch.empty(n, dtype=F64, device=device)
    done_ptr = c_void_p(state.data_ptr() + _lib.CG_DONE * 8)
    dad_ptr = c_void_p(state.data_ptr() + _lib.CG_DAD * 8)
    with ws.owned_by("CG (callable operator)"):
        Ax = as_vector(callable_A(x), n)
        if Ax.data_ptr() == x.data_ptr():
            Ax = Ax.clone()
        if shift_t is not None:
            check(lib.dsea_shift_dot(ws.handle, _ptr(x), _ptr(Ax), _ptr(shift_t), _ptr(ws.scal[1:2]), None, n, st),
                  "dsea_shift_dot")
        check(lib.dsea_cg_init(ws.handle, _ptr(b), _ptr(Ax), _ptr(r), _ptr(d), _ptr(state), n, st), "dsea_cg_init")
        check(lib.dsea_cg_init_check(ws.handle, _ptr(state), float(eps), st), "dsea_cg_init_check")
        issued = 0
        host = state.cpu()
        polls = 1
        while host[_lib.CG_DONE].item() == 0.0 and issued < cap:
            chunk = min(int(poll_every), cap - issued)
            for it in range(chunk):
                Ad = as_vector(callable_A(d), n)
                if Ad.data_ptr() == d.data_ptr() or not Ad.is_contiguous():
                    Ad = Ad.clone()
                if CALLABLE_CG_FUSED_STEP:
                    # one call = the three fused launches of dsea_cg_run's streaming iteration (bit-identical to the native
                    # operand's streaming form) instead of four phase calls and six launches
                    check(lib.dsea_cg_step(ws.handle, _ptr(x), _ptr(r), _ptr(d), _ptr(Ad), _ptr(shift_t), _ptr(state),
                                           float(eps), issued + it, n, st), "dsea_cg_step")
                    continue
                check(lib.dsea_shift_dot(ws.handle, _ptr(d), _ptr(Ad), _ptr(shift_t), dad_ptr, done_ptr, n, st),
                      "dsea_shift_dot")
                check(lib.dsea_cg_update(ws.handle, _ptr(x), _ptr(r), _ptr(d), _ptr(Ad), _ptr(state), n, st),
                      "dsea_cg_update")
                check(lib.dsea_cg_check(ws.handle, _ptr(state), float(eps), st), "dsea_cg_check")
                check(lib.dsea_cg_direction(ws.handle, _ptr(r), _ptr(d), _ptr(state), n, st), "dsea_cg_direction")
            issued += chunk
            host = state.cpu()
            polls += 1
    last_cg.polls, last_cg.form = polls, "streaming (callable operand: phase calls from Python)"
    last_cg.iters = int(host[_lib.CG_ITERS].item())
    last_cg.resnorm = float(host[_lib.CG_RESNORM].item())
    last_cg.converged = host[_lib.CG_DONE].item() != 0.0
    return x


# --------------------------------------------------------------------------- small helpers
def spmv(native, x, shift=None, out=None):
    """y = A x (- shift x) through the native operator handle."""
    lib = _lib.load()
    n = native.n
    x = as_vector(x, n)
    y = out if out is not None else torch.empty(n, dtype=F64, device=x.device)
    check(lib.dsea_spmv(native.handle, None, _ptr(x), _ptr(y), _ptr(shift), None, None, _stream(x.device)),
          "dsea_spmv")
    return y


def project_out(v, a):
    """v - (a.v) a on the device (reference CG.py:59,122; symeig.py:27,80)."""
    lib = _lib.load()
    n = v.numel()
    v, a = as_vector(v, n), as_vector(a, n)
    ws = Workspace.get(n, 8, v.device)
    out = torch.empty(n, dtype=F64, device=v.device)
    check(lib.dsea_project_out(ws.handle, _ptr(v), _ptr(a), _ptr(out), None, n, _stream(v.device)),
          "dsea_project_out")
    return out
