"""Shared test helpers: pinned draws (the same injection the golden script used on the reference)."""
import os

import numpy as np
import torch

from dominantsparseeigenad_amd.synthetic import normal_vector


class SeedDraws:
    """draw #c -> normal_vector(n, base + c); callable as oracle ``draw(n, dtype)``.
    ``offset``: first global row of a slab (row-partitioned runs draw their slab of the same global vector)."""

    def __init__(self, base, device="cpu", offset=0):
        self.base = int(base)
        self.count = 0
        self.device = device
        self.offset = int(offset)

    def __call__(self, n, dtype=torch.float64):
        v = torch.from_numpy(normal_vector(int(n), self.base + self.count, offset=self.offset)).to(dtype).to(self.device)
        self.count += 1
        return v


class PatchRandn:
    """Context manager replacing ``torch.randn`` by SeedDraws -- pins the product modules exactly
    the way tests/golden/make_golden.py pinned the reference (same call order: q0, dummy, x0...)."""

    def __init__(self, base, offset=0):
        self.draws = SeedDraws(base, offset=offset)
        self._orig = None

    def _randn(self, *size, dtype=None, device=None, **kw):
        assert len(size) == 1 and isinstance(size[0], int), size
        v = self.draws(size[0], dtype or torch.float64)
        return v.to(device) if device is not None else v

    def __enter__(self):
        self._orig = torch.randn
        torch.randn = self._randn
        return self.draws

    def __exit__(self, *exc):
        torch.randn = self._orig


def sym_from_seed(n, seed, scale=1.0):
    M = torch.from_numpy(normal_vector(n * n, seed).reshape(n, n)) * scale
    return M + M.T


def unit(n, seed):
    t = torch.from_numpy(normal_vector(n, seed))
    return t / t.norm()


def signed_close(a, b, atol, rtol=0.0):
    """max |a -/+ b| <= atol + rtol*max|b| for the better of the two signs; returns (ok, err, sign)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ep, em = np.max(np.abs(a - b)), np.max(np.abs(a + b))
    err, sgn = (ep, 1.0) if ep <= em else (em, -1.0)
    return err <= atol + rtol * np.max(np.abs(b)), err, sgn


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


# ---- explicit sparse matrices as parameters (tests/test_gpu_csr_param.py, tests/test_partitioned_gloo.py)
def banded_spd(n, hb, seed):
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    diags, offs = [], []
    for d in range(1, hb + 1):
        if rng.rand() < 0.7 or d == 1:
            band = rng.randn(n - d) * 0.3
            diags += [band, band]
            offs += [d, -d]
    M = sp.diags(diags, offs, shape=(n, n), format="csr")
    rowsum = np.asarray(abs(M).sum(axis=1)).ravel()
    M = (M + sp.diags(rowsum + 1.0 + np.linspace(0.0, 3.0, n), 0)).tocsr()      # diagonally dominant: SPD, gapped bottom
    M.sort_indices()
    return M


def eigh_reference(M_rowptr, M_cols, vals, n, t, w_E=1.0, w_psi=1.0, psi_like=None, autograd=True):
    """loss = w_E E0 + w_psi psi.t from torch.linalg.eigh of the SYMMETRISED dense matrix built from vals (CPU, fp64);
    returns (E0, psi, d loss / d vals) -- the tied-pair adjoint (chain rule through (A + A^T)/2).
    autograd=True : torch's own eigh backward (needs a spectrum without degeneracies ANYWHERE: it forms 1/(lam_i - lam_j)
                    for all pairs, and the TFIM spectrum is degenerate above the ground state -> NaN);
    autograd=False: first-order perturbation theory written out from the eigh factors,
                    A-bar = w_E psi psi^T + w_psi v psi^T,  v = sum_{j>0} u_j (u_j . t) / (lam_0 - lam_j), then symmetrised."""
    rows = torch.repeat_interleave(torch.arange(n), M_rowptr.cpu()[1:] - M_rowptr.cpu()[:-1])
    cols = M_cols.cpu().long()
    vals = vals.detach().cpu().clone().requires_grad_(autograd)
    A = torch.zeros((n, n), dtype=torch.float64).index_put((rows, cols), vals, accumulate=True)
    lam, U = torch.linalg.eigh(0.5 * (A + A.T))
    psi = U[:, 0]
    sgn = 1.0 if psi_like is None or float(psi.detach() @ psi_like.cpu()) > 0 else -1.0
    if autograd:
        loss = w_E * lam[0] + (w_psi * sgn * (psi @ t.cpu()) if w_psi else 0.0)
        (g,) = torch.autograd.grad(loss, vals)
        return lam[0].detach(), sgn * psi.detach(), g
    p0 = sgn * psi
    v = U[:, 1:] @ ((U[:, 1:].T @ t.cpu()) / (lam[0] - lam[1:]))
    g = w_E * p0[rows] * p0[cols] + w_psi * 0.5 * (v[rows] * p0[cols] + v[cols] * p0[rows])
    return lam[0], p0, g


class _ResultChannel:
    """what a spawned worker writes its result into (``ret[rank] = value``): a pipe of the SPAWN context.  No manager
    process: ``multiprocessing.Manager()`` FORKS the calling process -- here a pytest process that holds an initialised HIP
    runtime -- and the forked copies linger until garbage collection."""

    def __init__(self, queue):
        self.queue = queue

    def __setitem__(self, rank, value):
        self.queue.put((rank, value))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_collect(fn, args, nprocs, port_index=None):
    """``_spawn_collect_once`` with ONE thing retried: the rendezvous port.  A port found free can be taken by another process
    of the host before rank 0 listens on it (seen once in eight full runs of round 6: ``EADDRINUSE`` -- the GPU boxes share a
    network namespace with other jobs); with ``port_index`` given, such a start is repeated on a fresh port (twice at most)."""
    args = tuple(args)
    for attempt in range(3):
        try:
            return _spawn_collect_once(fn, args, nprocs)
        except Exception as exc:  # noqa: BLE001
            busy = "EADDRINUSE" in str(exc) or "address already in use" in str(exc).lower()
            if port_index is None or not busy or attempt == 2:
                raise
            args = args[:port_index] + (_free_port(),) + args[port_index + 1:]


def _spawn_collect_once(fn, args, nprocs):
    """``torch.multiprocessing.spawn(fn, args + (ret,), nprocs)`` where every worker does ``ret[rank] = value``; returns
    {rank: value}.  The parent reads while the workers run (a result larger than the pipe buffer would otherwise block its
    writer) and a crashed worker raises here instead of leaving the parent waiting."""
    import multiprocessing
    import torch.multiprocessing as mp
    if os.environ.get("DSEA_TEST_RESULT_CHANNEL", "") == "manager":
        # EXPERIMENT ONLY (docs/design/12-round6.md: is this what stalled two round-5 runs?): the round-4 way -- a
        # multiprocessing.Manager() per multi-rank case, which FORKS the pytest process (it holds an initialised HIP runtime)
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(fn, args=tuple(args) + (ret,), nprocs=nprocs, join=True)
        return dict(ret)
    # Workers come from a FORK SERVER that has imported torch / numpy / scipy and nothing else: a fresh process per worker as
    # with "spawn" (no state shared between cases, HIP is initialised by the worker itself -- the server never touches the
    # GPU), minus the ~1 s of imports per worker (measured on the GPU box, tools/probes/spawn_cost.py: 4 ranks 1.7-2.4 s ->
    # 0.3-0.5 s; ~100 spawns in `pytest -m gpu`).  The package itself is NOT preloaded: it reads environment switches at import.
    method = os.environ.get("DSEA_TEST_START_METHOD", "forkserver")
    if method == "forkserver":
        multiprocessing.set_forkserver_preload(["torch", "torch.distributed", "numpy", "scipy.sparse", "scipy.linalg"])
    ctx = mp.get_context(method)
    queue = ctx.SimpleQueue()
    pc = mp.start_processes(fn, args=tuple(args) + (_ResultChannel(queue),), nprocs=nprocs, join=False, start_method=method)
    out, done = {}, False
    while len(out) < nprocs and not done:
        while not queue.empty():
            rank, value = queue.get()
            out[rank] = value
        if len(out) < nprocs:
            done = pc.join(timeout=0.05)          # raises if a worker failed
    while not queue.empty():
        rank, value = queue.get()
        out[rank] = value
    while not done:
        done = pc.join(timeout=1.0)
    return out
