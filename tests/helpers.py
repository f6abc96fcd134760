"""Shared test helpers: pinned draws (the same injection the golden script used on the reference)."""
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


class _ResultChannel:
    """what a spawned worker writes its result into (``ret[rank] = value``): a pipe of the SPAWN context.  No manager
    process: ``multiprocessing.Manager()`` FORKS the calling process -- here a pytest process that holds an initialised HIP
    runtime -- and the forked copies linger until garbage collection."""

    def __init__(self, queue):
        self.queue = queue

    def __setitem__(self, rank, value):
        self.queue.put((rank, value))


def spawn_collect(fn, args, nprocs):
    """``torch.multiprocessing.spawn(fn, args + (ret,), nprocs)`` where every worker does ``ret[rank] = value``; returns
    {rank: value}.  The parent reads while the workers run (a result larger than the pipe buffer would otherwise block its
    writer) and a crashed worker raises here instead of leaving the parent waiting."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.SimpleQueue()
    pc = mp.spawn(fn, args=tuple(args) + (_ResultChannel(queue),), nprocs=nprocs, join=False)
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
