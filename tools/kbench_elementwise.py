#!/usr/bin/env python3
"""Elementwise phase kernels (grid-stride, several tiles per thread on large slabs) at 2^20 and 2^25 rows: achieved
bandwidth of dsea_axpy (2 reads + 1 write), dsea_scale_store (1 read + 1 write + bf16), dsea_dot (2 reads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
dev = torch.device("cuda:0"); lib = _lib.load(); st = _stream(dev)


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for lg in (20, 23, 25):
    n = 1 << lg
    ws = Workspace.get(n, 8, dev)
    x = torch.randn(n, dtype=torch.float64, device=dev); y = torch.randn(n, dtype=torch.float64, device=dev)
    q = torch.empty(n, dtype=torch.float64, device=dev)
    a = torch.tensor([0.5], dtype=torch.float64, device=dev); nrm2 = torch.tensor([2.0], dtype=torch.float64, device=dev)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    t = timeit(lambda: lib.dsea_axpy(ws.handle, 1.0, _ptr(a), _ptr(x), _ptr(y), n, st))
    print("n=2^%d  dsea_axpy        %.1f us  %.0f GB/s" % (lg, t, 24.0 * n / t / 1e3))
    t = timeit(lambda: lib.dsea_scale_store(ws.handle, _ptr(x), _ptr(nrm2), _ptr(q), None, n, st))
    print("n=2^%d  dsea_scale_store %.1f us  %.0f GB/s" % (lg, t, 16.0 * n / t / 1e3))
    t = timeit(lambda: lib.dsea_dot(ws.handle, _ptr(x), _ptr(y), n, _ptr(out), st))
    print("n=2^%d  dsea_dot         %.1f us  %.0f GB/s (incl. the second-stage launch)" % (lg, t, 16.0 * n / t / 1e3))
    t = timeit(lambda: y.add_(x, alpha=0.5))
    print("n=2^%d  torch y += a x   %.1f us  %.0f GB/s" % (lg, t, 24.0 * n / t / 1e3))
