#!/usr/bin/env python3
"""Fixed cost vs streaming rate of the two basis passes: ONE basis allocation, the passes timed at several i.
   python tools/kbench_slope.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ctypes import c_void_p
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
dev = torch.device("cuda:0"); lib = _lib.load()
n, imax = 1 << 20, 200
Q = torch.randn((imax + 1, n), dtype=torch.float64, device=dev)
u = torch.randn(n, dtype=torch.float64, device=dev); r = torch.empty(n, dtype=torch.float64, device=dev)
c = torch.zeros(imax + 2, dtype=torch.float64, device=dev); ab = torch.tensor([0.5, 0.25], dtype=torch.float64, device=dev)
nrm2 = torch.zeros(1, dtype=torch.float64, device=dev)
ws = Workspace.get(n, imax + 1, dev); st = _stream(dev)


def timeit(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# the shadow pass: a bf16 copy of the basis, coefficients at rounding level so that the premise holds
Qs = Q.to(torch.bfloat16)
c_small = torch.full((imax + 2,), 1e-17, dtype=torch.float64, device=dev); c_small[-2:] = 1.0


def shadow_pass(i):
    cc = c_small.clone(); cc[i] = 1.0          # c[i] = ||r||^2 (scale of the premise test)
    _lib.check(lib.dsea_ws_set_shadow(ws.handle, _ptr(Qs), n, imax + 1, 1e-12), "shadow")
    t = timeit(lambda: lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), n, n, i, _ptr(cc), _ptr(r), _ptr(nrm2), st))
    _lib.check(lib.dsea_ws_set_shadow(ws.handle, None, 0, 0, 0.0), "shadow off")
    return t


rows = []
for i in (4, 8, 20, 52, 100, 148, 200):
    t1 = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), n, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st))
    t2 = timeit(lambda: lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), n, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st))
    t3 = shadow_pass(i)
    rows.append((i, t1, t2, t3))
    print("i=%3d  dots pass + finalize %.1f us   fp64 correction pass + finalize %.1f us   bf16-shadow correction pass + finalize %.1f us" % (i, t1, t2, t3))
A = np.array([[1.0, r_[0]] for r_ in rows[2:]])
for name, col, esz in (("dots", 1, 8.0), ("correction", 2, 8.0), ("shadow", 3, 2.0)):
    y = np.array([r_[col] for r_ in rows[2:]])
    (a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
    print("%-10s fixed %.1f us + %.3f us per basis vector  (%.0f GB/s streaming rate)" % (name, a, b, esz * n / b / 1e3))
