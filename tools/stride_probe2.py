#!/usr/bin/env python3
"""Stride between basis vectors vs the speed of the two passes, on ONE allocation (placement fixed): the same buffer
is read as a basis with leading dimension n + pad.   python tools/stride_probe2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_void_p
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
dev = torch.device("cuda:0"); lib = _lib.load()
n, i = 1 << 20, 200
PADS = [0, 2, 16, 32, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
buf = torch.randn((i + 1) * (n + max(PADS)), dtype=torch.float64, device=dev)
u = torch.randn(n, dtype=torch.float64, device=dev); r = torch.empty(n, dtype=torch.float64, device=dev)
c = torch.zeros(i + 2, dtype=torch.float64, device=dev); ab = torch.tensor([0.5, 0.25], dtype=torch.float64, device=dev)
nrm2 = torch.zeros(1, dtype=torch.float64, device=dev)
ws = Workspace.get(n, i + 1, dev); st = _stream(dev)


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rnd in range(2):
    for pad in PADS:
        ldq = n + pad
        t1 = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(buf), ldq, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st))
        t2 = timeit(lambda: lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(buf), ldq, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st))
        print("round %d  ldq = n + %-5d dots %.1f us   correction %.1f us" % (rnd, pad, t1, t2))
