#!/usr/bin/env python3
"""Micro-benchmark of the TFIM mat-vec (plain, with dot partials) per LDS tile size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import _lib, engine
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
from dominantsparseeigenad_amd.operators import TFIMOperator
dev = torch.device("cuda:0")
lib = _lib.load()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << L
g = torch.tensor([1.0], dtype=torch.float64, device=dev)
op = TFIMOperator(L, dev, g=g)
x = torch.randn(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
out = torch.zeros(1, dtype=torch.float64, device=dev)
ws = Workspace.get(n, 8, dev); st = _stream(dev)
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ref = None
for T in (8, 9, 10, 11, 12):
    lib.dsea_op_set_tuning(op.handle, 1, T)
    t = timeit(lambda: lib.dsea_spmv(op.handle, ws.handle, _ptr(x), _ptr(y), None, _ptr(out), None, st))
    if ref is None: ref = y.clone()
    print("T=%2d  spmv+dot+finalize %.2f us   (%.0f GB/s algorithmic)  maxdiff %.1e" % (T, t, 16.0 * n / t / 1e3, float((y - ref).abs().max())))
t = timeit(lambda: y.copy_(x))
print("torch copy 8 MiB: %.2f us" % t)
