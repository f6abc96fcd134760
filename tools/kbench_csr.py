#!/usr/bin/env python3
"""TFIM as a device CSR matrix (21 nnz/row, int32 cols): mat-vec time vs the matrix-free kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
from dominantsparseeigenad_amd.operators import TFIMOperator, CSROperator
dev = torch.device("cuda:0"); lib = _lib.load()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << L
g = torch.tensor([1.0], dtype=torch.float64, device=dev)
op = TFIMOperator(L, dev, g=g)
csr = op.to_csr(layout="csr")
sell = op.to_csr(layout="sell")
x = torch.randn(n, dtype=torch.float64, device=dev); y1 = torch.empty_like(x); y2 = torch.empty_like(x)
out = torch.zeros(1, dtype=torch.float64, device=dev); ws = Workspace.get(n, 8, dev); st = _stream(dev)
def timeit(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t1 = timeit(lambda: lib.dsea_spmv(op.handle, ws.handle, _ptr(x), _ptr(y1), None, _ptr(out), None, st))
for G in (4, 8, 16, 32):
    lib.dsea_op_set_tuning(csr.handle, 2, G)
    tt = timeit(lambda: lib.dsea_spmv(csr.handle, ws.handle, _ptr(x), _ptr(y2), None, _ptr(out), None, st))
    print("G=%d: %.1f us" % (G, tt))
lib.dsea_op_set_tuning(csr.handle, 2, 0)
t2 = timeit(lambda: lib.dsea_spmv(csr.handle, ws.handle, _ptr(x), _ptr(y2), None, _ptr(out), None, st))
t3 = timeit(lambda: lib.dsea_spmv(sell.handle, ws.handle, _ptr(x), _ptr(y2), None, _ptr(out), None, st))
print("SELL-64: %.1f us  maxdiff %.1e" % (t3, float((y1 - y2).abs().max())))
nnz = csr.vals.numel()
bytes_csr = nnz * 12 + (n + 1) * 8 + 16 * n
print("matrix-free %.1f us | CSR %.1f us  (%.0f GB/s of %d MB algorithmic)  maxdiff %.1e" % (t1, t2, bytes_csr / t2 / 1e3, bytes_csr / 1e6, float((y1 - y2).abs().max())))
