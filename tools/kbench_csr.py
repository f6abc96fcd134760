#!/usr/bin/env python3
"""TFIM as an explicit device matrix (21 nnz/row, fp64 values + int32 columns): mat-vec time of the CSR kernels and of the
SELL-64 kernel at every unroll setting (1 = the round-5 kernel) vs the matrix-free kernel, plus the two kernels that make
the matrix a parameter (dsea_op_sddmm, dsea_op_update_vals); first of all the three 16-bit-column layouts side by side --
unpacked (dsea_op_create_sell16), packed two slice columns to a lane (dsea_op_create_sell16p2, the default), value-coded
(dsea_op_create_sell16v8) -- back to back and from a cold Infinity Cache (``--coded``: only that).
Evidence file: profiles/r06_kbench_csr.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_void_p
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
from dominantsparseeigenad_amd.operators import TFIMOperator
dev = torch.device("cuda:0"); lib = _lib.load()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << L
g = torch.tensor([1.0], dtype=torch.float64, device=dev)
op = TFIMOperator(L, dev, g=g)
csr = op.to_csr(layout="csr")
CODED_ONLY = "--coded" in sys.argv
x = torch.randn(n, dtype=torch.float64, device=dev); y1 = torch.empty_like(x); y2 = torch.empty_like(x); y3 = torch.empty_like(x)
out = torch.zeros(1, dtype=torch.float64, device=dev); ws = Workspace.get(n, 8, dev); st = _stream(dev)
def timeit(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
big = torch.randn(1 << 27, dtype=torch.float64, device=dev)      # 1 GiB: streamed between two timed launches = cold Infinity Cache
wsb = Workspace.get(1 << 27, 2, dev)
def timeit_cold(fn, reps=40):
    """the launch alone, after a 1 GiB stream has pushed the operand out of the 256 MiB Infinity Cache (the state a mat-vec
    finds inside a Lanczos step, whose basis passes stream 1.1 GB between two mat-vecs)"""
    tot = 0.0
    for i in range(reps + 3):
        lib.dsea_probe_stream(wsb.handle, _ptr(big), None, big.numel(), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3: tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
nnz = csr.vals.numel()
bytes_sell = nnz * 12 + (n // 64 + 1) * 8 + 16 * n
t1 = timeit(lambda: lib.dsea_spmv(op.handle, ws.handle, _ptr(x), _ptr(y1), None, _ptr(out), None, st))
print("L = %d, n = %d, nnz = %d; matrix-free %.1f us" % (L, n, nnz, t1))
if "--csr" in sys.argv:
    for G in (4, 8, 16, 32, 0):
        lib.dsea_op_set_tuning(csr.handle, 2, G)
        tt = timeit(lambda: lib.dsea_spmv(csr.handle, ws.handle, _ptr(x), _ptr(y2), None, _ptr(out), None, st))
        print("CSR G=%d: %.1f us" % (G, tt))
os.environ["DSEA_SELL_PACK2"] = "0"
sell = op.to_csr(layout="sell", col16=False, values="plain")
sell16 = op.to_csr(layout="sell", col16=True, values="plain")
os.environ["DSEA_SELL_PACK2"] = "1"
sell16p = op.to_csr(layout="sell", col16=True, values="plain")      # per-element arrays packed two slice columns to a lane
assert sell16p._pack2 and not sell16._pack2
lib.dsea_op_set_tuning(sell.handle, 3, 1)
lib.dsea_spmv(sell.handle, ws.handle, _ptr(x), _ptr(y3), None, _ptr(out), None, st)
bytes16 = nnz * 10 + nnz // 64 * 4 + (n // 64 + 1) * 8 + 16 * n
# value-coded operand (dsea_op_create_sell16v8): 16-bit column deltas + 8-bit value codes
sellv = op.to_csr(layout="sell", col16=True, values="coded")
bytesv = nnz * 3 + nnz // 64 * 4 + (n // 64 + 1) * 8 + 16 * n + 2048
lib.dsea_spmv(sell16.handle, ws.handle, _ptr(x), _ptr(y3), None, _ptr(out), None, st)
for rnd in range(2):
    for label, o, moved in (("16-bit columns, fp64 values ", sell16, bytes16), ("the same, packed by two     ", sell16p, bytes16 * 22 // 21),
                            ("16-bit columns, 8-bit codes ", sellv, bytesv)):
        _lib.check(lib.dsea_op_set_tuning(o.handle, 4, 1), "tune")
        tt = timeit(lambda: lib.dsea_spmv(o.handle, ws.handle, _ptr(x), _ptr(y2), None, None, None, st))
        tc = timeit_cold(lambda: lib.dsea_spmv(o.handle, ws.handle, _ptr(x), _ptr(y2), None, None, None, st))
        print("SELL-64 %s (%d distinct values): back to back %6.1f us | cold %6.1f us  (moves %d MB: %.0f GB/s cold)  bit-identical to the fp64-value operand: %s"
              % (label, int((sellv._vtab != 0).sum()) if o is sellv else -1, tt, tc, moved / 1e6, moved / tc / 1e3, bool(torch.equal(y2, y3))))
# the parameter kernels on the DEFAULT layout (packed by two): sampled outer product and in-place refresh
_v1 = torch.randn(n, dtype=torch.float64, device=dev); _gb = torch.empty(nnz, dtype=torch.float64, device=dev)
_rp = c_void_p(sell16p.rowptr.data_ptr())
for flags, name in ((0, "plain"), (2, "symmetric")):
    tt = timeit(lambda: lib.dsea_op_sddmm(sell16p.handle, _rp, _ptr(_v1), _ptr(x), 1.0, flags, _ptr(_gb), st), reps=50)
    print("dsea_op_sddmm (%s, packed SELL): %.1f us" % (name, tt))
tt = timeit(lambda: lib.dsea_op_update_vals(sell16p.handle, _rp, _ptr(sell16p.vals), st), reps=50)
print("dsea_op_update_vals (packed SELL): %.1f us  (%.0f GB/s of 16 B per non-zero)" % (tt, nnz * 16 / tt / 1e3))
if CODED_ONLY:
    sys.exit(0)
lib.dsea_spmv(sell.handle, ws.handle, _ptr(x), _ptr(y3), None, _ptr(out), None, st)
for rnd in range(int(os.environ.get('SELL_ROUNDS', '2'))):
    for label, o, U, xcd in (("round-5 kernel        ", sell, 1, 0), ("2 columns in flight   ", sell, 2, 0), ("4 columns in flight   ", sell, 4, 0),
                             ("8 columns in flight   ", sell, 8, 0), ("8 columns, XCD map    ", sell, 8, 1),
                             ("16-bit columns        ", sell16, 0, 0), ("16-bit columns, XCD map", sell16, 0, 1)):
        _lib.check(lib.dsea_op_set_tuning(o.handle, 3, U), "tune")
        _lib.check(lib.dsea_op_set_tuning(o.handle, 4, xcd), "tune")
        tt = timeit(lambda: lib.dsea_spmv(o.handle, ws.handle, _ptr(x), _ptr(y2), None, None, None, st))
        tc = timeit_cold(lambda: lib.dsea_spmv(o.handle, ws.handle, _ptr(x), _ptr(y2), None, None, None, st))
        moved = bytes16 if o is sell16 else bytes_sell
        print("SELL-64 %s: back to back %6.1f us | cold Infinity Cache %6.1f us = %5.0f GB/s of the %d MB algorithmic = %.3f of 8 TB/s (moved %d MB: %.0f GB/s)  bit-identical to round 5: %s  maxdiff vs matrix-free %.1e"
              % (label, tt, tc, bytes_sell / tc / 1e3, bytes_sell / 1e6, bytes_sell / tc / 1e3 / 8000,
                 moved / 1e6, moved / tc / 1e3, bool(torch.equal(y2, y3)), float((y1 - y2).abs().max())))
        lib.dsea_op_set_tuning(o.handle, 4, 0)
lib.dsea_op_set_tuning(sell.handle, 3, 0)
# the operand as a parameter
v1 = torch.randn(n, dtype=torch.float64, device=dev); v2 = torch.randn(n, dtype=torch.float64, device=dev)
gbar = torch.empty(nnz, dtype=torch.float64, device=dev)
rp = c_void_p(sell.rowptr.data_ptr())
for flags, name in ((0, "plain"), (2, "symmetric")):
    tt = timeit(lambda: lib.dsea_op_sddmm(sell.handle, rp, _ptr(v1), _ptr(v2), 1.0, flags, _ptr(gbar), st), reps=50)
    print("dsea_op_sddmm (%s, SELL): %.1f us  (%.0f GB/s of cols 4 B + out 8 B per non-zero)" % (name, tt, nnz * 12 / tt / 1e3))
    tt = timeit(lambda: lib.dsea_op_sddmm(csr.handle, None, _ptr(v1), _ptr(v2), 1.0, flags, _ptr(gbar), st), reps=50)
    print("dsea_op_sddmm (%s, CSR):  %.1f us" % (name, tt))
ref = v1[torch.repeat_interleave(torch.arange(n, device=dev), 21)] * v2[csr.colidx.long()]
lib.dsea_op_sddmm(sell.handle, rp, _ptr(v1), _ptr(v2), 1.0, 0, _ptr(gbar), st)
print("sddmm vs torch gathers: equal =", bool(torch.equal(gbar, ref)))
tt = timeit(lambda: lib.dsea_op_update_vals(sell.handle, rp, _ptr(sell.vals), st), reps=50)
print("dsea_op_update_vals (SELL): %.1f us  (%.0f GB/s of 16 B per non-zero)" % (tt, nnz * 16 / tt / 1e3))
