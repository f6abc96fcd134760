#!/usr/bin/env python3
"""Where the time of one Arnoldi cycle goes (BASELINE config 4 shape: transfer matrix D = 512, m = 200 vectors):
the library loop (dsea_arnoldi_extend) alone, the mat-vec alone, the host's small eigen-solve, DGKS second passes."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dominantsparseeigenad_amd import _lib, engine, krylov
from dominantsparseeigenad_amd.operators import TransferOperator
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0"); n = D * D; lib = _lib.load()
torch.manual_seed(0)
A = torch.randn(2, D, D, dtype=torch.float64, device=dev) / D ** 0.5
op = TransferOperator(A)
ws = engine.Workspace.get(n, m + 2, dev); st = engine._stream(dev); ldv = engine.round_up(n, 32)
V = torch.zeros((m + 1, ldv), dtype=torch.float64, device=dev); Hd = torch.zeros((m, m + 1), dtype=torch.float64, device=dev)
v = torch.randn(n, dtype=torch.float64, device=dev); V[0, :n] = v / v.norm()
P = lambda t: ctypes.c_void_p(t.data_ptr())
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(lib.dsea_arnoldi_extend(op.handle, ws.handle, None, P(V), ldv, 0, m, P(Hd), m + 1, st), "extend")
    torch.cuda.synchronize(); t1 = time.perf_counter()
cnt = ctypes.c_int64(0); lib.dsea_arnoldi_second_passes(ws.handle, ctypes.byref(cnt), st)
print("dsea_arnoldi_extend 0..%d at n=%d: %.2f ms (%.1f us/step), second Gram-Schmidt passes: %d" % (m, n, (t1 - t0) * 1e3, (t1 - t0) / m * 1e6, cnt.value))
x = V[0, :n].clone(); y = torch.empty_like(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): lib.dsea_spmv(op.handle, None, P(x), P(y), None, None, None, st)
torch.cuda.synchronize(); print("transfer mat-vec: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
B = Hd.cpu().numpy()[:m, :m].T.copy()
t0 = time.perf_counter(); th, yv, ev = krylov._wanted_pair(B, "LM"); t1 = time.perf_counter()
print("host: eigvals + inverse iteration of the %d x %d Hessenberg matrix: %.2f ms  theta=%.12f" % (m, m, (t1 - t0) * 1e3, th))
t0 = time.perf_counter(); Hh = Hd.cpu(); t1 = time.perf_counter(); print("D2H of H: %.3f ms" % ((t1 - t0) * 1e3))
G = torch.matmul(V[:m, :n], V[:m, :n].T); print("orthonormality ||V V^T - I||_max = %.2e" % float((G - torch.eye(m, dtype=torch.float64, device=dev)).abs().max()))
torch.manual_seed(1)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lam, xr = krylov.arnoldi_dominant(op, n, m, dev, "LM")
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("arnoldi_dominant: %.2f ms, %d cycle(s), lambda=%.12f" % ((t1 - t0) * 1e3, krylov.last('arnoldi_cycles'), lam))
krylov.STAGE_LOG = []
torch.cuda.synchronize(); t0 = time.perf_counter()
lam, xr = krylov.arnoldi_dominant(op, n, m, dev, "LM")
torch.cuda.synchronize(); t1 = time.perf_counter()
print("one more run, %.2f ms, stages [columns, ms to H on host, host ms, relative residual]:" % ((t1 - t0) * 1e3))
for row in krylov.STAGE_LOG: print("   ", ["%.3g" % v if v is not None else None for v in row])
