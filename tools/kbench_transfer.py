#!/usr/bin/env python3
"""Formulations of the transfer-matrix mat-vec r -> sum_s A_s r A_s^T (D = 512, d = 2) timed against each other."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd.operators import TransferOperator
dev = torch.device("cuda:0")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = 2
torch.manual_seed(0)
A = torch.randn(d, D, D, dtype=torch.float64, device=dev) / D ** 0.5
AT = A.transpose(1, 2).contiguous()
r = torch.randn(D, D, dtype=torch.float64, device=dev)
def timeit(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
op = TransferOperator(A); opT = TransferOperator(A, transpose=True)
v = r.reshape(-1).contiguous()
Aflat = A.reshape(d * D, D)
A2 = A.permute(1, 0, 2).reshape(D, d * D).contiguous()
T = torch.empty(d, D, D, dtype=torch.float64, device=dev)
print("libdsea TransferOperator            %.1f us   transpose %.1f us" % (timeit(lambda: op(v)), timeit(lambda: opT(v))))
print("torch bmm(A,r) then bmm(.,A^T).sum  %.1f us" % timeit(lambda: torch.matmul(torch.matmul(A, r), AT).sum(0)))
print("torch one 512^3 matmul              %.1f us" % timeit(lambda: torch.matmul(A[0], r)))
print("torch bmm 2 x 512^3                 %.1f us" % timeit(lambda: torch.matmul(A, r)))
print("torch (dD x D) @ (D x D)            %.1f us" % timeit(lambda: torch.matmul(Aflat, r)))
Tc = torch.matmul(A, r)
print("torch bmm(T, A^T) 2 x 512^3         %.1f us" % timeit(lambda: torch.matmul(Tc, AT)))
Tcat = Tc.permute(1, 0, 2).reshape(D, d * D).contiguous()
print("torch (D x dD) @ (dD x D)           %.1f us" % timeit(lambda: torch.matmul(Tcat, A2.T)))
print("torch baddbmm-style: T0 A0^T + T1 A1^T via addmm %.1f us" % timeit(lambda: torch.addmm(torch.matmul(Tc[0], AT[0]), Tc[1], AT[1])))
