#!/usr/bin/env python3
"""Row f-4: dense DominantSymeig at scale (reference symeig.py:15-31) -- forward (Lanczos, k vectors) + backward (CG
with the shift inside the kernels, rank-1 grad_A) on a dense symmetric n x n CUDA tensor: native loops on the
hand-written upper-triangle mat-vec vs the generic path with torch.matmul (rocBLAS GEMV) as the mat-vec."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.symeig import DominantSymeig
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.float64
torch.manual_seed(0)
A0 = torch.randn(n, n, dtype=dtype, device=dev); A0 = (A0 + A0.T) / 2
t = torch.randn(n, dtype=dtype, device=dev); t = t / t.norm()
for flag, name in ((True, "native upper-triangle operand"), (False, "torch.matmul callable")):
    engine.DENSE_SYMMETRIC_KERNEL = flag
    for rep in range(3):
        A = A0.clone().requires_grad_(True)
        torch.manual_seed(1); torch.cuda.synchronize(); t0 = time.perf_counter()
        lam, psi = DominantSymeig.apply(A, k, dev)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        (gA,) = torch.autograd.grad(lam + psi.matmul(t), A)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("n=%d k=%d %s  %-30s forward %.1f ms  backward %.1f ms (%d CG iterations)  lambda=%.10f" % (
        n, k, str(dtype).split(".")[1], name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, engine.last_cg.iters, lam.item()))
engine.DENSE_SYMMETRIC_KERNEL = True
