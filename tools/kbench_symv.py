#!/usr/bin/env python3
"""Dense symmetric mat-vec: hand-written upper-triangle kernel (k_symv_upper + k_symv_reduce) vs rocBLAS GEMV
(torch.matmul) on the full matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd.operators import SymmetricDenseOperator
dev = torch.device("cuda:0")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in [int(a) for a in sys.argv[1:]] or (256, 1024, 4096, 8192, 16384):
    A = torch.randn(n, n, dtype=torch.float64, device=dev); A = A + A.T
    x = torch.randn(n, dtype=torch.float64, device=dev)
    op = SymmetricDenseOperator(A)
    t1 = timeit(lambda: op(x)); t2 = timeit(lambda: torch.matmul(A, x))
    err = float((op(x) - A @ x).abs().max() / (A @ x).abs().max())
    print("n=%6d  upper-triangle kernel %8.1f us (%.0f GB/s of the half matrix)   rocBLAS gemv %8.1f us (%.0f GB/s of the full matrix)   rel diff %.1e"
          % (n, t1, 4.0 * n * n / t1 / 1e3, t2, 8.0 * n * n / t2 / 1e3, err))
