#!/usr/bin/env python3
"""general dense operand: the hand-written GEMV (default) against rocBLAS (DSEA_DENSE_GEMV=0), us per mat-vec and GB/s.
   python tools/kbench_gemv.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd.operators import DenseOperator
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [1000, 4096, 8192, 16384]:
    torch.manual_seed(0)
    G = torch.randn(n, n, dtype=torch.float64, device=dev)
    v = torch.randn(n, dtype=torch.float64, device=dev)
    op, opT = DenseOperator(G), DenseOperator(G, transpose=True)
    err = float((op(v) - G @ v).abs().max()), float((opT(v) - G.T @ v).abs().max())
    def timeit(f, reps=50):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t, tT, tt = timeit(lambda: op(v)), timeit(lambda: opT(v)), timeit(lambda: torch.mv(G, v))
    print("n=%6d  %s  A x %8.1f us (%5.0f GB/s)   A^T x %8.1f us   torch.mv %8.1f us   max err %.1e / %.1e" % (
        n, "rocBLAS " if os.environ.get("DSEA_DENSE_GEMV") == "0" else "dsea    ", t, 8.0 * n * n / t / 1e3, tT, tt, err[0], err[1]))
