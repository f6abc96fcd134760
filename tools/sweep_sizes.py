#!/usr/bin/env python3
"""Forward (Lanczos) time of native operators over sizes, to spot regimes where a geometry heuristic is off:
   python tools/sweep_sizes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dominantsparseeigenad_amd.Lanczos import symeigLanczos
from dominantsparseeigenad_amd.operators import Stencil3Operator, TFIMOperator
from dominantsparseeigenad_amd import engine
dev = torch.device("cuda:0"); F64 = torch.float64


def fwd(op, n, k):
    q0 = torch.randn(n, dtype=F64, device=dev)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


for N, k in ((10000, 100), (30000, 200), (100000, 300), (300000, 300), (1000000, 300), (3000000, 200), (10000000, 100), (30000000, 60)):
    x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
    op = Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2)
    t = fwd(op, N, k)
    print("stencil N=%-9d k=%-3d forward %8.2f ms  %6.1f us/step  %6.0f GB/s algorithmic" % (N, k, t * 1e3, t / k * 1e6, 8.0 * N * (k * k + 12 * k) / t / 1e9))
    del op; engine.BasisArena.release(); torch.cuda.empty_cache()
for L, k in ((20, 50), (20, 100), (20, 400), (22, 100), (24, 60)):
    n = 1 << L
    op = TFIMOperator(L, dev); op.g = torch.tensor([1.0], dtype=F64, device=dev)
    t = fwd(op, n, k)
    print("TFIM L=%d k=%-3d forward %8.2f ms  %6.1f us/step  %6.0f GB/s algorithmic" % (L, k, t * 1e3, t / k * 1e6, 8.0 * n * (k * k + 12 * k) / t / 1e9))
    del op; engine.BasisArena.release(); torch.cuda.empty_cache()
