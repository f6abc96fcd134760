#!/usr/bin/env python3
"""forward / backward wall split of the headline workload (host timers around synchronised phases)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dominantsparseeigenad_amd.symeig as symeig
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
from dominantsparseeigenad_amd import engine
from bench import PinnedRandn
dev = torch.device("cuda:0")
L, k = 20, 200
n = 1 << L
g = torch.tensor([1.0], dtype=torch.float64, device=dev, requires_grad=True)
op = TFIMOperator(L, dev); op.g = g
symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)
f = symeig.DominantSparseSymeig.apply
draws = [torch.from_numpy(normal_vector(n, 12355 + c)).to(dev) for c in range(3)]
t = torch.from_numpy(normal_vector(n, 12346)).to(dev); t = t / t.norm()
fw, bw = [], []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with PinnedRandn(draws):
        E0, psi = f(g, k, n, dev)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss = E0 + psi.matmul(t)
        (gl,) = torch.autograd.grad(loss, g)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    fw.append(t1 - t0); bw.append(t2 - t1)
print("forward  %.3f ms (min %.3f)" % (1e3 * sorted(fw)[len(fw)//2], 1e3 * min(fw)))
print("backward %.3f ms (min %.3f)  cg iters %d" % (1e3 * sorted(bw)[len(bw)//2], 1e3 * min(bw), engine.last_cg.iters))
# host-side pieces of the forward
import numpy as np
a = torch.randn(k, dtype=torch.float64, device=dev); b = torch.randn(k - 1, dtype=torch.float64, device=dev).abs()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): engine.tridiag_extreme(a, b, "min")
print("tridiag_extreme (D2H + LAPACK): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
