#!/usr/bin/env python3
"""Headline workload with the operator handed over as an opaque Python callable (the reference's calling
convention, e.g. examples/TFIM/E0.py:60): the mat-vec is user torch code, all other vector work are phase calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dominantsparseeigenad_amd.symeig as symeig
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
from dominantsparseeigenad_amd import engine
from bench import PinnedRandn
dev = torch.device("cuda:0"); L, k = 20, 200; n = 1 << L
g = torch.tensor([1.0], dtype=torch.float64, device=dev, requires_grad=True)
op = TFIMOperator(L, dev); op.g = g
draws = [torch.from_numpy(normal_vector(n, 12355 + c)).to(dev) for c in range(3)]
t = torch.from_numpy(normal_vector(n, 12346)).to(dev); t = t / t.norm()
for name, A, hook in (("native operator", op.H, op.Hadjoint_to_gadjoint),
                      ("opaque python callable", (lambda v: op.H(v)), (lambda v1, v2: op.pHpg(v2).matmul(v1)[None]))):
    symeig.setDominantSparseSymeig(A, hook)
    f = symeig.DominantSparseSymeig.apply
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with PinnedRandn(draws):
            E0, psi = f(g, k, n, dev)
            (gl,) = torch.autograd.grad(E0 + psi.matmul(t), g)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-24s fwd+bwd %.2f ms (min of 6)  E0/L=%.15f  dloss=%.12f  cg=%d" % (name, 1e3 * min(ts), E0.item() / L, gl.item(), engine.last_cg.iters))
