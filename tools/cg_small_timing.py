#!/usr/bin/env python3
"""us per CG iteration on README-sized problems (fixed iteration count, eps = 0).   python tools/cg_small_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator, Stencil3Operator
from dominantsparseeigenad_amd.synthetic import normal_vector

dev = torch.device("cuda:0")
iters = 400
cases = [("TFIM L=%d" % L, TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=torch.float64, device=dev)), 1 << L)
         for L in (8, 10, 11, 12, 13, 14, 16, 18, 19, 20)]
for N in (300, 1000, 4096):
    x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
    cases.append(("stencil N=%d" % N, Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2), N))
for name, op, n in cases:
    b = torch.from_numpy(normal_vector(n, 2)).to(dev)
    x0 = torch.from_numpy(normal_vector(n, 3)).to(dev)
    shift = torch.tensor(-30.0, dtype=torch.float64, device=dev)
    out = []
    for modes in ((-1,), (0,)):
        ws = engine.Workspace.get(n, 8, dev)
        ws.set_persist(modes[0])
        best = 1e30
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            engine.cg(b, x0, native=op, shift=shift, eps=0.0, maxiter=iters, poll_every=iters)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        ws.set_persist(-1)
        out.append(best / engine.last_cg.iters * 1e6)
    print("%-16s  default %6.2f us/iteration   streaming kernels %6.2f us/iteration" % (name, out[0], out[1]))
