#!/usr/bin/env python3
"""Mid-size single-launch Lanczos (csrc/dsea_lanczos_persist_mid.hip; 3-point stencil, 8192 < N <= 131072) against the
multi-launch kernels: ms per k-step run and us per step, best of 4, same start vector; extreme Ritz value of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import Stencil3Operator
from dominantsparseeigenad_amd.Lanczos import symeigLanczos
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0")
cases = [(10000, 300), (20000, 300), (50000, 300), (100000, 100), (100000, 300), (131072, 300), (100000, 500)]
if len(sys.argv) > 2:
    cases = [(int(sys.argv[1]), int(sys.argv[2]))]
for N, k in cases:
    x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
    op = Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2)
    q0 = torch.from_numpy(normal_vector(N, 1)).to(dev)
    out = []
    for mode, name in (("force", "single launch"), ("small", "multi-launch")):
        engine.LANCZOS_PERSIST = mode
        best = 1e30
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lam, psi = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=N, q0=q0)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        lp = engine.lanczos_lp_stats(N, dev)
        out.append("%s %.2f ms = %.1f us/step (theta %.9f, shadow steps %d / fp64 %d)" % (name, best * 1e3, best / k * 1e6, lam.item(), lp[0], lp[1]))
    engine.LANCZOS_PERSIST = True
    print("stencil N=%d k=%d: %s" % (N, k, " | ".join(out)), flush=True)
