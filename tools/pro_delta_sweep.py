#!/usr/bin/env python3
"""Threshold of the partial re-orthogonalisation: steps selected, forward time and Ritz-pair quality against the full schedule.
    python tools/pro_delta_sweep.py [--L 20] [--k 200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.Lanczos import symeigLanczos
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); F64 = torch.float64
L = int(sys.argv[sys.argv.index("--L") + 1]) if "--L" in sys.argv else 20
k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 200
n = 1 << L
engine.LANCZOS_PERSIST = False
op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=F64, device=dev))
q0 = torch.from_numpy(normal_vector(n, 7)).to(dev)
lo_f, v_f = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
for delta in (1.5e-8, 1e-9, 1e-10, 1e-11, 1.8e-12, 1e-13):
    engine.PARTIAL_REORTH = delta
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lo, v = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    engine.PARTIAL_REORTH = None
    s = 1.0 if float(v @ v_f) > 0 else -1.0
    print("L=%d k=%d delta %.1e: %3d of %d steps, forward %.2f ms, |dE0| %.1e, max|dpsi| %.1e, residual %.1e" % (
        L, k, delta, engine.last_reorth_steps, k - 1, best * 1e3, abs(lo.item() - lo_f.item()), float((v_f - s * v).abs().max()),
        float((op(v) - lo * v).norm())))
