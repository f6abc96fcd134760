#!/usr/bin/env python3
"""Partial re-orthogonalisation (reorth="partial") against the reference's full re-orthogonalisation: extreme Ritz pair,
orthogonality of the basis, steps re-orthogonalised, time.   python tools/partial_reorth_check.py [--big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator, Stencil3Operator
from dominantsparseeigenad_amd.Lanczos import symeigLanczos, Lanczos
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); F64 = torch.float64
cases = [("TFIM L=10", lambda: TFIMOperator(10, dev, g=torch.tensor([1.0], dtype=F64, device=dev)), 1 << 10, 120),
         ("TFIM L=14", lambda: TFIMOperator(14, dev, g=torch.tensor([1.0], dtype=F64, device=dev)), 1 << 14, 200),
         ("TFIM L=16 g=1.5", lambda: TFIMOperator(16, dev, g=torch.tensor([1.5], dtype=F64, device=dev)), 1 << 16, 200),
         ("stencil N=1000", lambda: Stencil3Operator(1000, 2.0 / 1000, 0.5 * torch.linspace(-1, 1, 1000, dtype=F64, device=dev) ** 2), 1000, 300),
         ("stencil N=100000", lambda: Stencil3Operator(100000, 2.0 / 100000, 0.5 * torch.linspace(-1, 1, 100000, dtype=F64, device=dev) ** 2), 100000, 300)]
if "--big" in sys.argv:
    cases = [("TFIM L=20", lambda: TFIMOperator(20, dev, g=torch.tensor([1.0], dtype=F64, device=dev)), 1 << 20, 200)]
if "--L" in sys.argv:      # --L 28 --k 100: the one-GPU anchor of the strong-scaling curve
    LL = int(sys.argv[sys.argv.index("--L") + 1]); kk = int(sys.argv[sys.argv.index("--k") + 1])
    cases = [("TFIM L=%d" % LL, lambda: TFIMOperator(LL, dev, g=torch.tensor([1.0], dtype=F64, device=dev)), 1 << LL, kk)]
REPS = 2 if "--L" in sys.argv else 3
engine.LANCZOS_PERSIST = False
for name, mk, n, k in cases:
    op = mk()
    q0 = torch.from_numpy(normal_vector(n, 7)).to(dev)
    res = {}
    for mode in ("full", "partial"):
        best = 1e30
        for _ in range(REPS):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lam, psi = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0, reorth=mode)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[mode] = (lam.item(), psi.clone(), best, engine.last_reorth_steps)
    lf, pf, tf, _ = res["full"]; lp, pp, tp, steps = res["partial"]
    if torch.dot(pf, pp) < 0: pp = -pp
    resid = (op(pp) - lp * pp).norm().item() if callable(op) else float("nan")
    residf = (op(pf) - lf * pf).norm().item() if callable(op) else float("nan")
    line = "%-18s k=%3d  full %.3f ms  partial %.3f ms (%d of %d steps re-orthogonalised)  |dE0| %.1e  max|dpsi| %.1e  resid full %.1e partial %.1e" % (
        name, k, tf * 1e3, tp * 1e3, steps, k - 1, abs(lf - lp), (pf - pp).abs().max().item(), residf, resid)
    if n <= 1 << 16:
        engine.PARTIAL_REORTH = 0.0
        Qk, T = Lanczos(op, k, dev, sparse=True, dim=n, q0=q0)
        engine.PARTIAL_REORTH = None
        G = Qk.T @ Qk - torch.eye(k, dtype=F64, device=dev)
        Qf, Tf = Lanczos(op, k, dev, sparse=True, dim=n, q0=q0)
        ev, evf = torch.linalg.eigvalsh(T), torch.linalg.eigvalsh(Tf)
        line += "  ||QtQ-I||max %.1e  max|dT[:20,:20]| %.1e  lowest 5 Ritz values differ by %.1e" % (
            G.abs().max().item(), (T - Tf)[:20, :20].abs().max().item(), (ev[:5] - evf[:5]).abs().max().item())
    print(line)
