#!/usr/bin/env python3
"""One-exchange (Chronopoulos-Gear) single-launch TFIM CG against the two-exchange form (CG.py's recurrences, bit-identical
to the streaming kernels): deviation of the iterates, recurrence residual and TRUE residual ||b - A'x|| after a fixed number
of iterations, and converged runs at the reference's eps = 1e-7 and at 1e-12."""
import sys
sys.path.insert(0, '.')
import torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); F64 = torch.float64

def solve(op, b, x0, shift, mode, **kw):
    ws = engine.Workspace.get(op.n, 8, dev)
    ws.set_persist(mode)
    try:
        x = engine.cg(b, x0, native=op, shift=shift, **kw)
    finally:
        ws.set_persist(-1)
    return x, engine.last_cg.iters, engine.last_cg.resnorm

for L, sh in ((14, -19.2), (17, -23.1), (20, -25.49), (20, -27.0)):
    n = 1 << L
    op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=F64, device=dev))
    b = torch.from_numpy(normal_vector(n, 500 + L)).to(dev)
    x0 = torch.from_numpy(normal_vector(n, 600 + L)).to(dev)
    shift = torch.tensor(sh, dtype=F64, device=dev)
    true_res = lambda x: float((op.H(x) - shift * x - b).norm())
    print("L=%d shift=%.2f  ||b||=%.1f" % (L, sh, float(b.norm())))
    for its in (10, 20, 30, 40, 50, 80, 120):
        xr, _, rr = solve(op, b, x0, shift, 200, eps=0.0, maxiter=its)
        xm, _, rm = solve(op, b, x0, shift, -1, eps=0.0, maxiter=its)
        print("  its %3d: max|x1 - x2| / max|x| %.1e   recurrence ||r|| two-exch %.2e one-exch %.2e   TRUE ||b - A'x|| two-exch %.2e one-exch %.2e"
              % (its, float((xr - xm).abs().max() / xr.abs().max()), rr, rm, true_res(xr), true_res(xm)))
    for eps in (1e-7, 1e-10, 1e-12):
        xr, ir, rr = solve(op, b, x0, shift, 200, eps=eps, maxiter=None)
        xm, im, rm = solve(op, b, x0, shift, -1, eps=eps, maxiter=None)
        print("  eps %.0e: iterations %d / %d   TRUE residual %.2e / %.2e   max|x1 - x2| / max|x| %.1e"
              % (eps, ir, im, true_res(xr), true_res(xm), float((xr - xm).abs().max() / xr.abs().max())))
