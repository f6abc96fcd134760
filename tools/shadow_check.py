#!/usr/bin/env python3
"""bf16 shadow vs all-fp64 correction pass across couplings (incl. the quasi-degenerate g < 1 regime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.Lanczos import symeigLanczos, Lanczos
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 18
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 1 << L
q0 = torch.from_numpy(normal_vector(n, 5)).to(dev)
for gval in (0.3, 0.5, 0.9, 1.0, 1.5, 3.0):
    op = TFIMOperator(L, dev, g=torch.tensor([gval], dtype=torch.float64, device=dev))
    out = {}
    for name, use in (("fp64", False), ("shadow", True)):
        engine.USE_SHADOW = use
        lam, psi = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
        stats = engine.lanczos_lp_stats(n, dev) if use else (0, 0)
        Qk, T = Lanczos(op, k, dev, sparse=True, dim=n, q0=q0)
        G = Qk.T @ Qk
        orth = float((G - torch.eye(k, dtype=torch.float64, device=dev)).abs().max())
        res = float((op.H(psi) - lam * psi).norm())
        out[name] = (lam.item(), psi, stats, orth, res)
        del Qk, G
    d_lam = abs(out["fp64"][0] - out["shadow"][0]) / abs(out["fp64"][0])
    s = 1.0 if float(out["fp64"][1] @ out["shadow"][1]) > 0 else -1.0
    d_psi = float((out["fp64"][1] - s * out["shadow"][1]).abs().max())
    print("g=%.1f  E0 rel diff %.1e  psi max diff %.1e | orth fp64 %.1e shadow %.1e | resid fp64 %.1e shadow %.1e | lp/fallback %s" % (
        gval, d_lam, d_psi, out["fp64"][3], out["shadow"][3], out["fp64"][4], out["shadow"][4], out["shadow"][2]))
