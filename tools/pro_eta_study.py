#!/usr/bin/env python3
"""How much of the basis would Simon's eta-selection touch?  Replays the omega recurrence of k_pro_update on the host from the
alphas / betas of a full-schedule run and, at every triggered step, counts the basis vectors whose estimate exceeds eta.
    python tools/pro_eta_study.py [--L 20] [--k 200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.Lanczos import Lanczos
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); F64 = torch.float64
L = int(sys.argv[sys.argv.index("--L") + 1]) if "--L" in sys.argv else 20
k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 200
n = 1 << L
engine.LANCZOS_PERSIST = False
op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=F64, device=dev))
q0 = torch.from_numpy(normal_vector(n, 7)).to(dev)
Qk, T = Lanczos(op, k, dev, sparse=True, dim=n, q0=q0)
al = torch.diagonal(T).cpu().numpy(); be = torch.diagonal(T, 1).cpu().numpy()
del Qk
eps1, delta = 64 * 2.220446049250313e-16, 1e-10
for eta in (1e-10 / 30, 1e-12, 1e-13):
    om = np.zeros((2, k + 1)); force = False; anorm = 0.0
    fracs, trig_steps = [], []
    for i in range(1, k):
        bcur = be[i - 1]; a = al[i - 1]; bprev = be[i - 2] if i >= 2 else 0.0
        anorm = max(anorm, abs(a) + bcur + bprev)
        o1, o2 = om[(i - 1) & 1], om[i & 1]
        new = np.zeros(i)
        for kk in range(i):
            if kk == i - 1:
                new[kk] = eps1 * anorm / bcur
            else:
                w1p = 1.0 if kk + 1 == i - 1 else o1[kk + 1]
                w1m = o1[kk - 1] if kk > 0 else 0.0
                w2k = 1.0 if kk == i - 2 else o2[kk]
                t = be[kk] * w1p + (al[kk] - a) * o1[kk] - bprev * w2k + (be[kk - 1] * w1m if kk > 0 else 0.0)
                d = eps1 * ((be[kk] + bcur) + anorm)
                new[kk] = (t + np.copysign(d, t)) / bcur
        mx = np.abs(new).max()
        trig = not (mx <= delta)
        if trig or force:
            sel = np.abs(new) > eta
            lo, hi = (np.where(sel)[0].min(), np.where(sel)[0].max()) if sel.any() else (0, -1)
            fracs.append((sel.sum() / i, (hi - lo + 1) / i))
            trig_steps.append(i)
            new[:] = eps1
        force = trig
        o2[:i] = new
    f = np.array(fracs)
    print("L=%d k=%d eta %.1e: %d re-orthogonalised steps; estimates above eta: %.0f%% of the basis on average (covering interval %.0f%%)"
          % (L, k, eta, len(fracs), 100 * f[:, 0].mean(), 100 * f[:, 1].mean()))
