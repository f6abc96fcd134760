#!/usr/bin/env python3
"""BASELINE configs[2] in its well-posed restatement: 1-D Schroedinger stencil, N = 100000.
Lanczos k = 300 forward time, and CG on the shifted SPD system over a FIXED 1000 iterations (us / iteration and
algorithmic GB/s = 11 vectors per iteration) -- persistent single-launch form vs streaming 3-launch form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import Stencil3Operator
from dominantsparseeigenad_amd.Lanczos import symeigLanczos
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 300
h = 2.0 / N
x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
op = Stencil3Operator(N, h, 0.5 * x ** 2)
q0 = torch.from_numpy(normal_vector(N, 1)).to(dev)
if os.environ.get("C3_SPLIT"):      # A/B: force the split width (and with it one sub-tile per block in the dots pass)
    engine.Workspace.get(N, k, dev).set_split(int(os.environ["C3_SPLIT"]))
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lam, psi = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=N, q0=q0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("Lanczos N=%d k=%d: %.2f ms  (%.1f us/step; algorithmic %.0f GB/s)  theta=%.6f" % (
    N, k, (t1 - t0) * 1e3, (t1 - t0) / k * 1e6, 8.0 * N * (k * k + 12 * k) / (t1 - t0) / 1e9, lam.item()))
if os.environ.get("C3_LANCZOS_ONLY"):
    sys.exit(0)
b = torch.from_numpy(normal_vector(N, 2)).to(dev); x0 = torch.from_numpy(normal_vector(N, 3)).to(dev)
shift = torch.tensor(-1.0, dtype=torch.float64, device=dev)
ws = engine.Workspace.get(N, 8, dev)
for mode, name in ((0, "streaming 3-launch"), (-1, "persistent auto"), (1, "persistent ppt=1 nvb=4"), (2, "persistent ppt=2 nvb=4"), (21, "persistent ppt=1 nvb=2"), (22, "persistent ppt=2 nvb=2"), (11, "persistent ppt=1 nvb=1"), (12, "persistent ppt=2 nvb=1")):
    ws.set_persist(mode)
    best = 1e30
    try:
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            xs = engine.cg(b, x0, native=op, shift=shift, eps=0.0, maxiter=1000, poll_every=1000)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("CG %-26s fixed %d iterations: %.2f ms  %.2f us/iteration  %.0f GB/s algorithmic" % (
            name, engine.last_cg.iters, best * 1e3, best / 1000 * 1e6, 11 * 8.0 * N * 1000 / best / 1e9))
    except Exception as exc:
        print("CG %-26s failed: %s" % (name, exc))
ws.set_persist(-1)
