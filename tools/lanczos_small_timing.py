#!/usr/bin/env python3
"""us per Lanczos step on README-sized problems: single-launch form vs multi-launch kernels.
   python tools/lanczos_small_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.Lanczos import symeigLanczos
from dominantsparseeigenad_amd.operators import TFIMOperator, Stencil3Operator
from dominantsparseeigenad_amd.synthetic import normal_vector

dev = torch.device("cuda:0")


def time_it(op, k, n, q0, reps=5):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


cases = [("TFIM L=%d" % L, TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=torch.float64, device=dev)), 1 << L, k)
         for L, k in ((8, 200), (10, 300), (11, 300), (12, 300), (13, 300))]
for N in (300, 1000, 4096, 8192):
    x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
    cases.append(("stencil N=%d" % N, Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2), N, min(300, N)))
print("%-18s %5s  %14s %14s   speed-up   E0 deviation" % ("problem", "k", "single-launch", "multi-launch"))
for name, op, n, k in cases:
    q0 = torch.from_numpy(normal_vector(n, 5)).to(dev)
    res = {}
    for on in (True, False):
        engine.LANCZOS_PERSIST = "force" if on else False
        symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
        t = time_it(op, k, n, q0)
        lo, _ = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
        res[on] = (t, lo.item())
    engine.LANCZOS_PERSIST = True
    print("%-18s %5d  %8.2f us/step %8.2f us/step   %5.2fx    %.1e" % (
        name, k, res[True][0] / k * 1e6, res[False][0] / k * 1e6, res[False][0] / res[True][0],
        abs(res[True][1] - res[False][1]) / abs(res[False][1])))
