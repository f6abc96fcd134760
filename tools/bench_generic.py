#!/usr/bin/env python3
"""Headline workload with the operator handed over the way the reference's callers do (examples/TFIM/E0.py:59-62): an opaque
Python callable -- the mat-vec is user code, every other vector operation a phase call of include/dsea.h issued from Python.
    python tools/bench_generic.py [--mode native|callable|tables] [--reps 6]
Under `rocprofv3 --kernel-trace` + tools/trace_gaps.py (tools/gpu_evidence.sh callable) it gives the host time outside kernels
per Lanczos step / CG iteration: profiles/r06_callable_operand_trace.txt."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dominantsparseeigenad_amd.symeig as symeig
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
from dominantsparseeigenad_amd import engine
from bench import PinnedRandn, _ReferenceStyleTFIM
ap = argparse.ArgumentParser(); ap.add_argument("--mode", default="all"); ap.add_argument("--reps", type=int, default=6)
args = ap.parse_args()
dev = torch.device("cuda:0"); L, k = 20, 200; n = 1 << L
g = torch.tensor([1.0], dtype=torch.float64, device=dev, requires_grad=True)
op = TFIMOperator(L, dev); op.g = g
draws = [torch.from_numpy(normal_vector(n, 12355 + c)).to(dev) for c in range(3)]
t = torch.from_numpy(normal_vector(n, 12346)).to(dev); t = t / t.norm()
hook = lambda v1, v2: op.pHpg(v2).matmul(v1)[None]
cases = {"native": ("native operator", op.H, op.Hadjoint_to_gadjoint),
         "callable": ("lambda around the native mat-vec", (lambda v: op.H(v)), hook)}
if args.mode in ("tables", "all"):
    cases["tables"] = ("reference-style torch gather tables", _ReferenceStyleTFIM(L, g, dev).H, hook)
for key, (name, A, hk) in cases.items():
    if args.mode not in ("all", key):
        continue
    symeig.setDominantSparseSymeig(A, hk)
    f = symeig.DominantSparseSymeig.apply
    ts = []
    for it in range(args.reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with PinnedRandn(draws):
            E0, psi = f(g, k, n, dev)
            (gl,) = torch.autograd.grad(E0 + psi.matmul(t), g)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-36s fwd+bwd %.2f ms (min of %d)  E0/L=%.15f  dloss=%.12f  cg=%d (%s, %d polls)"
          % (name, 1e3 * min(ts), args.reps, E0.item() / L, gl.item(), engine.last_cg.iters, engine.last_cg.form, getattr(engine.last_cg, "polls", 0)))
