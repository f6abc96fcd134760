#!/usr/bin/env python3
"""Does a HIP graph of the whole native Lanczos run (4 launches per step, no host sync inside) buy anything over the
plain stream launches?  Config-3 shape (3-point stencil N = 1e5, k = 300) and a README-sized TFIM (L = 12, k = 100)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dominantsparseeigenad_amd import _lib, engine
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream, round_up
from dominantsparseeigenad_amd.operators import Stencil3Operator, TFIMOperator
dev = torch.device("cuda:0"); lib = _lib.load(); F64 = torch.float64


def case(name, op, n, k):
    ldq = round_up(n, 32)
    Q = torch.empty((k, ldq), dtype=F64, device=dev); Qs = torch.empty((k, ldq), dtype=torch.bfloat16, device=dev)
    al = torch.empty(k, dtype=F64, device=dev); be = torch.empty(k, dtype=F64, device=dev)
    q0 = torch.randn(n, dtype=F64, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ws = Workspace.get(n, k, dev)
        st = _stream(dev)
        _lib.check(lib.dsea_ws_set_shadow(ws.handle, _ptr(Qs), ldq, k, 1e-12), "shadow")
        run = lambda: _lib.check(lib.dsea_lanczos_run(op.handle, ws.handle, k, _ptr(q0), _ptr(Q), ldq, _ptr(al), _ptr(be), st), "run")
        for _ in range(3): run()
        side.synchronize()
        best = 1e9
        for _ in range(5):
            side.synchronize(); t0 = time.perf_counter(); run(); side.synchronize(); best = min(best, time.perf_counter() - t0)
        a_ref = al.clone()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=side):
                run()
        except Exception as exc:
            print(name, "capture failed:", type(exc).__name__, str(exc)[:200]); return
        side.synchronize()
        bg = 1e9
        for _ in range(5):
            side.synchronize(); t0 = time.perf_counter(); g.replay(); side.synchronize(); bg = min(bg, time.perf_counter() - t0)
        print("%-28s stream launches %.3f ms (%.1f us/step)   graph replay %.3f ms (%.1f us/step)   same alphas: %s" % (
            name, best * 1e3, best / k * 1e6, bg * 1e3, bg / k * 1e6, bool(torch.equal(a_ref, al))))


N = 100000
x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
case("stencil N=1e5 k=300", Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2), N, 300)
for L, k in ((12, 100), (16, 200)):
    op = TFIMOperator(L, dev); op.g = torch.tensor([1.0], dtype=F64, device=dev)
    case("TFIM L=%d k=%d" % (L, k), op, 1 << L, k)
