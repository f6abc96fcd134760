#!/usr/bin/env python3
"""Is the partial-reorth loop bound by the host's enqueue rate?  Host time inside dsea_lanczos_run vs total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import engine, _lib
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); F64 = torch.float64
lib = _lib.load()
L, k = 20, 200
n = 1 << L
op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=F64, device=dev))
q0 = torch.from_numpy(normal_vector(n, 7)).to(dev)
ws = engine.Workspace.get(n, k, dev)
ldq = n
Q = torch.empty((k, ldq), dtype=F64, device=dev); al = torch.zeros(k, dtype=F64, device=dev); be = torch.zeros(k, dtype=F64, device=dev)
st = engine._stream(dev)
for mode in (0, 1):
    engine.check(lib.dsea_ws_set_partial_reorth(ws.handle, mode, 0.0), "set")
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        engine.check(lib.dsea_lanczos_run(op.handle, ws.handle, k, engine._ptr(q0), engine._ptr(Q), ldq, engine._ptr(al), engine._ptr(be), st), "run")
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("partial=%d: host enqueue %.3f ms, total %.3f ms" % (mode, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
