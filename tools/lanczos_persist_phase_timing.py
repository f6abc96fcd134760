#!/usr/bin/env python3
"""Per-phase time of k_lanczos_persist (workgroup 0's view, us per step).  Needs the timing build of the library (the
kernel then leaves its phase clocks in alphas[0..4]):
    make -C dominantsparseeigenad_amd/csrc libdsea_TIM.so
    DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_TIM.so python tools/lanczos_persist_phase_timing.py"""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np, torch
from dominantsparseeigenad_amd import engine, _lib
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); F64 = torch.float64
lib = _lib.load()
for L, k in ((8, 200), (10, 300), (12, 300)):
    n = 1 << L
    op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=F64, device=dev))
    q0 = torch.from_numpy(normal_vector(n, 5)).to(dev)
    ws = engine.Workspace.get(n, k, dev)
    ldq = (n + 31) // 32 * 32
    Q = torch.empty((k, ldq), dtype=F64, device=dev); al = torch.zeros(k, dtype=F64, device=dev); be = torch.zeros(k, dtype=F64, device=dev)
    st = engine._stream(dev)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.dsea_lanczos_run(op.handle, ws.handle, k, engine._ptr(q0), engine._ptr(Q), ldq, engine._ptr(al), engine._ptr(be), st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    d = al[:5].cpu().numpy()
    print("L=%d k=%d: %.2f us/step | E3 (norm + rows) %.2f  normalise+matvec+EA %.2f  three-term+dots %.2f  E2 (coefficients) %.2f  correction %.2f" % (L, k, dt / k * 1e6, *d))
