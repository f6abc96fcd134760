#!/usr/bin/env python3
"""Per-phase time of k_lanczos_persist_mid (one workgroup's view, us per step averaged over the run).  Needs the timing build
(the kernel then leaves its phase clocks in alphas[0..7]):
    make -C dominantsparseeigenad_amd/csrc libdsea_TIM.so
    DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_TIM.so python tools/lanczos_mid_phase_timing.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from dominantsparseeigenad_amd import engine, _lib
from dominantsparseeigenad_amd.operators import Stencil3Operator
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); F64 = torch.float64
lib = _lib.load()
for N, k in ((10000, 300), (100000, 80), (100000, 300)):
    x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
    op = Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2)
    q0 = torch.from_numpy(normal_vector(N, 5)).to(dev)
    ws = engine.Workspace.get(N, k, dev)
    ldq = (N + 31) // 32 * 32
    Q = torch.empty((k, ldq), dtype=F64, device=dev); al = torch.zeros(k, dtype=F64, device=dev); be = torch.zeros(k, dtype=F64, device=dev)
    Qs = torch.empty((k, ldq), dtype=torch.bfloat16, device=dev)
    lib.dsea_ws_set_shadow(ws.handle, engine._ptr(Qs), ldq, k, 1e-12)
    st = engine._stream(dev)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lib.dsea_lanczos_run(op.handle, ws.handle, k, engine._ptr(q0), engine._ptr(Q), ldq, engine._ptr(al), engine._ptr(be), st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.dsea_ws_set_shadow(ws.handle, None, 0, 0, 0.0)
    d = al[:8].cpu().numpy()
    print("N=%d k=%d: %.2f us/step | X1 (publish, gather, B1) %.2f  q/u/three-term/stores %.2f  dots..B2 %.2f  X2a publish+owner %.2f  "
          "X2b gather..B3 %.2f  premise %.2f  correction (to B4) %.2f  combine %.2f  (sum %.2f)"
          % (N, k, dt / k * 1e6, *d, d.sum()))
