#!/usr/bin/env python3
"""Per-phase time of k_cg_persist_tfim_big (one workgroup's view, us per iteration).  Needs a library built with
-DDSEA_CGB_TIMING (the kernel then leaves its phase clocks in the first doubles of the d buffer):
    make -C dominantsparseeigenad_amd/csrc libdsea_TIM.so
    DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_TIM.so python tools/cg_persist_phase_timing.py"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0")
for L in (16, 20):
    n = 1 << L
    op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=torch.float64, device=dev))
    b = torch.from_numpy(normal_vector(n, 2)).to(dev); x0 = torch.from_numpy(normal_vector(n, 3)).to(dev)
    shift = torch.tensor(-30.0, dtype=torch.float64, device=dev)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        engine.cg(b, x0, native=op, shift=shift, eps=0.0, maxiter=400, poll_every=400)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ws = engine.Workspace.get(n, 8, dev)
    # dbuf[0] = workspace vector 2: read its first 6 doubles through the workspace buffer layout: easier via ctypes? use torch view
    import ctypes
    lib = engine._lib.load()
    nbytes = ctypes.c_size_t(); lib.dsea_ws_bytes(ws.n, ws.kmax, ctypes.byref(nbytes))
    buf = ws.buffer.view(torch.float64)
    npad = (ws.n + 255) // 256 * 256
    vec_off = buf.numel() - 4 * npad
    dbg = buf[vec_off + 2 * npad: vec_off + 2 * npad + 6].cpu().numpy()
    if engine.CG_TFIM_REFERENCE_RECURRENCES:
        print("L=%d two-exchange form: %.2f us/iteration total | wait partners %.2f  matvec %.2f  S1 (d.Ad) %.2f  update+S2 (r.r) %.2f  direction+publish d %.2f  loop top %.2f" % (
            L, dt / 400 * 1e6, dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5]))
    else:
        print("L=%d one-exchange form: %.2f us/iteration total | publish r %.2f  wait partners %.2f  matvec %.2f  exchange (gamma, delta) %.2f  vector updates %.2f" % (
            L, dt / 400 * 1e6, dbg[0], dbg[1], dbg[2], dbg[3], dbg[4]))
