#!/usr/bin/env python3
"""the two basis passes at one i, a few launches each (to be run under rocprofv3 --pmc ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_void_p
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
dev = torch.device("cuda:0"); lib = _lib.load()
n, i = 1 << 20, int(sys.argv[1]) if len(sys.argv) > 1 else 200
Q = torch.randn((i + 1, n), dtype=torch.float64, device=dev)
u = torch.randn(n, dtype=torch.float64, device=dev); r = torch.empty(n, dtype=torch.float64, device=dev)
c = torch.zeros(i + 2, dtype=torch.float64, device=dev); ab = torch.tensor([0.5, 0.25], dtype=torch.float64, device=dev)
nrm2 = torch.zeros(1, dtype=torch.float64, device=dev)
ws = Workspace.get(n, i + 1, dev); st = _stream(dev)
for _ in range(5):
    lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), n, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st)
    lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), n, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st)
torch.cuda.synchronize()
