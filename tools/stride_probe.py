#!/usr/bin/env python3
"""Is the power-of-two stride between basis vectors (ldq = n = 2^20 doubles = 8 MiB) costing the dots pass anything?
Several FRESH allocations per leading dimension (placement of a 1.7 GB buffer moves the time by a few per cent on its
own, so single measurements cannot be compared).   python tools/stride_probe.py [--reps 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_void_p
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--i", type=int, default=199)
ap.add_argument("--pads", default="0,512,1024,1536"); args = ap.parse_args()
dev = torch.device("cuda:0"); lib = _lib.load()
n, i = 1 << 20, args.i
u = torch.randn(n, dtype=torch.float64, device=dev); r = torch.empty(n, dtype=torch.float64, device=dev)
c = torch.zeros(i + 2, dtype=torch.float64, device=dev); ab = torch.tensor([0.5, 0.25], dtype=torch.float64, device=dev)
ws = Workspace.get(n, i + 1, dev); st = _stream(dev)


def timeit(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


hold = []
res = {}
for rep in range(args.reps):
    for pad in [int(p) for p in args.pads.split(",")]:
        ldq = n + pad
        Q = torch.randn((i + 1, ldq), dtype=torch.float64, device=dev)
        t = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st))
        res.setdefault(pad, []).append(t)
        del Q
        torch.cuda.empty_cache()
    hold.append(torch.empty((rep + 1) * (53 << 20), dtype=torch.uint8, device=dev))
for pad, ts in res.items():
    print("ldq = n + %-5d rdots+finalize at i=%d: %s  min %.1f  median %.1f us" % (pad, i, " ".join("%.1f" % t for t in ts), min(ts), sorted(ts)[len(ts) // 2]))
