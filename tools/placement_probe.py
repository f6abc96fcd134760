#!/usr/bin/env python3
"""Placement of the basis: time the dots pass (i = 199, n = 2^20) on several allocations of the SAME shape that are
alive at the same time (hence on different physical pages), plus the shadow pass on its bf16 copy.
   python tools/placement_probe.py [--cands 8]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_void_p
from dominantsparseeigenad_amd import _lib
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream
ap = argparse.ArgumentParser(); ap.add_argument("--cands", type=int, default=9); ap.add_argument("--i", type=int, default=199)
ap.add_argument("--n-log2", type=int, default=20); args = ap.parse_args()
dev = torch.device("cuda:0"); lib = _lib.load()
n, i = 1 << args.n_log2, args.i
u = torch.randn(n, dtype=torch.float64, device=dev); r = torch.empty(n, dtype=torch.float64, device=dev)
c = torch.zeros(i + 2, dtype=torch.float64, device=dev); ab = torch.tensor([0.5, 0.25], dtype=torch.float64, device=dev)
ws = Workspace.get(n, i + 1, dev); st = _stream(dev)


def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cands = []
for cnd in range(args.cands):
    Q = torch.empty((i + 1, n), dtype=torch.float64, device=dev)
    kind = ("normal", "untouched", "zeros")[cnd % 3]
    if kind == "normal": Q.normal_()
    if kind == "zeros": Q.zero_()
    t = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), n, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st))
    t3 = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), n, n, 31, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st), reps=10)
    cands.append(Q)
    from dominantsparseeigenad_amd.engine import _dots_probe_us
    tp = _dots_probe_us(Q.view(torch.uint8).reshape(-1), i + 1, n, n, dev)
    print("[%s, engine probe %.1f us] " % (kind, tp), end="")
    print("candidate %d  Q @ %#x  dots pass i=%d: %.1f us (%.0f GB/s)   short probe i=31: %.1f us" % (cnd, Q.data_ptr(), i, t, (i + 5) * 8.0 * n / t / 1e3, t3))
# the same candidates again, in reverse order: is the mode a property of the allocation?
for cnd in reversed(range(args.cands)):
    Q = cands[cnd]
    t = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), n, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st))
    print("candidate %d again: %.1f us" % (cnd, t))
