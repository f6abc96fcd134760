#!/usr/bin/env python3
"""Micro-benchmark of the two basis-streaming kernels (reorth pair) at fixed i, per rows-per-lane variant.
   python tools/kbench.py [--n-log2 20] [--i 199] [--reps 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_void_p
from dominantsparseeigenad_amd import _lib, engine
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream

ap = argparse.ArgumentParser()
ap.add_argument("--n-log2", type=int, default=20)
ap.add_argument("--n", type=int, default=0, help="rows (overrides --n-log2)")
ap.add_argument("--splits", default="4,8,16")
ap.add_argument("--no-ceiling", action="store_true")
ap.add_argument("--i", type=int, default=199)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rpls", default="2,4,8,16")
ap.add_argument("--ldq-pad", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
n, i = (args.n or (1 << args.n_log2)), args.i
ldq = (n + 31) // 32 * 32 + args.ldq_pad
Q = torch.randn((i + 1, ldq), dtype=torch.float64, device=dev)
u = torch.randn(n, dtype=torch.float64, device=dev)
r = torch.empty(n, dtype=torch.float64, device=dev)
c = torch.zeros(i + 2, dtype=torch.float64, device=dev)
ab = torch.tensor([0.5, 0.25], dtype=torch.float64, device=dev)
nrm2 = torch.zeros(1, dtype=torch.float64, device=dev)
ws = Workspace.get(n, i + 1, dev)
st = _stream(dev)
GB = 8.0 * n / 1e9

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps

# copy ceiling for reference
if not args.no_ceiling:
    big = torch.empty(1 << 28, dtype=torch.float64, device=dev); big2 = torch.empty_like(big)
    t = timeit(lambda: big2.copy_(big))
    print("torch copy 2 GiB->2 GiB: %.3f ms  %.0f GB/s (read+write)" % (t, 2 * big.numel() * 8 / t / 1e6))
    t = timeit(lambda: big.sum())
    print("torch sum 2 GiB: %.3f ms  %.0f GB/s (read)" % (t, big.numel() * 8 / t / 1e6))
    del big, big2
for sw in [int(x) for x in args.splits.split(",") if x]:
    ws.set_rows_per_lane(0); ws.set_split(sw)
    t1 = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st))
    t2 = timeit(lambda: lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), ldq, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st))
    print("split W=%2d  rdots(+finalize) %.1f us  %.0f GB/s | axpy_norm(+finalize) %.1f us  %.0f GB/s" % (
        sw, t1 * 1e3, (i + 5) * GB / t1 * 1e3, t2 * 1e3, (i + 2) * GB / t2 * 1e3))
ws.set_split(0)
for rpl in [int(x) for x in args.rpls.split(",") if x]:
    ws.set_rows_per_lane(rpl)
    t1 = timeit(lambda: lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(ab), c_void_p(ab.data_ptr() + 8), _ptr(r), _ptr(c), st))
    t2 = timeit(lambda: lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), ldq, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st))
    print("rpl=%2d  rdots(+finalize) %.1f us  %.0f GB/s | axpy_norm(+finalize) %.1f us  %.0f GB/s" % (
        rpl, t1 * 1e3, (i + 5) * GB / t1 * 1e3, t2 * 1e3, (i + 2) * GB / t2 * 1e3))
