#!/usr/bin/env python3
"""Config-4 shape (transfer matrix D = 512, A and A^T): 107 Arnoldi columns of BOTH factorisations
  (a) as today: one dsea_arnoldi_extend per side, two host threads, two streams;
  (b) the same, one side after the other on one stream;
  (c) dsea_arnoldi_extend_pair: lock step, the orthogonalisation of both sides in the same launches, one stream.
Device time only (no stage tests): what a lock-step driver could gain.   python tools/kbench_arnoldi_pair.py [columns]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import _lib, engine
from dominantsparseeigenad_amd.engine import _ptr
from dominantsparseeigenad_amd.operators import TransferOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0"); lib = _lib.load(); F64 = torch.float64
D, d = 512, 2; n = D * D; m = int(sys.argv[1]) if len(sys.argv) > 1 else 107
A = torch.from_numpy(normal_vector(d * D * D, 5).reshape(d, D, D)).to(dev) / (D ** 0.5)
ops = [TransferOperator(A, transpose=False), TransferOperator(A, transpose=True)]
ldv = engine.round_up(n, 32); ldh = m + 1
wss = [engine.Workspace(n, m + 2, dev) for _ in range(2)]
for ws in wss:
    _lib.check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 1), "opt")
def fresh():
    Vs = [torch.zeros((m + 1, ldv), dtype=F64, device=dev) for _ in range(2)]
    Hs = [torch.zeros((m, ldh), dtype=F64, device=dev) for _ in range(2)]
    for s in range(2):
        v = torch.from_numpy(normal_vector(n, 70 + s)).to(dev); Vs[s][0, :n] = v / v.norm()
    return Vs, Hs
def st(stream): from ctypes import c_void_p; return c_void_p(stream.cuda_stream)
def run_two_streams(Vs, Hs, streams):
    def side(s):
        with torch.cuda.stream(streams[s]):
            _lib.check(lib.dsea_arnoldi_extend(ops[s].handle, wss[s].handle, None, _ptr(Vs[s]), ldv, 0, m, _ptr(Hs[s]), ldh, st(streams[s])), "extend")
    th = [threading.Thread(target=side, args=(s,)) for s in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
def run_serial(Vs, Hs, stream):
    for s in range(2):
        _lib.check(lib.dsea_arnoldi_extend(ops[s].handle, wss[s].handle, None, _ptr(Vs[s]), ldv, 0, m, _ptr(Hs[s]), ldh, st(stream)), "extend")
def run_pair(Vs, Hs, stream):
    _lib.check(lib.dsea_arnoldi_extend_pair(ops[0].handle, ops[1].handle, wss[0].handle, wss[1].handle, _ptr(Vs[0]), _ptr(Vs[1]), ldv, 0, m,
                                            _ptr(Hs[0]), _ptr(Hs[1]), ldh, st(stream)), "extend_pair")
main = torch.cuda.current_stream(dev); side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
ref = None
for name, fn in (("two streams (as today)", lambda V, H: run_two_streams(V, H, side)), ("serial, one stream", lambda V, H: run_serial(V, H, main)),
                 ("pair launches, one stream", lambda V, H: run_pair(V, H, main))):
    ts = []
    for rep in range(6):
        Vs, Hs = fresh(); torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(Vs, Hs); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    if ref is None: ref = (Vs, Hs)
    same = all(torch.equal(Hs[s], ref[1][s]) and torch.equal(Vs[s], ref[0][s]) for s in range(2))
    print("%-28s %d columns x 2 sides: %.2f ms (min of 6; all: %s)   H, V bit-identical to the first mode: %s"
          % (name, m, min(ts), " ".join("%.2f" % t for t in ts), same))
