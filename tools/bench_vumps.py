#!/usr/bin/env python3
"""BASELINE config 4 shape: dominant eigen-triple of an MPS transfer matrix (bond dimension D, n = D^2) through
DominantSparseEig on the device (krylov.py) -- forward (two Arnoldi solves) + backward (two GMRES solves).
   python tools/bench_vumps.py [D] [k]
The mat-vec is sum_s A_s r A_s^T (reference examples/TFIM_vumps/general.py:59-66) as batched fp64 GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dominantsparseeigenad_amd.eig as eig
from dominantsparseeigenad_amd import krylov
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0"); d = 2; n = D * D
torch.manual_seed(0)
A = (torch.randn(d, D, D, dtype=torch.float64, device=dev) / D ** 0.5).requires_grad_(True)
Ad = A.detach(); AdT = Ad.transpose(1, 2).contiguous()
count = [0]
def fr(v):
    count[0] += 1
    return torch.matmul(torch.matmul(Ad, v.reshape(D, D)), AdT).sum(0).reshape(-1)
def fl(v):
    count[0] += 1
    return torch.matmul(torch.matmul(AdT, v.reshape(D, D)), Ad).sum(0).reshape(-1)
def hook(pieces):
    gA = torch.zeros_like(Ad)
    for u, v in pieces:
        um, vm = u.reshape(D, D), v.reshape(D, D)
        gA = gA + torch.matmul(torch.matmul(um, Ad), vm.T) + torch.matmul(torch.matmul(um.T, Ad), vm)
    return gA
NATIVE = os.environ.get("DSEA_VUMPS_CALLABLE", "0") != "1"
if NATIVE:   # loops + batched-GEMM mat-vec inside libdsea
    from dominantsparseeigenad_amd.operators import TransferOperator
    op, opT = TransferOperator(Ad), TransferOperator(Ad, transpose=True)
else:        # opaque torch callables: mat-vec = user code, every other stage a library call
    op, opT = krylov.TorchLinearOperator((n, n), fr, dev), krylov.TorchLinearOperator((n, n), fl, dev)
print("operand form:", "native TransferOperator" if NATIVE else "torch callable")
eig.setDominantSparseEig(op, opT, hook)
tv1 = torch.randn(n, dtype=torch.float64, device=dev); tv2 = torch.randn(n, dtype=torch.float64, device=dev)
for it in range(3):
    count[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lam, l, r = eig.DominantSparseEig.apply(A, k)
    torch.cuda.synchronize(); t1 = time.perf_counter(); c1 = count[0]
    loss = lam.sum() + (l * tv1).sum() * (r * tv2).sum()   # b != 0 in both adjoint solves
    (gA,) = torch.autograd.grad(loss, A)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("D=%d n=%d k=%d  forward %.1f ms (%d mat-vecs)  backward %.1f ms (%d mat-vecs)  lambda=%.12f" % (
        D, n, k, (t1 - t0) * 1e3, c1, (t2 - t1) * 1e3, count[0] - c1, lam.item()))
res = float((fr(r.detach()) - lam.detach() * r.detach()).norm())
print("eigen-residual ||G r - lambda r|| = %.2e" % res)

if D <= 128:   # the same problem through the host branch = the reference's arithmetic (scipy ARPACK + gmres)
    import numpy as np
    from scipy.sparse.linalg import LinearOperator
    An = Ad.cpu().numpy()
    Gr = LinearOperator((n, n), matvec=lambda v: np.einsum("kij,kmn,jn->im", An, An, v.reshape(D, D), optimize="greedy").reshape(-1))
    Gl = LinearOperator((n, n), matvec=lambda v: np.einsum("kij,kmn,im->jn", An, An, v.reshape(D, D), optimize="greedy").reshape(-1))
    def hook_np(pieces):
        gA = np.zeros_like(An)
        for u, v in pieces:
            um, vm = u.reshape(D, D), v.reshape(D, D)
            gA = gA + np.einsum("im,jn,kmn->kij", um, vm, An, optimize="greedy") + np.einsum("mi,nj,kmn->kij", um, vm, An, optimize="greedy")
        return torch.from_numpy(gA)
    Ac = torch.from_numpy(An).requires_grad_(True)
    eig.setDominantSparseEig(Gr, Gl, hook_np)
    t0 = time.perf_counter()
    lam_c, l_c, r_c = eig.DominantSparseEig.apply(Ac, k)
    tf = time.perf_counter()
    s = 1.0 if float(r_c @ r.detach().cpu()) > 0 else -1.0
    loss = lam_c.sum() + (l_c * s * tv1.cpu()).sum() * (r_c * s * tv2.cpu()).sum()
    (gc,) = torch.autograd.grad(loss, Ac)
    tb = time.perf_counter()
    print("host branch (scipy, %d threads): forward %.2f s  backward %.2f s ; lambda diff %.1e ; grad rel diff %.1e" % (
        torch.get_num_threads(), tf - t0, tb - tf, abs(lam_c.item() - lam.item()), float((gc - gA.cpu()).abs().max() / gc.abs().max())))
