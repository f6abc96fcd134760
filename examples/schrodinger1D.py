"""Hamiltonian engineering of a 1-D Schroedinger problem: optimise the potential V(x) so that the ground state
matches a target wave function (counterpart of reference examples/schrodinger1D.py).  The forward pass is
DominantSparseSymeig on the matrix-free stencil operator (native HIP kernel on a CUDA device), the backward pass
the projected CG adjoint; LBFGS drives the loop as in the reference (:101-127).

    python examples/schrodinger1D.py [--N 300] [--k 300] [--iters 10] [--device cuda]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import DominantSparseEigenAD.symeig as symeig  # noqa: E402


class Schrodinger1D(torch.nn.Module):
    def __init__(self, xmin, xmax, N, xmesh):
        super().__init__()
        self.xmesh, self.N = xmesh, N
        self.h = (xmax - xmin) / N
        self.potential = torch.nn.Parameter(0.5 * xmesh ** 2)
        self._op = None
        if xmesh.is_cuda:
            from dominantsparseeigenad_amd.operators import Stencil3Operator
            self._op = Stencil3Operator(N, self.h, self.potential)

    def Hsparse(self, v):
        if self._op is not None:
            return self._op.H(v)
        zero = torch.zeros(1, dtype=v.dtype, device=v.device)
        return -0.5 / self.h ** 2 * (-2 * v + torch.cat((v[1:], zero)) + torch.cat((zero, v[:-1]))) + self.potential * v

    @staticmethod
    def Hadjoint_to_padjoint(v1, v2):
        return v1 * v2

    def forward_sparseAD(self, target, k):
        A = self._op if self._op is not None else self.Hsparse
        symeig.setDominantSparseSymeig(A, self.Hadjoint_to_padjoint)
        _, self.psi0 = symeig.DominantSparseSymeig.apply(self.potential, k, self.N, self.potential.device)
        return 1.0 - (self.psi0.abs() * target).sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=300)
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args()
    dev = torch.device(args.device)
    xmin, xmax, N = -1.0, 1.0, args.N
    xm = np.linspace(xmin, xmax, num=N, endpoint=False)
    tgt = np.zeros(N)
    idx = np.abs(xm) < 0.5
    tgt[idx] = 1.0 - np.abs(xm[idx])
    tgt /= np.linalg.norm(tgt)
    xmesh = torch.from_numpy(xm).to(dev)
    target = torch.from_numpy(tgt).to(dev)
    model = Schrodinger1D(xmin, xmax, N, xmesh)
    opt = torch.optim.LBFGS(model.parameters(), max_iter=10, tolerance_change=1e-7, tolerance_grad=1e-7,
                            line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        t0 = time.time()
        loss = model.forward_sparseAD(target, args.k)
        t1 = time.time()
        loss.backward()
        print("forward %.3f s  backward %.3f s" % (t1 - t0, time.time() - t1))
        return loss

    losses = []
    for it in range(args.iters):
        loss = opt.step(closure)
        losses.append(loss.item())
        print(it, loss.item())
    return losses


if __name__ == "__main__":
    main()
