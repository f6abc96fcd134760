"""Variational uniform-MPS optimisation of the infinite 1-D TFIM with a general real rank-3 tensor A (bond
dimension D) -- counterpart of reference examples/TFIM_vumps/general.py (BASELINE config 4).  The energy per
site needs the dominant eigen-triple (lambda, l, r) of the D^2 x D^2 transfer matrix "Gong":

    matrix_forward   DominantEig on the explicit transfer matrix            (reference :47-57)
    sparse_forward   DominantSparseEig with the transfer matrix as an operator  v -> sum_s A_s v A_s^T
                     (two small GEMMs per application)                        (reference :59-100)

On a CUDA device both run the device Arnoldi / GMRES of ``dominantsparseeigenad_amd.krylov`` (the reference is
host-only here); on the CPU they take the reference's SciPy route through ``LinearOperator``s.

    python examples/TFIM_vumps/general.py [--g 1.0] [--D 5] [--k 10] [--epochs 20] [--device cuda]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from DominantSparseEigenAD.eig import DominantEig  # noqa: E402
import DominantSparseEigenAD.eig as eig  # noqa: E402


class TFIM(torch.nn.Module):
    def __init__(self, D, k, device=torch.device("cpu")):
        super().__init__()
        self.d, self.D, self.k = 2, int(D), int(k)
        self.device = torch.device(device)

    def seth(self, g):
        """nearest-neighbour Hamiltonian h_{ab,cd} of H = -sum (g sx + sz sz)   (reference :20-33)"""
        h = torch.zeros(2, 2, 2, 2, dtype=torch.float64)
        h[0, 0, 0, 0] = h[1, 1, 1, 1] = -1.0
        h[0, 1, 0, 1] = h[1, 0, 1, 0] = 1.0
        for idx in ((1, 0, 0, 0), (0, 1, 0, 0), (1, 1, 0, 1), (0, 0, 0, 1),
                    (0, 0, 1, 0), (1, 1, 1, 0), (0, 1, 1, 1), (1, 0, 1, 1)):
            h[idx] = -g / 2
        self.h = h.to(self.device)

    def setparameters(self, initA=None):
        A = torch.randn(self.d, self.D, self.D, dtype=torch.float64) if initA is None else initA
        self.A = torch.nn.Parameter(A.to(self.device))

    def _energy(self, l, r, lam):
        A = self.A
        upper = torch.einsum("amk,bkn->abmn", torch.einsum("aik,im->amk", A, l), torch.einsum("bkj,jn->bkn", A, r))
        lower = torch.einsum("cml,dln->cdmn", A, A)
        return torch.einsum("abcd,abcd", torch.einsum("abmn,cdmn->abcd", upper, lower), self.h) / lam ** 2

    def matrix_forward(self):
        D = self.D
        Gong = torch.einsum("kij,kmn->imjn", self.A, self.A).reshape(D * D, D * D)
        lam, l, r = DominantEig.apply(Gong, self.k)
        return self._energy(l.reshape(D, D), r.reshape(D, D), lam).reshape(())

    def sparse_forward(self):
        D, dev = self.D, self.device
        Ad = self.A.detach()
        if dev.type == "cuda":
            # the transfer matrix as a NATIVE operand: Arnoldi / GMRES loops and the batched-GEMM mat-vec all
            # inside libdsea (reference :59-66 builds scipy LinearOperators around numpy einsums)
            from dominantsparseeigenad_amd.operators import TransferOperator
            G, GT = TransferOperator(Ad), TransferOperator(Ad, transpose=True)

            def hook(pieces):
                gA = torch.zeros_like(Ad)
                for u, v in pieces:
                    um, vm = u.reshape(D, D), v.reshape(D, D)
                    gA = gA + torch.matmul(torch.matmul(um, Ad), vm.T) + torch.matmul(torch.matmul(um.T, Ad), vm)
                return gA
        else:
            from scipy.sparse.linalg import LinearOperator
            An = Ad.numpy()
            G = LinearOperator((D * D, D * D), matvec=lambda v: np.einsum("kij,kmn,jn->im", An, An, v.reshape(D, D),
                                                                          optimize="greedy").reshape(-1))
            GT = LinearOperator((D * D, D * D), matvec=lambda v: np.einsum("kij,kmn,im->jn", An, An, v.reshape(D, D),
                                                                           optimize="greedy").reshape(-1))

            def hook(pieces):
                gA = np.zeros_like(An)
                for u, v in pieces:
                    um, vm = u.reshape(D, D), v.reshape(D, D)
                    gA = gA + np.einsum("im,jn,kmn->kij", um, vm, An, optimize="greedy") \
                            + np.einsum("mi,nj,kmn->kij", um, vm, An, optimize="greedy")
                return torch.from_numpy(gA)
        eig.setDominantSparseEig(G, GT, hook)
        lam, l, r = eig.DominantSparseEig.apply(self.A, self.k)
        return self._energy(l.reshape(D, D), r.reshape(D, D), lam).reshape(())


def optimise(g, D, k, epochs, device, initA=None, verbose=True):
    model = TFIM(D, k, device)
    model.seth(g)
    model.setparameters(initA)
    opt = torch.optim.LBFGS(model.parameters(), max_iter=20, tolerance_grad=0.0, tolerance_change=0.0,
                            line_search_fn="strong_wolfe")

    def closure():
        E0 = model.sparse_forward()
        opt.zero_grad()
        E0.backward()
        return E0

    E0 = None
    for epoch in range(epochs):
        t0 = time.time()
        E0 = opt.step(closure)
        if verbose:
            print("iter %3d  E0 = %.12f   %.2f s" % (epoch, E0.item(), time.time() - t0))
    return E0.item(), model


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--g", type=float, default=1.0)
    ap.add_argument("--D", type=int, default=5)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args()
    torch.manual_seed(42)
    E0, _ = optimise(args.g, args.D, args.k, args.epochs, torch.device(args.device))
    print("g = %.2f  D = %d:  E0 = %.12f" % (args.g, args.D, E0))
