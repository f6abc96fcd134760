"""Variational uniform-MPS optimisation of the infinite 1-D TFIM with a rank-3 tensor A that is symmetric in its two
virtual indices, so that the D^2 x D^2 transfer matrix is symmetric -- counterpart of reference
examples/TFIM_vumps/symmetric.py, the caller of the DENSE primitive:

    E0(A) = <h> / lambda_max^2 ,   (lambda_max, v) = dominant eigenpair of Gong = sum_s A_s (x) A_s
          = DominantSymeig.apply(-Gong, k)                                   (reference :40-48)

On a CUDA device the dense tensor becomes a native operand of libdsea (Lanczos forward and the CG adjoint solve run
inside the library; ``operators.dense_symmetric_operand``); on the CPU it takes the torch plumbing path.

    python examples/TFIM_vumps/symmetric.py [--g 1.0] [--D 20] [--k 100] [--epochs 50] [--device cuda]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from DominantSparseEigenAD.symeig import DominantSymeig  # noqa: E402


class TFIM(torch.nn.Module):
    def __init__(self, D, k, device=torch.device("cpu")):
        super().__init__()
        self.d, self.D, self.k = 2, int(D), int(k)
        self.device = torch.device(device)

    def seth(self, g):
        """nearest-neighbour Hamiltonian h_{ab,cd} of H = -sum (g sx + sz sz)   (reference :21-33)"""
        h = torch.zeros(2, 2, 2, 2, dtype=torch.float64)
        h[0, 0, 0, 0] = h[1, 1, 1, 1] = -1.0
        h[0, 1, 0, 1] = h[1, 0, 1, 0] = 1.0
        for idx in ((1, 0, 0, 0), (0, 1, 0, 0), (1, 1, 0, 1), (0, 0, 0, 1),
                    (0, 0, 1, 0), (1, 1, 1, 0), (0, 1, 1, 1), (1, 0, 1, 1)):
            h[idx] = -g / 2
        self.h = h.to(self.device)

    def setparameters(self, initA=None):
        A = torch.randn(self.d, self.D, self.D, dtype=torch.float64) if initA is None else initA
        self.A = torch.nn.Parameter((0.5 * (A + A.permute(0, 2, 1))).to(self.device))     # reference :36-38

    def forward(self):
        D = self.D
        A = 0.5 * (self.A + self.A.permute(0, 2, 1))
        Gong = torch.einsum("kij,kmn->imjn", A, A).reshape(D * D, D * D)
        lam, v = DominantSymeig.apply(-Gong, self.k)          # smallest eigenvalue of -Gong = -lambda_max
        v = v.reshape(D, D)
        # <h> with the environment v (x) v: the seven-tensor contraction of reference :46-47 in pairwise steps
        left = torch.einsum("aik,im->amk", A, v)
        right = torch.einsum("bkj,jn->bkn", A, v)
        upper = torch.einsum("amk,bkn->abmn", left, right)
        lower = torch.einsum("cml,dln->cdmn", A, A)
        return torch.einsum("abcd,abcd", torch.einsum("abmn,cdmn->abcd", upper, lower), self.h) / lam ** 2


def optimise(g, D, k, epochs, device, initA=None, verbose=True):
    model = TFIM(D, k, device)
    model.seth(g)
    model.setparameters(initA)
    opt = torch.optim.LBFGS(model.parameters(), max_iter=10, tolerance_grad=1e-7)     # reference :67

    def closure():
        E0 = model()
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
    ap.add_argument("--D", type=int, default=20)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args()
    torch.manual_seed(42)
    E0, _ = optimise(args.g, args.D, args.k, args.epochs, torch.device(args.device))
    print("g = %.2f  D = %d:  E0 = %.12f" % (args.g, args.D, E0))
