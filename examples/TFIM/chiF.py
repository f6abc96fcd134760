"""Fidelity susceptibility chi_F(g) of the 1-D TFIM (counterpart of reference examples/TFIM/chiF.py):
    chiF_perturbation  full-spectrum perturbation formula (:11-25)
    chiF_sparseAD      -d^2/dg^2 log <psi0(g0)|psi0(g)> through DominantSparseSymeig (:40-53)

    python examples/TFIM/chiF.py [--N 10] [--k 300] [--points 11] [--device cuda]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from TFIM import TFIM  # noqa: E402


def chiF_perturbation(model):
    Es, psis = torch.linalg.eigh(model.Hmatrix.detach())
    psi0 = psis[:, 0]
    model.setpHpg()
    num = psi0.matmul(model.pHpgmatrix).matmul(psis)[1:] ** 2
    den = (Es[0] - Es[1:]) ** 2
    return (num / den).sum().item()


def chiF_sparseAD(model, k):
    import DominantSparseEigenAD.symeig as symeig
    symeig.setDominantSparseSymeig(model.H, model.Hadjoint_to_gadjoint)
    E0, psi0 = symeig.DominantSparseSymeig.apply(model.g, k, model.dim, model.device)
    logF = torch.log(psi0.detach().matmul(psi0))
    dlogF, = torch.autograd.grad(logF, model.g, create_graph=True)
    d2logF, = torch.autograd.grad(dlogF, model.g)
    return E0, psi0, -d2logF.item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=10)
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--points", type=int, default=11)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--reorth", choices=["full", "partial", "twice", "none"], default="full",
                    help="Lanczos re-orthogonalisation of the sparse primitive on the GPU: 'full' = the reference's schedule "
                         "(Lanczos.py:66); the others are options the reference lacks (DESIGN.md section 8)")
    args = ap.parse_args()
    import DominantSparseEigenAD.Lanczos as _LZ
    _LZ.REORTH_DEFAULT = args.reorth
    model = TFIM(args.N, torch.device(args.device))
    out = []
    for gval in np.linspace(0.5, 1.5, num=args.points):
        model.g = torch.tensor([gval], dtype=torch.float64, device=model.device, requires_grad=True)
        _, _, chi = chiF_sparseAD(model, args.k)
        row = [gval, chi]
        if args.dense:
            model.setHmatrix()
            row.append(chiF_perturbation(model))
        out.append(row)
        print(" ".join("% .10f" % v for v in row))
    return out


if __name__ == "__main__":
    main()
