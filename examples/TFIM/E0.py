"""Ground-state energy per site of the 1-D TFIM and its first two derivatives in g, four ways
(counterpart of reference examples/TFIM/E0.py):

    E0_analytic   closed form + AD                      (reference E0.py:9-23)
    E0_torchAD    AD through the full eigensolver       (:25-36, torch.linalg.eigh instead of the removed torch.symeig)
    E0_matrixAD   DominantSymeig on the dense matrix    (:38-51)
    E0_sparseAD   DominantSparseSymeig, matrix-free     (:53-67)  <- the MI355X hot path on a CUDA device

    python examples/TFIM/E0.py [--N 10] [--k 300] [--points 11] [--device cuda] [--plot]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from TFIM import TFIM  # noqa: E402


def E0_analytic(model):
    ks = torch.linspace(-(model.N - 1) / 2, (model.N - 1) / 2, steps=model.N, device=model.device,
                        dtype=torch.float64) / model.N * 2 * np.pi
    E0 = -0.5 * (2 * torch.sqrt(model.g ** 2 - 2 * model.g * torch.cos(ks) + 1)).sum()
    dE0, = torch.autograd.grad(E0, model.g, create_graph=True)
    d2E0, = torch.autograd.grad(dE0, model.g)
    return E0.item() / model.N, dE0.item() / model.N, d2E0.item() / model.N


def E0_torchAD(model):
    Es, _ = torch.linalg.eigh(model.Hmatrix)
    E0 = Es[0]
    dE0, = torch.autograd.grad(E0, model.g, create_graph=True)
    d2E0, = torch.autograd.grad(dE0, model.g, retain_graph=True)
    return E0.item() / model.N, dE0.item() / model.N, d2E0.item() / model.N


def E0_matrixAD(model, k):
    from DominantSparseEigenAD.symeig import DominantSymeig
    E0, _ = DominantSymeig.apply(model.Hmatrix, k, model.device)
    dE0, = torch.autograd.grad(E0, model.g, create_graph=True)
    d2E0, = torch.autograd.grad(dE0, model.g)
    return E0.item() / model.N, dE0.item() / model.N, d2E0.item() / model.N


def E0_sparseAD(model, k):
    import DominantSparseEigenAD.symeig as symeig
    symeig.setDominantSparseSymeig(model.H, model.Hadjoint_to_gadjoint)
    E0, _ = symeig.DominantSparseSymeig.apply(model.g, k, model.dim, model.device)
    dE0, = torch.autograd.grad(E0, model.g, create_graph=True)
    d2E0, = torch.autograd.grad(dE0, model.g)
    return E0.item() / model.N, dE0.item() / model.N, d2E0.item() / model.N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=10)
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--points", type=int, default=11)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--dense", action="store_true", help="also run the dense variants (N <= 12)")
    ap.add_argument("--plot", action="store_true")
    ap.add_argument("--reorth", choices=["full", "partial", "twice", "none"], default="full",
                    help="Lanczos re-orthogonalisation of the sparse primitive on the GPU: 'full' = the reference's schedule "
                         "(Lanczos.py:66); the others are options the reference lacks (DESIGN.md section 8)")
    args = ap.parse_args()
    import DominantSparseEigenAD.Lanczos as _LZ
    _LZ.REORTH_DEFAULT = args.reorth
    model = TFIM(args.N, torch.device(args.device))
    gs = np.linspace(0.5, 1.5, num=args.points)
    rows = []
    for gval in gs:
        model.g = torch.tensor([gval], dtype=torch.float64, device=model.device, requires_grad=True)
        row = [gval, *E0_analytic(model), *E0_sparseAD(model, args.k)]
        if args.dense:
            model.setHmatrix()
            row += [*E0_torchAD(model), *E0_matrixAD(model, args.k)]
        rows.append(row)
        print(" ".join("% .12f" % v for v in row))
    if args.plot:
        import matplotlib.pyplot as plt
        rows = np.array(rows)
        for col, name in ((1, "E0/N"), (2, "dE0/dg /N"), (3, "d2E0/dg2 /N")):
            plt.figure()
            plt.plot(rows[:, 0], rows[:, col], label="analytic")
            plt.plot(rows[:, 0], rows[:, col + 3], "o", label="DominantSparseSymeig")
            plt.xlabel("g"), plt.ylabel(name), plt.legend()
        plt.show()
    return rows


if __name__ == "__main__":
    main()
