"""Hamiltonian engineering with an EXPLICIT sparse matrix whose stored non-zeros are the parameters: the 1-D problem of
examples/schrodinger1D.py (reference examples/schrodinger1D.py:101-127), but the Hamiltonian is a CSR matrix -- the
potential on the diagonal AND the hopping amplitudes on the two off-diagonals are learned, so that the ground state
matches a target wave function.

What it shows of the drop-in (README "The matrix as a parameter"):
  * ``vals`` is a leaf tensor on the device; the adjoint A-bar = v1 v2^T of reference symeig.py:82-84 reaches it through
    the sampled outer product ``op.Aadjoint_to_valsadjoint_symmetric`` (include/dsea.h: dsea_op_sddmm) -- the symmetric
    form because the pairs (i, j), (j, i) of a symmetric operand move together;
  * the optimiser steps ``vals`` in place; the operator follows through the tensor's version counter
    (dsea_op_update_vals: the SELL copy is rewritten in place, nothing is rebuilt).

    python examples/sparse_matrix_parameters.py [--N 300] [--k 300] [--iters 10]          (needs the GPU)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import DominantSparseEigenAD.symeig as symeig  # noqa: E402


def tridiagonal_pattern(N):
    """CSR pattern of a tridiagonal N x N matrix (sorted columns) and, per stored element, its offset from the diagonal"""
    rows = np.repeat(np.arange(N), 3)
    cols = rows + np.tile(np.array([-1, 0, 1]), N)
    keep = (cols >= 0) & (cols < N)
    rows, cols = rows[keep], cols[keep]
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=N), out=rowptr[1:])
    return rowptr, cols.astype(np.int32), cols - rows


class SparseHamiltonian(torch.nn.Module):
    """H = kinetic stencil + potential, stored as CSR; every stored element is a parameter"""

    def __init__(self, xmin, xmax, N, xmesh):
        super().__init__()
        from dominantsparseeigenad_amd.operators import CSROperator
        self.N = N
        h = (xmax - xmin) / N
        rowptr, cols, offs = tridiagonal_pattern(N)
        x_of = np.repeat(xmesh.cpu().numpy(), np.diff(rowptr))
        init = np.where(offs == 0, 1.0 / h ** 2 + 0.5 * x_of ** 2, -0.5 / h ** 2)
        dev = xmesh.device
        self.vals = torch.nn.Parameter(torch.from_numpy(init).to(dev))
        self.op = CSROperator(torch.from_numpy(rowptr).to(dev), torch.from_numpy(cols).to(dev), self.vals, N)

    def forward_sparseAD(self, target, k):
        symeig.setDominantSparseSymeig(self.op, self.op.Aadjoint_to_valsadjoint_symmetric)
        self.E0, self.psi0 = symeig.DominantSparseSymeig.apply(self.vals, k, self.N, self.vals.device)
        return 1.0 - (self.psi0.abs() * target).sum()

    def dense(self, vals=None):
        """the symmetrised dense matrix of ``vals`` (the cross-check of tests/test_gpu_csr_param.py goes through torch.linalg.eigh)"""
        vals = self.vals if vals is None else vals
        rows = torch.repeat_interleave(torch.arange(self.N, device=vals.device), self.op.rowptr[1:] - self.op.rowptr[:-1])
        A = torch.zeros((self.N, self.N), dtype=vals.dtype, device=vals.device).index_put((rows, self.op.colidx.long()), vals)
        return 0.5 * (A + A.T)


def target_wavefunction(xm):
    tgt = np.zeros(len(xm))
    idx = np.abs(xm) < 0.5
    tgt[idx] = 1.0 - np.abs(xm[idx])
    return tgt / np.linalg.norm(tgt)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=300)
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args(argv)
    dev = torch.device("cuda")
    xmin, xmax, N = -1.0, 1.0, args.N
    xm = np.linspace(xmin, xmax, num=N, endpoint=False)
    xmesh = torch.from_numpy(xm).to(dev)
    target = torch.from_numpy(target_wavefunction(xm)).to(dev)
    model = SparseHamiltonian(xmin, xmax, N, xmesh)
    opt = torch.optim.LBFGS(model.parameters(), max_iter=10, tolerance_change=1e-7, tolerance_grad=1e-7, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = model.forward_sparseAD(target, args.k)
        loss.backward()
        return loss

    losses = []
    for it in range(args.iters):
        loss = opt.step(closure)
        losses.append(loss.item())
        print(it, loss.item())
    return model, losses


if __name__ == "__main__":
    main()
