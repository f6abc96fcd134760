"""Oracle operators (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

CPU restatement of the two user operators that ARE the sparse mat-vec of the
benchmark configurations:

  * TFIMTables  -- reference examples/TFIM/TFIM.py:39-51 (index tables),
                   :58-65 (dH/dg), :91-98 (H), :100-101 (adjoint hook)
  * Stencil3    -- reference examples/schrodinger1D.py:18-34

Both keep the reference's data layout on purpose (an (n, L) int64 gather
table for TFIM), because that is what the reference's CPU path pays for and
the CPU baseline should be timed on the same kind of work.
"""
from __future__ import annotations

import numpy as np
import torch


class TFIMTables:
    """H = -sum_i (g sx_i + sz_i sz_{i+1}), periodic chain of L sites, dimension 2**L."""

    def __init__(self, L, g=None):
        self.L = int(L)
        self.dim = 1 << self.L
        self.g = g
        self.diag = self._build_diag()
        self.flips = self._build_flips()

    def _build_diag(self):
        # TFIM.py:39-46 -- bit j of the basis index is spin j (MSB first in the
        # reference's column order, which is irrelevant for a cyclic sum of
        # neighbour products): diag = -sum_j s_j s_{j+1 mod L}, s = 1 - 2*bit.
        idx = np.arange(self.dim, dtype=np.int64)[:, None]
        bits = (idx >> np.arange(self.L, dtype=np.int64)[::-1]) & 1
        s = 1 - 2 * bits
        s_next = np.concatenate((s[:, 1:], s[:, :1]), axis=1)
        d = -(s * s_next).sum(axis=1)
        return torch.from_numpy(d).to(torch.float64)

    def _build_flips(self):
        # TFIM.py:48-51 -- flips[i, j] = i XOR (1 << j)
        masks = torch.tensor([1 << j for j in range(self.L)], dtype=torch.int64)
        return torch.arange(self.dim, dtype=torch.int64)[:, None] ^ masks

    # TFIM.py:58-65
    def dHdg(self, v):
        return -v[self.flips].sum(dim=1)

    # TFIM.py:91-98
    def H(self, v):
        return v * self.diag - self.g * v[self.flips].sum(dim=1)

    # TFIM.py:100-101 : gbar = v1^T (dH/dg) v2, returned with shape (1,)
    def adjoint_hook(self, v1, v2):
        return self.dHdg(v2).matmul(v1)[None]

    def dense(self):
        """Dense matrix (small L only); TFIM.py:67-89 without the 1e-12 noise term."""
        n = self.dim
        M = torch.diag(self.diag)
        off = torch.zeros(n, n, dtype=torch.float64)
        off[self.flips.T, torch.arange(n)] = 1.0
        return M - self.g * off


def tfim_diag_closed_form(L):
    """diag_i = -(L - 2*popcount(i ^ rotl_L(i,1))): the table-free identity the
    HIP operator uses; the golden test checks it against TFIMTables.diag."""
    n = 1 << L
    i = np.arange(n, dtype=np.uint64)
    rot = ((i << np.uint64(1)) | (i >> np.uint64(L - 1))) & np.uint64(n - 1) if L > 0 else i
    x = i ^ rot
    pop = np.zeros(n, dtype=np.int64)
    for b in range(L):
        pop += ((x >> np.uint64(b)) & np.uint64(1)).astype(np.int64)
    return -(L - 2 * pop).astype(np.float64)


def tfim_analytic_E0(L, g):
    """Closed-form ground-state energy, reference examples/TFIM/E0.py:15-18 (returns total E0)."""
    ks = torch.linspace(-(L - 1) / 2, (L - 1) / 2, steps=L, dtype=torch.float64) / L * 2 * np.pi
    eps_k = 2 * torch.sqrt(g ** 2 - 2 * g * torch.cos(ks) + 1)
    return -0.5 * eps_k.sum()


class Stencil3:
    """H v = -0.5/h^2 (-2 v + v_{+1} + v_{-1}) + V o v with zero (Dirichlet) padding.

    reference examples/schrodinger1D.py:18-27; hook v1 o v2 at :29-34.
    """

    def __init__(self, n, h, potential):
        self.n = int(n)
        self.h = float(h)
        self.potential = potential

    def H(self, v):
        zero = torch.zeros(1, dtype=v.dtype)
        up = torch.cat((v[1:], zero))
        down = torch.cat((zero, v[:-1]))
        return -0.5 / self.h ** 2 * (-2 * v + up + down) + self.potential * v

    @staticmethod
    def adjoint_hook(v1, v2):
        return v1 * v2

    def dense(self):
        n = self.n
        K = -0.5 / self.h ** 2 * (
            torch.diag(-2 * torch.ones(n, dtype=torch.float64))
            + torch.diag(torch.ones(n - 1, dtype=torch.float64), diagonal=1)
            + torch.diag(torch.ones(n - 1, dtype=torch.float64), diagonal=-1)
        )
        return K + torch.diag(self.potential)
