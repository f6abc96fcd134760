"""1-D transverse-field Ising chain, H = -sum_i (g sx_i + sz_i sz_{i+1}), periodic -- model class with the
attribute names of reference examples/TFIM/TFIM.py (N, dim, device, g, H, pHpg, Hadjoint_to_gadjoint,
setHmatrix, setpHpg, Hmatrix, pHpgmatrix), so the driver scripts here read like the reference's.

On a CUDA device the matrix-free operator is the native HIP kernel of
``dominantsparseeigenad_amd.operators.TFIMOperator`` (index arithmetic, no tables: the reference builds an
(n, L) int64 gather table, 168 MB at L = 20).  On the CPU the same index arithmetic is evaluated with torch ops.
"""
import numpy as np
import torch


class TFIM(object):
    def __init__(self, N, device=torch.device("cpu")):
        self.N = int(N)
        self.dim = 1 << self.N
        self.device = torch.device(device)
        self._g = None
        self._op = None
        if self.device.type == "cuda":
            from dominantsparseeigenad_amd.operators import TFIMOperator
            self._op = TFIMOperator(self.N, self.device)
        else:
            idx = torch.arange(self.dim, dtype=torch.int64)
            rot = ((idx << 1) | (idx >> (self.N - 1))) & (self.dim - 1)
            x = idx ^ rot
            pop = torch.zeros_like(x)
            for b in range(self.N):
                pop += (x >> b) & 1
            self.diag_elements = (-(self.N - 2 * pop)).to(torch.float64)
            self._idx = idx

    # ---- parameter
    @property
    def g(self):
        return self._g

    @g.setter
    def g(self, value):
        self._g = value
        if self._op is not None:
            self._op.g = value

    # ---- matrix-free operators
    def _flip_sum(self, v):
        s = torch.zeros_like(v)
        for j in range(self.N):
            s = s + v[self._idx ^ (1 << j)]
        return s

    def pHpg(self, v):
        """dH/dg v = -sum_i sx_i v"""
        if self._op is not None:
            return self._op.pHpg(v)
        return -self._flip_sum(v)

    def H(self, v):
        if self._op is not None:
            return self._op.H(v)
        return v * self.diag_elements - self.g * self._flip_sum(v)

    def Hadjoint_to_gadjoint(self, v1, v2):
        return self.pHpg(v2).matmul(v1)[None]

    @property
    def _native_methods(self):  # lets setDominantSparseSymeig(model.H, ...) find the native operator
        if self._op is None:
            raise AttributeError("_native_methods")
        return ("H",)

    @property
    def handle(self):
        return self._op.handle

    @property
    def n(self):
        return self.dim

    # ---- dense forms (small N only)
    def _dense(self, diag_scale, g):
        n = self.dim
        idx = torch.arange(n, dtype=torch.int64)
        rot = ((idx << 1) | (idx >> (self.N - 1))) & (n - 1)
        x = idx ^ rot
        pop = torch.zeros_like(x)
        for b in range(self.N):
            pop += (x >> b) & 1
        M = torch.diag((-(self.N - 2 * pop)).to(torch.float64)) * diag_scale
        M = M.to(self.device)
        off = torch.zeros(n, n, dtype=torch.float64, device=self.device)
        for j in range(self.N):
            off[(idx ^ (1 << j)).to(self.device), idx.to(self.device)] = 1.0
        return M - g * off

    def setHmatrix(self):
        """Dense Hamiltonian (differentiable in g), plus the reference's 1e-12 symmetric noise that keeps
        torch's own second derivative through eigh finite (reference TFIM.py:82-89)."""
        noise = 1e-12 * torch.randn(self.dim, self.dim, dtype=torch.float64, device=self.device)
        self.Hmatrix = self._dense(1.0, self.g) + 0.5 * (noise + noise.T)

    def setpHpg(self):
        self.pHpgmatrix = self._dense(0.0, 1.0)
