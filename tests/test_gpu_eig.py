"""Row f-1: the non-symmetric primitives on the device (krylov.py: restarted Arnoldi + GMRES on the HIP
orthogonalisation kernels) against the reference's arithmetic for this path, which is SciPy's ARPACK / gmres
on the host (reference eig.py:29-30,54-57) -- reached here through the package's host branch.
Eigenvectors carry a gauge (sign of r; l.r = 1, r.r = 1 fixed by the primitive, eig.py:36): compared after
sign alignment, gradients through gauge-invariant losses (as reference tests/test_gradient.py does)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dominantsparseeigenad_amd.eig import DominantEig  # noqa: E402
import dominantsparseeigenad_amd.eig as eig  # noqa: E402
from dominantsparseeigenad_amd import krylov  # noqa: E402

F64 = torch.float64
cuda = torch.device("cuda:0")


def _transfer(D, d, seed):
    rng = np.random.RandomState(seed)
    A = rng.randn(d, D, D)
    return A, np.einsum("kij,kmn->imjn", A, A).reshape(D * D, D * D)


def _align(l, r, l_ref, r_ref):
    s = 1.0 if float(torch.dot(r, r_ref)) > 0 else -1.0
    return l * s, r * s


@pytest.mark.parametrize("D,k", [(5, 25), (12, 60)])
def test_dominant_eig_device_matches_scipy_path(D, k):
    _, Gong = _transfer(D, 2, 3)
    n = D * D
    torch.manual_seed(0)
    a = torch.randn(1, dtype=F64)
    Arandom = torch.randn(n, n, dtype=F64)
    Gc = torch.from_numpy(Gong).requires_grad_(True)
    lam_c, l_c, r_c = DominantEig.apply(Gc, k)
    (g_c,) = torch.autograd.grad((a * lam_c + l_c @ Arandom @ r_c).sum(), Gc)
    Gd = torch.from_numpy(Gong).to(cuda).requires_grad_(True)
    lam_d, l_d, r_d = DominantEig.apply(Gd, k)
    assert lam_d.shape == (1,) and lam_d.is_cuda
    (g_d,) = torch.autograd.grad((a.to(cuda) * lam_d + l_d @ Arandom.to(cuda) @ r_d).sum(), Gd)
    assert abs(lam_d.item() - lam_c.item()) < 1e-11 * abs(lam_c.item())
    l_a, r_a = _align(l_d.detach().cpu(), r_d.detach().cpu(), l_c.detach(), r_c.detach())
    assert float((r_a - r_c.detach()).abs().max()) < 1e-9
    assert float((l_a - l_c.detach()).abs().max()) < 1e-8 * float(l_c.detach().abs().max())
    assert abs(float(l_d.detach() @ r_d.detach()) - 1.0) < 1e-12 and abs(float(r_d.detach().norm()) - 1.0) < 1e-12
    assert float((g_d.cpu() - g_c).abs().max()) < 1e-7 * float(g_c.abs().max())
    w = np.linalg.eigvals(Gong)
    assert abs(lam_d.item() - w[np.argmax(np.abs(w))].real) < 1e-10 * abs(lam_d.item())


def test_dominant_sparse_eig_device_operator():
    """VUMPS-shaped operand (reference examples/TFIM_vumps/general.py:57-75): transfer-matrix mat-vec as two
    small GEMMs, adjoint hook on device tensors; agrees with the dense device primitive and with scipy."""
    D, d, k = 16, 2, 60
    An, Gong = _transfer(D, d, 5)
    n = D * D
    A = torch.from_numpy(An).to(cuda).requires_grad_(True)
    Ad = A.detach()

    def fr(v):
        return torch.einsum("kij,kmn,jn->im", Ad, Ad, v.reshape(D, D)).reshape(-1)

    def fl(v):
        return torch.einsum("kij,kmn,im->jn", Ad, Ad, v.reshape(D, D)).reshape(-1)

    def hook(pieces):
        gA = torch.zeros_like(Ad)
        for u, v in pieces:
            um, vm = u.reshape(D, D), v.reshape(D, D)
            gA = gA + torch.einsum("im,jn,kmn->kij", um, vm, Ad) + torch.einsum("mi,nj,kmn->kij", um, vm, Ad)
        return gA

    op, opT = krylov.TorchLinearOperator((n, n), fr, cuda), krylov.TorchLinearOperator((n, n), fl, cuda)
    eig.setDominantSparseEig(op, opT, hook)
    lam, l, r = eig.DominantSparseEig.apply(A, k)
    (gA,) = torch.autograd.grad(lam.sum(), A)
    # dense device primitive through autograd of the einsum that builds the transfer matrix
    A2 = torch.from_numpy(An).to(cuda).requires_grad_(True)
    G2 = torch.einsum("kij,kmn->imjn", A2, A2).reshape(n, n)
    lam2, l2, r2 = DominantEig.apply(G2, k)
    (gA2,) = torch.autograd.grad(lam2.sum(), A2)
    assert abs(lam.item() - lam2.item()) < 1e-11 * abs(lam2.item())
    assert float((gA - gA2).abs().max()) < 1e-8 * float(gA2.abs().max())
    w = np.linalg.eigvals(Gong)
    assert abs(lam.item() - w[np.argmax(np.abs(w))].real) < 1e-10 * abs(lam.item())


def test_gmres_device_matches_scipy():
    from scipy.sparse.linalg import gmres as sgmres
    rng = np.random.RandomState(9)
    n = 300
    M = rng.randn(n, n) / np.sqrt(n) + 2.5 * np.eye(n)
    b = rng.randn(n)
    xs, info = sgmres(M, b, rtol=1e-12, atol=1e-12)
    Md = torch.from_numpy(M).to(cuda)
    x = krylov.gmres(lambda v: Md @ v, torch.from_numpy(b).to(cuda))
    assert info == 0
    assert float((Md @ x - torch.from_numpy(b).to(cuda)).norm()) <= 1.01e-12 * np.linalg.norm(b) + 1e-12
    assert float((x.cpu() - torch.from_numpy(xs)).abs().max()) < 1e-9
