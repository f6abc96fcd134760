#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REAL REFERENCE.

Run in the build container only (the reference lives at /root/reference and does
not travel to the GPU box):

    python tests/golden/make_golden.py            # small cases   (~1 min)
    python tests/golden/make_golden.py --big      # + TFIM L=16 / L=20 scalars (~3 min, 6 GB RSS)

What it does
  * imports ``DominantSparseEigenAD`` from /root/reference read-only, with the two
    in-process compatibility shims SURVEY.md section 8c describes (no reference file
    is modified or copied):
        torch.symeig       -> torch.linalg.eigh      (removed from torch >= 1.13)
        scipy gmres(tol=)  -> gmres(rtol=)           (scipy >= 1.14)
  * pins every ``torch.randn`` draw the reference makes (Lanczos.py:52,59; CG.py:58,121)
    to index-keyed synthetic vectors (dominantsparseeigenad_amd.synthetic.normal_vector)
    whose seeds are stored in the fixture, so tests can regenerate the inputs
  * stores only data: seeds / small inputs and the reference's outputs.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from dominantsparseeigenad_amd.synthetic import normal_vector  # noqa: E402

# ---------------------------------------------------------------- shims
torch.symeig = lambda A, eigenvectors=False, upper=True: torch.linalg.eigh(A, UPLO="U" if upper else "L")
import scipy.sparse.linalg as _ssl  # noqa: E402

_gmres_orig = _ssl.gmres


def _gmres_compat(A, b, *args, tol=None, **kw):
    if tol is not None and "rtol" not in kw:
        kw["rtol"] = tol
    return _gmres_orig(A, b, *args, **kw)


_ssl.gmres = _gmres_compat
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "examples", "TFIM"))
import DominantSparseEigenAD.Lanczos as ref_lanczos  # noqa: E402
import DominantSparseEigenAD.CG as ref_cg  # noqa: E402
import DominantSparseEigenAD.symeig as ref_symeig  # noqa: E402

_randn_orig = torch.randn


class PinnedDraws:
    """Replace torch.randn by a queue of synthetic vectors: draw #c uses seed base+c."""

    def __init__(self, base_seed):
        self.base = int(base_seed)
        self.count = 0

    def __call__(self, *size, dtype=None, device=None, **kw):
        n = size[0] if len(size) == 1 and isinstance(size[0], int) else None
        if n is None:
            raise RuntimeError("unexpected randn shape %r in pinned run" % (size,))
        v = torch.from_numpy(normal_vector(n, self.base + self.count)).to(dtype or torch.float64)
        self.count += 1
        return v

    def __enter__(self):
        torch.randn = self
        return self

    def __exit__(self, *exc):
        torch.randn = _randn_orig


def sym_from_seed(n, seed, scale=1.0):
    M = torch.from_numpy(normal_vector(n * n, seed).reshape(n, n)) * scale
    return M + M.T


def save(name, **arrays):
    out = {}
    for key, val in arrays.items():
        if isinstance(val, torch.Tensor):
            val = val.detach().cpu().numpy()
        out[key] = np.asarray(val)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-36s %8.1f kB" % (name + ".npz", os.path.getsize(path) / 1e3))


# ---------------------------------------------------------------- cases
def case_dense_symeig(n, k, tag):
    """config C1: DominantSymeig on a dense symmetric matrix (symeig.py:15-31)."""
    A = sym_from_seed(n, 7001)
    t = torch.from_numpy(normal_vector(n, 7002))
    t = t / t.norm()
    with PinnedDraws(7100):
        Q, T = ref_lanczos.Lanczos(A, k)
    alphas, betas = torch.diagonal(T).clone(), torch.diagonal(T, 1).clone()
    Ag = A.clone().requires_grad_(True)
    with PinnedDraws(7100) as d:
        lam, psi = ref_symeig.DominantSymeig.apply(Ag, k)
        loss = lam + psi.matmul(t)
        (gA,) = torch.autograd.grad(loss, Ag)
        ndraw = d.count
    save(
        "dense_symeig_" + tag,
        n=n, k=k, seed_A=7001, seed_t=7002, seed_draw=7100, ndraw=ndraw,
        alphas=alphas, betas=betas, Q_first8=Q[:, :8], lam=lam, psi=psi, loss=loss,
        gradA_psi=gA.matmul(psi.detach()), gradA_fro=gA.norm(),
        gradA_row0=gA[0], gradA_col0=gA[:, 0],
    )


def case_lanczos_minmax(n, k):
    """test_Lanczos.py:6-31 semantics (A = 0.1*rand, A += A^T, min & max pair), pinned."""
    R = torch.from_numpy((np.abs(normal_vector(n * n, 7201)) % 1.0).reshape(n, n)) * 0.1
    A = R + R.T
    with PinnedDraws(7210):
        lo, vlo, hi, vhi = ref_lanczos.symeigLanczos(A, k)
    save("lanczos_minmax", n=n, k=k, seed_A=7201, seed_draw=7210, lo=lo, vlo=vlo, hi=hi, vhi=vhi)


def case_cg():
    """test_CG.py:5-47 semantics, pinned inputs."""
    # full rank SPD, n = 100
    n = 100
    G = torch.from_numpy(normal_vector(n * n, 7301).reshape(n, n))
    U, _ = torch.linalg.qr(G)
    diag = 1.0 + 10.0 * torch.from_numpy((np.abs(normal_vector(n, 7302)) % 1.0))
    A = U @ torch.diag(diag) @ U.T
    A = 0.5 * (A + A.T)
    b = torch.from_numpy(normal_vector(n, 7303))
    x0 = torch.from_numpy(normal_vector(n, 7304))
    calls = [0]
    Acount = lambda v: (calls.__setitem__(0, calls[0] + 1), A.matmul(v))[1]  # noqa: E731
    x = ref_cg.CG_torch(Acount, b, x0, sparse=True)
    save("cg_fullrank", n=n, A=A, b=b, x0=x0, x=x, matvecs=calls[0])
    # rank n-1, n = 300: A - lambda_min I, b and x0 projected (test_CG.py:29-47)
    n = 300
    S = sym_from_seed(n, 7311)
    w, V = torch.linalg.eigh(S)
    lam, psi = w[0], V[:, 0]
    Ap = S - lam * torch.eye(n, dtype=torch.float64)
    b = torch.from_numpy(normal_vector(n, 7312))
    b = b - psi.matmul(b) * psi
    x0 = torch.from_numpy(normal_vector(n, 7313))
    x0 = x0 - psi.matmul(x0) * psi
    calls = [0]
    Acount = lambda v: (calls.__setitem__(0, calls[0] + 1), Ap.matmul(v))[1]  # noqa: E731
    x = ref_cg.CG_torch(Acount, b, x0, sparse=True)
    save("cg_lowrank", n=n, seed_S=7311, lam=lam, psi=psi, b=b, x0=x0, x=x, matvecs=calls[0])


def case_symeig_potential(N, k):
    """test_symeig.py:5-46 semantics: H = K + diag(potential), gradient wrt potential."""
    K = sym_from_seed(N, 7401)
    target = torch.from_numpy(normal_vector(N, 7402))
    potential = torch.from_numpy(normal_vector(N, 7403)).requires_grad_(True)
    H = K + torch.diag(potential)
    with PinnedDraws(7410) as d:
        _, psi = ref_symeig.DominantSymeig.apply(H, k)
        loss = 1.0 - psi.matmul(target)
        (gp,) = torch.autograd.grad(loss, potential)
        ndraw = d.count
    # the reference test's own ground truth: AD through the full eigensolver
    w, V = torch.linalg.eigh(H)
    loss_full = 1.0 - V[:, 0].matmul(target)
    (gp_full,) = torch.autograd.grad(loss_full, potential)
    save("symeig_potential", N=N, k=k, seed_K=7401, seed_target=7402, seed_potential=7403,
         seed_draw=7410, ndraw=ndraw, psi=psi, loss=loss, grad=gp,
         psi_full=V[:, 0], loss_full=loss_full, grad_full=gp_full)


def _tfim_model(L, g):
    from TFIM import TFIM  # reference examples/TFIM/TFIM.py
    model = TFIM(L)
    model.g = torch.tensor([g], dtype=torch.float64, requires_grad=True)
    return model


def case_tfim(L, k, g, tag, second_order=True, store_psi=True, seed=7500):
    """examples/TFIM/E0.py:53-67 (E0, dE0, d2E0) and chiF.py:40-53 (chi_F), pinned."""
    t0 = time.time()
    model = _tfim_model(L, g)
    n = model.dim
    tvec = torch.from_numpy(normal_vector(n, seed + 1))
    tvec = tvec / tvec.norm()
    ref_symeig.setDominantSparseSymeig(model.H, model.Hadjoint_to_gadjoint)
    f = ref_symeig.DominantSparseSymeig.apply
    out = dict(L=L, k=k, g=g, seed_t=seed + 1)
    # (1) energy and its derivatives
    with PinnedDraws(seed + 10) as d:
        E0, psi = f(model.g, k, n)
        (dE0,) = torch.autograd.grad(E0, model.g, create_graph=second_order)
        out.update(seed_draw_E=seed + 10, E0=E0, dE0=dE0)
        if second_order:
            (d2E0,) = torch.autograd.grad(dE0, model.g)
            out.update(d2E0=d2E0)
        out.update(ndraw_E=d.count)
    # (2) loss = E0 + psi.t  (the benchmark's loss: b != 0 in the adjoint solve)
    with PinnedDraws(seed + 10) as d:
        E0b, psib = f(model.g, k, n)
        loss = E0b + psib.matmul(tvec)
        (gl,) = torch.autograd.grad(loss, model.g)
        out.update(loss=loss, dloss=gl, psi_dot_t=psib.matmul(tvec), ndraw_loss=d.count)
    # (3) fidelity susceptibility
    if second_order:
        with PinnedDraws(seed + 10) as d:
            E0c, psic = f(model.g, k, n)
            logF = torch.log(psic.detach().matmul(psic))
            (dlogF,) = torch.autograd.grad(logF, model.g, create_graph=True)
            (d2logF,) = torch.autograd.grad(dlogF, model.g)
            out.update(chiF=-d2logF, ndraw_chi=d.count)
    if store_psi:
        out.update(psi=psi)
    else:
        out.update(psi_head=psi[:64], psi_norm=psi.norm(), psi_sum=psi.sum())
    save("tfim_" + tag, **out)
    print("   tfim %s took %.1f s" % (tag, time.time() - t0))


def case_schrodinger(N, k):
    """examples/schrodinger1D.py:64-73 semantics (forward_sparseAD + backward), pinned."""
    xmin, xmax = -1.0, 1.0
    xmesh = torch.from_numpy(np.linspace(xmin, xmax, num=N, endpoint=False))
    h = (xmax - xmin) / N
    tgt = np.zeros(N)
    xm = xmesh.numpy()
    idx = np.abs(xm) < 0.5
    tgt[idx] = 1.0 - np.abs(xm[idx])
    tgt /= np.linalg.norm(tgt)
    target = torch.from_numpy(tgt)
    potential = (0.5 * xmesh ** 2).clone().requires_grad_(True)

    def Hsparse(v):  # operator of schrodinger1D.py:18-27 evaluated by the reference's formula
        zero = torch.zeros(1, dtype=torch.float64)
        return -0.5 / h ** 2 * (-2 * v + torch.cat((v[1:], zero)) + torch.cat((zero, v[:-1]))) + potential * v

    ref_symeig.setDominantSparseSymeig(Hsparse, lambda v1, v2: v1 * v2)
    with PinnedDraws(7610) as d:
        E, psi = ref_symeig.DominantSparseSymeig.apply(potential, k, N)
        loss = 1.0 - (psi.abs() * target).sum()
        (gp,) = torch.autograd.grad(loss, potential)
        ndraw = d.count
    save("schrodinger", N=N, k=k, h=h, seed_draw=7610, ndraw=ndraw, target=target, E=E, psi=psi, loss=loss, grad=gp)


def _gauge(l, r):
    """the primitives fix l.r = 1, r.r = 1 (eig.py:36); the sign of r is free: make its largest component positive"""
    sgn = 1.0 if r[np.argmax(np.abs(r))] > 0 else -1.0
    return l * sgn, r * sgn


def case_dominant_eig(D=5, d=2, k=25):
    """reference tests/test_gradient.py:5-22 (the gradcheck case of DominantEig) with seeded inputs: the dominant
    eigen-triple of the D^2 x D^2 transfer matrix of a random rank-3 tensor and the gradient of a gauge-invariant
    loss, computed by reference eig.py:27-62 (ARPACK eigs + scipy gmres on the host)."""
    import DominantSparseEigenAD.eig as ref_eig
    n = D * D
    A = normal_vector(d * D * D, 7701).reshape(d, D, D)
    Gong = np.einsum("kij,kmn->imjn", A, A).reshape(n, n)
    a = float(normal_vector(1, 7702)[0])
    M = normal_vector(n * n, 7703).reshape(n, n)
    G = torch.from_numpy(Gong).requires_grad_(True)
    lam, l, r = ref_eig.DominantEig.apply(G, k)
    loss = a * lam + l.matmul(torch.from_numpy(M)).matmul(r)
    (gG,) = torch.autograd.grad(loss.sum(), G)
    lg, rg = _gauge(l.detach().numpy(), r.detach().numpy())
    save("dominant_eig_D%d" % D, D=D, d=d, k=k, seed_A=7701, seed_a=7702, seed_M=7703, A=A, eigval=lam, l=lg, r=rg,
         loss=loss.sum(), grad_Gong=gG)


def case_dominant_sparse_eig(D=10, d=2, k=50):
    """reference eig.py:64-152 (DominantSparseEig) on the transfer matrix given as scipy LinearOperators, the operand
    form of reference examples/TFIM_vumps/general.py:59-74 (r -> sum_s A_s r A_s^T, l -> sum_s A_s^T l A_s, and the map
    of the adjoint pieces (u, v) to A-bar), with seeded inputs."""
    import DominantSparseEigenAD.eig as ref_eig
    from scipy.sparse.linalg import LinearOperator
    n = D * D
    A0 = normal_vector(d * D * D, 7711).reshape(d, D, D)
    M = normal_vector(n * n, 7713).reshape(n, n)
    a = float(normal_vector(1, 7712)[0])

    def right(v):
        r = v.reshape(D, D)
        return sum(A0[s] @ r @ A0[s].T for s in range(d)).reshape(-1)

    def left(v):
        lm = v.reshape(D, D)
        return sum(A0[s].T @ lm @ A0[s] for s in range(d)).reshape(-1)

    def pieces_to_Abar(pieces):
        # Gong-bar = sum u v^T with Gong[(i,m),(j,n)] = sum_s A[s,i,j] A[s,m,n]
        gA = np.zeros_like(A0)
        for u, v in pieces:
            U, Vm = u.reshape(D, D), v.reshape(D, D)
            for s in range(d):
                gA[s] += U @ A0[s] @ Vm.T + U.T @ A0[s] @ Vm
        return torch.from_numpy(gA)

    ref_eig.setDominantSparseEig(LinearOperator((n, n), matvec=right), LinearOperator((n, n), matvec=left),
                                 pieces_to_Abar)
    At = torch.from_numpy(A0).requires_grad_(True)
    lam, l, r = ref_eig.DominantSparseEig.apply(At, k)
    loss = a * lam + l.matmul(torch.from_numpy(M)).matmul(r)
    (gA,) = torch.autograd.grad(loss.sum(), At)
    lg, rg = _gauge(l.detach().numpy(), r.detach().numpy())
    save("dominant_sparse_eig_D%d" % D, D=D, d=d, k=k, seed_A=7711, seed_a=7712, seed_M=7713, A=A0, eigval=lam, l=lg,
         r=rg, loss=loss.sum(), grad_A=gA)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    todo = {
        "dense32": lambda: case_dense_symeig(256, 32, "n256_k32"),
        "dense256": lambda: case_dense_symeig(256, 256, "n256_k256"),
        "minmax": lambda: case_lanczos_minmax(400, 160),
        "cg": case_cg,
        "potential": lambda: case_symeig_potential(300, 300),
        "tfim10a": lambda: case_tfim(10, 300, 1.0, "L10_k300_g1.0"),
        "tfim10b": lambda: case_tfim(10, 300, 1.5, "L10_k300_g1.5"),
        "tfim12": lambda: case_tfim(12, 200, 1.0, "L12_k200_g1.0"),
        "schrodinger": lambda: case_schrodinger(300, 300),
        "eig": case_dominant_eig,
        "sparse_eig": case_dominant_sparse_eig,
    }
    big = {
        "tfim16": lambda: case_tfim(16, 200, 1.0, "L16_k200_g1.0", second_order=False, store_psi=False, seed=12345),
        "tfim20": lambda: case_tfim(20, 200, 1.0, "L20_k200_g1.0", second_order=False, store_psi=False, seed=12345),
    }
    if args.big:
        todo.update(big)
    for name, fn in todo.items():
        if args.only and name not in args.only.split(","):
            continue
        fn()


if __name__ == "__main__":
    main()
