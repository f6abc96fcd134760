"""Pin the CPU oracle against golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only.  Tolerances: the oracle restates the
same torch-CPU fp64 ops, so agreement is expected at rounding level; 1e-12 is
asserted (1e-9 for second derivatives, whose own seed-to-seed noise in the
reference is ~1e-9, SURVEY.md section 0 hazard 2)."""
import numpy as np
import pytest
import torch

import oracle
from oracle.operators import tfim_diag_closed_form, tfim_analytic_E0
from oracle.adjoint import make_dense_dominant_symeig
from helpers import SeedDraws, sym_from_seed, unit, signed_close, rel
from dominantsparseeigenad_amd.synthetic import normal_vector

TOL = 1e-12


@pytest.mark.parametrize("tag", ["n256_k32", "n256_k256"])
def test_dense_symeig_c1(golden, tag):
    gd = golden("dense_symeig_" + tag)
    n, k = int(gd["n"]), int(gd["k"])
    A = sym_from_seed(n, int(gd["seed_A"]))
    t = unit(n, int(gd["seed_t"]))
    Q, alphas, betas = oracle.lanczos_tridiag(A, k, draw=SeedDraws(int(gd["seed_draw"])))
    assert rel(alphas, gd["alphas"]) < TOL
    assert rel(betas, gd["betas"]) < TOL
    assert rel(Q[:, :8], gd["Q_first8"]) < 1e-11
    Ag = A.clone().requires_grad_(True)
    draws = SeedDraws(int(gd["seed_draw"]))
    lam, psi = make_dense_dominant_symeig(draw=draws).apply(Ag, k)
    ok, err, sgn = signed_close(psi.detach(), gd["psi"], 1e-11)
    assert ok, err
    loss = lam + psi.matmul(t * sgn)
    (gA,) = torch.autograd.grad(loss, Ag)
    assert draws.count == int(gd["ndraw"])
    assert abs(lam.item() - float(gd["lam"])) < TOL * abs(float(gd["lam"]))
    assert abs(loss.item() - float(gd["loss"])) < 1e-11
    assert rel(gA[0], gd["gradA_row0"] * 1.0) < 1e-9 or rel(gA[0], gd["gradA_row0"]) < 1e-9
    assert abs(gA.norm().item() - float(gd["gradA_fro"])) < 1e-9 * float(gd["gradA_fro"])


def test_lanczos_minmax(golden):
    gd = golden("lanczos_minmax")
    n, k = int(gd["n"]), int(gd["k"])
    R = torch.from_numpy((np.abs(normal_vector(n * n, int(gd["seed_A"]))) % 1.0).reshape(n, n)) * 0.1
    A = R + R.T
    lo, vlo, hi, vhi = oracle.symeig_lanczos(A, k, draw=SeedDraws(int(gd["seed_draw"])))
    assert abs(lo.item() - float(gd["lo"])) < TOL * abs(float(gd["hi"]))
    assert abs(hi.item() - float(gd["hi"])) < TOL * abs(float(gd["hi"]))
    assert signed_close(vlo, gd["vlo"], 1e-10)[0]
    assert signed_close(vhi, gd["vhi"], 1e-10)[0]


def test_cg_fullrank(golden):
    gd = golden("cg_fullrank")
    A, b, x0 = (torch.from_numpy(gd[key]) for key in ("A", "b", "x0"))
    st = {}
    x = oracle.cg_solve(A, b, x0, stats=st)
    assert rel(x, gd["x"]) < TOL
    assert st["matvecs"] == int(gd["matvecs"])
    assert torch.allclose(A.matmul(x), b)  # the reference test's own assertion, test_CG.py:27


def test_cg_lowrank(golden):
    gd = golden("cg_lowrank")
    n = int(gd["n"])
    S = sym_from_seed(n, int(gd["seed_S"]))
    Ap = S - float(gd["lam"]) * torch.eye(n, dtype=torch.float64)
    b, x0, psi = (torch.from_numpy(gd[key]) for key in ("b", "x0", "psi"))
    st = {}
    x = oracle.cg_solve(Ap, b, x0, stats=st)
    assert rel(x, gd["x"]) < 1e-10
    assert st["matvecs"] == int(gd["matvecs"])
    assert (Ap.matmul(x) - b).abs().max() < 1e-6 and abs(x.matmul(psi).item()) < 1e-6  # test_CG.py:42-47


def test_symeig_potential(golden):
    gd = golden("symeig_potential")
    N, k = int(gd["N"]), int(gd["k"])
    K = sym_from_seed(N, int(gd["seed_K"]))
    target = torch.from_numpy(normal_vector(N, int(gd["seed_target"])))
    potential = torch.from_numpy(normal_vector(N, int(gd["seed_potential"]))).requires_grad_(True)
    H = K + torch.diag(potential)
    draws = SeedDraws(int(gd["seed_draw"]))
    _, psi = make_dense_dominant_symeig(draw=draws).apply(H, k)
    ok, err, sgn = signed_close(psi.detach(), gd["psi"], 1e-10)
    assert ok, err
    loss = 1.0 - psi.matmul(target) * sgn
    (gp,) = torch.autograd.grad(loss, potential)
    assert abs(loss.item() - float(gd["loss"])) < 1e-10
    assert rel(gp, gd["grad"]) < 1e-9
    # and the reference test's own ground truth (full eigensolver AD), test_symeig.py:43-46
    ok2, _, sgn2 = signed_close(psi.detach(), gd["psi_full"], 1e-6)
    assert ok2
    assert np.allclose(gp.numpy() * sgn2, gd["grad_full"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("tag", ["L10_k300_g1.0", "L10_k300_g1.5", "L12_k200_g1.0"])
def test_tfim(golden, tag):
    gd = golden("tfim_" + tag)
    L, k, g = int(gd["L"]), int(gd["k"]), float(gd["g"])
    model = oracle.TFIMTables(L)
    n = model.dim
    assert np.array_equal(model.diag.numpy(), tfim_diag_closed_form(L))
    model.g = torch.tensor([g], dtype=torch.float64, requires_grad=True)
    tvec = unit(n, int(gd["seed_t"]))
    # E0, dE0, d2E0 (examples/TFIM/E0.py:53-67)
    draws = SeedDraws(int(gd["seed_draw_E"]))
    f = oracle.make_sparse_dominant_symeig(model.H, model.adjoint_hook, draw=draws).apply
    E0, psi = f(model.g, k, n)
    (dE0,) = torch.autograd.grad(E0, model.g, create_graph=True)
    (d2E0,) = torch.autograd.grad(dE0, model.g)
    assert draws.count == int(gd["ndraw_E"])
    assert abs(E0.item() - float(gd["E0"])) < TOL * abs(float(gd["E0"]))
    assert signed_close(psi.detach(), gd["psi"], 1e-11)[0]
    assert abs(dE0.item() - float(gd["dE0"][0])) < 1e-11 * abs(float(gd["dE0"][0]))
    assert abs(d2E0.item() - float(gd["d2E0"][0])) < 1e-9 * abs(float(gd["d2E0"][0]))
    # loss = E0 + psi.t
    draws = SeedDraws(int(gd["seed_draw_E"]))
    f = oracle.make_sparse_dominant_symeig(model.H, model.adjoint_hook, draw=draws).apply
    E0, psi = f(model.g, k, n)
    sgn = signed_close(psi.detach(), gd["psi"], 1e-11)[2]
    loss = E0 + psi.matmul(tvec) * sgn
    (gl,) = torch.autograd.grad(loss, model.g)
    assert abs(loss.item() - float(gd["loss"])) < 1e-11
    assert abs(gl.item() - float(gd["dloss"][0])) < 1e-10 * abs(float(gd["dloss"][0]))
    # chi_F (examples/TFIM/chiF.py:40-53)
    draws = SeedDraws(int(gd["seed_draw_E"]))
    f = oracle.make_sparse_dominant_symeig(model.H, model.adjoint_hook, draw=draws).apply
    E0, psi = f(model.g, k, n)
    logF = torch.log(psi.detach().matmul(psi))
    (dlogF,) = torch.autograd.grad(logF, model.g, create_graph=True)
    (d2logF,) = torch.autograd.grad(dlogF, model.g)
    assert abs(-d2logF.item() - float(gd["chiF"][0])) < 1e-8 * abs(float(gd["chiF"][0]))
    # closed form (E0.py:9-23): E0 agrees to ~1e-15 once converged
    if L == 10:
        assert abs(E0.item() - tfim_analytic_E0(L, torch.tensor(g, dtype=torch.float64)).item()) < 1e-10


def test_tfim_reference_curves():
    """The reference's own stored curves examples/TFIM/datas/E0_N_10.npz (copied as data into
    tests/golden/ref_datas/; per-site E0, dE0, d2E0 on 100 g-points, produced by E0_sparseAD,
    E0.py:53-67,111-113).  SURVEY.md section 8c: the stored curves were made with a less-converged
    setting and pin results at ~1e-7 near g=1 and ~1e-15 (E0) at the end points."""
    import os
    from conftest import GOLDEN
    cur = np.load(os.path.join(GOLDEN, "ref_datas", "E0_N_10.npz"))
    L, k = 10, 300
    model = oracle.TFIMTables(L)
    for idx, tolE, told in ((0, 1e-13, 1e-7), (50, 1e-6, 1e-5), (99, 1e-13, 1e-7)):
        g = float(cur["gs"][idx])
        model.g = torch.tensor([g], dtype=torch.float64, requires_grad=True)
        f = oracle.make_sparse_dominant_symeig(model.H, model.adjoint_hook, draw=SeedDraws(4242)).apply
        E0, _ = f(model.g, k, model.dim)
        (dE0,) = torch.autograd.grad(E0, model.g, create_graph=True)
        (d2E0,) = torch.autograd.grad(dE0, model.g)
        assert abs(E0.item() / L - cur["E0s"][idx]) < tolE * abs(cur["E0s"][idx]), (g, E0.item() / L)
        assert abs(dE0.item() / L - cur["dE0s"][idx]) < told * abs(cur["dE0s"][idx])
        assert abs(d2E0.item() / L - cur["d2E0s"][idx]) < 50 * told * abs(cur["d2E0s"][idx])
        # closed form, E0.py:9-23
        assert abs(E0.item() - tfim_analytic_E0(L, torch.tensor(g, dtype=torch.float64)).item()) < 1e-11


def test_schrodinger(golden):
    gd = golden("schrodinger")
    N, k, h = int(gd["N"]), int(gd["k"]), float(gd["h"])
    xmesh = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False))
    potential = (0.5 * xmesh ** 2).clone().requires_grad_(True)
    op = oracle.Stencil3(N, h, potential)
    target = torch.from_numpy(gd["target"])
    draws = SeedDraws(int(gd["seed_draw"]))
    f = oracle.make_sparse_dominant_symeig(op.H, op.adjoint_hook, draw=draws).apply
    E, psi = f(potential, k, N)
    loss = 1.0 - (psi.abs() * target).sum()
    (gp,) = torch.autograd.grad(loss, potential)
    assert draws.count == int(gd["ndraw"])
    assert abs(E.item() - float(gd["E"])) < 1e-11 * abs(float(gd["E"]))
    assert signed_close(psi.detach(), gd["psi"], 1e-9)[0]
    assert abs(loss.item() - float(gd["loss"])) < 1e-9
    assert rel(gp, gd["grad"]) < 1e-6  # CG here runs into the n-iteration cap: noisy


def test_host_branch_of_eig_reproduces_reference_fixtures():
    """DominantEig / DominantSparseEig on HOST operands keep the reference's own third-party calls (ARPACK eigs, scipy
    gmres; eig.py:29-30,54-57): they must reproduce the outputs the reference produced for the seeded D = 5 / D = 10
    transfer matrices (tests/golden/make_golden.py: case_dominant_eig, case_dominant_sparse_eig)."""
    import os
    from scipy.sparse.linalg import LinearOperator
    import dominantsparseeigenad_amd.eig as eig
    from dominantsparseeigenad_amd.synthetic import normal_vector
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

    def gauge(l, r):
        s = 1.0 if r[np.argmax(np.abs(r))] > 0 else -1.0
        return l * s, r * s

    gd = np.load(os.path.join(here, "dominant_eig_D5.npz"))
    D, k = int(gd["D"]), int(gd["k"])
    n = D * D
    Gong = np.einsum("kij,kmn->imjn", gd["A"], gd["A"]).reshape(n, n)
    a = float(normal_vector(1, int(gd["seed_a"]))[0])
    M = torch.from_numpy(normal_vector(n * n, int(gd["seed_M"])).reshape(n, n))
    G = torch.from_numpy(Gong).requires_grad_(True)
    lam, l, r = eig.DominantEig.apply(G, k)
    loss = a * lam + l.matmul(M).matmul(r)
    (gG,) = torch.autograd.grad(loss.sum(), G)
    lg, rg = gauge(l.detach().numpy(), r.detach().numpy())
    assert abs(lam.item() - float(gd["eigval"][0])) < 1e-12 * abs(float(gd["eigval"][0]))
    assert np.max(np.abs(rg - gd["r"])) < 1e-10 and np.max(np.abs(lg - gd["l"])) < 1e-9 * np.max(np.abs(gd["l"]))
    assert np.max(np.abs(gG.numpy() - gd["grad_Gong"])) < 1e-8 * np.max(np.abs(gd["grad_Gong"]))

    gd = np.load(os.path.join(here, "dominant_sparse_eig_D10.npz"))
    D, d, k = int(gd["D"]), int(gd["d"]), int(gd["k"])
    n = D * D
    A0 = gd["A"]
    a = float(normal_vector(1, int(gd["seed_a"]))[0])
    M = torch.from_numpy(normal_vector(n * n, int(gd["seed_M"])).reshape(n, n))
    right = lambda v: sum(A0[s] @ v.reshape(D, D) @ A0[s].T for s in range(d)).reshape(-1)     # noqa: E731
    left = lambda v: sum(A0[s].T @ v.reshape(D, D) @ A0[s] for s in range(d)).reshape(-1)      # noqa: E731

    def hook(pieces):
        gA = np.zeros_like(A0)
        for u, v in pieces:
            U, Vm = u.reshape(D, D), v.reshape(D, D)
            for s in range(d):
                gA[s] += U @ A0[s] @ Vm.T + U.T @ A0[s] @ Vm
        return torch.from_numpy(gA)

    eig.setDominantSparseEig(LinearOperator((n, n), matvec=right), LinearOperator((n, n), matvec=left), hook)
    At = torch.from_numpy(A0).requires_grad_(True)
    lam, l, r = eig.DominantSparseEig.apply(At, k)
    loss = a * lam + l.matmul(M).matmul(r)
    (gA,) = torch.autograd.grad(loss.sum(), At)
    lg, rg = gauge(l.detach().numpy(), r.detach().numpy())
    assert abs(lam.item() - float(gd["eigval"][0])) < 1e-12 * abs(float(gd["eigval"][0]))
    assert np.max(np.abs(rg - gd["r"])) < 1e-10 and np.max(np.abs(lg - gd["l"])) < 1e-9 * np.max(np.abs(gd["l"]))
    assert np.max(np.abs(gA.numpy() - gd["grad_A"])) < 1e-8 * np.max(np.abs(gd["grad_A"]))


def test_tight_eps_headline_fixture_is_consistent_with_the_reference_fixture(golden):
    """tests/golden/tfim_L20_k200_g1.0_eps1e-12.npz (pinned oracle at eps = 1e-12, make_tight_adjoint.py) against the
    REFERENCE's own outputs for the same injected vectors at its hard-coded eps = 1e-7
    (tfim_L20_k200_g1.0.npz, make_golden.py --big): the forward is independent of eps and must agree to rounding; the
    adjoints may differ by what eps = 1e-7 leaves undetermined (~eps/gap) and no more; dE0/dg at
    eps = 1e-12 equals the closed form (E0.py:9-23) to 1e-10."""
    import torch
    from oracle.operators import tfim_analytic_E0
    tight, ref = golden("tfim_L20_k200_g1.0_eps1e-12"), golden("tfim_L20_k200_g1.0")
    assert int(tight["seed_draw"]) == int(ref["seed_draw_E"]) and int(tight["seed_t"]) == int(ref["seed_t"])
    assert abs(float(tight["E0"]) - float(ref["E0"])) < 1e-13 * abs(float(ref["E0"]))
    assert np.max(np.abs(tight["psi_head"] - ref["psi_head"])) < 1e-13
    assert abs(float(tight["psi_dot_t"]) - float(ref["psi_dot_t"])) < 1e-13
    for key in ("dloss", "dE0"):
        dev = abs(float(tight[key]) - float(np.ravel(ref[key])[0])) / abs(float(tight[key]))
        assert dev < 2e-8, (key, dev)
    gt = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
    (dE_an,) = torch.autograd.grad(tfim_analytic_E0(20, gt), gt)
    assert abs(float(tight["dE0"]) - dE_an.item()) < 1e-10 * abs(dE_an.item())
    assert int(tight["cg_iters_loss"]) > 90          # eps = 1e-12 runs longer than the reference's ~90 iterations
