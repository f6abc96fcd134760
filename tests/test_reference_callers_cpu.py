"""Drop-in proof with the REFERENCE'S OWN CALLERS (build container only; skipped where /root/reference is absent, so it
never runs on the GPU box and nothing of the reference travels): the reference's example scripts are imported UNCHANGED
from where they lie, with this repository first on ``sys.path`` -- so their ``import DominantSparseEigenAD.symeig`` /
``from DominantSparseEigenAD.eig import DominantEig`` resolve to this repository's drop-in package -- and their
functions are run against the curves the reference stores next to them.

  examples/TFIM/E0.py:38-67     E0_matrixAD (DominantSymeig on the dense matrix), E0_sparseAD (DominantSparseSymeig on
                                model.H with model.Hadjoint_to_gadjoint): E0, dE0/dg, d2E0/dg2 per site
  examples/TFIM/chiF.py:40-53   chiF_sparseAD: second derivative of log <psi0(g)|psi0(g')>
  examples/TFIM_vumps/general.py:59-108   the VUMPS-style optimisation through DominantSparseEig (host branch = SciPy)

Tolerances are the levels SURVEY.md 8c measured for the stored curves (produced by the reference with a less converged
setting than k = 300 reaches today): E0, dE0, d2E0 1e-9 at g = 0.5 / 1.5 and 1e-7 near the critical point; chi_F 2e-7 / 1e-6;
VUMPS energies 2e-6.
The only shim is ``torch.symeig`` (removed from torch; the example files call it at import-independent places only)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples", "TFIM")),
                                reason="the reference tree is only present in the build container")


def _load(path, name):
    """import a reference example file under a private module name, with the repository FIRST on sys.path"""
    d = os.path.dirname(path)
    old = list(sys.path)
    sys.path[:] = [ROOT, d] + [p for p in old if p not in (ROOT, d)]
    try:
        for shadow in ("TFIM",):                       # both example directories have their own TFIM module/class
            sys.modules.pop(shadow, None)
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.path[:] = old


@pytest.fixture()
def symeig_shim(monkeypatch):
    monkeypatch.setattr(torch, "symeig", lambda A, eigenvectors=False, upper=True:
                        torch.linalg.eigh(A, UPLO="U" if upper else "L"), raising=False)


def _drop_in_is_ours():
    import DominantSparseEigenAD.symeig as symeig
    assert os.path.realpath(symeig.__file__).startswith(os.path.realpath(ROOT)), symeig.__file__


@pytest.mark.parametrize("g,tol", [(0.5, 1e-9), (1.5, 1e-9), (1.0, 1e-7)])
def test_reference_E0_callers_on_the_drop_in(symeig_shim, g, tol):
    E0mod = _load(os.path.join(REF, "examples", "TFIM", "E0.py"), "ref_example_E0")
    _drop_in_is_ours()
    stored = np.load(os.path.join(REF, "examples", "TFIM", "datas", "E0_N_10.npz"))
    idx = int(np.argmin(np.abs(stored["gs"] - g)))
    gval = float(stored["gs"][idx])
    torch.manual_seed(7)
    model = E0mod.TFIM(10, torch.device("cpu"))        # the reference's own model class (gather tables, Hmatrix)
    model.g = torch.Tensor([gval]).to(model.device, dtype=torch.float64)
    model.g.requires_grad_(True)
    sparse = E0mod.E0_sparseAD(model, 300)
    model.setHmatrix()
    dense = E0mod.E0_matrixAD(model, 300)
    analytic = E0mod.E0_analytic(model)
    want = (stored["E0s"][idx], stored["dE0s"][idx], stored["d2E0s"][idx])
    for got in (sparse, dense):
        for a, b in zip(got, want):
            assert abs(a - b) <= tol * max(1.0, abs(b)), (g, got, want)
    # and against the closed form the same file carries (E0.py:9-23)
    assert abs(sparse[0] - analytic[0]) < 1e-12 and abs(sparse[1] - analytic[1]) < 1e-7 and abs(sparse[2] - analytic[2]) < 1e-6


@pytest.mark.parametrize("g,tol", [(0.5, 2e-7), (1.5, 2e-7), (1.0, 1e-6)])
def test_reference_chiF_caller_on_the_drop_in(symeig_shim, g, tol):
    chiF = _load(os.path.join(REF, "examples", "TFIM", "chiF.py"), "ref_example_chiF")
    _drop_in_is_ours()
    stored = np.load(os.path.join(REF, "examples", "TFIM", "datas", "chiF_N_10.npz"))
    idx = int(np.argmin(np.abs(stored["gs"] - g)))
    torch.manual_seed(11)
    model = chiF.TFIM(10, torch.device("cpu"))
    model.g = torch.Tensor([float(stored["gs"][idx])]).to(model.device, dtype=torch.float64)
    model.g.requires_grad_(True)
    E0, psi0, chi = chiF.chiF_sparseAD(model, 300)
    assert abs(chi - stored["chiFs"][idx]) <= tol * abs(stored["chiFs"][idx]), (chi, stored["chiFs"][idx])
    model.setHmatrix()
    assert abs(chiF.chiF_perturbation(model) - chi) <= 1e-6 * abs(chi)          # the file's own full-spectrum formula


def test_reference_vumps_caller_on_the_drop_in():
    """general.py's TFIM module at D = 5, k = 10, g = 1.0: the reference's optimisation loop (LBFGS, strong Wolfe; a third
    of its 60 outer iterations is enough at this size) through ``eig.setDominantSparseEig`` / ``DominantSparseEig.apply``
    and, for one evaluation, through ``DominantEig`` on the explicit transfer matrix -- against the stored energy."""
    gen = _load(os.path.join(REF, "examples", "TFIM_vumps", "general.py"), "ref_example_vumps_general")
    import DominantSparseEigenAD.eig as eig
    assert os.path.realpath(eig.__file__).startswith(os.path.realpath(ROOT))
    stored = np.load(os.path.join(REF, "examples", "TFIM_vumps", "datas", "E0s_general", "g_1.00.npz"))
    assert int(stored["Ds"][0]) == 5
    torch.manual_seed(3)
    model = gen.TFIM(5, 10)
    model.seth(1.0)
    model.setparameters()
    opt = torch.optim.LBFGS(model.parameters(), max_iter=20, tolerance_grad=0.0, tolerance_change=0.0,
                            line_search_fn="strong_wolfe")

    def closure():
        E0 = model.sparse_forward()
        opt.zero_grad()
        E0.backward()
        return E0

    E0 = None
    for _ in range(20):
        E0 = opt.step(closure)
    assert abs(E0.item() - float(stored["E0s"][0])) < 2e-6 * abs(float(stored["E0s"][0])), (E0.item(), stored["E0s"][0])
    # the dense primitive on the same tensor agrees with the sparse one (general.py:44-55 vs :96-108)
    Em, Es = model.matrix_forward(), model.sparse_forward()
    assert abs(Em.item() - Es.item()) < 1e-11
    (gm,) = torch.autograd.grad(Em, model.A)
    (gs,) = torch.autograd.grad(Es, model.A)
    assert float((gm - gs).abs().max()) < 1e-9 * max(1.0, float(gm.abs().max()))


# ------------------------------------------------------------------ the reference's OWN UNIT TESTS on the drop-in
# (round-5 verdict, Next 6).  /root/reference/DominantSparseEigenAD/tests/test_{Lanczos,CG,symeig,gradient}.py are loaded
# by path -- nothing of them is copied or travels -- with this repository first on sys.path, so their
# ``from DominantSparseEigenAD.Lanczos import symeigLanczos`` etc. bind to the drop-in package, and every test function
# in them is called.  Shims: ``torch.symeig`` (removed from torch; the tests use it as their dense reference) and the
# ``.T`` of a 1-D tensor the CG test takes (a deprecation warning turned error by nothing here -- left as is).
# CUDA-gated functions skip here (no GPU in the build container) exactly as they do in the reference's CI.
REF_TESTS = os.path.join(REF, "DominantSparseEigenAD", "tests")
_REF_TEST_FILES = ("test_Lanczos.py", "test_CG.py", "test_symeig.py", "test_gradient.py")


def _ref_test_functions():
    out = []
    for fname in _REF_TEST_FILES:
        path = os.path.join(REF_TESTS, fname)
        if not os.path.exists(path):
            continue
        import ast
        tree = ast.parse(open(path).read())
        out += [(fname, node.name) for node in tree.body if isinstance(node, ast.FunctionDef) and node.name.startswith("test_")]
    return out


@pytest.mark.parametrize("fname,func", _ref_test_functions() or [("-", "-")])
def test_reference_unit_tests_on_the_drop_in(symeig_shim, fname, func):
    if fname == "-":
        pytest.skip("the reference's tests are only present in the build container")
    mod = _load(os.path.join(REF_TESTS, fname), "ref_unit_" + fname[:-3])
    bound = [getattr(mod, name) for name in ("symeigLanczos", "CG_torch", "DominantSymeig", "DominantEig") if hasattr(mod, name)]
    assert bound, "the reference test imports none of the primitives?"
    for obj in bound:                                   # what the reference's test calls IS this repository's code
        src = sys.modules[obj.__module__].__file__
        assert os.path.realpath(src).startswith(os.path.realpath(ROOT)), (obj, src)
    fn = getattr(mod, func)
    for mark in getattr(fn, "pytestmark", []):
        if mark.name == "skipif" and mark.args and mark.args[0]:
            pytest.skip("reference skipif: " + str(mark.kwargs.get("reason", "")))
    torch.manual_seed(2026)
    np.random.seed(2026)
    fn()


# ------------------------------------------------------------------ examples/schrodinger1D.py and TFIM_vumps/symmetric.py
@pytest.fixture()
def pyplot_stub(monkeypatch):
    """schrodinger1D.py imports matplotlib.pyplot at the top for its plotting loop (not run here); give it a stand-in when
    matplotlib is not installed"""
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:       # noqa: BLE001
        import types
        from unittest import mock
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = mock.MagicMock()
        monkeypatch.setitem(sys.modules, "matplotlib", mpl)
        monkeypatch.setitem(sys.modules, "matplotlib.pyplot", mpl.pyplot)


def test_reference_schrodinger1d_compute_functions_on_the_drop_in(symeig_shim, pyplot_stub):
    """examples/schrodinger1D.py:36-71: the model's three forward computations at the script's own setting (N = 300, k = 300,
    triangle target) -- full diagonalisation with torch, DominantSymeig on the dense matrix, DominantSparseSymeig on
    ``Hsparse`` with ``Hadjoint_to_padjoint`` -- agree in the loss and in the gradient w.r.t. the potential; then two LBFGS
    steps of the script's optimisation loop through the sparse primitive lower the loss."""
    mod = _load(os.path.join(REF, "examples", "schrodinger1D.py"), "ref_example_schrodinger1D")
    _drop_in_is_ours()
    xmin, xmax, N, k = -1.0, 1.0, 300, 300
    xm = np.linspace(xmin, xmax, num=N, endpoint=False)
    target = np.zeros(N)
    idx = np.abs(xm) < 0.5
    target[idx] = 1.0 - np.abs(xm[idx])
    target /= np.linalg.norm(target)
    xmesh, target = torch.from_numpy(xm).to(torch.float64), torch.from_numpy(target).to(torch.float64)
    torch.manual_seed(5)
    model = mod.Schrodinger1D(xmin, xmax, N, xmesh)
    res = {}
    for name, fwd in (("torch", lambda: model.forward_torch(target)), ("matrix", lambda: model.forward_matrixAD(target, k)),
                      ("sparse", lambda: model.forward_sparseAD(target, k))):
        model.potential.grad = None
        loss = fwd()
        loss.backward()
        res[name] = (loss.item(), model.potential.grad.clone())
    for name in ("matrix", "sparse"):
        assert abs(res[name][0] - res["torch"][0]) < 1e-9, (name, res[name][0], res["torch"][0])
        # the adjoint solve of this operator runs into CG's n-iteration cap: gradients agree to ~1e-5
        gerr = float((res[name][1] - res["torch"][1]).abs().max()) / float(res["torch"][1].abs().max())
        assert gerr < 1e-4, (name, gerr)
    opt = torch.optim.LBFGS(model.parameters(), max_iter=10, tolerance_change=1e-7, tolerance_grad=1e-7, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = model.forward_sparseAD(target, k)
        loss.backward()
        return loss

    first = opt.step(closure).item()
    second = opt.step(closure).item()
    last = model.forward_sparseAD(target, k).item()
    assert last < second <= first and last < 0.5 * first, (first, second, last)


def test_reference_vumps_symmetric_caller_on_the_drop_in():
    """examples/TFIM_vumps/symmetric.py:11-49: the symmetric-MPS model's forward (DominantSymeig on -Gong) against the same
    energy from a full diagonalisation of its transfer matrix, gradient included; then its optimisation loop (LBFGS) at
    D = 8 for g = 1.0 against the stored exact energy per site (datas/E0_sum.npz) at the accuracy a D = 8 MPS reaches."""
    mod = _load(os.path.join(REF, "examples", "TFIM_vumps", "symmetric.py"), "ref_example_vumps_symmetric")
    _drop_in_is_ours()
    stored = np.load(os.path.join(REF, "examples", "TFIM_vumps", "datas", "E0_sum.npz"))
    i = int(np.argmin(np.abs(stored["gs"] - 1.0)))
    torch.manual_seed(4)
    D, k = 8, 64
    model = mod.TFIM(D, k)
    model.seth(float(stored["gs"][i]))
    model.setparameters()

    def dense_energy():
        A = 0.5 * (model.A + model.A.permute(0, 2, 1))
        Gong = torch.einsum("kij,kmn->imjn", A, A).reshape(D ** 2, D ** 2)
        lam, U = torch.linalg.eigh(-Gong)
        v = U[:, 0].reshape(D, D)
        return torch.einsum("aik,bkj,abcd,cml,dln,im,jn", A, A, model.h, A, A, v, v) / lam[0] ** 2

    E_prim = model()
    (g_prim,) = torch.autograd.grad(E_prim, model.A)
    E_full = dense_energy()
    (g_full,) = torch.autograd.grad(E_full, model.A)
    assert abs(E_prim.item() - E_full.item()) < 1e-10 * abs(E_full.item())
    assert float((g_prim - g_full).abs().max()) < 1e-7 * float(g_full.abs().max())
    opt = torch.optim.LBFGS(model.parameters(), max_iter=10, tolerance_grad=1e-7)

    def closure():
        E0 = model()
        opt.zero_grad()
        E0.backward()
        return E0

    E0 = None
    for _ in range(25):
        E0 = opt.step(closure)
    assert abs(E0.item() - float(stored["E0s"][i])) < 2e-4 * abs(float(stored["E0s"][i])), (E0.item(), float(stored["E0s"][i]))
