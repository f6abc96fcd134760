"""Partial re-orthogonalisation (``reorth="partial"``, include/dsea.h dsea_ws_set_partial_reorth): an OPTION the reference
lacks (it re-orthogonalises on every step, Lanczos.py:66; SURVEY.md 8 f-4 lists "selective reorth").  Same Krylov
process and stored basis; a one-block kernel advances the omega estimates of q_i . q_k per step and only the steps it
selects run the dots / correction kernels over the basis.  Held here to the reference's schedule on the same inputs:
extreme Ritz pair and gradient to the path's 1e-10, the leading block of T to rounding, the basis semi-orthogonal."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dominantsparseeigenad_amd import _lib, engine  # noqa: E402
from dominantsparseeigenad_amd import Lanczos as LZ  # noqa: E402
from dominantsparseeigenad_amd.Lanczos import symeigLanczos, Lanczos  # noqa: E402
from dominantsparseeigenad_amd.operators import CSROperator, TFIMOperator, Stencil3Operator  # noqa: E402
import dominantsparseeigenad_amd.symeig as symeig  # noqa: E402
from helpers import unit  # noqa: E402

F64 = torch.float64
DELTA = 1e-10        # the option's default threshold (csrc/dsea_internal.h DSEA_PRO_DELTA_DEFAULT)


def dev():
    return torch.device("cuda:0")


def _stencil(n, grad=False):
    xm = torch.from_numpy(np.linspace(-1.0, 1.0, num=n, endpoint=False)).to(dev())
    V = (0.5 * xm ** 2)
    if grad:
        V.requires_grad_(True)
    return Stencil3Operator(n, 2.0 / n, V), V


def _basis(A, k, n, q0, partial):
    old = engine.PARTIAL_REORTH
    engine.PARTIAL_REORTH = 0.0 if partial else None
    try:
        Qk, T = Lanczos(A, k, dev(), sparse=True, dim=n, q0=q0)
    finally:
        engine.PARTIAL_REORTH = old
    return Qk, T, engine.last_reorth_steps


@pytest.mark.parametrize("case", ["tfim-L10-g1.0", "tfim-L12-g1.5", "tfim-L14-g1.0", "tfim-L13-g0.8", "sell-L10", "stencil-1000",
                                  "stencil-300-k280", "stencil-20000"])
def test_partial_reorthogonalisation_matches_the_reference_schedule(case):
    if case.startswith("tfim"):
        L = int(case.split("-")[1][1:])
        gval = float(case.split("-")[2][1:])
        n, k = 1 << L, 200
        op = TFIMOperator(L, dev(), g=torch.tensor([gval], dtype=F64, device=dev()))
    elif case == "sell-L10":
        L, n, k = 10, 1 << 10, 150
        t = TFIMOperator(L, dev(), g=torch.tensor([1.0], dtype=F64, device=dev()))
        dense = torch.stack([t(torch.eye(n, dtype=F64, device=dev())[j]) for j in range(n)]).T.cpu()
        op = CSROperator.from_dense(0.5 * (dense + dense.T), dev(), layout="sell")
    else:
        n = int(case.split("-")[1])
        k = int(case.split("-")[2][1:]) if case.count("-") == 2 else 300
        op, _ = _stencil(n)
    q0 = unit(n, 41).to(dev())
    lo_f, v_f = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
    lo_p, v_p = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0, reorth="partial")
    steps = engine.last_reorth_steps
    assert steps is not None and 0 <= steps <= k - 1
    Qp, Tp, steps2 = _basis(op, k, n, q0, True)
    Qf, Tf, none = _basis(op, k, n, q0, False)
    assert steps2 == steps and none is None
    tn = float(Tf.abs().max())             # ~ ||A||: what rounding errors of a Lanczos run are relative to (2.5e5 for the stencil)
    assert abs(lo_f.item() - lo_p.item()) < 1e-12 * tn
    res_f = float((op(v_f) - lo_f * v_f).norm())
    res_p = float((op(v_p) - lo_p * v_p).norm())
    assert abs(float(v_p.norm()) - 1.0) < 1e-12
    assert res_p < 4.0 * res_f + 1e-11 * tn
    sgn = 1.0 if float(v_f @ v_p) > 0 else -1.0
    if res_f < 1e-9 * tn:                  # a converged pair: the vectors agree to the path's tolerance
        assert float((v_f - sgn * v_p).abs().max()) < 1e-10
    # the basis: semi-orthogonal to the threshold; the leading block of T as in the reference's schedule
    orth = float((Qp.T @ Qp - torch.eye(k, dtype=F64, device=dev())).abs().max())
    lead = float((Tp - Tf)[:12, :12].abs().max())
    print("%s: %d of %d steps re-orthogonalised, ||Q^T Q - I||_max %.1e, |dT[:12,:12]| %.1e, |dE0| %.1e, residual %.1e (full %.1e)"
          % (case, steps, k - 1, orth, lead, abs(lo_f.item() - lo_p.item()), res_p, res_f))
    assert orth < 10 * DELTA
    assert lead < 1e-11 * tn
    ev_p, ev_f = torch.linalg.eigvalsh(Tp), torch.linalg.eigvalsh(Tf)
    assert abs(float(ev_p[0] - ev_f[0])) < 1e-12 * tn and abs(float(ev_p[-1] - ev_f[-1])) < 1e-12 * tn
    if case.startswith("tfim"):
        assert steps < (k - 1) // 2        # the point of the option: most steps are not re-orthogonalised


def test_partial_reorthogonalisation_through_the_primitive_and_thresholds():
    """module-level switch (the apply signature is the reference's): E0, psi . t and dloss/dg agree with the reference's
    schedule at 1e-10; threshold knob: a tiny delta re-orthogonalises (almost) every step, a huge one never."""
    L, k = 12, 200
    n = 1 << L
    op = TFIMOperator(L, dev(), g=torch.tensor([1.0], dtype=F64, device=dev(), requires_grad=True))
    symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)
    tvec = unit(n, 43).to(dev())
    out = {}
    from dominantsparseeigenad_amd import CG
    old_eps = CG.EPS_DEFAULT
    CG.EPS_DEFAULT = 1e-12
    try:
        for mode in ("full", "partial"):
            LZ.REORTH_DEFAULT = mode
            try:
                torch.manual_seed(5)
                E, psi = symeig.DominantSparseSymeig.apply(op.g, k, n, dev())
                s = 1.0 if float(psi.detach()[0]) > 0 else -1.0
                loss = E + s * psi.matmul(tvec)
                (gr,) = torch.autograd.grad(loss, op.g)
                out[mode] = (E.item(), s * float(psi.detach() @ tvec), float(gr))
            finally:
                LZ.REORTH_DEFAULT = "full"
    finally:
        CG.EPS_DEFAULT = old_eps
    (Ef, pf, gf), (Ep, pp, gp) = out["full"], out["partial"]
    print("E0 %.3e  psi.t %.3e  dloss/dg %.3e (relative deviations)" % (abs(Ef - Ep) / abs(Ef), abs(pf - pp), abs(gf - gp) / abs(gf)))
    assert abs(Ef - Ep) < 1e-12 * abs(Ef) and abs(pf - pp) < 1e-10 and abs(gf - gp) < 1e-10 * abs(gf)
    assert engine.PARTIAL_REORTH is None and LZ.REORTH_DEFAULT == "full"
    q0 = unit(n, 44).to(dev())
    counts = {}
    for delta in (1e-15, 0.0, 1.5e-8, 1e-4, 1e300):
        engine.PARTIAL_REORTH = delta
        try:
            lo, v = symeigLanczos(op.H, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
        finally:
            engine.PARTIAL_REORTH = None
        counts[delta] = engine.last_reorth_steps
        assert abs(lo.item() - Ef) < 1e-11 * abs(Ef)          # the extreme pair survives even without any re-orthogonalisation
    print("steps re-orthogonalised by threshold:", counts)
    assert counts[1e-15] >= k - 2 and counts[1e300] == 0 and counts[1e-15] >= counts[0.0] >= counts[1.5e-8] >= counts[1e-4] >= 1


@pytest.mark.parametrize("form", ["callable", "native-csr", "dense-primitive", "dense-primitive-fp32"])
def test_partial_reorthogonalisation_through_the_phase_calls(form):
    """operands without a fused Lanczos tail -- a user's Python mat-vec (the reference's calling convention), plain CSR, the
    dense primitive's tensor -- take the option through dsea_lanczos_partial_step: one phase call per step around the
    caller's mat-vec, decisions on the device.  Hard case on purpose: a matrix with a cluster of well-separated low
    eigenvalues, where Ritz values converge early and an un-re-orthogonalised Lanczos run loses orthogonality at once."""
    n, k = 600, 150
    gen = torch.Generator().manual_seed(11)
    U, _ = torch.linalg.qr(torch.randn(n, n, dtype=F64, generator=gen))
    ev = torch.cat([torch.tensor([-50.0, -40.0, -30.0, -20.0], dtype=F64), torch.linspace(0.0, 10.0, n - 4, dtype=F64)])
    A = ((U * ev) @ U.T)
    A = (0.5 * (A + A.T)).to(dev())
    q0 = unit(n, 47).to(dev())
    if form == "callable":
        op, kw = (lambda v: A @ v), dict(sparse=True, dim=n)
    elif form == "native-csr":
        op, kw = CSROperator.from_dense(A.cpu(), dev(), layout="csr"), dict(sparse=True, dim=n)
    elif form == "dense-primitive-fp32":     # Lanczos.py:47: the dense path follows A.dtype (fp64 loops on promoted operands)
        A = A.to(torch.float32)
        op, kw = A, dict()
    else:
        op, kw = A, dict()
    fp32 = form.endswith("fp32")
    lo_f, v_f = symeigLanczos(op, k, dev(), extreme="min", q0=q0, **kw)
    lo_p, v_p = symeigLanczos(op, k, dev(), extreme="min", q0=q0, reorth="partial", **kw)
    steps = engine.last_reorth_steps
    engine.PARTIAL_REORTH = 0.0
    try:
        Qp, Tp = Lanczos(op, k, dev(), q0=q0, **kw)
    finally:
        engine.PARTIAL_REORTH = None
    engine.PARTIAL_REORTH = 1e300           # never re-orthogonalise: what the option protects against
    try:
        Qn, Tn = Lanczos(op, k, dev(), q0=q0, **kw)
    finally:
        engine.PARTIAL_REORTH = None
    eye = torch.eye(k, dtype=F64, device=dev())
    orth, orth_none = float((Qp.T @ Qp - eye).abs().max()), float((Qn.T @ Qn - eye).abs().max())
    sgn = 1.0 if float(v_f @ v_p) > 0 else -1.0
    print("%s: %d of %d steps re-orthogonalised, ||Q^T Q - I||_max %.1e (never re-orthogonalised: %.1e), |dE0| %.1e, max|dpsi| %.1e"
          % (form, steps, k - 1, orth, orth_none, abs(lo_f.item() - lo_p.item()), float((v_f - sgn * v_p).abs().max())))
    assert 1 <= steps < k - 1
    if fp32:                                 # outputs are rounded to the tensor's dtype
        assert lo_p.dtype == torch.float32 and v_p.dtype == torch.float32
        assert abs(lo_f.item() - lo_p.item()) < 1e-5 and abs(lo_p.item() + 50.0) < 1e-3
        assert float((v_f - sgn * v_p).abs().max()) < 1e-5 and orth < 1e-5
        return
    assert abs(lo_f.item() - lo_p.item()) < 1e-12 * 50.0 and abs(lo_p.item() + 50.0) < 1e-10
    assert float((v_f - sgn * v_p).abs().max()) < 1e-10
    assert orth < 10 * DELTA
    assert orth_none > 1e-3                  # the hard case is hard: without the option's passes orthogonality is gone
    # the spectrum of T has no spurious copies of the converged eigenvalues (the signature of lost orthogonality)
    evT = torch.linalg.eigvalsh(Tp)
    assert int((evT < -45.0).sum()) == 1 and int(((evT > -45.0) & (evT < -35.0)).sum()) == 1


def test_partial_reorthogonalisation_says_where_it_does_not_apply():
    n, k = 256, 40
    A = torch.randn(n, n, dtype=F64)
    A = (A + A.T)
    with pytest.raises(NotImplementedError):
        symeigLanczos(lambda v: A @ v, k, torch.device("cpu"), extreme="min", sparse=True, dim=n, reorth="partial")
    with pytest.raises(ValueError):
        symeigLanczos(lambda v: A @ v, k, dev(), extreme="min", sparse=True, dim=n, reorth="sometimes")
    A = A.to(dev())
    old = engine.REORTH_PASSES
    engine.REORTH_PASSES = 2
    try:
        with pytest.raises(NotImplementedError):
            symeigLanczos(lambda v: A @ v, k, dev(), extreme="min", sparse=True, dim=n, reorth="partial")
    finally:
        engine.REORTH_PASSES = old
    lib = _lib.load()
    ws = engine.Workspace.get(n, k, dev())
    assert lib.dsea_ws_set_partial_reorth(ws.handle, 1, -1.0) == _lib.ERR_ARG
    assert lib.dsea_ws_set_partial_reorth(None, 1, 0.0) == _lib.ERR_ARG
    engine.check(lib.dsea_ws_set_partial_reorth(ws.handle, 0, 0.0), "dsea_ws_set_partial_reorth")
    ws.partial_reorth = None
    # ... and the default path is untouched afterwards
    lo, _ = symeigLanczos(lambda v: A @ v, k, dev(), extreme="min", sparse=True, dim=n)
    assert torch.isfinite(lo)


@pytest.mark.parametrize("n,k", [(1, 1), (2, 1), (2, 2), (3, 3), (5, 2), (64, 64), (65, 70), (129, 129), (200, 230)])
def test_partial_reorthogonalisation_edge_sizes_and_breakdown(n, k):
    """degenerate sizes (k = 1: no step at all; k >= n: the Krylov space runs out and the device-side breakdown record
    stops the run -- Lanczos.py:69-70 would divide by beta ~ 0) give what the full schedule gives"""
    import warnings
    op, _ = _stencil(n)
    q0 = unit(n, 51).to(dev())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lo_f, v_f = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
        brk_f = engine.last_break
        lo_p, v_p = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0, reorth="partial")
        brk_p = engine.last_break
    tn = 2.0 / (2.0 / max(n, 2)) ** 2 + 1.0
    assert torch.isfinite(lo_p) and torch.isfinite(v_p).all()
    assert abs(lo_f.item() - lo_p.item()) < 1e-11 * tn
    if k > n:
        assert brk_f > 0 and brk_p > 0 and abs(brk_f - brk_p) <= 1        # both stop where the Krylov space ends
    sgn = 1.0 if float(v_f @ v_p) >= 0 else -1.0
    res_f, res_p = float((op(v_f) - lo_f * v_f).norm()), float((op(v_p) - lo_p * v_p).norm())
    assert res_p < 4.0 * res_f + 1e-10 * tn
    if res_f < 1e-10 * tn:
        assert float((v_f - sgn * v_p).abs().max()) < 1e-9


@pytest.mark.slow
def test_partial_reorthogonalisation_at_the_headline_size():
    """TFIM L = 20, k = 200: same E0 / eigenvector as the reference's schedule, a fraction of its basis traffic"""
    L, k = 20, 200
    n = 1 << L
    op = TFIMOperator(L, dev(), g=torch.tensor([1.0], dtype=F64, device=dev()))
    q0 = unit(n, 45).to(dev())
    lo_f, v_f = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
    lo_p, v_p = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0, reorth="partial")
    steps = engine.last_reorth_steps
    sgn = 1.0 if float(v_f @ v_p) > 0 else -1.0
    print("L=20: %d of %d steps re-orthogonalised, |dE0| %.1e, max|dpsi| %.1e" % (steps, k - 1, abs(lo_f.item() - lo_p.item()),
                                                                                  float((v_f - sgn * v_p).abs().max())))
    assert abs(lo_f.item() - lo_p.item()) < 1e-12 * abs(lo_f.item())
    assert float((v_f - sgn * v_p).abs().max()) < 1e-10
    assert steps < 60
    assert abs(lo_p.item() / L - (-1.2745494843182374)) < 1e-10       # closed form


@pytest.mark.parametrize("tag", ["L16_k200_g1.0", "L20_k200_g1.0"])
def test_partial_reorthogonalisation_against_the_reference_outputs(tag):
    """the option held to what the REFERENCE ITSELF returned for the same injected vectors at full size (tests/golden,
    generated by importing the reference): E0, the head and the sum of psi, the loss and both gradients at the tolerances of
    tests/test_gpu_parity.py::test_headline_sizes_against_reference_scalars"""
    import os
    from helpers import PatchRandn
    gd = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tfim_" + tag + ".npz"))
    L, k, g = int(gd["L"]), int(gd["k"]), float(gd["g"])
    n = 1 << L
    op = TFIMOperator(L, dev())
    op.g = torch.tensor([g], dtype=F64, device=dev(), requires_grad=True)
    symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)
    f = symeig.DominantSparseSymeig.apply
    tvec = unit(n, int(gd["seed_t"])).to(dev())
    LZ.REORTH_DEFAULT = "partial"
    try:
        with PatchRandn(int(gd["seed_draw_E"])):
            E0, psi = f(op.g, k, n, dev())
            steps = engine.last_reorth_steps
            sgn = 1.0 if float(psi.detach()[:64].cpu() @ torch.from_numpy(gd["psi_head"])) > 0 else -1.0
            loss = E0 + psi.matmul(tvec) * sgn
            (gl,) = torch.autograd.grad(loss, op.g)
        with PatchRandn(int(gd["seed_draw_E"])):
            E0b, _ = f(op.g, k, n, dev())
            (dE0,) = torch.autograd.grad(E0b, op.g)
    finally:
        LZ.REORTH_DEFAULT = "full"
    head = psi.detach()[:64].cpu().numpy() * sgn
    print("%s: %d of %d steps re-orthogonalised; vs the reference: E0 %.1e  psi head %.1e  loss %.1e  dloss/dg %.1e  dE0/dg %.1e (relative)"
          % (tag, steps, k - 1, abs(E0.item() - float(gd["E0"])) / abs(float(gd["E0"])),
             np.max(np.abs(head - gd["psi_head"])) / np.max(np.abs(gd["psi_head"])),
             abs(loss.item() - float(gd["loss"])) / abs(float(gd["loss"])),
             abs(gl.item() - float(gd["dloss"][0])) / abs(float(gd["dloss"][0])),
             abs(dE0.item() - float(gd["dE0"][0])) / abs(float(gd["dE0"][0]))))
    assert steps is not None and steps < (k - 1) // 2
    assert abs(E0.item() - float(gd["E0"])) < 1e-10 * abs(float(gd["E0"]))
    assert np.max(np.abs(head - gd["psi_head"])) < 1e-9 * np.max(np.abs(gd["psi_head"]))
    assert abs(float(psi.detach().sum()) * sgn - float(gd["psi_sum"])) < 1e-9 * abs(float(gd["psi_sum"]))
    assert abs(loss.item() - float(gd["loss"])) < 1e-10 * abs(float(gd["loss"]))
    assert abs(gl.item() - float(gd["dloss"][0])) < 2e-8 * abs(float(gd["dloss"][0]))      # CG eps = 1e-7 (CG.py:25)
    assert abs(dE0.item() - float(gd["dE0"][0])) < 2e-8 * abs(float(gd["dE0"][0]))
