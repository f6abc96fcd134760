"""Kernel-level parity on the MI355X: every C-ABI phase against a plain fp64 torch/numpy evaluation of
the same reference expression, on ragged sizes (odd n, n < one wave tile, n not a multiple of anything)
and every rows-per-lane variant of the basis-streaming kernels.  Tolerance: 1e-13 relative to the
operand norms (sum-order differences only)."""
from ctypes import byref, c_void_p

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dominantsparseeigenad_amd import _lib, engine  # noqa: E402
from dominantsparseeigenad_amd.engine import Workspace, _ptr, _stream  # noqa: E402
from dominantsparseeigenad_amd.synthetic import normal_vector  # noqa: E402

F64 = torch.float64
SIZES = [1, 2, 3, 63, 64, 65, 127, 129, 300, 1000, 4097, 100000, 262144 + 6]


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X (no fallback)"
    return torch.device("cuda:0")


def vec(n, seed):
    return torch.from_numpy(normal_vector(n, seed)).to(dev())


def basis(k, n, seed):
    ldq = engine.round_up(n, 32)
    Q = torch.zeros((k, ldq), dtype=F64, device=dev())
    Q[:, :n] = torch.from_numpy(normal_vector(k * n, seed).reshape(k, n)).to(dev())
    return Q, ldq


@pytest.mark.parametrize("n", SIZES)
def test_dot_axpy_scale_project(n):
    lib = _lib.load()
    ws = Workspace.get(n, 8, dev())
    st = _stream(dev())
    x, y = vec(n, 1), vec(n, 2)
    out = torch.zeros(1, dtype=F64, device=dev())
    _lib.check(lib.dsea_dot(ws.handle, _ptr(x), _ptr(y), n, _ptr(out), st))
    ref = float(np.dot(x.cpu().numpy(), y.cpu().numpy()))
    scale = float(x.norm() * y.norm())
    assert abs(out.item() - ref) <= 1e-13 * scale
    # axpy with device scalar
    a = torch.tensor([0.37], dtype=F64, device=dev())
    y2 = y.clone()
    _lib.check(lib.dsea_axpy(ws.handle, -2.0, _ptr(a), _ptr(x), _ptr(y2), n, st))
    assert torch.allclose(y2, y + (-2.0 * 0.37) * x, rtol=1e-14, atol=1e-14)
    # normalise
    nrm2 = torch.zeros(1, dtype=F64, device=dev())
    q = torch.empty(n, dtype=F64, device=dev())
    beta = torch.zeros(1, dtype=F64, device=dev())
    _lib.check(lib.dsea_nrm2sq(ws.handle, _ptr(x), n, _ptr(nrm2), st))
    _lib.check(lib.dsea_scale_store(ws.handle, _ptr(x), _ptr(nrm2), _ptr(q), _ptr(beta), n, st))
    assert abs(beta.item() - x.norm().item()) <= 1e-13 * x.norm().item()
    assert torch.allclose(q, x / x.norm(), rtol=1e-13, atol=1e-15)
    # projection  v - (a.v) a
    p = engine.project_out(y, q)
    assert torch.allclose(p, y - torch.dot(q, y) * q, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("rpl", [0, 2, 4, 8, 16])
@pytest.mark.parametrize("n,i", [(1, 1), (3, 2), (129, 5), (1000, 37), (4097, 9), (100000, 23), (300000, 6)])
def test_reorth_pair(n, i, rpl):
    """Lanczos.py:61 + :66: r = u - a q1 - b q2 ; c = Q^T r ; r -= Q c ; ||r||^2."""
    lib = _lib.load()
    ws = Workspace.get(n, 64, dev())
    ws.set_rows_per_lane(rpl)
    try:
        st = _stream(dev())
        Q, ldq = basis(i, n, 11)
        u = vec(n, 12)
        ab = torch.tensor([0.7, -1.3], dtype=F64, device=dev())
        r = torch.empty(n, dtype=F64, device=dev())
        c = torch.zeros(i + 1, dtype=F64, device=dev())   # i coefficients + r.r
        beta_ptr = c_void_p(ab.data_ptr() + 8) if i >= 2 else c_void_p(None)
        _lib.check(lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(ab), beta_ptr, _ptr(r),
                                          _ptr(c), st))
        Qn = Q[:, :n]
        r_ref = u - 0.7 * Qn[i - 1] - ((-1.3) * Qn[i - 2] if i >= 2 else 0.0)
        assert torch.equal(r, r_ref)  # same rounding sequence as the torch expression
        c_ref = Qn @ r_ref
        tol = 1e-13 * float(r_ref.norm()) * float(Qn.norm(dim=1).max())
        assert float((c[:i] - c_ref).abs().max()) <= tol
        assert abs(c[i].item() - float(r_ref.dot(r_ref))) <= 1e-13 * float(r_ref.dot(r_ref))
        c = c[:i].clone()
        nrm2 = torch.zeros(1, dtype=F64, device=dev())
        _lib.check(lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), ldq, n, i, _ptr(c), _ptr(r), _ptr(nrm2), st))
        r2_ref = r_ref - Qn.T @ c
        assert float((r - r2_ref).abs().max()) <= 1e-12 * float(r_ref.abs().max() + (Qn.T @ c).abs().max())
        assert abs(nrm2.item() - float(r2_ref.dot(r2_ref))) <= 1e-12 * float(r2_ref.dot(r2_ref))
        # Ritz combination (Lanczos.py:99, one column)
        out = torch.empty(n, dtype=F64, device=dev())
        _lib.check(lib.dsea_ritz_combine(ws.handle, _ptr(Q), ldq, n, i, _ptr(c), _ptr(out), st))
        assert float((out - Qn.T @ c).abs().max()) <= 1e-12 * float((Qn.T @ c).abs().max() + 1e-300)
    finally:
        ws.set_rows_per_lane(0)


def test_run_to_run_determinism():
    """No atomics anywhere: two runs give bit-identical results (the reference is bitwise repeatable)."""
    lib = _lib.load()
    n, i = 100000, 17
    ws = Workspace.get(n, 64, dev())
    st = _stream(dev())
    Q, ldq = basis(i, n, 21)
    u = vec(n, 22)
    a = torch.tensor([0.3], dtype=F64, device=dev())
    outs = []
    for _ in range(2):
        r = torch.empty(n, dtype=F64, device=dev())
        c = torch.zeros(i + 1, dtype=F64, device=dev())
        _lib.check(lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(a), None, _ptr(r), _ptr(c), st))
        outs.append(c.clone())
    assert torch.equal(outs[0], outs[1])


def test_alignment_is_rejected():
    lib = _lib.load()
    n = 1000
    ws = Workspace.get(n, 8, dev())
    x = vec(n + 1, 5)
    out = torch.zeros(1, dtype=F64, device=dev())
    rc = lib.dsea_dot(ws.handle, c_void_p(x.data_ptr() + 8), _ptr(x), n, _ptr(out), _stream(dev()))
    assert rc == -2
    assert lib.dsea_dot(None, _ptr(x), _ptr(x), n, _ptr(out), _stream(dev())) == -1


@pytest.mark.parametrize("n", [1, 2, 5, 300, 4099])
def test_cg_phases_match_reference_sequence(n):
    """The CG phase kernels reproduce CG.py:24-41 step by step on a dense SPD system."""
    lib = _lib.load()
    ws = Workspace.get(n, 8, dev())
    st = _stream(dev())
    G = torch.from_numpy(normal_vector(n * n, 31).reshape(n, n)).to(dev())
    A = G @ G.T / n + torch.eye(n, dtype=F64, device=dev())
    b, x0 = vec(n, 32), vec(n, 33)
    x = engine.cg(b, x0, callable_A=lambda v: A @ v, eps=1e-9)
    assert float((A @ x - b).norm()) < 1e-8
    # oracle iterate-for-iterate: same number of iterations as the host formulation
    import oracle
    stt = {}
    xo = oracle.cg_solve(A.cpu(), b.cpu(), x0.cpu(), eps=1e-9, stats=stt)
    assert engine.last_cg.iters == stt["iters"]
    assert float((x.cpu() - xo).abs().max()) <= 1e-10 * float(xo.abs().max())


@pytest.mark.parametrize("waves", [4, 8, 16])
@pytest.mark.parametrize("n,i", [(1, 1), (129, 5), (1000, 37), (100000, 23), (40000, 201)])
def test_reorth_pair_split_form(n, i, waves):
    """Small-n form of the two passes (a block of W waves per 128-row tile, basis vectors split between the
    waves): same results as the torch expressions, for every W, ragged n, i not a multiple of 4."""
    lib = _lib.load()
    ws = Workspace.get(n, 256, dev())
    ws.set_split(waves)
    try:
        st = _stream(dev())
        Q, ldq = basis(i, n, 61)
        u = vec(n, 62)
        ab = torch.tensor([0.7, -1.3], dtype=F64, device=dev())
        r = torch.empty(n, dtype=F64, device=dev())
        c = torch.zeros(i + 1, dtype=F64, device=dev())
        beta_ptr = c_void_p(ab.data_ptr() + 8) if i >= 2 else c_void_p(None)
        _lib.check(lib.dsea_lanczos_rdots(ws.handle, _ptr(Q), ldq, n, i, _ptr(u), _ptr(ab), beta_ptr, _ptr(r),
                                          _ptr(c), st))
        Qn = Q[:, :n]
        r_ref = u - 0.7 * Qn[i - 1] - ((-1.3) * Qn[i - 2] if i >= 2 else 0.0)
        assert torch.equal(r, r_ref)
        c_ref = Qn @ r_ref
        assert float((c[:i] - c_ref).abs().max()) <= 1e-13 * float(r_ref.norm()) * float(Qn.norm(dim=1).max())
        assert abs(c[i].item() - float(r_ref.dot(r_ref))) <= 1e-13 * float(r_ref.dot(r_ref))
        cc = c[:i].clone()
        nrm2 = torch.zeros(1, dtype=F64, device=dev())
        _lib.check(lib.dsea_lanczos_axpy_norm(ws.handle, _ptr(Q), ldq, n, i, _ptr(cc), _ptr(r), _ptr(nrm2), st))
        r2_ref = r_ref - Qn.T @ cc
        assert float((r - r2_ref).abs().max()) <= 1e-12 * float(r_ref.abs().max() + (Qn.T @ cc).abs().max())
        assert abs(nrm2.item() - float(r2_ref.dot(r2_ref))) <= 1e-12 * float(r2_ref.dot(r2_ref))
        out = torch.empty(n, dtype=F64, device=dev())
        _lib.check(lib.dsea_ritz_combine(ws.handle, _ptr(Q), ldq, n, i, _ptr(cc), _ptr(out), st))
        assert float((out - Qn.T @ cc).abs().max()) <= 1e-12 * float((Qn.T @ cc).abs().max() + 1e-300)
    finally:
        ws.set_split(-1)


@pytest.mark.parametrize("n", [1, 65, 1000, 4097, 100000])
def test_cg_step_around_a_callable_equals_the_phase_calls_and_the_native_streaming_form(monkeypatch, n):
    """dsea_cg_step (one call per iteration around the CALLER'S mat-vec: reference CG.py:31-40 with Amap a Python function)
    against the four phase calls it replaces and against dsea_cg_run's streaming form on the same operator given natively (the
    same update / direction kernels; the d.Ad partials come from different kernels): same iteration count, iterates to 1e-13."""
    import scipy.sparse as sp
    from dominantsparseeigenad_amd.operators import CSROperator
    rng = np.random.RandomState(n)
    M = sp.diags([rng.rand(n) + 3.0] + ([rng.randn(n - 1) * 0.3] * 2 if n > 1 else []), [0, 1, -1][:(3 if n > 1 else 1)], format="csr")
    op = CSROperator.from_scipy(M, dev(), layout="csr")
    b, x0 = vec(n, 9100), vec(n, 9101)
    shift = torch.tensor([-0.25], dtype=F64, device=dev())
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(engine, "CALLABLE_CG_FUSED_STEP", fused)
        outs[fused] = (engine.cg(b, x0, callable_A=lambda v: op(v), shift=shift, eps=1e-11, maxiter=60).clone(), engine.last_cg.iters,
                       engine.last_cg.resnorm)
    assert outs[True][1] == outs[False][1]
    scale = float(outs[False][0].abs().max())
    assert float((outs[True][0] - outs[False][0]).abs().max()) <= 1e-13 * scale
    ws = Workspace.get(n, 8, dev())
    ws.set_persist(0)                                   # the native operand's STREAMING form
    try:
        xn = engine.cg(b, x0, native=op, shift=shift, eps=1e-11, maxiter=60)
    finally:
        ws.set_persist(-1)
    assert engine.last_cg.iters == outs[True][1]
    assert float((xn - outs[True][0]).abs().max()) <= 1e-13 * scale


@pytest.mark.parametrize("n,k,shadow", [(1, 1, False), (3, 3, False), (1000, 40, False), (4097, 64, True), (100000, 50, True),
                                        (1 << 20, 24, True)])
def test_lanczos_step_around_a_callable_equals_the_phase_calls(monkeypatch, n, k, shadow):
    """dsea_lanczos_callable_step / _alpha (two calls per step around the CALLER'S mat-vec: reference Lanczos.py:60-75 with Amap a
    Python function) against the five phase calls they replace: T to rounding, the basis to 1e-12, with and without the bf16
    shadow of the basis (split and wave-owned geometries, the lp pass from 2^20 rows)."""
    import scipy.sparse as sp
    from dominantsparseeigenad_amd.operators import CSROperator
    rng = np.random.RandomState(n + k)
    diags = [rng.rand(n) + 1.0] + ([rng.randn(n - 1) * 0.4] * 2 if n > 1 else [])
    M = sp.diags(diags, [0, 1, -1][:len(diags)], format="csr")
    op = CSROperator.from_scipy(M, dev(), layout="csr")
    q0 = vec(n, 9200)
    monkeypatch.setattr(engine, "USE_SHADOW", shadow)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(engine, "CALLABLE_LANCZOS_FUSED_STEP", fused)
        Q, ldq, alphas, betas = engine.lanczos(None, k, n, dev(), q0, callable_A=lambda v: op(v))
        res[fused] = (Q[:, :n].clone(), alphas.clone(), betas.clone())
    scale = float(res[False][1].abs().max())
    assert float((res[True][1] - res[False][1]).abs().max()) <= 1e-13 * scale
    if k > 1:
        assert float((res[True][2] - res[False][2]).abs().max()) <= 1e-13 * scale
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-12
