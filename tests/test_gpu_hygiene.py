"""Robustness items: on-device Lanczos breakdown record, per-stream workspaces, per-operator tuning knobs,
persistent basis arena."""
import warnings
from ctypes import c_void_p

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dominantsparseeigenad_amd import _lib, engine  # noqa: E402
from dominantsparseeigenad_amd.Lanczos import symeigLanczos, Lanczos  # noqa: E402
from dominantsparseeigenad_amd.operators import CSROperator, TFIMOperator, Stencil3Operator  # noqa: E402
from helpers import unit  # noqa: E402

F64 = torch.float64


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("layout", ["sell", "csr"])
def test_device_breakdown_record_stops_the_run(layout):
    """Reference Lanczos.py:69-70 divides by beta whatever it is.  The native loop compares beta with the running
    |alpha|,|beta| scale ON THE DEVICE, records the step and turns its remaining launches into no-ops
    (sell: fused tail kernel; csr: scale/store kernel)."""
    n = 40
    A = torch.diag(torch.arange(1, n + 1, dtype=F64))
    op = CSROperator.from_dense(A, dev(), layout=layout)
    q0 = torch.zeros(n, dtype=F64)
    q0[[3, 7, 11]] = torch.tensor([1.0, 2.0, -1.0], dtype=F64)      # 3-dimensional invariant subspace
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lo, vlo = symeigLanczos(op, 10, dev(), extreme="min", sparse=True, dim=n, q0=q0.to(dev()))
    assert engine.last_break == 3
    assert any("breakdown" in str(w.message) for w in rec)
    assert abs(lo.item() - 4.0) < 1e-12 and torch.isfinite(vlo).all()
    assert abs(vlo.norm().item() - 1.0) < 1e-13
    assert float((op(vlo) - lo * vlo).norm()) < 1e-12
    # exact breakdown at the first step: A = I
    eye = CSROperator.from_dense(torch.eye(6, dtype=F64), dev(), layout=layout)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lo, v = symeigLanczos(eye, 4, dev(), extreme="min", sparse=True, dim=6, q0=unit(6, 11).to(dev()))
    assert engine.last_break == 1 and abs(lo.item() - 1.0) < 1e-14 and torch.isfinite(v).all()
    # and no false alarm on a healthy run
    op2 = TFIMOperator(10, dev(), g=torch.tensor([1.0], dtype=F64, device=dev()))
    symeigLanczos(op2, 60, dev(), extreme="min", sparse=True, dim=1024, q0=unit(1024, 5).to(dev()))
    assert engine.last_break == 0


def test_workspace_is_per_stream_and_streams_do_not_interfere():
    n = 1 << 14
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        w1 = engine.Workspace.get(n, 8, dev())
    with torch.cuda.stream(s2):
        w2 = engine.Workspace.get(n, 8, dev())
    assert w1 is not w2 and w1.buffer.data_ptr() != w2.buffer.data_ptr()
    assert engine.Workspace.get(n, 8, dev()) is not w1                      # default stream: a third one
    # two CG solves of the same size in flight on two streams: each equals its own sequential result
    op = TFIMOperator(14, dev(), g=torch.tensor([1.3], dtype=F64, device=dev()))
    shift = torch.tensor(-30.0, dtype=F64, device=dev())
    bs = [unit(n, 40 + c).to(dev()) for c in range(2)]
    x0 = torch.zeros(n, dtype=F64, device=dev())
    ref = [engine.cg(b, x0, native=op, shift=shift, eps=1e-10) for b in bs]
    torch.cuda.synchronize()
    outs = [None, None]
    for rep in range(3):
        with torch.cuda.stream(s1):
            outs[0] = engine.cg(bs[0], x0, native=op, shift=shift, eps=1e-10, poll_every=1000)
        with torch.cuda.stream(s2):
            outs[1] = engine.cg(bs[1], x0, native=op, shift=shift, eps=1e-10, poll_every=1000)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])


def test_tuning_knobs_are_per_operator():
    lib = _lib.load()
    L = 13
    g = torch.tensor([0.9], dtype=F64, device=dev())
    a, b = TFIMOperator(L, dev(), g=g), TFIMOperator(L, dev(), g=g)
    _lib.check(lib.dsea_op_set_tuning(a.handle, 1, 8), "dsea_op_set_tuning")
    assert lib.dsea_op_set_tuning(a.handle, 1, 99) != 0 and lib.dsea_op_set_tuning(a.handle, 7, 1) != 0
    x = unit(1 << L, 77).to(dev())
    ya, yb = a(x), b(x)            # tile 2^8 vs the default 2^11: same operator, neighbour sums in another order
    assert float((ya - yb).abs().max()) < 1e-13


def test_basis_arena_is_reused_and_not_handed_to_users():
    n, k = 1 << 12, 40
    op = TFIMOperator(12, dev(), g=torch.tensor([1.0], dtype=F64, device=dev()))
    q0 = unit(n, 3).to(dev())
    engine.BasisArena.release()
    lo1, v1 = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
    bufs = {key: t.data_ptr() for key, t in engine.BasisArena._bufs.items()}
    assert any(key[2] == "Q" for key in bufs)
    lo2, v2 = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
    assert {key: t.data_ptr() for key, t in engine.BasisArena._bufs.items()} == bufs       # same buffers again
    assert torch.equal(v1, v2) and lo1.item() == lo2.item()
    Qk, T = Lanczos(op, k, dev(), sparse=True, dim=n, q0=q0)                                 # user-visible basis
    assert Qk.untyped_storage().data_ptr() not in bufs.values()
    assert float((Qk.T @ Qk - torch.eye(k, dtype=F64, device=dev())).abs().max()) < 1e-13


@pytest.mark.parametrize("kind", ["tfim", "stencil"])
def test_basisfree_two_pass_lanczos_matches_full_reorthogonalisation(kind):
    """reorth='none': no stored basis, no re-orthogonalisation, Ritz vector from a replayed recurrence.  The extreme
    pair agrees with the full-reorthogonalisation path (the reference's algorithm, Lanczos.py:66) to rounding --
    stated tolerance: eigenvalue 1e-12 relative, eigenvector 1e-8 -- and satisfies the eigen-equation."""
    from dominantsparseeigenad_amd import Lanczos as LZ
    import dominantsparseeigenad_amd.symeig as symeig
    if kind == "tfim":
        n, k = 1 << 14, 200
        op = TFIMOperator(14, dev(), g=torch.tensor([1.0], dtype=F64, device=dev(), requires_grad=True))
        A, hook, g = op.H, op.Hadjoint_to_gadjoint, op.g
    else:
        n, k = 3000, 600
        xm = torch.from_numpy(np.linspace(-1.0, 1.0, num=n, endpoint=False)).to(dev())
        V = (0.5 * xm ** 2).requires_grad_(True)
        op = Stencil3Operator(n, 2.0 / n, V)
        A, hook, g = op.H, op.Hadjoint_to_padjoint, V
    q0 = unit(n, 21).to(dev())
    lo_f, v_f = symeigLanczos(A, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
    lo_n, v_n = symeigLanczos(A, k, dev(), extreme="min", sparse=True, dim=n, q0=q0, reorth="none")
    assert abs(lo_f.item() - lo_n.item()) < 1e-12 * abs(lo_f.item())
    sgn = 1.0 if float(v_f @ v_n) > 0 else -1.0
    assert float((v_f - sgn * v_n).abs().max()) < 1e-8
    assert abs(float(v_n.norm()) - 1.0) < 1e-13
    res_f = float((op(v_f) - lo_f * v_f).norm())
    res_n = float((op(v_n) - lo_n * v_n).norm())
    assert res_n < 10 * res_f + 1e-9 * abs(lo_f.item())
    both = symeigLanczos(A, k, dev(), extreme="both", sparse=True, dim=n, q0=q0, reorth="none")
    hi_f = symeigLanczos(A, k, dev(), extreme="max", sparse=True, dim=n, q0=q0)[0]
    assert abs(both[0].item() - lo_f.item()) < 1e-12 * abs(lo_f.item())
    assert abs(both[2].item() - hi_f.item()) < 1e-10 * abs(hi_f.item())
    # through the primitive: module-level switch (the apply signature is the reference's), gradient unchanged
    symeig.setDominantSparseSymeig(A, hook)
    tvec = unit(n, 22).to(dev())
    grads = []
    for mode in ("full", "none"):
        LZ.REORTH_DEFAULT = mode
        try:
            torch.manual_seed(4)
            E, psi = symeig.DominantSparseSymeig.apply(g, k, n, dev())
            s2 = 1.0 if float(psi.detach() @ v_f) > 0 else -1.0
            (gr,) = torch.autograd.grad(E + s2 * psi.matmul(tvec), g)
            grads.append(gr)
        finally:
            LZ.REORTH_DEFAULT = "full"
    assert float((grads[0] - grads[1]).abs().max()) < 1e-6 * float(grads[0].abs().max())


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 129, 1000, 2049])
def test_symmetric_dense_matvec_reads_upper_triangle_only(n):
    """hand-written symmetric mat-vec (each 64 x 64 upper tile loaded once) vs torch.matmul; the strictly lower
    triangle is never read: poisoning it with NaN changes nothing"""
    from dominantsparseeigenad_amd.operators import SymmetricDenseOperator
    rng = np.random.RandomState(n)
    M = rng.randn(n, n)
    A = torch.from_numpy(M + M.T).to(dev())
    x = torch.from_numpy(rng.randn(n)).to(dev())
    ref = A @ x
    y = SymmetricDenseOperator(A)(x)
    assert float((y - ref).abs().max()) <= 1e-13 * max(float(ref.abs().max()), 1.0) * max(n, 8) ** 0.5
    Ap = A.clone()
    Ap[torch.tril(torch.ones(n, n, dtype=torch.bool, device=dev()), diagonal=-1)] = float("nan")
    assert torch.equal(SymmetricDenseOperator(Ap)(x), y)
    shift = torch.tensor(0.37, dtype=F64, device=dev())
    ys = engine.spmv(SymmetricDenseOperator(A)._H, x, shift=shift)
    assert float((ys - (ref - 0.37 * x)).abs().max()) <= 1e-12 * max(float(ref.abs().max()), 1.0)


def test_symmetric_matvec_reads_fp32_matrices_without_promotion():
    from dominantsparseeigenad_amd.operators import SymmetricDenseOperator, dense_symmetric_operand
    n = 2500
    rng = np.random.RandomState(1)
    M = rng.randn(n, n).astype(np.float32)
    A = torch.from_numpy(M + M.T).to(dev())
    x = torch.from_numpy(rng.randn(n)).to(dev())
    op = dense_symmetric_operand(A)
    assert isinstance(op, SymmetricDenseOperator) and op.A.dtype == torch.float32       # the matrix stays fp32
    ref = A.double() @ x
    assert float((op(x) - ref).abs().max()) <= 1e-12 * float(ref.abs().max())          # fp64 arithmetic on fp32 data


def test_dense_primitive_runs_native_loops_and_matches_gemv_path(monkeypatch):
    """DominantSymeig on a dense CUDA tensor: native in-library loops on the upper-triangle operator vs the generic
    path with torch.matmul (rocBLAS GEMV) as mat-vec"""
    from dominantsparseeigenad_amd.symeig import DominantSymeig
    from helpers import sym_from_seed, PatchRandn
    n, k = 500, 120
    A0 = sym_from_seed(n, 8101).to(dev())
    t = unit(n, 8102).to(dev())
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(engine, "DENSE_SYMMETRIC_KERNEL", flag)
        A = A0.clone().requires_grad_(True)
        with PatchRandn(8110):
            lam, psi = DominantSymeig.apply(A, k, dev())
            sgn = 1.0 if float(psi.detach()[:10].sum()) > 0 else -1.0
            (gA,) = torch.autograd.grad(lam + sgn * psi.matmul(t), A)
        outs.append((lam.item(), sgn * psi.detach(), gA))
    assert abs(outs[0][0] - outs[1][0]) < 1e-12 * abs(outs[1][0])
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 1e-9
    assert float((outs[0][2] - outs[1][2]).abs().max()) < 1e-6 * float(outs[1][2].abs().max())


def test_warm_start_reaches_the_same_pair_with_fewer_vectors(tmp_path):
    """Lanczos.WARM_START: the eigenvector of a neighbouring coupling as start vector -- with 25 vectors
    it is orders of magnitude closer than a random start; the attribute is consumed by one run; RNG consumption
    unchanged."""
    from dominantsparseeigenad_amd import Lanczos as LZ
    L = 14
    n = 1 << L
    op = TFIMOperator(L, dev(), g=torch.tensor([1.00], dtype=F64, device=dev()))
    lo0, v0 = symeigLanczos(op.H, 200, dev(), extreme="min", sparse=True, dim=n, q0=unit(n, 31).to(dev()))
    op.g = torch.tensor([1.02], dtype=F64, device=dev())
    lo_ref, v_ref = symeigLanczos(op.H, 200, dev(), extreme="min", sparse=True, dim=n, q0=unit(n, 32).to(dev()))
    torch.manual_seed(0)
    LZ.WARM_START = v0
    lo_w, v_w = symeigLanczos(op.H, 25, dev(), extreme="min", sparse=True, dim=n)
    state_after = torch.cuda.get_rng_state(dev())
    assert LZ.WARM_START is None
    torch.manual_seed(0)
    lo_c, v_c = symeigLanczos(op.H, 25, dev(), extreme="min", sparse=True, dim=n)      # cold, same k, same draws
    assert torch.equal(torch.cuda.get_rng_state(dev()), state_after)
    err_w, err_c = abs(lo_w.item() - lo_ref.item()), abs(lo_c.item() - lo_ref.item())
    assert err_w < 1e-8 * abs(lo_ref.item()) and err_c > 100.0 * err_w + 1e-12, (err_w, err_c)
    sgn = 1.0 if float(v_w @ v_ref) > 0 else -1.0
    assert float((v_w - sgn * v_ref).abs().max()) < 1e-4
    # on-disk CSR operand: scipy.sparse.save_npz -> CSROperator.from_npz
    import scipy.sparse as sp
    M = sp.random(500, 500, density=0.02, random_state=3, format="csr")
    M = M + M.T
    path = str(tmp_path / "m.npz")
    sp.save_npz(path, M)
    x = unit(500, 33).to(dev())
    y = CSROperator.from_npz(path, dev())(x)
    assert float((y.cpu() - torch.from_numpy(M @ x.cpu().numpy())).abs().max()) < 1e-13


def test_lazy_rank_one_adjoint_of_a_dense_matrix_through_the_sparse_primitive():
    """the rank-1 adjoint A-bar = v1 v2^T without the n x n gradient.  A(p) = A0 + p A1 dense symmetric:
    the dense primitive materialises grad_A (symeig.py:29) and autograd contracts it with A1; the sparse primitive on
    the SAME dense operand hands (v1, v2) to the hook, which returns v1^T A1 v2.  Same numbers."""
    from dominantsparseeigenad_amd.operators import SymmetricDenseOperator
    from dominantsparseeigenad_amd.symeig import DominantSymeig
    import dominantsparseeigenad_amd.symeig as symeig
    from helpers import sym_from_seed, PatchRandn
    n, k = 700, 150
    A0, A1 = sym_from_seed(n, 8201).to(dev()), sym_from_seed(n, 8202, scale=0.1).to(dev())
    t = unit(n, 8203).to(dev())
    p = torch.tensor([0.3], dtype=F64, device=dev(), requires_grad=True)
    with PatchRandn(8210):
        lam, psi = DominantSymeig.apply(A0 + p * A1, k, dev())
        sgn = 1.0 if float(psi.detach()[:16].sum()) > 0 else -1.0
        (g_dense,) = torch.autograd.grad(lam + sgn * psi.matmul(t), p)
    op = SymmetricDenseOperator((A0 + p.detach() * A1))
    symeig.setDominantSparseSymeig(op, lambda v1, v2: (v1.matmul(A1.matmul(v2)))[None])
    with PatchRandn(8210):
        lam2, psi2 = symeig.DominantSparseSymeig.apply(p, k, n, dev())
        sgn2 = 1.0 if float(psi2.detach()[:16].sum()) > 0 else -1.0
        (g_lazy,) = torch.autograd.grad(lam2 + sgn2 * psi2.matmul(t), p)
    assert abs(lam.item() - lam2.item()) < 1e-12 * abs(lam.item())
    assert abs(g_dense.item() - g_lazy.item()) < 1e-7 * abs(g_dense.item()), (g_dense.item(), g_lazy.item())


def test_basis_arena_measures_the_placement_of_a_new_basis(monkeypatch):
    """A new arena basis above PLACEMENT_MIN_BYTES is chosen among PLACEMENT_TRIES simultaneously alive candidates by
    timing the dots pass on each (engine.BasisArena); results do not depend on which candidate wins."""
    L, k = 14, 64
    n = 1 << L
    op = TFIMOperator(L, dev())
    op.g = torch.tensor([1.0], dtype=F64, device=dev())
    q0 = unit(n, 77).to(dev())
    out = []
    for tries in (1, 3):
        engine.BasisArena.release()
        torch.cuda.empty_cache()
        monkeypatch.setattr(engine.BasisArena, "PLACEMENT_MIN_BYTES", 1 << 20)
        monkeypatch.setattr(engine.BasisArena, "PLACEMENT_TRIES", tries)
        engine.BasisArena.last_placement = None
        lo, v = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
        out.append((lo.item(), v.clone()))
        if tries == 1:
            assert engine.BasisArena.last_placement is None
        else:
            assert len(engine.BasisArena.last_placement) == 3 and min(engine.BasisArena.last_placement) > 0.0
            # the arena is reused: no new measurement on the second call
            engine.BasisArena.last_placement = None
            symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
            assert engine.BasisArena.last_placement is None
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
    engine.BasisArena.release()


def test_more_than_2048_krylov_vectors():
    """The dots pass keeps one LDS row of i + 1 partial sums per wave: beyond 2048 vectors a block carries two waves
    instead of four (beyond 4096 one), and kmax is capped at 8000 (include/dsea.h)."""
    from ctypes import byref, c_size_t
    L, k = 18, 2100
    n = 1 << L
    op = TFIMOperator(L, dev(), g=torch.tensor([1.0], dtype=F64, device=dev()))
    q0 = unit(n, 5).to(dev())
    engine.BasisArena.release()
    Qk, T = Lanczos(op, k, dev(), sparse=True, dim=n, q0=q0)
    # orthonormality of a sample of the basis, early / middle / late vectors alike
    idx = torch.tensor([0, 1, 2, 500, 1023, 1024, 2047, 2048, 2049, 2098, 2099], device=dev())
    S = Qk[:, idx]
    G = S.T @ Qk
    G[torch.arange(idx.numel(), device=dev()), idx] -= 1.0
    assert float(G.abs().max()) < 1e-12
    lo = torch.linalg.eigvalsh(T)[0].item()
    ks = (2 * np.arange(L) + 1) * np.pi / L
    exact = float(-0.5 * (2 * np.sqrt(2 - 2 * np.cos(ks))).sum())
    assert abs(lo - exact) < 1e-11 * abs(exact)
    nbytes = c_size_t()
    lib = _lib.load()
    assert lib.dsea_ws_bytes(1 << 12, 8000, byref(nbytes)) == 0
    assert lib.dsea_ws_bytes(1 << 12, 8001, byref(nbytes)) == -1
    del Qk
    torch.cuda.empty_cache()


def test_nested_solver_on_the_same_workspace_raises():
    """A workspace is not re-entrant (include/dsea.h).  A solver started from inside the user mat-vec of another
    solver of the same size on the same stream used to share -- and silently corrupt -- its scalar / CG-state buffers;
    it now raises.  The same nested solve on ANOTHER STREAM has a workspace of its own and is fine."""
    from dominantsparseeigenad_amd.CG import CG_torch
    n = 512
    A = torch.diag(torch.linspace(1.0, 3.0, n, dtype=F64)).to(dev())
    b = unit(n, 8).to(dev())
    side = torch.cuda.Stream()

    def matvec_with_nested_solve(v):
        CG_torch(lambda w: A @ w, b, torch.zeros_like(b), sparse=True)          # same n, same stream
        return A @ v

    with pytest.raises(RuntimeError, match="not re-entrant"):
        symeigLanczos(matvec_with_nested_solve, 8, dev(), extreme="min", sparse=True, dim=n, q0=unit(n, 9).to(dev()))
    assert engine.Workspace.get(n, 8, dev()).busy is None                         # released on the way out

    def matvec_with_nested_solve_on_a_side_stream(v):
        done = torch.cuda.Event()
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.current_stream())
            CG_torch(lambda w: A @ w, b, torch.zeros_like(b), sparse=True)
            done.record(side)
        torch.cuda.current_stream().wait_event(done)
        return A @ v

    lo, _ = symeigLanczos(matvec_with_nested_solve_on_a_side_stream, 8, dev(), extreme="min", sparse=True, dim=n,
                          q0=unit(n, 9).to(dev()))
    ref, _ = symeigLanczos(lambda v: A @ v, 8, dev(), extreme="min", sparse=True, dim=n, q0=unit(n, 9).to(dev()))
    assert lo.item() == ref.item()


def test_basis_arena_shrinks_and_caps_its_placement_probe(monkeypatch):
    """engine.BasisArena: a request far below what the arena holds replaces the buffer (one large solve does not pin its
    basis for the life of the process); the placement probe's extra candidates are capped by PLACEMENT_BUDGET_BYTES."""
    engine.BasisArena.release()
    monkeypatch.setattr(engine.BasisArena, "SHRINK_MIN_BYTES", 1 << 20)
    big = engine.BasisArena.get(dev(), "Q", 64 << 20)
    assert big.numel() == 64 << 20
    same = engine.BasisArena.get(dev(), "Q", 32 << 20)            # within the ratio: reused
    assert same.data_ptr() == big.data_ptr() and same.numel() == 64 << 20
    del big, same
    small = engine.BasisArena.get(dev(), "Q", 4 << 20)            # < 1/4 of the capacity: replaced by a small one
    assert small.numel() == 4 << 20
    del small
    engine.BasisArena.release()
    # probe budget: with a budget below one candidate no extra candidate is taken
    calls = []
    monkeypatch.setattr(engine.BasisArena, "PLACEMENT_MIN_BYTES", 1 << 20)
    monkeypatch.setattr(engine.BasisArena, "PLACEMENT_BUDGET_BYTES", 1 << 20)
    engine.BasisArena.get(dev(), "Q", 8 << 20, probe=lambda b: calls.append(1) or 1.0)
    assert calls == []
    engine.BasisArena.release()
    monkeypatch.setattr(engine.BasisArena, "PLACEMENT_BUDGET_BYTES", 64 << 20)
    engine.BasisArena.get(dev(), "Q", 8 << 20, probe=lambda b: calls.append(1) or 1.0)
    assert len(calls) == 3
    engine.BasisArena.release()


def test_rocblas_handle_table_serves_more_streams_than_it_holds():
    """the rocBLAS handle table of the GEMM-shaped operands was keyed by stream only, held 16 handles and
    refused the 17th stream.  It is keyed by (device, stream) and evicts the least recently used handle: 40 streams
    (PyTorch's pool alone has 32) all get a correct GEMV."""
    from dominantsparseeigenad_amd.operators import DenseOperator
    n = 96
    A = torch.from_numpy(np.random.RandomState(0).randn(n, n)).to(dev())
    v = unit(n, 4).to(dev())
    want = A @ v
    op = DenseOperator(A)
    streams = [torch.cuda.Stream() for _ in range(40)]
    for rep in range(2):
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                y = op(v)
            s.synchronize()
            assert float((y - want).abs().max()) < 1e-13


def test_shadow_is_dropped_when_it_does_not_fit(monkeypatch):
    """engine.shadow_fits: the bf16 shadow is an optimisation; when the device cannot hold it next to the basis the
    solve runs with the all-fp64 correction pass instead of failing in the allocator (L = 28, k = 100 on one GPU)."""
    n, k = 1 << 14, 48
    op = TFIMOperator(14, dev())
    op.g = torch.tensor([1.0], dtype=F64, device=dev())
    q0 = unit(n, 6).to(dev())
    lo1, v1 = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
    lp_on = engine.lanczos_lp_stats(n, dev())
    monkeypatch.setattr(engine, "shadow_fits", lambda *a, **kw: False)
    lo2, v2 = symeigLanczos(op, k, dev(), extreme="min", sparse=True, dim=n, q0=q0)
    lp_off = engine.lanczos_lp_stats(n, dev())
    assert lp_on[0] + lp_on[1] == k - 1 or lp_on == (0, 0)        # (small slabs may use the fp64 split kernels)
    assert lp_off == (0, 0)
    assert abs(lo1.item() - lo2.item()) < 1e-13 * abs(lo1.item()) and float((v1 - v2).abs().max()) < 1e-11


@pytest.mark.parametrize("form", ["native-fused", "native-csr", "callable"])
def test_cgs2_option_repeats_the_gram_schmidt_pass(form):
    """``reorth="twice"`` / ``engine.REORTH_PASSES = 2`` (include/dsea.h dsea_ws_set_reorth_passes): the Gram-Schmidt pass
    of Lanczos.py:66 applied twice per step -- an option the reference lacks.  Same Krylov process: the extreme Ritz pair
    agrees with the default to rounding and the basis is orthonormal to rounding level on the reference's own hard case
    (k ~ n on the 1-D Schroedinger stencil, tests/test_Lanczos.py:100-112).  Measured there: the reference's ONE pass
    already reaches 1.3e-15 (its three-term update removes the O(1) components first, so the pass only measures
    rounding-level residue) -- the second pass is insurance, never the default."""
    N = 300
    h = 2.0 / N
    xm = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False))
    V = (0.5 * xm ** 2).to(dev())
    st = Stencil3Operator(N, h, V)
    if form == "native-fused":
        A = st
    elif form == "native-csr":
        dense = torch.stack([st.H(torch.eye(N, dtype=F64, device=dev())[j]) for j in range(N)]).T.cpu()
        A = CSROperator.from_dense(0.5 * (dense + dense.T), dev(), layout="csr")
    else:
        A = lambda v: st.H(v)       # noqa: E731  (opaque callable: phase calls around the user's mat-vec)
    q0 = unit(N, 31).to(dev())
    k = N - 20
    old_persist = engine.LANCZOS_PERSIST
    engine.LANCZOS_PERSIST = False          # (the single-launch form keeps the reference's one pass)
    try:
        lo1, v1 = symeigLanczos(A, k, dev(), extreme="min", sparse=True, dim=N, q0=q0)
        lo2, v2 = symeigLanczos(A, k, dev(), extreme="min", sparse=True, dim=N, q0=q0, reorth="twice")
        Q1, _ = Lanczos(A, k, dev(), sparse=True, dim=N, q0=q0)
        engine.REORTH_PASSES = 2
        try:
            Q2, T2 = Lanczos(A, k, dev(), sparse=True, dim=N, q0=q0)
        finally:
            engine.REORTH_PASSES = 1
    finally:
        engine.LANCZOS_PERSIST = old_persist
    assert engine.REORTH_PASSES == 1
    assert abs(lo1.item() - lo2.item()) < 1e-10 * abs(lo1.item())
    eye = torch.eye(k, dtype=F64, device=dev())
    o1 = float((Q1.T @ Q1 - eye).abs().max())
    o2 = float((Q2.T @ Q2 - eye).abs().max())
    print("%s: ||Q^T Q - I||_max one pass %.2e, two passes %.2e" % (form, o1, o2))
    assert o2 < 5e-14 and o2 <= o1 * 1.5 + 1e-15
    r1, r2 = float((A(v1) - lo1 * v1).norm()), float((A(v2) - lo2 * v2).norm())
    print("   eigen-residual one pass %.2e, two passes %.2e (||A|| ~ %.1e)" % (r1, r2, 2.0 / h ** 2))
    assert r2 <= 2.0 * r1 + 1e-9 * (2.0 / h ** 2)
