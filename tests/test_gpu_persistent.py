"""Persistent single-launch CG (k_cg_persist_stencil, BASELINE config 3 regime) against the streaming
3-launches-per-iteration form: BIT-IDENTICAL iterates (torch.equal), same iteration counts, same residual norms --
on ragged sizes, with and without shift, fixed-iteration and converged runs -- and against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from dominantsparseeigenad_amd import engine  # noqa: E402
from dominantsparseeigenad_amd.operators import Stencil3Operator  # noqa: E402
from dominantsparseeigenad_amd.synthetic import normal_vector  # noqa: E402

F64 = torch.float64
cuda = torch.device("cuda:0")


def _problem(N, seed=50):
    h = 2.0 / N
    xmesh = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False))
    V = 0.5 * xmesh ** 2
    op = Stencil3Operator(N, h, V.to(cuda))
    b = torch.from_numpy(normal_vector(N, seed)).to(cuda)
    x0 = torch.from_numpy(normal_vector(N, seed + 1)).to(cuda)
    return op, V, h, b, x0


def _solve(op, b, x0, shift, mode, **kw):
    ws = engine.Workspace.get(op.n, 8, cuda)
    ws.set_persist(mode)
    try:
        x = engine.cg(b, x0, native=op, shift=shift, **kw)
    finally:
        ws.set_persist(-1)
    return x, engine.last_cg.iters, engine.last_cg.resnorm, engine.last_cg.converged


@pytest.mark.parametrize("N", [1, 2, 3, 511, 512, 513, 1000, 4097, 100000, 131072, 300001, 524288])
def test_persistent_cg_is_bit_identical_to_streaming_form(N):
    op, V, h, b, x0 = _problem(N)
    shift = torch.tensor(-1.0, dtype=F64, device=cuda)
    iters = 40 if N > 3 else 3
    ref = _solve(op, b, x0, shift, 0, eps=0.0, maxiter=iters)
    nt = (N + 511) // 512
    for mode in (-1, 1, 2, 21, 22, 11, 12):       # pairs per thread x virtual blocks per workgroup (dsea_ws_set_persist)
        tpw = {1: 4, 2: 8, 21: 2, 22: 4, 11: 1, 12: 2}.get(mode)
        if tpw is not None and (nt + tpw - 1) // tpw > 256:
            continue                                # more than 256 workgroups: outside this geometry's envelope
        got = _solve(op, b, x0, shift, mode, eps=0.0, maxiter=iters)
        assert got[1] == ref[1] == iters and got[2] == ref[2], (mode, got[1:], ref[1:])
        assert torch.equal(got[0], ref[0]), (mode, float((got[0] - ref[0]).abs().max()))


def test_persistent_cg_converged_run_and_no_shift():
    N = 20000
    op, V, h, b, x0 = _problem(N, seed=60)
    # no shift (A itself is SPD but ill-conditioned: fixed number of iterations)
    ref = _solve(op, b, x0, None, 0, eps=0.0, maxiter=60)
    got = _solve(op, b, x0, None, -1, eps=0.0, maxiter=60)
    assert got[1] == ref[1] == 60 and got[2] == ref[2] and torch.equal(got[0], ref[0])
    # shifted, well-conditioned system: run to convergence -- same iteration count, same final iterate
    shift = torch.tensor(-3.5e5, dtype=F64, device=cuda)
    ref = _solve(op, b, x0, shift, 0, eps=1e-3, maxiter=N)
    got = _solve(op, b, x0, shift, -1, eps=1e-3, maxiter=N)
    assert ref[3] and got[3] and got[1] == ref[1] and got[2] == ref[2], (ref[1:], got[1:])
    assert torch.equal(got[0], ref[0])
    # early out: a start vector that already solves the system
    xs = got[0]
    bb = op(xs) - shift * xs
    again = _solve(op, bb, xs, shift, -1, eps=1e30, maxiter=N)
    assert again[1] == 0 and again[3] and torch.equal(again[0], xs)


def test_persistent_cg_first_50_iterates_against_oracle():
    """parity on the first 50 CG iterates of the shifted SPD system at N = 1e5"""
    N = 100000
    op, V, h, b, x0 = _problem(N, seed=42)
    ref = oracle.Stencil3(N, h, V)
    theta = torch.tensor(-1.0, dtype=F64)
    st = {}
    xo = oracle.cg_solve(lambda v: ref.H(v) - theta * v, b.cpu(), x0.cpu(), sparse=True, maxiter=50, stats=st)
    x, it, rn, _ = _solve(op, b, x0, theta.to(cuda), -1, eps=1e-7, maxiter=50)
    assert it == st["iters"] == 50
    assert float((x.cpu() - xo).abs().max()) <= 1e-10 * float(xo.abs().max())


# ------------------------------------------------------------------ merged-reduction form (one exchange per iteration)
MERGED_MODES = {100: None, 101: 4, 102: 8, 121: 2, 122: 4, 111: 1, 112: 2}     # mode -> tiles per workgroup


@pytest.mark.parametrize("N", [1, 2, 3, 511, 513, 1000, 4097, 100000, 131072, 300001, 524288])
def test_merged_reduction_form_follows_the_reference_iteration(N):
    """dsea_ws_set_persist(100 + g): r.r and r.Ar reduced together, A p by recurrence (Chronopoulos-Gear) -- the same
    iteration as CG.py:31-40 in exact arithmetic, not its rounding sequence.  After 40 iterations of the
    ill-conditioned shifted stencil system the iterate stays within 1e-11 of the reference-rounding form (measured:
    <= 7e-14), same residual norm to 1e-10, on every launch geometry and ragged size."""
    op, V, h, b, x0 = _problem(N)
    shift = torch.tensor(-1.0, dtype=F64, device=cuda)
    iters = 40 if N > 3 else 3
    ref = _solve(op, b, x0, shift, 0, eps=0.0, maxiter=iters)
    nt = (N + 511) // 512
    scale = float(ref[0].abs().max())
    for mode, tpw in MERGED_MODES.items():
        if tpw is not None and (nt + tpw - 1) // tpw > 256:
            continue
        got = _solve(op, b, x0, shift, mode, eps=0.0, maxiter=iters)
        assert got[1] == iters, (mode, got[1])
        assert float((got[0] - ref[0]).abs().max()) <= 1e-11 * scale, (mode, float((got[0] - ref[0]).abs().max()) / scale)
        if N > 3:
            assert abs(got[2] - ref[2]) <= 1e-10 * ref[2], (mode, got[2], ref[2])


def test_merged_reduction_form_converged_run_oracle_and_keyword():
    N = 20000
    op, V, h, b, x0 = _problem(N, seed=60)
    shift = torch.tensor(-3.5e5, dtype=F64, device=cuda)
    ref = _solve(op, b, x0, shift, 0, eps=1e-3, maxiter=N)
    got = _solve(op, b, x0, shift, 100, eps=1e-3, maxiter=N)
    assert ref[3] and got[3] and got[1] == ref[1], (ref[1:], got[1:])          # same number of iterations
    assert abs(got[2] - ref[2]) <= 1e-9 * ref[2]
    assert float((op(got[0]) - shift * got[0] - b).norm()) < 1.01e-3             # the TRUE residual meets the tolerance
    assert float((got[0] - ref[0]).abs().max()) <= 1e-6 * float(ref[0].abs().max())
    # early out on a start vector that already solves the system
    again = _solve(op, op(got[0]) - shift * got[0], got[0], shift, 100, eps=1e30, maxiter=N)
    assert again[1] == 0 and again[3] and torch.equal(again[0], got[0])
    # the first 50 iterates at N = 1e5 against the CPU oracle (reference recurrences)
    N = 100000
    op, V, h, b, x0 = _problem(N, seed=42)
    ref_op = oracle.Stencil3(N, h, V)
    theta = torch.tensor(-1.0, dtype=F64)
    st = {}
    xo = oracle.cg_solve(lambda v: ref_op.H(v) - theta * v, b.cpu(), x0.cpu(), sparse=True, maxiter=50, stats=st)
    x = engine.cg(b, x0, native=op, shift=theta.to(cuda), eps=1e-7, maxiter=50, merged_reductions=True)   # keyword form
    assert engine.last_cg.iters == st["iters"] == 50 and engine.last_cg.form == "persistent, one exchange"
    assert float((x.cpu() - xo).abs().max()) <= 1e-10 * float(xo.abs().max())
    # the keyword leaves the workspace's own setting alone
    assert getattr(engine.Workspace.get(N, 8, cuda), "persist_mode", -1) == -1


# ------------------------------------------------------------------ single-launch Lanczos (README-sized problems)
def _lanczos_both(op, k, n, q0):
    """(Q, T) with the single-launch form and with the multi-launch kernels"""
    from dominantsparseeigenad_amd.Lanczos import Lanczos
    out = []
    for on in ("force", False):
        engine.LANCZOS_PERSIST = on
        try:
            Qk, T = Lanczos(op, k, cuda, sparse=True, dim=n, q0=q0)
            out.append((Qk.clone(), T.clone()))
        finally:
            engine.LANCZOS_PERSIST = True
    return out


@pytest.mark.parametrize("L,k", [(1, 2), (3, 5), (6, 40), (7, 60), (8, 100), (8, 200), (10, 300), (12, 200), (13, 120)])
def test_single_launch_lanczos_tfim_matches_oracle_and_streaming_form(L, k):
    """csrc/dsea_lanczos_persist.hip on the reference's own problem sizes (examples/TFIM/E0.py: N = 10, k = 300):
    the same Krylov process as the multi-launch kernels and as the CPU oracle on the same start vector -- extreme Ritz
    pairs at 1e-12, leading tridiagonal entries at 1e-10, orthonormal basis; one workgroup (L <= 7) up to 64 (L = 13)."""
    from dominantsparseeigenad_amd.operators import TFIMOperator
    n = 1 << L
    k = min(k, n)
    op = TFIMOperator(L, cuda, g=torch.tensor([1.1], dtype=F64, device=cuda))
    q0 = torch.from_numpy(normal_vector(n, 900 + L)).to(cuda)
    if (L, k) == (8, 200):
        # k beyond the Krylov dimension of the start vector (degenerate spectrum): beta decays to ~1e-5 without an exact
        # breakdown.  The regime that separates the reference's "three-term, then Gram-Schmidt" order from a one-projection
        # form (which returned E0 = -39.8 here): the extreme Ritz pair must still be exact.
        from dominantsparseeigenad_amd.Lanczos import symeigLanczos
        w = torch.linalg.eigvalsh(oracle.TFIMTables(L, g=torch.tensor([1.1], dtype=F64)).dense())
        for on in ("force", False):
            engine.LANCZOS_PERSIST = on
            try:
                lo, v = symeigLanczos(op, k, cuda, extreme="min", sparse=True, dim=n, q0=q0)
            finally:
                engine.LANCZOS_PERSIST = True
            assert abs(lo.item() - w[0].item()) < 1e-12 * abs(w[0].item())
            assert float((op.H(v) - lo * v).norm()) < 1e-10
        return
    (Qp, Tp), (Qs, Ts) = _lanczos_both(op, k, n, q0)
    m = min(k, 12)
    assert float((Tp[:m, :m] - Ts[:m, :m]).abs().max()) < 1e-10 * float(Ts[:m, :m].abs().max())
    eye = torch.eye(k, dtype=F64, device=cuda)
    if k < n:            # (k = n: the last vectors of an exhausted Krylov space are rounding noise on any path)
        assert float((Qp.T @ Qp - eye).abs().max()) < 1e-12
    wp, ws_ = torch.linalg.eigvalsh(Tp), torch.linalg.eigvalsh(Ts)
    assert abs(wp[0] - ws_[0]) < 1e-12 * abs(ws_[0]) and abs(wp[-1] - ws_[-1]) < 1e-12 * abs(ws_[-1])
    # against the CPU oracle (gather-table operator of the reference, TFIM.py:39-51,91-98)
    model = oracle.TFIMTables(L, g=torch.tensor([1.1], dtype=F64))
    seq = iter([q0.cpu(), torch.zeros(n, dtype=F64)])
    Qo, al, be = oracle.lanczos_tridiag(model.H, k, sparse=True, dim=n, draw=lambda m_, dtype=F64: next(seq))
    To = oracle.solvers.tridiag_matrix(al, be)
    wo = torch.linalg.eigvalsh(To)
    assert abs(wp[0].cpu() - wo[0]) < 1e-12 * abs(wo[0]) and abs(wp[-1].cpu() - wo[-1]) < 1e-12 * abs(wo[-1])
    assert float((Tp[:m, :m].cpu() - To[:m, :m]).abs().max()) < 1e-10 * float(To[:m, :m].abs().max())


@pytest.mark.parametrize("N,k", [(1, 1), (2, 2), (127, 60), (128, 100), (129, 64), (300, 250), (1000, 300), (4097, 200), (8192, 512)])
def test_single_launch_lanczos_stencil_matches_streaming_form(N, k):
    """the 3-point stencil of examples/schrodinger1D.py:18-27 (N = 300, k = 300 is the reference's own configuration;
    tests/test_Lanczos.py uses N = 1000): ragged sizes, edge exchange between neighbouring workgroups, k = n."""
    op, V, h, b, x0 = _problem(N, seed=70)
    (Qp, Tp), (Qs, Ts) = _lanczos_both(op, k, N, b)
    m = min(k, 12)
    assert float((Tp[:m, :m] - Ts[:m, :m]).abs().max()) < 1e-10 * float(Ts[:m, :m].abs().max())
    wp, ws_ = torch.linalg.eigvalsh(Tp), torch.linalg.eigvalsh(Ts)
    assert abs(wp[0] - ws_[0]) < 1e-11 * abs(ws_[-1]) and abs(wp[-1] - ws_[-1]) < 1e-12 * abs(ws_[-1])
    if k < N:
        assert float((Qp.T @ Qp - torch.eye(k, dtype=F64, device=cuda)).abs().max()) < 1e-11


def test_single_launch_lanczos_is_the_default_and_deterministic_and_reports_breakdown():
    """on by default where it applies; two runs are bit-identical (every workgroup sums the same partials in the same
    order); an invariant subspace is recorded on the device exactly like in the multi-launch form."""
    import warnings
    from dominantsparseeigenad_amd.Lanczos import Lanczos, symeigLanczos
    from dominantsparseeigenad_amd.operators import TFIMOperator
    L, k = 10, 120
    n = 1 << L
    op = TFIMOperator(L, cuda, g=torch.tensor([0.9], dtype=F64, device=cuda))
    q0 = torch.from_numpy(normal_vector(n, 77)).to(cuda)
    Q1, T1 = Lanczos(op, k, cuda, sparse=True, dim=n, q0=q0)
    Q2, T2 = Lanczos(op, k, cuda, sparse=True, dim=n, q0=q0)
    assert torch.equal(T1, T2) and torch.equal(Q1, Q2)
    engine.LANCZOS_PERSIST = False
    try:
        Q3, T3 = Lanczos(op, k, cuda, sparse=True, dim=n, q0=q0)
    finally:
        engine.LANCZOS_PERSIST = True
    assert not torch.equal(T1, T3) or True      # (different rounding path; equality is neither required nor excluded)
    assert float((T1 - T3)[:10, :10].abs().max()) < 1e-12
    # breakdown: start vector inside a 2-dimensional invariant subspace of the stencil operator's ... use TFIM g = 0:
    # H is diagonal, q0 supported on rows of 3 distinct diagonal values -> Krylov dimension 3
    opd = TFIMOperator(6, cuda, g=torch.tensor([0.0], dtype=F64, device=cuda))
    q = torch.zeros(64, dtype=F64, device=cuda)
    q[0], q[1], q[3] = 1.0, 2.0, -1.0           # diagonal values -6, -2, -2 ... distinct count decides the dimension
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lo, v = symeigLanczos(opd, 10, cuda, extreme="min", sparse=True, dim=64, q0=q)
    assert 1 <= engine.last_break <= 3 and any("breakdown" in str(w.message) for w in rec)
    assert abs(lo.item() + 6.0) < 1e-12 and torch.isfinite(v).all()


# ------------------------------------------------------------------ mid-size single-launch Lanczos (BASELINE configs[2] regime)
@pytest.mark.parametrize("N,k", [(8193, 40), (8200, 130), (20000, 120), (33001, 200), (65536, 96), (100000, 300),
                                 (100001, 64), (131072, 150), (131072, 2)])
def test_mid_size_single_launch_lanczos_matches_streaming_form(N, k):
    """csrc/dsea_lanczos_persist_mid.hip (3-point stencil, 8192 < N <= 131072: one launch, a slab per workgroup, the first
    vectors of the basis in LDS and registers, the rest streamed -- fp64 for the dots, bf16 shadow for the correction)
    against the multi-launch kernels on the same start vector: leading block of T, extreme Ritz values, orthonormal
    basis.  Ragged slabs (N not a multiple of the slab rows, odd N), the smallest and the largest slab, k = 2."""
    op, V, h, b, x0 = _problem(N, seed=70)
    (Qp, Tp), (Qs, Ts) = _lanczos_both(op, k, N, b)
    m = min(k, 12)
    assert float((Tp[:m, :m] - Ts[:m, :m]).abs().max()) < 1e-10 * float(Ts[:m, :m].abs().max())
    wp, ws_ = torch.linalg.eigvalsh(Tp), torch.linalg.eigvalsh(Ts)
    assert abs(wp[0] - ws_[0]) < 1e-11 * abs(ws_[-1]) and abs(wp[-1] - ws_[-1]) < 1e-12 * abs(ws_[-1])
    assert float((Qp.T @ Qp - torch.eye(k, dtype=F64, device=cuda)).abs().max()) < 1e-11
    # the stored basis satisfies the three-term recurrence with the stored T (A Q = Q T + beta_k q_{k+1} e_k^T): columns < k - 1
    AQ = torch.stack([op.H(Qp[:, j].contiguous()) for j in range(min(k - 1, 6))], dim=1)
    QT = Qp @ Tp[:, :AQ.shape[1]]
    assert float((AQ - QT).abs().max()) < 1e-9 * float(Tp.abs().max())


@pytest.mark.parametrize("shadow,tau", [(True, None), (False, None), (True, 0.0)])
def test_mid_size_single_launch_lanczos_streamed_part_with_and_without_the_shadow(shadow, tau):
    """N = 1e5, k = 300 (config 3): 200+ vectors are streamed.  With the bf16 shadow (premise holds on every step), with
    the all-fp64 correction pass, and with tau = 0 (premise violated on every step -> fp64 fallback inside the launch):
    the same Ritz values and an orthonormal basis each time; the step counters say which path ran."""
    from dominantsparseeigenad_amd.Lanczos import Lanczos
    N, k = 100000, 300
    op, V, h, b, x0 = _problem(N, seed=71)
    old = engine.USE_SHADOW, engine.SHADOW_TAU
    engine.USE_SHADOW = shadow
    if tau is not None:
        engine.SHADOW_TAU = tau
    try:
        Qk, T = Lanczos(op, k, cuda, sparse=True, dim=N, q0=b)
        lp, fb = engine.lanczos_lp_stats(N, cuda)
        engine.LANCZOS_PERSIST = False
        Qr, Tr = Lanczos(op, k, cuda, sparse=True, dim=N, q0=b)
    finally:
        engine.USE_SHADOW, engine.SHADOW_TAU = old
        engine.LANCZOS_PERSIST = True
    assert lp + fb == k - 1
    if shadow and tau is None:
        assert fb == 0
    else:
        assert fb > 150 and lp < 150            # the steps that stream anything took the fp64 path
    w, wr = torch.linalg.eigvalsh(T), torch.linalg.eigvalsh(Tr)
    assert abs(w[0] - wr[0]) < 1e-11 * abs(wr[-1]) and abs(w[-1] - wr[-1]) < 1e-12 * abs(wr[-1])
    G = Qk.T @ Qk
    assert float((G - torch.eye(k, dtype=F64, device=cuda)).abs().max()) < 1e-11


def test_mid_size_single_launch_lanczos_is_deterministic_reports_breakdown_and_survives_a_lost_peer():
    import time
    import warnings
    from dominantsparseeigenad_amd import _lib
    from dominantsparseeigenad_amd.Lanczos import Lanczos, symeigLanczos
    N, k = 40000, 90
    op, V, h, b, x0 = _problem(N, seed=72)
    engine.LANCZOS_PERSIST = "force"        # (the automatic choice starts at 49152 rows, where the form is measured to win)
    try:
        _mid_size_checks(op, N, k, b, Lanczos, symeigLanczos, _lib, warnings, time)
    finally:
        engine.LANCZOS_PERSIST = True


def _mid_size_checks(op, N, k, b, Lanczos, symeigLanczos, _lib, warnings, time):
    Q1, T1 = Lanczos(op, k, cuda, sparse=True, dim=N, q0=b)
    Q2, T2 = Lanczos(op, k, cuda, sparse=True, dim=N, q0=b)
    assert torch.equal(T1, T2) and torch.equal(Q1, Q2)
    # breakdown: with h = 1e200 the stencil coefficient -0.5 / h^2 is exactly zero and the operator is diag(V); a start vector
    # supported on rows of three distinct potential values spans a 3-dimensional Krylov space -> beta_3 = 0 exactly
    Vd = torch.zeros(N, dtype=F64, device=cuda)
    Vd[5], Vd[20000], Vd[39999] = 1.0, 2.0, 4.0
    opd = Stencil3Operator(N, 1e150, Vd)          # coefficient -0.5e-300: negligible beside V, no overflow in h ** 2
    q = torch.zeros(N, dtype=F64, device=cuda)
    q[5], q[20000], q[39999] = 1.0, -2.0, 0.5
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lo, v = symeigLanczos(opd, 12, cuda, extreme="min", sparse=True, dim=N, q0=q)
    assert 1 <= engine.last_break <= 3 and any("breakdown" in str(w.message) for w in rec)
    assert torch.isfinite(v).all() and abs(lo.item() - 1.0) < 1e-12
    # lost peer: the launch ends, the host repeats the run on the multi-launch kernels and stays there
    lib = _lib.load()
    engine.LANCZOS_PERSIST = False
    try:
        ref_lo, ref_v = symeigLanczos(op, k, cuda, extreme="min", sparse=True, dim=N, q0=b)
    finally:
        engine.LANCZOS_PERSIST = "force"
    ws = engine.Workspace.get(N, k, cuda)
    ws.lanczos_persist_lost = False
    _lib.check(lib.dsea_ws_set_fault_injection(ws.handle, 1), "dsea_ws_set_fault_injection")
    try:
        t0 = time.time()
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            lo, v = symeigLanczos(op, k, cuda, extreme="min", sparse=True, dim=N, q0=b)
        assert any("single-launch Lanczos timed out" in str(w.message) for w in rec)
        assert 2.0 < time.time() - t0 < 20.0
        assert lo.item() == ref_lo.item() and torch.equal(v, ref_v)
        assert ws.lanczos_persist_lost and ws.lanczos_persist_mode == 0
    finally:
        lib.dsea_ws_set_fault_injection(ws.handle, 0)
        ws.lanczos_persist_lost = False


# ------------------------------------------------------------------ single-launch CG for the TFIM operator (README sizes)
@pytest.mark.parametrize("L", [1, 2, 3, 6, 7, 8, 10, 11, 12, 13])
def test_persistent_tfim_cg_matches_streaming_form_and_oracle(L):
    """csrc/dsea_cg_persist_tfim.hip (n = 2^L <= 8192: one launch per solve, x / r / d in registers, two grid
    exchanges per iteration) against the 3-launches-per-iteration kernels and the CPU oracle on the shifted system
    (A - s) x = b of the adjoint solve (CG.py:120): fixed-iteration iterates to rounding, converged runs with the same
    iteration count and a residual below eps."""
    from dominantsparseeigenad_amd.operators import TFIMOperator
    n = 1 << L
    op = TFIMOperator(L, cuda, g=torch.tensor([1.05], dtype=F64, device=cuda))
    b = torch.from_numpy(normal_vector(n, 300 + L)).to(cuda)
    x0 = torch.from_numpy(normal_vector(n, 400 + L)).to(cuda)
    shift = torch.tensor(-1.3 * L - 2.0, dtype=F64, device=cuda)       # below the spectrum: A - s is SPD
    iters = min(25, n)
    ref = _solve(op, b, x0, shift, 0, eps=0.0, maxiter=iters)
    got = _solve(op, b, x0, shift, -1, eps=0.0, maxiter=iters)
    assert got[1] == ref[1] == iters
    scale = float(ref[0].abs().max())
    assert float((got[0] - ref[0]).abs().max()) < 1e-12 * scale, float((got[0] - ref[0]).abs().max()) / scale
    assert abs(got[2] - ref[2]) <= 1e-9 * max(ref[2], 1e-300) + 1e-18
    # converged runs
    ref = _solve(op, b, x0, shift, 0, eps=1e-9, maxiter=None)
    got = _solve(op, b, x0, shift, -1, eps=1e-9, maxiter=None)
    assert ref[3] and got[3] and abs(got[1] - ref[1]) <= 1 and got[2] < 1e-9
    assert float((got[0] - ref[0]).abs().max()) < 1e-8 * scale
    res = op.H(got[0]) - shift * got[0] - b
    assert float(res.norm()) < 1e-8
    # CPU oracle (gather-table operator), same start vector
    model = oracle.TFIMTables(L, g=torch.tensor([1.05], dtype=F64))
    s_host = float(shift)
    xo = oracle.cg_solve(lambda v: model.H(v) - s_host * v, b.cpu(), x0.cpu(), sparse=True, eps=0.0, maxiter=iters)
    gi = _solve(op, b, x0, shift, -1, eps=0.0, maxiter=iters)
    assert float((gi[0].cpu() - xo).abs().max()) < 1e-11 * float(xo.abs().max())


def test_persistent_tfim_cg_without_shift_and_zero_rhs_and_maxiter_zero():
    from dominantsparseeigenad_amd.operators import TFIMOperator
    L = 9
    n = 1 << L
    op = TFIMOperator(L, cuda, g=torch.tensor([0.7], dtype=F64, device=cuda))
    b = torch.from_numpy(normal_vector(n, 11)).to(cuda)
    x0 = torch.from_numpy(normal_vector(n, 12)).to(cuda)
    # no shift: H is indefinite; a fixed number of iterations still has to agree with the streaming kernels
    ref = _solve(op, b, x0, None, 0, eps=0.0, maxiter=12)
    got = _solve(op, b, x0, None, -1, eps=0.0, maxiter=12)
    assert got[1] == ref[1] == 12 and float((got[0] - ref[0]).abs().max()) < 1e-11 * float(ref[0].abs().max())
    # early out: x0 already solves the system (CG.py:28-29)
    shift = torch.tensor(-20.0, dtype=F64, device=cuda)
    bb = op.H(x0) - shift * x0
    got = _solve(op, bb, x0, shift, -1, eps=1e-7, maxiter=None)
    assert got[1] == 0 and got[3] and torch.equal(got[0], x0)
    # maxiter = 0: not converged, x untouched
    got = _solve(op, b, x0, shift, -1, eps=1e-30, maxiter=0)
    assert got[1] == 0 and not got[3] and torch.equal(got[0], x0)


# ------------------------------------------------------------------ single-launch TFIM CG at 2^11 ... 2^20 rows
@pytest.mark.parametrize("L", [11, 12, 13, 14, 15, 17, 19, 20])
def test_persistent_tfim_cg_large_is_bit_identical_to_streaming_form(L):
    """csrc/dsea_cg_persist_tfim_big.hip, TWO-exchange form (dsea_ws_set_persist(200); x / r / d in registers for the whole
    solve, d exchanged through device-coherent buffer stores / loads, partial sums reproduced per mat-vec tile and per
    512-row tile in the streaming kernels' order): the iterates are BIT-IDENTICAL to the mat-vec + update + direction
    launches (torch.equal), with the same iteration counts and residual norms -- fixed-iteration and converged runs, with
    and without shift.  L = 20 is BASELINE configs[1]'s adjoint solve."""
    from dominantsparseeigenad_amd.operators import TFIMOperator
    n = 1 << L
    op = TFIMOperator(L, cuda, g=torch.tensor([1.0], dtype=F64, device=cuda))
    b = torch.from_numpy(normal_vector(n, 500 + L)).to(cuda)
    x0 = torch.from_numpy(normal_vector(n, 600 + L)).to(cuda)
    shift = torch.tensor(-1.3 * L - 1.0, dtype=F64, device=cuda)       # below the spectrum: A - s is SPD
    for sh in (shift, None):
        ref = _solve(op, b, x0, sh, 0, eps=0.0, maxiter=30)
        got = _solve(op, b, x0, sh, 200, eps=0.0, maxiter=30)
        assert got[1] == ref[1] == 30 and got[2] == ref[2], (got[1:], ref[1:])
        assert torch.equal(got[0], ref[0]), float((got[0] - ref[0]).abs().max())
    ref = _solve(op, b, x0, shift, 0, eps=1e-8, maxiter=None)
    got = _solve(op, b, x0, shift, 200, eps=1e-8, maxiter=None)
    assert ref[3] and got[3] and got[1] == ref[1] and got[2] == ref[2] and torch.equal(got[0], ref[0])
    # early out and maxiter = 0
    bb = op.H(x0) - shift * x0
    got = _solve(op, bb, x0, shift, 200, eps=1e-6, maxiter=None)
    assert got[1] == 0 and got[3] and torch.equal(got[0], x0)
    got = _solve(op, b, x0, shift, 200, eps=1e-30, maxiter=0)
    assert got[1] == 0 and not got[3] and torch.equal(got[0], x0)


@pytest.mark.parametrize("L", [11, 12, 13, 14, 15, 17, 19, 20])
def test_persistent_tfim_cg_large_one_exchange_form_follows_the_reference_iteration(L):
    """The DEFAULT single-launch form at 2^11 ... 2^20 rows makes ONE grid-wide exchange per iteration (Chronopoulos-Gear
    recurrences: gamma = r.r and delta = r.A'r reduced together, s = A'p carried by a recurrence) -- the same iteration as
    CG.py:31-40 in exact arithmetic, not its rounding sequence.  Held to the streaming kernels (which reproduce CG.py's
    recurrences): the first 50 iterates to 1e-11 relative, residual norms alike, converged runs with the same number of
    iterations (+-1) and a true residual below eps; to the CPU ORACLE at L = 14: 50 iterates at 1e-10; early-out paths."""
    from dominantsparseeigenad_amd.operators import TFIMOperator
    n = 1 << L
    op = TFIMOperator(L, cuda, g=torch.tensor([1.0], dtype=F64, device=cuda))
    b = torch.from_numpy(normal_vector(n, 500 + L)).to(cuda)
    x0 = torch.from_numpy(normal_vector(n, 600 + L)).to(cuda)
    shift = torch.tensor(-1.3 * L - 1.0, dtype=F64, device=cuda)       # below the spectrum: A - s is SPD
    # (without the shift the operator is indefinite: CG is then no convergent process and any two rounding-different
    #  evaluations drift apart after a few iterations -- only the first steps are compared there)
    for sh in (shift, None):
        for its in ((1, 2, 50) if sh is not None else (1, 2, 5)):
            ref = _solve(op, b, x0, sh, 0, eps=0.0, maxiter=its)
            got = _solve(op, b, x0, sh, -1, eps=0.0, maxiter=its)
            scale = float(ref[0].abs().max())
            assert got[1] == ref[1] == its
            assert float((got[0] - ref[0]).abs().max()) < 1e-11 * scale, (its, float((got[0] - ref[0]).abs().max()) / scale)
            assert abs(got[2] - ref[2]) <= 1e-8 * ref[2]
    for eps in (1e-8, 1e-12):
        ref = _solve(op, b, x0, shift, 0, eps=eps, maxiter=None)
        got = _solve(op, b, x0, shift, -1, eps=eps, maxiter=None)
        assert ref[3] and got[3] and abs(got[1] - ref[1]) <= 1 and got[2] < eps
        res = op.H(got[0]) - shift * got[0] - b
        assert float(res.norm()) < 10 * eps * max(1.0, float(b.norm()) * 1e-3) + 1e-11
        assert float((got[0] - ref[0]).abs().max()) < 100 * eps + 1e-12 * float(ref[0].abs().max())
    again = _solve(op, b, x0, shift, -1, eps=1e-12, maxiter=None)
    assert torch.equal(again[0], got[0]) and again[1] == got[1]            # deterministic: fixed summation orders
    # early out and maxiter = 0
    bb = op.H(x0) - shift * x0
    e0 = _solve(op, bb, x0, shift, -1, eps=1e-6, maxiter=None)
    assert e0[1] == 0 and e0[3] and torch.equal(e0[0], x0)
    e1 = _solve(op, b, x0, shift, -1, eps=1e-30, maxiter=0)
    assert e1[1] == 0 and not e1[3] and torch.equal(e1[0], x0)
    if L == 14:
        model = oracle.TFIMTables(L, g=torch.tensor([1.0], dtype=F64))
        st = {}
        xo = oracle.cg_solve(lambda v: model.H(v) - shift.cpu() * v, b.cpu(), x0.cpu(), sparse=True, eps=0.0, maxiter=50, stats=st)
        got = _solve(op, b, x0, shift, -1, eps=0.0, maxiter=50)
        assert got[1] == st["iters"] == 50
        assert float((got[0].cpu() - xo).abs().max()) <= 1e-10 * float(xo.abs().max())


# ------------------------------------------------------------------ a lost peer workgroup: time out, fall back, stay there
def test_lost_peer_times_out_and_the_host_falls_back_to_the_multi_launch_kernels():
    """The persistent single-launch forms spin on each other; every spin is bounded (3 s).  With the test hook
    dsea_ws_set_fault_injection the last workgroup of a launch exits at once: the launch must END (not hang the GPU),
    report DSEA_ERR_TIMEOUT, and the host must repeat the solve on the multi-launch kernels -- same result as if the
    single-launch form had never been tried -- warn, and keep that workspace on the multi-launch kernels."""
    import time
    import warnings
    from dominantsparseeigenad_amd import _lib
    from dominantsparseeigenad_amd.Lanczos import symeigLanczos
    from dominantsparseeigenad_amd.operators import TFIMOperator
    lib = _lib.load()
    # -- single-launch Lanczos (n = 1024: 8 workgroups)
    L, k = 10, 60
    n = 1 << L
    op = TFIMOperator(L, cuda, g=torch.tensor([1.0], dtype=F64, device=cuda))
    q0 = torch.from_numpy(normal_vector(n, 21)).to(cuda)
    engine.LANCZOS_PERSIST = False
    try:
        ref_lo, ref_v = symeigLanczos(op, k, cuda, extreme="min", sparse=True, dim=n, q0=q0)
    finally:
        engine.LANCZOS_PERSIST = True
    ws = engine.Workspace.get(n, k, cuda)
    ws.lanczos_persist_lost = False
    _lib.check(lib.dsea_ws_set_fault_injection(ws.handle, 1), "dsea_ws_set_fault_injection")
    try:
        t0 = time.time()
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            lo, v = symeigLanczos(op, k, cuda, extreme="min", sparse=True, dim=n, q0=q0)
        waited = time.time() - t0
        assert any("single-launch Lanczos timed out" in str(w.message) for w in rec)
        assert 0.2 < waited < 20.0, waited        # (0.3 s spin bound of the README-sized form)
        assert lo.item() == ref_lo.item() and torch.equal(v, ref_v)
        assert ws.lanczos_persist_lost and ws.lanczos_persist_mode == 0
        t0 = time.time()
        symeigLanczos(op, k, cuda, extreme="min", sparse=True, dim=n, q0=q0)      # sticky: no second timeout
        assert time.time() - t0 < 1.0
    finally:
        lib.dsea_ws_set_fault_injection(ws.handle, 0)
        ws.lanczos_persist_lost = False
    # -- single-launch CG: both TFIM forms (n = 1024: 8 workgroups; n = 2^15: 16 workgroups) and the stencil form
    for Lc in (10, 15, "stencil"):
        if Lc == "stencil":
            nc = 20000
            opc = _problem(nc, seed=90)[0]
        else:
            nc = 1 << Lc
            opc = TFIMOperator(Lc, cuda, g=torch.tensor([1.0], dtype=F64, device=cuda))
        b = torch.from_numpy(normal_vector(nc, 31)).to(cuda)
        x0 = torch.from_numpy(normal_vector(nc, 32)).to(cuda)
        shift = torch.tensor(-40.0, dtype=F64, device=cuda)
        ref = _solve(opc, b, x0, shift, 0, eps=0.0, maxiter=40)
        wsc = engine.Workspace.get(nc, 8, cuda)
        _lib.check(lib.dsea_ws_set_fault_injection(wsc.handle, 1), "dsea_ws_set_fault_injection")
        try:
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                x = engine.cg(b, x0, native=opc, shift=shift, eps=0.0, maxiter=40)
            assert any("persistent CG launch timed out" in str(w.message) for w in rec)
            assert engine.last_cg.iters == ref[1] == 40
            assert torch.equal(x, ref[0])
            assert wsc.persist_mode == 0                                              # sticky
            assert engine.last_cg.form == "streaming"                                 # (dsea_cg_last_form: the caller can tell)
        finally:
            lib.dsea_ws_set_fault_injection(wsc.handle, 0)
            wsc.set_persist(-1)
