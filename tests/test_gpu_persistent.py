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
    assert engine.last_cg.iters == st["iters"] == 50
    assert float((x.cpu() - xo).abs().max()) <= 1e-10 * float(xo.abs().max())
    # the keyword leaves the workspace's own setting alone
    assert getattr(engine.Workspace.get(N, 8, cuda), "persist_mode", -1) == -1
