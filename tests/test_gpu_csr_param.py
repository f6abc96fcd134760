"""The explicit-matrix operand as a PARAMETER of the primitives (reference README.md:88-126, symeig.py:29,56-64,82-84:
the adjoint A-bar = v1 v2^T is pushed to the parameters that produced A; for a sparse A whose parameters are its
stored non-zeros that is the sampled outer product vals-bar[e] = v1[row e] v2[col e]).

  * dsea_op_sddmm / dsea_op_update_vals against torch index arithmetic, bit for bit (ragged rows, both layouts,
    32-bit and 16-bit columns);
  * d E0 / d vals and d (E0 + psi.t) / d vals through DominantSparseSymeig + CSROperator.Aadjoint_to_valsadjoint against
    (i) the dense primitive's A-bar (symeig.py:29) sampled on the pattern and (ii) torch.linalg.eigh autograd on the
    symmetrised dense matrix, at 1e-10 with the CG tolerance at 1e-12 (tests/test_gpu_parity.py explains why);
  * second order once; update-then-solve == rebuild-then-solve bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import PatchRandn, unit, rel, banded_spd as _banded_spd, eigh_reference as _eigh_reference  # noqa: E402
from dominantsparseeigenad_amd import engine  # noqa: E402
from dominantsparseeigenad_amd.operators import TFIMOperator, CSROperator  # noqa: E402
from dominantsparseeigenad_amd.synthetic import normal_vector  # noqa: E402
import dominantsparseeigenad_amd.symeig as symeig  # noqa: E402
import dominantsparseeigenad_amd.CG as CG  # noqa: E402

F64 = torch.float64
TOL = 1e-10


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X (no fallback)"
    return torch.device("cuda:0")


def _ragged(n=1037, seed=3, density=0.02):
    import scipy.sparse as sp
    rng = np.random.RandomState(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="lil")
    M[5, :] = 0
    M[:, 5] = 0                                      # an empty row / column
    M = sp.csr_matrix(M)
    M = (M + M.T).tocsr()
    M.sort_indices()
    return M


def _rows_of(M):
    return np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))


@pytest.mark.parametrize("layout,col16", [("sell", "auto"), ("sell", False), ("csr", False)])
def test_sddmm_and_update_kernels_bitwise(layout, col16):
    M = _ragged()
    n = M.shape[0]
    op = CSROperator.from_scipy(M, dev(), layout=layout, col16=col16)
    assert op.col16 == (layout == "sell" and col16 == "auto")
    rows, cols = torch.from_numpy(_rows_of(M)), torch.from_numpy(M.indices.astype("int64"))
    v1, v2 = torch.from_numpy(normal_vector(n, 7001)), torch.from_numpy(normal_vector(n, 7002))
    plain = v1[rows] * v2[cols]
    assert torch.equal(op.sddmm(v1.to(dev()), v2.to(dev())).cpu(), plain)
    sym = 0.5 * (v1[rows] * v2[cols] + v1[cols] * v2[rows])
    assert torch.equal(op.sddmm(v1.to(dev()), v2.to(dev()), symmetric=True).cpu(), sym)
    out = torch.from_numpy(normal_vector(M.nnz, 7003)).to(dev())
    base = out.cpu().clone()
    op.sddmm(v1.to(dev()), v2.to(dev()), out=out, alpha=-0.75, accumulate=True)
    assert torch.equal(out.cpu(), base + (-0.75) * plain)
    # in-place refresh == rebuild, bit for bit; the SELL padding stays zero
    x = torch.from_numpy(normal_vector(n, 7004)).to(dev())
    newv = torch.from_numpy(normal_vector(M.nnz, 7005)).to(dev())
    op.vals.copy_(newv)                               # in place: picked up through the tensor's version counter
    rebuilt = CSROperator(op.rowptr, op.colidx, newv.clone(), n, layout=layout, col16=col16)
    assert torch.equal(op(x), rebuilt(x))
    M2 = M.copy()
    M2.data = newv.cpu().numpy()
    assert rel(op(x).cpu(), torch.from_numpy(M2 @ x.cpu().numpy())) < 1e-13
    # binding ANOTHER tensor (``op.vals = new``: what a model that recomputes its matrix entries every step does)
    third = torch.from_numpy(normal_vector(M.nnz, 7006)).to(dev())
    op.vals = third
    assert torch.equal(op(x), CSROperator(op.rowptr, op.colidx, third.clone(), n, layout=layout, col16=col16)(x))
    with pytest.raises(ValueError):
        op.vals = third[:-1]


def test_sell16_falls_back_when_a_slice_column_is_too_wide():
    """a slice column spanning >= 65536 columns cannot use 16-bit deltas: 'auto' keeps 32-bit columns, True refuses"""
    import scipy.sparse as sp
    n = 70000
    rows = np.array([0, 1, 69999, 69998], dtype=np.int64)
    cols = np.array([69999, 1, 0, 69998], dtype=np.int64)
    M = sp.csr_matrix((np.ones(4), (rows, cols)), shape=(n, n))
    op = CSROperator.from_scipy(M, dev())
    assert not op.col16
    x = torch.from_numpy(normal_vector(n, 7010)).to(dev())
    assert rel(op(x).cpu(), torch.from_numpy(M @ x.cpu().numpy())) < 1e-14
    with pytest.raises(ValueError):
        CSROperator.from_scipy(M, dev(), col16=True)


@pytest.mark.parametrize("case", ["tfim-L12", "banded-spd"])
def test_gradient_wrt_nonzeros_vs_dense_primitive_and_eigh(monkeypatch, case):
    monkeypatch.setattr(CG, "EPS_DEFAULT", 1e-12)
    if case == "tfim-L12":
        L = 12
        n, k = 1 << L, 120
        src = TFIMOperator(L, dev(), g=torch.tensor([1.0], dtype=F64, device=dev())).to_csr()
        rowptr, colidx, vals0 = src.rowptr, src.colidx, src.vals.clone()
    else:
        n, k = 3000, 200
        M = _banded_spd(n, 9, 11)
        rowptr = torch.from_numpy(M.indptr.astype("int64")).to(dev())
        colidx = torch.from_numpy(M.indices.astype("int32")).to(dev())
        vals0 = torch.from_numpy(M.data.copy()).to(dev())
    vals = vals0.clone().requires_grad_(True)
    op = CSROperator(rowptr, colidx, vals, n)
    assert op.vals is vals
    t = unit(n, 7100).to(dev())
    rows = torch.repeat_interleave(torch.arange(n, device=dev()), rowptr[1:] - rowptr[:-1])
    cols = colidx.long()

    # (i) the dense primitive on the same matrix: its A-bar = v1 v2^T (symeig.py:29) sampled on the pattern
    Ad = torch.zeros((n, n), dtype=F64, device=dev()).index_put((rows, cols), vals0, accumulate=True).requires_grad_(True)
    with PatchRandn(7200):
        E0d, psid = symeig.DominantSymeig.apply(Ad, k)
        (gAd,) = torch.autograd.grad(E0d + psid @ t, Ad)
    dense_sampled = gAd[rows, cols]

    symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint)
    with PatchRandn(7200):
        E0, psi = symeig.DominantSparseSymeig.apply(op.vals, k, n)
        (g_plain,) = torch.autograd.grad(E0 + psi @ t, op.vals)
    assert engine.last_cg.converged and engine.last_cg.resnorm < 1e-12
    assert abs(E0.item() - E0d.item()) < 1e-12 * abs(E0d.item())
    sgn = 1.0 if float(psi.detach() @ psid.detach()) > 0 else -1.0
    assert sgn == 1.0                                   # same start vector, same algorithm: same sign
    scale = float(dense_sampled.abs().max())
    assert float((g_plain - dense_sampled).abs().max()) < TOL * scale, float((g_plain - dense_sampled).abs().max()) / scale

    # (ii) torch.linalg.eigh autograd on the symmetrised dense matrix: the tied-pair adjoint
    symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint_symmetric)
    for w_psi in (0.0, 1.0):
        with PatchRandn(7300):
            E0, psi = symeig.DominantSparseSymeig.apply(op.vals, k, n)
            loss = E0 + w_psi * (psi @ t) if w_psi else E0
            (g_sym,) = torch.autograd.grad(loss, op.vals)
        E_ref, psi_ref, g_ref = _eigh_reference(rowptr, colidx, vals0, n, t, 1.0, w_psi, psi_like=psi.detach(),
                                                autograd=(case == "banded-spd"))
        if case == "banded-spd":                        # the closed form used for the TFIM case, checked where autograd works
            _, _, g_pt = _eigh_reference(rowptr, colidx, vals0, n, t, 1.0, w_psi, psi_like=psi.detach(), autograd=False)
            assert float((g_pt - g_ref).abs().max()) < 1e-11 * float(g_ref.abs().max())
        assert abs(E0.item() - E_ref.item()) < 1e-12 * abs(E_ref.item())
        assert rel(psi.detach().cpu(), psi_ref) < 1e-9
        scale = float(g_ref.abs().max())
        err = float((g_sym.cpu() - g_ref).abs().max()) / scale
        assert err < TOL, (case, w_psi, err)
        print("%s: d(E0 + %g psi.t)/d vals vs eigh autograd: max abs err / max |g| = %.2e over %d non-zeros"
              % (case, w_psi, err, g_ref.numel()))


def test_second_order_through_the_nonzeros(monkeypatch):
    """d/d vals of (dE0/d vals . w): the hooks are differentiable (their backward is a mat-vec with the incoming gradient as
    the values of the same pattern), as the reference's torch-code hooks are (examples/TFIM/E0.py:63-64)"""
    monkeypatch.setattr(CG, "EPS_DEFAULT", 1e-12)
    n, k = 400, 400
    M = _banded_spd(n, 4, 5)
    rowptr = torch.from_numpy(M.indptr.astype("int64")).to(dev())
    colidx = torch.from_numpy(M.indices.astype("int32")).to(dev())
    vals = torch.from_numpy(M.data.copy()).to(dev()).requires_grad_(True)
    op = CSROperator(rowptr, colidx, vals, n)
    symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint_symmetric)
    w = torch.from_numpy(normal_vector(M.nnz, 7400)).to(dev())
    with PatchRandn(7500):
        E0, _ = symeig.DominantSparseSymeig.apply(op.vals, k, n)
        (g1,) = torch.autograd.grad(E0, op.vals, create_graph=True)
        (g2,) = torch.autograd.grad(g1 @ w, op.vals)
    # reference: eigh double backward on the symmetrised dense matrix
    v = vals.detach().cpu().clone().requires_grad_(True)
    rows = torch.repeat_interleave(torch.arange(n), rowptr.cpu()[1:] - rowptr.cpu()[:-1])
    A = torch.zeros((n, n), dtype=F64).index_put((rows, colidx.cpu().long()), v, accumulate=True)
    lam, _ = torch.linalg.eigh(0.5 * (A + A.T))
    (r1,) = torch.autograd.grad(lam[0], v, create_graph=True)
    (r2,) = torch.autograd.grad(r1 @ w.cpu(), v)
    assert float((g1.detach().cpu() - r1.detach()).abs().max()) < TOL * float(r1.detach().abs().max())
    err = float((g2.cpu() - r2).abs().max()) / float(r2.abs().max())
    assert err < 1e-8, err
    print("second order through the non-zeros: max abs err / max = %.2e" % err)


def test_optimiser_loop_update_then_solve_equals_rebuild_then_solve():
    """an optimiser steps the non-zeros in place; the next solve sees them (dsea_op_update_vals through the stored map,
    no rebuild) and equals a solve on an operator rebuilt from the new values, bit for bit"""
    n, k = 3000, 64
    M = _banded_spd(n, 6, 21)
    rowptr = torch.from_numpy(M.indptr.astype("int64")).to(dev())
    colidx = torch.from_numpy(M.indices.astype("int32")).to(dev())
    vals = torch.from_numpy(M.data.copy()).to(dev()).requires_grad_(True)
    op = CSROperator(rowptr, colidx, vals, n)
    symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint_symmetric)
    opt = torch.optim.SGD([vals], lr=0.05)
    E_hist = []
    for it in range(3):
        opt.zero_grad()
        with PatchRandn(7600 + it):
            E0, psi = symeig.DominantSparseSymeig.apply(op.vals, k, n)
        E0.backward()
        E_hist.append(E0.item())
        opt.step()                                      # in place on vals
    assert E_hist[2] < E_hist[1] < E_hist[0]            # descending the smallest eigenvalue
    with PatchRandn(7700):
        E_upd, psi_upd = symeig.DominantSparseSymeig.apply(op.vals, k, n)
    fresh = CSROperator(rowptr, colidx, vals.detach().clone(), n)
    symeig.setDominantSparseSymeig(fresh, fresh.Aadjoint_to_valsadjoint_symmetric)
    with PatchRandn(7700):
        E_new, psi_new = symeig.DominantSparseSymeig.apply(fresh.vals, k, n)
    assert E_upd.item() == E_new.item()
    assert torch.equal(psi_upd, psi_new)


def test_sell_kernel_beyond_one_trip_and_beyond_64_slice_columns():
    """two geometry edges of k_spmv_sell that the headline size (exactly 4096 blocks, 21 slice columns) does not reach:
    (a) more than 4096 blocks' worth of slices -- blocks walk several chunks, with the XCD-contiguous map on and off;
    (b) a slice wider than 64 columns -- the lane-held column bases of the 16-bit layout are reloaded per 64 columns."""
    import scipy.sparse as sp
    from dominantsparseeigenad_amd import _lib
    rng = np.random.RandomState(17)
    n = (1 << 20) + (1 << 18) + 37                       # 20481 slices -> 5121 chunks of four > 4096 blocks
    offs = [-700, -3, -1, 0, 1, 3, 700]
    M = sp.diags([rng.randn(n - abs(o)) for o in offs], offs, shape=(n, n), format="csr")
    M.sort_indices()
    x = torch.from_numpy(normal_vector(n, 7800)).to(dev())
    ref = torch.from_numpy(M @ x.cpu().numpy())
    for col16 in ("auto", False):
        op = CSROperator.from_scipy(M, dev(), col16=col16)
        for xcd in (1, 0):
            _lib.check(_lib.load().dsea_op_set_tuning(op._H.handle, _lib.TUNE_SELL_XCD_MAP, xcd), "dsea_op_set_tuning")
            assert rel(op(x).cpu(), ref) < 1e-13, (col16, xcd)
    # (b) 300 rows, rows 70 and 130 hold 150 / 200 non-zeros: their slices are 150 / 200 columns wide
    n2 = 300
    D = sp.lil_matrix((n2, n2))
    D.setdiag(1.0 + rng.rand(n2))
    for r, cnt in ((70, 150), (130, 200)):
        cols = rng.choice(n2, size=cnt, replace=False)
        D[r, cols] = rng.randn(cnt)
    D = sp.csr_matrix(D)
    D.sort_indices()
    x2 = torch.from_numpy(normal_vector(n2, 7801)).to(dev())
    ref2 = torch.from_numpy(D @ x2.cpu().numpy())
    for col16 in ("auto", False):
        op2 = CSROperator.from_scipy(D, dev(), col16=col16)
        assert op2.col16 == (col16 == "auto")
        assert rel(op2(x2).cpu(), ref2) < 1e-13, col16
        rows = torch.from_numpy(_rows_of(D))
        v1 = torch.from_numpy(normal_vector(n2, 7802))
        assert torch.equal(op2.sddmm(v1.to(dev()), x2).cpu(), v1[rows] * x2.cpu()[torch.from_numpy(D.indices.astype("int64"))])


def test_example_sparse_matrix_parameters_lbfgs_follows_the_dense_eigh_loop(monkeypatch):
    """examples/sparse_matrix_parameters.py (the inverse problem of reference examples/schrodinger1D.py:101-127 with the
    Hamiltonian's stored non-zeros as the parameters): first loss and gradient against torch.linalg.eigh autograd on the
    symmetrised dense matrix; three LBFGS steps -- in-place updates of the leaf, followed through its version counter --
    against the same optimiser driven by dense eigh from the same start"""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "examples", "sparse_matrix_parameters.py")
    spec = importlib.util.spec_from_file_location("sparse_matrix_parameters", path)
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    monkeypatch.setattr(CG, "EPS_DEFAULT", 1e-12)
    N, k = 200, 200
    xm = np.linspace(-1.0, 1.0, num=N, endpoint=False)
    xmesh = torch.from_numpy(xm).to(dev())
    target = torch.from_numpy(ex.target_wavefunction(xm)).to(dev())
    model = ex.SparseHamiltonian(-1.0, 1.0, N, xmesh)
    start = model.vals.detach().clone()

    def dense_loss(vals):
        lam, U = torch.linalg.eigh(model.dense(vals).cpu())
        return 1.0 - (U[:, 0].abs() * target.cpu()).sum()

    loss = model.forward_sparseAD(target, k)
    (g,) = torch.autograd.grad(loss, model.vals)
    ref_vals = start.clone().requires_grad_(True)
    ref_loss = dense_loss(ref_vals)
    (g_ref,) = torch.autograd.grad(ref_loss, ref_vals)
    assert abs(loss.item() - ref_loss.item()) < 1e-11
    assert float((g.cpu() - g_ref.cpu()).abs().max()) < TOL * float(g_ref.abs().max())

    def run(params, loss_of):
        opt = torch.optim.LBFGS([params], max_iter=10, tolerance_change=1e-7, tolerance_grad=1e-7, line_search_fn="strong_wolfe")

        def closure():
            opt.zero_grad()
            value = loss_of()
            value.backward()
            return value
        return [opt.step(closure).item() for _ in range(3)]

    ours = run(model.vals, lambda: model.forward_sparseAD(target, k))
    ref_vals = start.clone().requires_grad_(True)
    theirs = run(ref_vals, lambda: dense_loss(ref_vals))
    print("losses:", ours, "dense eigh:", theirs)
    assert ours[-1] < ours[0]
    for a, b in zip(ours, theirs):
        assert abs(a - b) < 1e-7 * max(1.0, abs(b)), (ours, theirs)
    # (the parameters themselves are NOT compared: psi is invariant under H -> c H and nearly so along other directions, and
    #  LBFGS wanders along them with the rounding -- measured 1e-3 relative between the two loops at equal losses)


@pytest.mark.parametrize("layout", ["sell", "csr"])
def test_operator_without_a_stored_entry(layout):
    """the zero matrix as an operand (a row slab that is all padding is one): mat-vec = 0, the sampled outer product and the
    value refresh are no-ops on empty arrays (torch reports a null pointer for an empty tensor; the C ABI refuses those)"""
    n = 130
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev())
    vals = torch.zeros(0, dtype=F64, device=dev(), requires_grad=True)
    op = CSROperator(rowptr, torch.zeros(0, dtype=torch.int32, device=dev()), vals, n, layout=layout)
    x = torch.from_numpy(normal_vector(n, 1)).to(dev())
    assert float(op(x).abs().max()) == 0.0
    assert op.sddmm(x, x).numel() == 0 and op.Aadjoint_to_valsadjoint_symmetric(x, x).numel() == 0
    op.refresh()


def _few_valued(n=1037, seed=5, levels=7):
    """ragged symmetric pattern (an empty row, n not a multiple of 64) whose entries take a handful of values, +0.0 and -0.0
    among them"""
    M = _ragged(n, seed)
    rng = np.random.RandomState(seed)
    table = np.concatenate([rng.randn(levels), [0.0, -0.0]])
    M.data = table[rng.randint(0, len(table), size=M.nnz)]
    return M


def test_value_coded_operand_is_bit_identical_to_the_fp64_value_operand(monkeypatch):
    """dsea_op_create_sell16v8 (8-bit value codes into a table of 256 doubles, metadata packed four slice columns to a lane)
    against dsea_op_create_sell16 on the same matrix: mat-vec (+ shift, + x.y), sampled outer product, and a whole
    DominantSparseSymeig forward + backward (fused Lanczos tail, CG) -- equal bit for bit"""
    from dominantsparseeigenad_amd.engine import Workspace
    M = _few_valued()
    n = M.shape[0]
    coded = CSROperator.from_scipy(M, dev(), values="coded")
    plain = CSROperator.from_scipy(M, dev(), values="plain")
    auto = CSROperator.from_scipy(M, dev())
    assert coded._coded and auto._coded and not plain._coded and coded.col16
    x = torch.from_numpy(normal_vector(n, 7900)).to(dev())
    y = coded(x)
    assert torch.equal(y, plain(x)) and torch.equal(y, auto(x))
    assert rel(y.cpu(), torch.from_numpy(M @ x.cpu().numpy())) < 1e-13
    # signed zeros survive the coding (the codes are taken from the bit patterns)
    assert torch.equal(coded._vtab[coded._codes.long()].view(torch.int64).sort().values.unique(),
                       torch.cat([torch.from_numpy(M.data), torch.zeros(1, dtype=F64)]).view(torch.int64).unique().to(dev()))
    v1 = torch.from_numpy(normal_vector(n, 7901)).to(dev())
    for sym in (False, True):
        assert torch.equal(coded.sddmm(v1, x, symmetric=sym), plain.sddmm(v1, x, symmetric=sym))
    # the TFIM matrix (21 per row at L = 12: the slice width 13 is padded to 16) through the whole primitive
    monkeypatch.setattr(CG, "EPS_DEFAULT", 1e-12)
    L, k = 12, 120
    tf = TFIMOperator(L, dev(), g=torch.tensor([1.0], dtype=F64, device=dev()))
    res = []
    for values in ("coded", "plain"):
        op = tf.to_csr(values=values)
        assert op._coded == (values == "coded")
        vals = op.vals.requires_grad_(True)            # (after the layout was chosen: 'auto' would keep a parameter uncoded)
        symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint_symmetric)
        t = unit(1 << L, 7902).to(dev())
        with PatchRandn(7903):
            E0, psi = symeig.DominantSparseSymeig.apply(vals, k, 1 << L, dev())
            (gv,) = torch.autograd.grad(E0 + psi @ t, vals)
        res.append((E0.detach().clone(), psi.detach().clone(), gv.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_value_coded_operand_selection_recoding_and_fallback():
    """'auto' codes a matrix of few distinct values unless it is a parameter; an in-place change that keeps the values few is
    re-coded in place (same handle), one that does not rebuilds the fp64 layout once -- each equal to a fresh operator"""
    M = _few_valued(n=500, seed=9)
    n = M.shape[0]
    x = torch.from_numpy(normal_vector(n, 7910)).to(dev())
    rp, ci = torch.from_numpy(M.indptr.astype("int64")).to(dev()), torch.from_numpy(M.indices.astype("int32")).to(dev())
    leaf = torch.from_numpy(M.data.copy()).to(dev()).requires_grad_(True)
    assert not CSROperator(rp, ci, leaf, n)._coded                               # a parameter: fp64 values
    assert CSROperator(rp, ci, leaf, n, values="coded")._coded                   # ... unless asked for
    many = torch.from_numpy(normal_vector(M.nnz, 7911)).to(dev())
    assert not CSROperator(rp, ci, many, n)._coded
    with pytest.raises(ValueError):
        CSROperator(rp, ci, many, n, values="coded")
    vals = torch.from_numpy(M.data.copy()).to(dev())
    op = CSROperator(rp, ci, vals, n)
    handle = op.handle.value
    with torch.no_grad():
        vals.mul_(-2.5)                                                          # still few values
    assert torch.equal(op(x), CSROperator(rp, ci, vals.clone(), n, values="plain")(x))
    assert op._coded and op.handle.value == handle
    with torch.no_grad():
        vals.copy_(many)                                                         # no longer codable
    assert torch.equal(op(x), CSROperator(rp, ci, many.clone(), n, values="plain")(x))
    assert not op._coded
    with torch.no_grad():
        vals.mul_(0.5)                                                           # from here on the ordinary refresh
    assert torch.equal(op(x), CSROperator(rp, ci, (many * 0.5), n, values="plain")(x))


def test_packed_and_unpacked_fp64_value_layouts_are_bit_identical(monkeypatch):
    """dsea_op_create_sell16p2 (the default with 16-bit deltas: values and deltas packed two slice columns to a lane, slices
    padded to even widths) against dsea_op_create_sell16 (DSEA_SELL_PACK2=0): mat-vec, in-place refresh, sampled outer
    product -- odd and even slice widths, an empty row, n not a multiple of 64"""
    M = _ragged()
    n = M.shape[0]
    monkeypatch.setenv("DSEA_SELL_PACK2", "0")
    flat = CSROperator.from_scipy(M, dev())
    monkeypatch.setenv("DSEA_SELL_PACK2", "1")
    packed = CSROperator.from_scipy(M, dev())
    assert packed._pack2 and not flat._pack2 and packed.col16 and flat.col16
    assert packed._sell_total >= flat._sell_total and packed._sell_total % 128 == 0
    x = torch.from_numpy(normal_vector(n, 7920)).to(dev())
    v1 = torch.from_numpy(normal_vector(n, 7921)).to(dev())
    assert torch.equal(packed(x), flat(x))
    for sym in (False, True):
        assert torch.equal(packed.sddmm(v1, x, symmetric=sym), flat.sddmm(v1, x, symmetric=sym))
    newv = torch.from_numpy(normal_vector(M.nnz, 7922)).to(dev())
    packed.vals.copy_(newv)
    flat.vals.copy_(newv)
    assert torch.equal(packed(x), flat(x))
    M2 = M.copy()
    M2.data = newv.cpu().numpy()
    assert rel(packed(x).cpu(), torch.from_numpy(M2 @ x.cpu().numpy())) < 1e-13
    twin = packed.with_vals(newv * 2.0)               # (the hooks' backward builds operators on the same structure)
    assert torch.equal(twin(x), CSROperator(packed.rowptr, packed.colidx, newv * 2.0, n)(x))


def test_parameter_kernels_on_slices_beyond_the_lds_cap():
    """a slice whose 64 rows hold more than 2048 non-zeros does not fit the LDS staging of dsea_op_sddmm / dsea_op_update_vals:
    they take their direct form (here 48 per row = 3072 per slice), plain, symmetric and accumulating"""
    import scipy.sparse as sp
    rng = np.random.RandomState(23)
    n = 200
    D = sp.lil_matrix((n, n))
    for r in range(n):
        cols = rng.choice(n, size=48 if r < 130 else 5, replace=False)
        D[r, cols] = rng.randn(len(cols))
    D = sp.csr_matrix(D)
    D.sort_indices()
    rows, cols = torch.from_numpy(_rows_of(D)), torch.from_numpy(D.indices.astype("int64"))
    v1, v2 = torch.from_numpy(normal_vector(n, 7930)), torch.from_numpy(normal_vector(n, 7931))
    plain = v1[rows] * v2[cols]
    sym = 0.5 * (v1[rows] * v2[cols] + v1[cols] * v2[rows])
    for col16 in ("auto", False):
        op = CSROperator.from_scipy(D, dev(), col16=col16)
        assert torch.equal(op.sddmm(v1.to(dev()), v2.to(dev())).cpu(), plain)
        assert torch.equal(op.sddmm(v1.to(dev()), v2.to(dev()), symmetric=True).cpu(), sym)
        base = torch.from_numpy(normal_vector(D.nnz, 7932))
        out = base.clone().to(dev())
        op.sddmm(v1.to(dev()), v2.to(dev()), out=out, alpha=1.25, accumulate=True)
        assert torch.equal(out.cpu(), base + 1.25 * plain)
        x = torch.from_numpy(normal_vector(n, 7933)).to(dev())
        newv = torch.from_numpy(normal_vector(D.nnz, 7934)).to(dev())
        op.vals.copy_(newv)
        D2 = D.copy()
        D2.data = newv.cpu().numpy()
        assert rel(op(x).cpu(), torch.from_numpy(D2 @ x.cpu().numpy())) < 1e-13
        assert torch.equal(op(x), CSROperator(op.rowptr, op.colidx, newv.clone(), n, col16=col16)(x))
