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


# ------------------------------------------------------------------ native operands + library loops
GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def _gauge(l, r):
    sgn = 1.0 if r[np.argmax(np.abs(r))] > 0 else -1.0
    return l * sgn, r * sgn


@pytest.mark.parametrize("D,d", [(3, 2), (16, 2), (33, 3), (64, 2)])
def test_transfer_and_dense_operators_match_einsum(D, d):
    """the batched-GEMM transfer mat-vec (both orientations) and the GEMV dense operand against torch einsums"""
    from dominantsparseeigenad_amd.operators import DenseOperator, TransferOperator
    rng = np.random.RandomState(D)
    A = torch.from_numpy(rng.randn(d, D, D)).to(cuda)
    v = torch.from_numpy(rng.randn(D * D)).to(cuda)
    fr = torch.einsum("kij,kmn,jn->im", A, A, v.reshape(D, D)).reshape(-1)     # general.py:59-61
    fl = torch.einsum("kij,kmn,im->jn", A, A, v.reshape(D, D)).reshape(-1)     # general.py:62-64
    scale = float(fr.abs().max())
    assert float((TransferOperator(A)(v) - fr).abs().max()) < 1e-13 * scale * D
    assert float((TransferOperator(A, transpose=True)(v) - fl).abs().max()) < 1e-13 * scale * D
    G = torch.einsum("kij,kmn->imjn", A, A).reshape(D * D, D * D)
    assert float((DenseOperator(G)(v) - G @ v).abs().max()) < 1e-13 * scale * D
    assert float((DenseOperator(G, transpose=True)(v) - G.T @ v).abs().max()) < 1e-13 * scale * D


@pytest.mark.parametrize("D,d,form", [(3, 2, "1"), (20, 2, "1"), (33, 3, "1"), (64, 1, "1"), (64, 3, "1"), (80, 2, "1"), (100, 3, "1"), (128, 3, "1"),
                                      (192, 2, "1"), (256, 1, "1"), (320, 2, "1"), (384, 2, "1"), (500, 1, "1"), (512, 2, "1"), (640, 1, "1"), (900, 1, "1"),
                                      (1024, 1, "1")])
def test_transfer_matvec_on_the_fp64_matrix_cores_matches_the_contraction_and_the_library_gemm_path(D, d, form):
    """csrc/dsea_transfer_mfma.hip (default up to D = 512, DSEA_TRANSFER_MFMA=1 forces it, =0 the library GEMMs; any D -- the reference's examples run 20 and 80 --
    zero-padded to a multiple of 64 with guarded reads of x and writes of y: two hand-written v_mfma_f64_16x16x4
    kernels -- T = [B_s] X as one stacked product, y = sum_s T_s B_s^T as ONE product over the inner dimension d D, no
    transpose, no slice sum; the waves split the inner dimension and take their fragments straight from global memory out of
    fragment-packed operands -- odd and even numbers of k blocks per wave) against the contraction of general.py:59-66 and against the default rocBLAS path, both
    orientations; an ASYMMETRIC operand so that a transposed tile or fragment cannot pass."""
    import os
    from dominantsparseeigenad_amd.operators import TransferOperator
    rng = np.random.RandomState(1000 + D)
    A = torch.from_numpy(rng.randn(d, D, D) + np.arange(D)[None, :, None] * 0.01).to(cuda)
    v = torch.from_numpy(rng.randn(D * D)).to(cuda)
    X = v.reshape(D, D)
    fr = sum(A[s] @ X @ A[s].T for s in range(d)).reshape(-1)                  # general.py:59-61
    fl = sum(A[s].T @ X @ A[s] for s in range(d)).reshape(-1)                  # general.py:62-64
    opr, opl = TransferOperator(A), TransferOperator(A, transpose=True)
    dr = opr(v).clone()                                                        # default: chosen by size
    os.environ["DSEA_TRANSFER_MFMA"] = "0"
    try:
        zr, zl = opr(v).clone(), opl(v).clone()                                # the library-GEMM path
        os.environ["DSEA_TRANSFER_MFMA"] = form
        yr, yl = opr(v).clone(), opl(v).clone()
        again = opr(v).clone()
    finally:
        del os.environ["DSEA_TRANSFER_MFMA"]
    # the default is the hand-written pair up to D = 768, the library GEMMs beyond (D = 900, 1024: the 64 x 64 tiles)
    if form == "1":
        assert torch.equal(dr, yr if D <= 768 else zr)
    sr, sl = float(fr.abs().max()), float(fl.abs().max())
    assert float((yr - fr).abs().max()) < 1e-13 * sr * D, float((yr - fr).abs().max()) / sr
    assert float((yl - fl).abs().max()) < 1e-13 * sl * D
    assert float((yr - zr).abs().max()) < 1e-13 * sr * D and float((yl - zl).abs().max()) < 1e-13 * sl * D
    assert torch.equal(again, yr)                                               # deterministic


def test_dominant_eig_device_matches_reference_fixture():
    """reference tests/test_gradient.py:5-22 case: outputs of the REFERENCE's DominantEig (ARPACK + scipy gmres,
    eig.py:27-62) stored by tests/golden/make_golden.py, reproduced by the library Arnoldi / GMRES on the GPU."""
    gd = np.load(__import__("os").path.join(GOLDEN, "dominant_eig_D5.npz"))
    D, k = int(gd["D"]), int(gd["k"])
    n = D * D
    from dominantsparseeigenad_amd.synthetic import normal_vector
    A = gd["A"]
    Gong = np.einsum("kij,kmn->imjn", A, A).reshape(n, n)
    a = float(normal_vector(1, int(gd["seed_a"]))[0])
    M = torch.from_numpy(normal_vector(n * n, int(gd["seed_M"])).reshape(n, n)).to(cuda)
    G = torch.from_numpy(Gong).to(cuda).requires_grad_(True)
    lam, l, r = DominantEig.apply(G, k)
    loss = a * lam + l.matmul(M).matmul(r)
    (gG,) = torch.autograd.grad(loss.sum(), G)
    lg, rg = _gauge(l.detach().cpu().numpy(), r.detach().cpu().numpy())
    assert abs(lam.item() - float(gd["eigval"][0])) < 1e-11 * abs(float(gd["eigval"][0]))
    assert np.max(np.abs(rg - gd["r"])) < 1e-9 and np.max(np.abs(lg - gd["l"])) < 1e-8 * np.max(np.abs(gd["l"]))
    assert abs(loss.item() - float(gd["loss"])) < 1e-9 * abs(float(gd["loss"]))
    assert float(np.max(np.abs(gG.cpu().numpy() - gd["grad_Gong"]))) < 1e-7 * float(np.max(np.abs(gd["grad_Gong"])))


@pytest.mark.parametrize("native", [True, False])
def test_dominant_sparse_eig_device_matches_reference_fixture(native):
    """the REFERENCE's DominantSparseEig on the D = 10 transfer matrix (eig.py:115-149, operand form of
    examples/TFIM_vumps/general.py:59-74) vs the device path: native TransferOperator (loops and batched-GEMM
    mat-vec inside libdsea) and an opaque torch callable (mat-vec = user code, everything else library calls)."""
    from dominantsparseeigenad_amd.operators import TransferOperator
    from dominantsparseeigenad_amd.synthetic import normal_vector
    gd = np.load(__import__("os").path.join(GOLDEN, "dominant_sparse_eig_D10.npz"))
    D, d, k = int(gd["D"]), int(gd["d"]), int(gd["k"])
    n = D * D
    A = torch.from_numpy(gd["A"]).to(cuda).requires_grad_(True)
    Ad = A.detach()
    a = float(normal_vector(1, int(gd["seed_a"]))[0])
    M = torch.from_numpy(normal_vector(n * n, int(gd["seed_M"])).reshape(n, n)).to(cuda)

    def hook(pieces):
        gA = torch.zeros_like(Ad)
        for u, v in pieces:
            um, vm = u.reshape(D, D), v.reshape(D, D)
            gA = gA + torch.matmul(torch.matmul(um, Ad), vm.T) + torch.matmul(torch.matmul(um.T, Ad), vm)
        return gA

    if native:
        op, opT = TransferOperator(Ad), TransferOperator(Ad, transpose=True)
    else:
        fr = lambda v: torch.einsum("kij,kmn,jn->im", Ad, Ad, v.reshape(D, D)).reshape(-1)   # noqa: E731
        fl = lambda v: torch.einsum("kij,kmn,im->jn", Ad, Ad, v.reshape(D, D)).reshape(-1)   # noqa: E731
        op, opT = krylov.TorchLinearOperator((n, n), fr, cuda), krylov.TorchLinearOperator((n, n), fl, cuda)
    eig.setDominantSparseEig(op, opT, hook)
    lam, l, r = eig.DominantSparseEig.apply(A, k)
    loss = a * lam + l.matmul(M).matmul(r)
    (gA,) = torch.autograd.grad(loss.sum(), A)
    lg, rg = _gauge(l.detach().cpu().numpy(), r.detach().cpu().numpy())
    assert abs(lam.item() - float(gd["eigval"][0])) < 1e-11 * abs(float(gd["eigval"][0]))
    assert np.max(np.abs(rg - gd["r"])) < 1e-9 and np.max(np.abs(lg - gd["l"])) < 1e-8 * np.max(np.abs(gd["l"]))
    assert abs(loss.item() - float(gd["loss"])) < 1e-9 * abs(float(gd["loss"]))
    assert float(np.max(np.abs(gA.cpu().numpy() - gd["grad_A"]))) < 1e-7 * float(np.max(np.abs(gd["grad_A"])))


def test_config4_transfer_matrix_D512_k200():
    """BASELINE configs[3]: non-symmetric dominant eigen-triple of the D = 512 transfer matrix (n = 262144; the dense
    matrix would be 550 GB), k = 200 -- size-independent properties: eigen-residuals of r and l, the gauge
    l.r = 1 / r.r = 1, GMRES solves of the adjoint to 1e-12, and the gradient against a directional finite
    difference of lambda."""
    import time
    from dominantsparseeigenad_amd.operators import TransferOperator
    D, d, k = 512, 2, 200
    n = D * D
    gen = torch.Generator(device="cpu").manual_seed(1234)
    A = (torch.randn(d, D, D, dtype=F64, generator=gen) / np.sqrt(D)).to(cuda).requires_grad_(True)
    Ad = A.detach()

    def hook(pieces):
        gA = torch.zeros_like(Ad)
        for u, v in pieces:
            um, vm = u.reshape(D, D), v.reshape(D, D)
            gA = gA + torch.matmul(torch.matmul(um, Ad), vm.T) + torch.matmul(torch.matmul(um.T, Ad), vm)
        return gA

    op, opT = TransferOperator(Ad), TransferOperator(Ad, transpose=True)
    eig.setDominantSparseEig(op, opT, hook)
    torch.manual_seed(3)
    lam, l, r = eig.DominantSparseEig.apply(A, k)      # warm-up (rocBLAS start-up)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lam, l, r = eig.DominantSparseEig.apply(A, k)
    torch.cuda.synchronize()
    t_fwd = time.perf_counter() - t0
    lv, ld_, rd = lam.item(), l.detach(), r.detach()
    assert float((op(rd) - lv * rd).norm()) <= 1e-12 * abs(lv)
    assert float((opT(ld_) - lv * ld_).norm()) <= 1e-12 * abs(lv) * float(ld_.norm())
    assert abs(float(ld_ @ rd) - 1.0) < 1e-12 and abs(float(rd.norm()) - 1.0) < 1e-12
    t0 = time.perf_counter()
    (gA,) = torch.autograd.grad(lam.sum(), A)
    torch.cuda.synchronize()
    t_bwd = time.perf_counter() - t0
    # d lambda / dA = l r^T contracted with dGong/dA: check along a random direction by central differences
    dirn = torch.randn(d, D, D, dtype=F64, generator=gen).to(cuda)
    dirn = dirn / dirn.norm()
    eps = 1e-5
    lams = []
    for sgn in (+1.0, -1.0):
        Ap = (Ad + sgn * eps * dirn).contiguous()
        lp, _ = krylov.arnoldi_dominant(TransferOperator(Ap), n, k, cuda, "LM", v0=rd)
        lams.append(lp)
    fd = (lams[0] - lams[1]) / (2 * eps)
    assert abs(float((gA * dirn).sum()) - fd) < 1e-6 * max(abs(fd), 1.0), (float((gA * dirn).sum()), fd)
    print("config 4: D=512 k=200 forward %.1f ms, backward %.1f ms" % (t_fwd * 1e3, t_bwd * 1e3))


def test_gmres_native_operator_and_shift():
    from dominantsparseeigenad_amd.operators import DenseOperator
    rng = np.random.RandomState(19)
    n = 500
    Mh = rng.randn(n, n) / np.sqrt(n) + 3.0 * np.eye(n)
    b = torch.from_numpy(rng.randn(n)).to(cuda)
    Md = torch.from_numpy(Mh).to(cuda)
    shift = torch.tensor(0.7, dtype=F64, device=cuda)
    x = krylov.gmres(DenseOperator(Md), b, shift=shift)
    assert float((Md @ x - 0.7 * x - b).norm()) <= 1.01e-12 * float(b.norm()) + 1e-12
    xT = krylov.gmres(DenseOperator(Md, transpose=True), b, shift=shift)
    assert float((Md.T @ xT - 0.7 * xT - b).norm()) <= 1.01e-12 * float(b.norm()) + 1e-12
    # already solved: zero right-hand side
    assert float(krylov.gmres(DenseOperator(Md), torch.zeros(n, dtype=F64, device=cuda)).abs().max()) == 0.0


def test_arnoldi_restarts_and_reports_non_convergence():
    """a small basis forces thick restarts; an impossible budget raises instead of returning an unconverged pair"""
    from dominantsparseeigenad_amd.operators import DenseOperator
    rng = np.random.RandomState(2)
    n = 400
    Mh = np.abs(rng.randn(n, n)) / n + np.diag(np.linspace(0.0, 1.0, n))   # non-negative: real dominant eigenvalue
    w = np.linalg.eigvals(Mh)
    lam_ref = w[np.argmax(np.abs(w))].real
    op = DenseOperator(torch.from_numpy(Mh).to(cuda))
    torch.manual_seed(5)
    lam, x = krylov.arnoldi_dominant(op, n, 12, cuda, "LM")          # ncv = 12: many restarts
    assert abs(lam - lam_ref) < 1e-9 * abs(lam_ref)
    with pytest.raises(krylov.ArnoldiNoConvergence):
        krylov.arnoldi_dominant(op, n, 4, cuda, "LM", max_restarts=1)


def test_staged_convergence_test_finds_the_same_pair_with_fewer_columns():
    """The wanted Ritz pair is tested after stages of the factorisation (krylov.STAGE_FIRST) instead of only once all
    ncv columns exist (ARPACK's schedule, reference eig.py:29): same eigenpair, fewer mat-vecs; a callable operand
    takes the same path; STAGE_FIRST = 0 restores the single full-length cycle."""
    from dominantsparseeigenad_amd.operators import TransferOperator
    D, d, k = 64, 2, 200
    n = D * D
    gen = torch.Generator().manual_seed(23)
    Ad = (torch.randn(d, D, D, dtype=F64, generator=gen) / D ** 0.5).to(cuda)
    v0 = torch.randn(n, dtype=F64, generator=gen).to(cuda)
    op = TransferOperator(Ad)
    saved = krylov.STAGE_FIRST
    try:
        krylov.STAGE_FIRST = 0
        lam_full, x_full = krylov.arnoldi_dominant(op, n, k, cuda, "LM", v0=v0)
        assert krylov.last('arnoldi_columns') == k and krylov.last('arnoldi_stages') == 1
        krylov.STAGE_FIRST = saved
        lam_st, x_st = krylov.arnoldi_dominant(op, n, k, cuda, "LM", v0=v0)
        cols = krylov.last('arnoldi_columns')
        assert cols < k and krylov.last('arnoldi_stages') >= 2, cols
        AdT = Ad.transpose(1, 2).contiguous()
        fr = lambda v: torch.matmul(torch.matmul(Ad, v.reshape(D, D)), AdT).sum(0).reshape(-1)  # noqa: E731
        lam_cb, x_cb = krylov.arnoldi_dominant(krylov.TorchLinearOperator((n, n), fr, cuda), n, k, cuda, "LM", v0=v0)
    finally:
        krylov.STAGE_FIRST = saved
    assert abs(lam_st - lam_full) <= 1e-12 * abs(lam_full) and abs(lam_cb - lam_full) <= 1e-12 * abs(lam_full)
    for x in (x_st, x_cb):
        s = 1.0 if float(x @ x_full) > 0 else -1.0
        assert float((s * x - x_full).abs().max()) < 1e-10
        assert float((op(x) - lam_st * x).norm()) <= 1e-12 * abs(lam_st)
    print("staged Arnoldi: converged with %d of %d columns" % (cols, k))


def test_optimistic_second_pass_hands_failing_steps_back_and_changes_nothing():
    """include/dsea.h dsea_ws_set_arnoldi_optimistic / dsea_arnoldi_status: with the option on (krylov's default for native
    operands) the second Gram-Schmidt pass is not enqueued -- 6 launches per Arnoldi step instead of 11 -- and a step that fails
    the DGKS test on the device is handed back and repeated with the pass.  On a matrix that is a rank-one term plus 1e-6 noise
    almost every step after the first fails the test (A v_j lies in the span of the basis up to 1e-6): the factorisation, and so
    the Ritz pair, must come out bit for bit as in the default mode; on the well-conditioned D = 16 transfer matrix no step
    comes back."""
    from dominantsparseeigenad_amd.operators import DenseOperator, TransferOperator
    rng = np.random.RandomState(77)
    n = 2048
    u, v = rng.randn(n), rng.randn(n)
    v = v + 0.5 * u                                          # (u.v > 0: the dominant eigenvalue is real)
    G = torch.from_numpy(np.outer(u, v) / n + 1e-6 * rng.randn(n, n)).to(cuda)
    op = DenseOperator(G)
    v0 = torch.from_numpy(rng.randn(n)).to(cuda)
    out = {}
    old = krylov.OPTIMISTIC_SECOND_PASS
    try:
        for mode in (False, True):
            krylov.OPTIMISTIC_SECOND_PASS = mode
            theta, x = krylov.arnoldi_dominant(op, n, 24, cuda, v0=v0)
            out[mode] = (theta, x.clone(), krylov.last("arnoldi_second_pass_redos"), krylov.last("arnoldi_columns"))
    finally:
        krylov.OPTIMISTIC_SECOND_PASS = old
    assert out[False][2] == 0 and out[True][2] >= 3, (out[False][2], out[True][2])
    assert out[True][0] == out[False][0] and torch.equal(out[True][1], out[False][1]) and out[True][3] == out[False][3]
    lam = float((v @ u) / n)
    assert abs(out[True][0] - lam) < 1e-4 * abs(lam)
    res = float((G @ out[True][1] - out[True][0] * out[True][1]).norm())
    assert res < 1e-10 * abs(lam)
    # GMRES (the adjoint solves, eig.py:54-57) honours the option as well: a failing step ends its cycle early -- a restart --
    # and the rest of the solve runs with the pass enqueued; both modes reach the tolerance
    b = torch.from_numpy(rng.randn(n)).to(cuda)
    sigma = torch.tensor([2.0 * lam], dtype=F64, device=cuda)
    sols = {}
    assert krylov.OPTIMISTIC_SECOND_PASS_GMRES is False          # (off by default: no gain measured on the adjoint systems)
    try:
        for mode in (False, True):
            krylov.OPTIMISTIC_SECOND_PASS_GMRES = mode
            xs = krylov.gmres(op, b, shift=sigma)
            sols[mode] = (xs, krylov.last("gmres_second_pass_fallbacks"), krylov.last("gmres_cycles"))
    finally:
        krylov.OPTIMISTIC_SECOND_PASS_GMRES = False
    assert sols[False][1] == 0 and sols[True][1] == 1, (sols[False][1:], sols[True][1:])
    for mode in (False, True):
        r = G @ sols[mode][0] - sigma * sols[mode][0] - b
        assert float(r.norm()) <= 1.01e-12 * max(float(b.norm()), 1.0), (mode, float(r.norm()))
    assert float((sols[True][0] - sols[False][0]).norm()) < 1e-10 * float(sols[False][0].norm())
    # the usual case: nothing comes back
    A = torch.from_numpy(np.random.RandomState(5).randn(2, 32, 32) / 32 ** 0.5).to(cuda)
    assert krylov.OPTIMISTIC_SECOND_PASS is True
    top = TransferOperator(A)
    theta, x = krylov.arnoldi_dominant(top, 1024, 30, cuda)
    assert krylov.last("arnoldi_second_pass_redos") == 0
    assert float((top(x) - theta * x).norm()) < 1e-10 * abs(theta)


def test_arnoldi_status_protocol_through_the_c_abi():
    """include/dsea.h, the optimistic-second-pass protocol step by step through ctypes: extend with the option on -> the
    first step that fails the DGKS test stops the run and dsea_arnoldi_status returns DSEA_ERR_SECOND_PASS with that step
    (and clears the record: the next status call is DSEA_OK) -> that step repeated with the option off -> the rest with
    the option on again; H and V equal the default mode's bit for bit.  Bad arguments are refused."""
    import ctypes
    from dominantsparseeigenad_amd import _lib, engine
    from dominantsparseeigenad_amd.engine import _ptr
    from dominantsparseeigenad_amd.operators import DenseOperator
    lib = _lib.load()
    rng = np.random.RandomState(78)
    n, m = 1024, 6
    u, v = rng.randn(n), rng.randn(n)
    G = torch.from_numpy(np.outer(u, v + 0.5 * u) / n + 1e-6 * rng.randn(n, n)).to(cuda)
    op = DenseOperator(G)
    lp = krylov._Loop(op, n, cuda, m + 2)
    ws, ldv, st = lp.ws, lp.ldv, lp.st
    v0 = torch.from_numpy(rng.randn(n)).to(cuda)
    v0 = v0 / v0.norm()

    def run(optimistic):
        V = torch.zeros((m + 1, ldv), dtype=F64, device=cuda)
        H = torch.zeros((m, m + 1), dtype=F64, device=cuda)
        V[0, :n] = v0
        brk, redo, log, j = ctypes.c_int(0), ctypes.c_int(-1), [], 0
        while j < m:
            _lib.check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 1 if optimistic else 0))
            try:
                _lib.check(lib.dsea_arnoldi_extend(op.handle, ws.handle, None, _ptr(V), ldv, j, m, _ptr(H), m + 1, st()))
            finally:
                _lib.check(lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 0))
            rc = lib.dsea_arnoldi_status(ws.handle, ctypes.byref(brk), ctypes.byref(redo), st())
            log.append((rc, brk.value, redo.value))
            if rc != _lib.ERR_SECOND_PASS:
                assert rc == 0, rc
                break
            assert optimistic and j <= redo.value < m and brk.value == 0
            # the record was cleared by the status call
            assert lib.dsea_arnoldi_status(ws.handle, ctypes.byref(brk), ctypes.byref(redo), st()) == 0 and redo.value == -1
            _lib.check(lib.dsea_arnoldi_extend(op.handle, ws.handle, None, _ptr(V), ldv, log[-1][2], log[-1][2] + 1, _ptr(H),
                                               m + 1, st()))
            j = log[-1][2] + 1
        torch.cuda.synchronize()
        return V.clone(), H.clone(), log

    Vd, Hd, logd = run(False)
    Vo, Ho, logo = run(True)
    assert logd == [(0, 0, -1)]
    assert [e[0] for e in logo].count(_lib.ERR_SECOND_PASS) >= 2 and logo[0][2] == 1, logo   # step 0 passes, step 1 comes back
    assert torch.equal(Vo, Vd) and torch.equal(Ho, Hd)
    Vh = Vd[:m + 1, :n]
    assert float((Vh @ Vh.T - torch.eye(m + 1, dtype=F64, device=cuda)).abs().max()) < 1e-12     # second passes did their work
    assert lib.dsea_ws_set_arnoldi_optimistic(ws.handle, 2) == _lib.ERR_ARG
    assert lib.dsea_ws_set_arnoldi_optimistic(None, 1) == _lib.ERR_ARG and lib.dsea_arnoldi_status(None, None, None, None) == _lib.ERR_ARG
    assert b"second Gram-Schmidt" in lib.dsea_error_string(_lib.ERR_SECOND_PASS)


@pytest.mark.parametrize("n", [1, 2, 25, 1000, 1001, 2050, 9000])
def test_dense_operand_gemv_kernels(n):
    """include/dsea.h dsea_op_create_dense: the general dense operand of the non-symmetric primitives (eig.py:28-30) is a
    hand-written row-streaming GEMV (k_gemv_rows: one row per wave up to n = 8192, four beyond; 16-byte accesses when n and
    lda are even, scalar otherwise), A^T through a transposed copy made once by operators.DenseOperator or, at the C ABI,
    through the column-strip kernel (k_gemv_cols) -- against torch.mv, deterministic, with the shift / dot tail of dsea_spmv."""
    import ctypes
    from dominantsparseeigenad_amd import _lib, engine
    from dominantsparseeigenad_amd.engine import _ptr
    from dominantsparseeigenad_amd.operators import DenseOperator
    rng = np.random.RandomState(900 + n)
    G = torch.from_numpy(rng.randn(n, n)).to(cuda)
    v = torch.from_numpy(rng.randn(n)).to(cuda)
    ref, refT = torch.mv(G, v), torch.mv(G.T, v)
    tol = 1e-14 * max(n, 16) ** 0.5 * float(G.abs().max()) * float(v.abs().max()) * 8
    op, opT = DenseOperator(G), DenseOperator(G, transpose=True)
    y, yT = op(v), opT(v)
    assert float((y - ref).abs().max()) <= tol and float((yT - refT).abs().max()) <= tol
    assert torch.equal(op(v), y) and torch.equal(opT(v), yT)                    # deterministic
    # the C entry point with transpose != 0 (no transposed copy): the column-strip kernel
    lib = _lib.load()
    raw = ctypes.c_void_p()
    Gc = G.contiguous()
    _lib.check(lib.dsea_op_create_dense(n, ctypes.c_void_p(Gc.data_ptr()), n, 1, ctypes.byref(raw)))
    try:
        ws = engine.Workspace.get(n, 8, cuda)
        out = torch.empty(n, dtype=F64, device=cuda)
        dot = torch.zeros(1, dtype=F64, device=cuda)
        shift = torch.tensor([0.37], dtype=F64, device=cuda)
        _lib.check(lib.dsea_spmv(raw, ws.handle, _ptr(v), _ptr(out), _ptr(shift), _ptr(dot), None, engine._stream(cuda)))
        torch.cuda.synchronize()
        want = refT - 0.37 * v
        assert float((out - want).abs().max()) <= tol
        assert abs(float(dot) - float(v @ want)) <= 1e-12 * max(1.0, float(v.abs().max()) * float(want.abs().max()) * n)
    finally:
        lib.dsea_op_destroy(raw)
