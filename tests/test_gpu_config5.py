"""BASELINE configs[4] (TFIM L = 28 row-partitioned over 8 GPUs) at the sizes ONE GPU of that run sees, kept green:

  * the PER-GPU LOAD of config 5 -- 2^25 rows, k = 200 (53.7 GB of fp64 basis + 13.4 GB bf16 shadow) -- on the
    DISTRIBUTED driver (``force_driver``), world size 1 over "nccl" (= RCCL: every all-reduce of the step is really
    issued), through the reference API;
  * the one-GPU ANCHOR of north_star's strong-scaling curve -- L = 28 (n = 2^28), k = 100, 215 GB of basis, no shadow --
    on the in-library single-GPU loops.

Neither size has a CPU oracle run (hours); they are held to SIZE-INDEPENDENT properties: E0 against the closed form of
reference examples/TFIM/E0.py:9-23 (1e-12 where k converges the pair), the eigen-residual ||H psi - E0 psi||,
normalisation, a converged CG adjoint solve whose gradient equals the closed-form dE0/dg (Hellmann-Feynman is exact
for the converged pair), and orthonormality of a sample of basis columns.  The multi-rank collectives themselves are covered at small sizes by
tests/test_gpu_partitioned.py and tests/test_partitioned_gloo.py."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.operators import tfim_analytic_E0  # noqa: E402

F64 = torch.float64
SAMPLE = (0, 1, 2, 3, 50, 99, 100, 101, 150, 197, 198, 199)


def _closed_form(L, g0=1.0):
    """E0 and dE0/dg of the periodic chain: free fermions with the momenta of the even-parity sector,
    k = (2m+1) pi / L.  For even L this is reference examples/TFIM/E0.py:9-23 (``tfim_analytic_E0``); for odd L
    (config 5's slab has L = 25) the reference's linspace would pick integer momenta, which is not the ground-state
    sector -- the formula below holds for both and equals the reference's at even L."""
    import math
    gt = torch.tensor(g0, dtype=F64, requires_grad=True)
    ks = (2.0 * torch.arange(L, dtype=F64) + 1.0) * math.pi / L
    E = -torch.sqrt(gt * gt - 2.0 * gt * torch.cos(ks) + 1.0).sum()
    (dE,) = torch.autograd.grad(E, gt)
    if L % 2 == 0:
        assert abs(E.item() - tfim_analytic_E0(L, torch.tensor(g0, dtype=F64)).item()) < 1e-12 * abs(E.item())
    return E.item(), dE.item()


def _case_config5_slab(rank, world, backend, dev, Lg, k):
    """2^Lg rows on this rank, the distributed driver, reference API + a direct look at the basis"""
    import dominantsparseeigenad_amd.CG as CG
    import dominantsparseeigenad_amd.symeig as symeig
    from dominantsparseeigenad_amd import engine
    from dominantsparseeigenad_amd.partitioned import PartitionedTFIMOperator
    n = 1 << Lg
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    g = torch.tensor([1.0], dtype=F64, device=dev, requires_grad=True)
    op = PartitionedTFIMOperator(Lg, g, dev)
    op.force_driver = True
    out = {"overlap": bool(op.overlap), "transposed": bool(op.transposed)}
    # (1) the Lanczos driver itself: basis, tridiagonal, orthonormality of a column sample, shadow path taken
    q0 = torch.randn(n, dtype=F64, device=dev, generator=gen)
    Q, ldq, alphas, betas = op.lanczos(k, q0, arena=True)
    cols = [c for c in SAMPLE if c < k]
    S = torch.stack([Q[c, :n] for c in cols])
    G = S @ S.T
    out["orth"] = float((G - torch.eye(len(cols), dtype=F64, device=dev)).abs().max())
    (lam, s), = engine.tridiag_extreme(alphas, betas, "min")
    psi = op.ritz_vector(Q, ldq, k, s)
    out["E0_driver"] = lam
    del Q, S, G
    # (2) the same through the reference API, with the adjoint (CG tightened: the property is checked at 1e-10)
    CG.EPS_DEFAULT = 1e-12
    symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)
    E0, psi = symeig.DominantSparseSymeig.apply(g, k, op.dim, dev)
    (dE0,) = torch.autograd.grad(E0, g)
    p = psi.detach()
    res = op.H(p) - E0.detach() * p
    out.update(E0=E0.item(), dE0=dE0.item(), resid=float(op.dot(res, res).sqrt()), norm=float(op.dot(p, p).sqrt()),
               cg_iters=int(op.last_cg_iters), cg_resnorm=float(op.last_cg_resnorm),
               cg_converged=bool(engine.last_cg.converged), overlap_fallbacks=int(op.overlap_fallbacks))
    # (3) the library driver's default CG for this operand carries r and A'p by recurrences (one all-reduce per iteration,
    # Chronopoulos-Gear); its stop is accepted only on the TRUE residual.  Checked here from outside: solve, then form
    # b - (H - E0) x with the operator (round-5 advisor: "a test at the config-5 slab size with eps = 1e-12")
    b = torch.randn(n, dtype=F64, device=dev, generator=gen)
    b = b - op.dot(p, b) * p
    b = b / op.dot(b, b).sqrt()
    x0 = torch.randn(n, dtype=F64, device=dev, generator=gen)
    x0 = x0 - op.dot(p, x0) * p
    x = op.solve_shifted(E0.detach(), b, x0, eps=1e-12)
    true_r = b - (op.H(x) - E0.detach() * x)
    out.update(cg2_true=float(op.dot(true_r, true_r).sqrt()), cg2_reported=float(op.last_cg_resnorm), cg2_iters=int(op.last_cg_iters),
               cg2_form=str(engine.last_cg.form))
    torch.cuda.synchronize()
    return out


def test_config5_per_gpu_load_on_the_distributed_driver_over_rccl():
    """2^25 rows x k = 200 on one rank over RCCL, force_driver: what each of the 8 GPUs of BASELINE configs[4] runs
    (minus the exchange partners)."""
    Lg, k = 25, 200
    o = _spawn_config5(Lg, k)
    E_an, dE_an = _closed_form(Lg)
    print("config-5 slab: E0 rel dev %.2e, dE0/dg rel dev %.2e, residual %.2e, orth %.2e, CG %d its to %.1e"
          % (abs(o["E0"] - E_an) / abs(E_an), abs(o["dE0"] - dE_an) / abs(dE_an), o["resid"], o["orth"], o["cg_iters"],
             o["cg_resnorm"]))
    assert abs(o["E0"] - E_an) < 1e-12 * abs(E_an), (o["E0"], E_an)
    assert abs(o["E0_driver"] - E_an) < 1e-12 * abs(E_an)
    assert o["resid"] < 1e-9 and abs(o["norm"] - 1.0) < 1e-12
    assert o["orth"] < 1e-12, o["orth"]
    assert o["cg_converged"] and o["cg_resnorm"] < 1e-12 and 0 < o["cg_iters"] < 2000
    assert abs(o["dE0"] - dE_an) < 1e-10 * abs(dE_an), (o["dE0"], dE_an)
    print("config-5 slab, one-reduction CG at eps 1e-12: %d iterations, reported %.3e, true ||b - (H - E0) x|| %.3e (%s)"
          % (o["cg2_iters"], o["cg2_reported"], o["cg2_true"], o["cg2_form"]))
    assert "one all-reduce" in o["cg2_form"]
    assert o["cg2_reported"] < 1e-12 and o["cg2_true"] < 1.01e-12, (o["cg2_reported"], o["cg2_true"])
    assert abs(o["cg2_true"] - o["cg2_reported"]) < 1e-14          # what is reported IS the true residual


def _worker_config5(rank, port, Lg, k, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ret[0] = _case_config5_slab(0, 1, "nccl", dev, Lg, k)
    finally:
        dist.destroy_process_group()


def _spawn_config5(Lg, k):
    """run the case in a fresh process (its 67 GB must not meet whatever the pytest process still caches)"""
    import torch.multiprocessing as mp
    from test_gpu_partitioned import _free_port
    from helpers import spawn_collect
    return spawn_collect(_worker_config5, (_free_port(), Lg, k), 1, port_index=0)[0]


@pytest.mark.slow
def test_L28_k100_one_gpu_anchor_properties():
    """The N = 1 point of the strong-scaling curve: TFIM L = 28, k = 100 with full re-orthogonalisation on ONE GPU
    (215 GB of basis; the bf16 shadow does not fit and is dropped by engine.shadow_fits).  k = 100 does not converge the
    ground state of the critical chain at this size to rounding level (measured: E0 4e-11 ... 1.1e-9 relative above the
    closed form depending on the start vector, eigen-residual 6e-4, dE0/dg 1.4e-4), so the properties are held at the
    level that Ritz pair has: E0 within 1e-8 of the closed form and above it
    (Ritz values approach from above), E0 error <= residual^2 / gap (Kato-Temple with the closed-form gap), unit norm,
    CG converged at the reference's tolerance, dE0/dg within 1e-3 of the closed form (first order in the eigenvector
    error)."""
    import dominantsparseeigenad_amd.CG as CG
    import dominantsparseeigenad_amd.symeig as symeig
    from dominantsparseeigenad_amd import engine
    from dominantsparseeigenad_amd.operators import TFIMOperator
    dev = torch.device("cuda:0")
    engine.BasisArena.release()
    engine.Workspace.clear_cache()
    torch.cuda.empty_cache()
    L, k = 28, 100
    n = 1 << L
    # (the 67 GB of the previous test's worker process come back to the driver a moment AFTER that process has exited:
    #  look again for a while before concluding that the memory is not there)
    import time
    for _ in range(60):
        free_b, total_b = torch.cuda.mem_get_info(dev)
        if free_b >= 8.0 * n * (k + 10):
            break
        time.sleep(0.5)
    if free_b < 8.0 * n * (k + 10):
        pytest.skip("needs %.0f GB of free HBM, %.0f GB free" % (8.0 * n * (k + 10) / 1e9, free_b / 1e9))
    old = CG.EPS_DEFAULT
    try:
        CG.EPS_DEFAULT = 1e-7           # the reference's tolerance (CG.py:25)
        op = TFIMOperator(L, dev)
        op.g = torch.tensor([1.0], dtype=F64, device=dev, requires_grad=True)
        symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)
        torch.manual_seed(7)
        E0, psi = symeig.DominantSparseSymeig.apply(op.g, k, n, dev)
        (dE0,) = torch.autograd.grad(E0, op.g)
        p = psi.detach()
        with torch.no_grad():
            resid = float((op.H(p) - E0.detach() * p).norm())
        nrm = float(p.norm())
        iters, conv, rn = engine.last_cg.iters, engine.last_cg.converged, engine.last_cg.resnorm
        del psi, p
    finally:
        CG.EPS_DEFAULT = old
        engine.BasisArena.release()
        engine.Workspace.clear_cache()
        torch.cuda.empty_cache()
    E_an, dE_an = _closed_form(L)
    print("L=28 k=100 one GPU: E0 rel dev %.2e, dE0/dg rel dev %.2e, residual %.2e, CG %d its to %.1e"
          % (abs(E0.item() - E_an) / abs(E_an), abs(dE0.item() - dE_an) / abs(dE_an), resid, iters, rn))
    assert -1e-12 * abs(E_an) < E0.item() - E_an < 1e-8 * abs(E_an), (E0.item(), E_an)
    gap = 2.0 * 3.141592653589793 / L * 0.5         # > lower bound of the excitation gap of the critical chain ~ pi/L
    assert resid < 5e-3 and E0.item() - E_an <= 2.0 * resid ** 2 / gap + 1e-12 * abs(E_an), (resid, E0.item() - E_an)
    assert abs(nrm - 1.0) < 1e-12
    assert conv and rn < 1e-7
    assert abs(dE0.item() - dE_an) < 1e-3 * abs(dE_an), (dE0.item(), dE_an)
