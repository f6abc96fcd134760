"""Row-partitioned EXPLICIT-MATRIX operand on the real HIP slab kernels (SURVEY.md 8e "tridiagonal/CSR-banded: halo";
include/dsea.h dsea_op_set_slab / dsea_pop_create_csr / dsea_pop_sddmm; partitioned.PartitionedCSROperator).
One MI355X: world 1 over real RCCL; world 2 / 3 / 4 with all ranks on cuda:0 -- the library driver through the callback
communicator (gloo staged through the host) and through its RCCL branch over the stand-in (tests/fake_rccl).

  * slab mat-vec (neighbour halo of hb elements, or the all-gather fallback) == the one-GPU CSROperator, bit for bit;
  * dsea_pop_sddmm == dsea_op_sddmm on the whole matrix, bit for bit (one-sided form);
  * E0 / psi / d(E0 + psi.t)/d vals behind the reference API at 1e-10 against the dense eigh factors;
  * RCCL branch == callback path, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from dominantsparseeigenad_amd.synthetic import normal_vector  # noqa: E402
from helpers import unit  # noqa: E402

FAKE_RCCL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _matrix(kind):
    import scipy.sparse as sp
    from helpers import banded_spd
    n = 3001
    M = banded_spd(n, 9, 41)
    if kind == "scattered":
        rng = np.random.RandomState(42)
        extra = sp.random(n, n, density=0.001, random_state=rng, format="csr") * 0.05
        M = (M + extra + extra.T).tocsr()
        M.sort_indices()
    return M


def _comm_for(dev, mode):
    """None = torch's process group (world 1 over RCCL); otherwise gloo staged through the host, carrying for "rccl" a pair
    of library-owned communicators created over the stand-in"""
    from dominantsparseeigenad_amd.partitioned import NativeComm, RankOrderedHostStagedComm
    if mode == "nccl":
        return None
    comm = RankOrderedHostStagedComm()
    if mode == "rccl":
        os.environ["DSEA_RCCL_LIB"] = FAKE_RCCL
        comm.native_comm = NativeComm.own(None, dev)
    return comm


def _case(rank, world, dev, kind, mode, python_driver, force=True):
    from helpers import PatchRandn
    import dominantsparseeigenad_amd.symeig as symeig
    import dominantsparseeigenad_amd.CG as CG
    from dominantsparseeigenad_amd.partitioned import PartitionedCSROperator, csr_partition
    if python_driver:
        os.environ["DSEA_DRIVER"] = "python"
    CG.EPS_DEFAULT = 1e-12
    M = _matrix(kind)
    n, k = M.shape[0], 200
    nloc, off, real = csr_partition(n, world, rank)
    sub = M[off:off + real]
    vals = torch.from_numpy(sub.data.copy()).to(dev).requires_grad_(True)
    op = PartitionedCSROperator(torch.from_numpy(sub.indptr.astype("int64")).to(dev), torch.from_numpy(sub.indices.astype("int64")).to(dev),
                                vals, n, dev, comm=_comm_for(dev, mode))
    op.force_driver = bool(force)
    pad = nloc * world - n
    x = op.slab(torch.cat([torch.from_numpy(normal_vector(n, 8300)), torch.zeros(pad, dtype=torch.float64)])).to(dev)
    v1 = op.slab(torch.cat([torch.from_numpy(normal_vector(n, 8301)), torch.zeros(pad, dtype=torch.float64)])).to(dev)
    y = op.H(x.clone())
    g_plain = op.Aadjoint_to_valsadjoint(v1, x)
    g_sym = op.Aadjoint_to_valsadjoint_symmetric(v1, x)
    t = op.slab(torch.cat([unit(n, 8100), torch.zeros(pad, dtype=torch.float64)])).to(dev)
    symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint_symmetric)
    with PatchRandn(8200, offset=off):
        E0, psi = symeig.DominantSparseSymeig.apply(vals, k, op.dim, dev)
        loss = E0 + op.dot(psi, t)
        (gv,) = torch.autograd.grad(loss, vals)
    with torch.no_grad():
        vals.mul_(1.5)                                   # what an optimiser step does: in place, seen through the version counter
    y_upd = op.H(x.clone())
    torch.cuda.synchronize()
    return dict(mode=op.mode, hb=op.hb, driver=op.driver, E=E0.item(), loss=loss.item(), y=y.cpu().numpy()[:real].copy(),
                y_upd=y_upd.cpu().numpy()[:real].copy(),
                psi=psi.detach().cpu().numpy()[:real].copy(), pad=float(psi.detach()[real:].abs().sum()),
                grad=gv.cpu().numpy().copy(), g_plain=g_plain.cpu().numpy().copy(), g_sym=g_sym.cpu().numpy().copy())


def _worker(rank, world, port, backend, args, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        ret[rank] = _case(rank, world, dev, *args)
    finally:
        dist.destroy_process_group()


def _run(world, kind, mode, python_driver=False, force=True):
    from helpers import spawn_collect
    backend = "nccl" if mode == "nccl" else "gloo"
    ret = spawn_collect(_worker, (world, _free_port(), backend, (kind, mode, python_driver, force)), world, port_index=1)
    assert len(ret) == world
    return [ret[r] for r in range(world)]


def _one_gpu(kind):
    from dominantsparseeigenad_amd.operators import CSROperator
    dev = torch.device("cuda:0")
    M = _matrix(kind)
    n = M.shape[0]
    op = CSROperator.from_scipy(M, dev)
    x = torch.from_numpy(normal_vector(n, 8300)).to(dev)
    v1 = torch.from_numpy(normal_vector(n, 8301)).to(dev)
    upd = CSROperator.from_scipy(M * 1.5, dev)
    return M, op(x).cpu().numpy(), op.sddmm(v1, x).cpu().numpy(), op.sddmm(v1, x, symmetric=True).cpu().numpy(), upd(x).cpu().numpy()


def _check(ret, kind, world):
    from helpers import eigh_reference
    M, y1, gp1, gs1, y_upd1 = _one_gpu(kind)
    n = M.shape[0]
    expect = "halo" if (kind == "banded" or world <= 2) else "gather"
    assert all(r["mode"] == expect for r in ret), [r["mode"] for r in ret]
    y = np.concatenate([r["y"] for r in ret])
    assert np.array_equal(y, y1)                                     # slab mat-vec == one-GPU operator, bit for bit
    assert np.array_equal(np.concatenate([r["y_upd"] for r in ret]), y_upd1)   # ... also after an in-place step of the non-zeros
    assert np.array_equal(np.concatenate([r["g_plain"] for r in ret]), gp1)
    gs = np.concatenate([r["g_sym"] for r in ret])
    assert np.max(np.abs(gs - gs1)) <= 4e-16 * np.max(np.abs(gs1))   # (two one-sided launches vs one: last-bit rounding)
    assert all(r["pad"] == 0.0 for r in ret)
    for r in ret:
        assert r["E"] == ret[0]["E"] and r["loss"] == ret[0]["loss"]
    psi = torch.from_numpy(np.concatenate([r["psi"] for r in ret]))
    grad = torch.from_numpy(np.concatenate([r["grad"] for r in ret]))
    E_ref, psi_ref, g_ref = eigh_reference(torch.from_numpy(M.indptr.astype("int64")), torch.from_numpy(M.indices.astype("int64")),
                                           torch.from_numpy(M.data.copy()), n, unit(n, 8100), 1.0, 1.0, psi_like=psi, autograd=False)
    assert abs(ret[0]["E"] - E_ref.item()) < 1e-12 * abs(E_ref.item())
    assert float((psi - psi_ref).abs().max()) < 1e-9
    err = float((grad - g_ref).abs().max()) / float(g_ref.abs().max())
    assert err < 1e-10, err
    return err


@pytest.mark.parametrize("world,kind,mode", [(1, "banded", "nccl"), (2, "banded", "callbacks"), (3, "banded", "callbacks"),
                                             (4, "banded", "callbacks"), (3, "scattered", "callbacks"), (4, "scattered", "callbacks")])
def test_partitioned_csr_library_driver(world, kind, mode):
    ret = _run(world, kind, mode)
    assert all("library" in r["driver"] for r in ret), ret[0]["driver"]
    err = _check(ret, kind, world)
    print("world %d %s (%s, hb = %d): d(E0 + psi.t)/d vals max abs err / max = %.2e" % (world, kind, ret[0]["mode"], ret[0]["hb"], err))


def test_partitioned_csr_world1_takes_the_one_gpu_loops():
    """world size 1 without ``force_driver``: the slab IS the matrix, so the in-library single-GPU loops run on the slab
    operator (a SELL operand in slab mode has no fused Lanczos tail: dsea_lanczos_run takes its unfused sequence)"""
    ret = _run(1, "banded", "nccl", False, False)
    _check(ret, "banded", 1)


@pytest.mark.parametrize("world,kind", [(2, "banded"), (4, "scattered")])
def test_partitioned_csr_rccl_branch_equals_callback_path(world, kind):
    """the library's RCCL branch (ncclGroupStart / Send / Recv / End for the halo and the all-gather, ncclAllReduce for the
    inner products) executing over the stand-in, against the callback communicator and the Python step driver"""
    if not os.path.exists(FAKE_RCCL):
        import subprocess
        subprocess.run(["make", "-C", os.path.dirname(FAKE_RCCL), "libfake_rccl.so"], capture_output=True, timeout=300)
    assert os.path.exists(FAKE_RCCL), "tests/fake_rccl/libfake_rccl.so is missing: __graft_entry__.build() builds it"
    rccl = _run(world, kind, "rccl")
    cb = _run(world, kind, "callbacks")
    assert all("rccl" in r["driver"] for r in rccl), rccl[0]["driver"]
    _check(rccl, kind, world)
    for r in range(world):
        for key in ("E", "loss"):
            assert rccl[r][key] == cb[r][key], key
        for key in ("y", "y_upd", "psi", "grad", "g_plain", "g_sym"):
            assert np.array_equal(rccl[r][key], cb[r][key]), key
    py = _run(world, kind, "callbacks", True)
    assert all(r["driver"] == "python" for r in py)
    _check(py, kind, world)


def _tiny_matrix():
    from helpers import banded_spd
    return banded_spd(5, 1, 7)


def _tiny_case(rank, world, dev):
    from dominantsparseeigenad_amd.partitioned import PartitionedCSROperator, RankOrderedHostStagedComm, csr_partition
    M = _tiny_matrix()
    n = M.shape[0]
    nloc, off, real = csr_partition(n, world, rank)
    sub = M[off:off + real] if real else M[0:0]
    vals = torch.from_numpy(sub.data.copy()).to(dev).requires_grad_(True)
    op = PartitionedCSROperator(torch.from_numpy(sub.indptr.astype("int64")).to(dev), torch.from_numpy(sub.indices.astype("int64")).to(dev),
                                vals, n, dev, comm=RankOrderedHostStagedComm())
    op.force_driver = True
    pad = nloc * world - n
    x = op.slab(torch.cat([torch.from_numpy(normal_vector(n, 8300)), torch.zeros(pad, dtype=torch.float64)])).to(dev)
    v1 = op.slab(torch.cat([torch.from_numpy(normal_vector(n, 8301)), torch.zeros(pad, dtype=torch.float64)])).to(dev)
    y = op.H(x.clone())
    g = op.Aadjoint_to_valsadjoint(v1, x)
    gs = op.Aadjoint_to_valsadjoint_symmetric(v1, x)
    with torch.no_grad():
        vals.mul_(1.5)
    y_upd = op.H(x.clone())
    torch.cuda.synchronize()
    return dict(real=real, nnz=int(vals.numel()), y=y.cpu().numpy()[:real].copy(), ypad=float(y[real:].abs().sum()),
                y_upd=y_upd.cpu().numpy()[:real].copy(), g=g.cpu().numpy().copy(), gs=gs.cpu().numpy().copy())


def _tiny_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        ret[rank] = _tiny_case(rank, world, dev)
    finally:
        dist.destroy_process_group()


def test_partitioned_csr_slab_that_is_all_padding():
    """5 rows on 4 ranks: slabs of 2 rows, the third holds one real row, the FOURTH NONE -- no stored entry, yet it takes
    part in every exchange (found by tools/fuzz_partitioned_csr.py: torch reports a null pointer for every empty tensor and
    the C ABI refuses null pointers; the host side keeps one addressable dummy element instead)"""
    from helpers import spawn_collect
    from dominantsparseeigenad_amd.operators import CSROperator
    world = 4
    ret = spawn_collect(_tiny_worker, (world, _free_port()), world, port_index=1)
    assert [ret[r]["real"] for r in range(world)] == [2, 2, 1, 0] and ret[3]["nnz"] == 0
    M = _tiny_matrix()
    dev = torch.device("cuda:0")
    op = CSROperator.from_scipy(M, dev)
    x = torch.from_numpy(normal_vector(5, 8300)).to(dev)
    v1 = torch.from_numpy(normal_vector(5, 8301)).to(dev)
    assert np.array_equal(np.concatenate([ret[r]["y"] for r in range(world)]), op(x).cpu().numpy())
    assert np.array_equal(np.concatenate([ret[r]["y_upd"] for r in range(world)]), CSROperator.from_scipy(M * 1.5, dev)(x).cpu().numpy())
    assert np.array_equal(np.concatenate([ret[r]["g"] for r in range(world)]), op.sddmm(v1, x).cpu().numpy())
    gs1 = op.sddmm(v1, x, symmetric=True).cpu().numpy()
    assert np.max(np.abs(np.concatenate([ret[r]["gs"] for r in range(world)]) - gs1)) <= 4e-16 * np.max(np.abs(gs1))
    assert all(ret[r]["ypad"] == 0.0 for r in range(world))
