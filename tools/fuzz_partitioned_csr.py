#!/usr/bin/env python3
"""Random configurations of the ROW-PARTITIONED EXPLICIT-MATRIX operand (partitioned.PartitionedCSROperator;
include/dsea.h dsea_op_set_slab / dsea_pop_create_csr / dsea_pop_sddmm) on ONE GPU -- ranks share the device, collectives
staged through the host over gloo -- against the one-GPU CSROperator on the whole matrix.  Not a test: run it on the GPU box
to look for edge cases (tests/test_gpu_partitioned_csr.py holds the fixed cases).

What is drawn: world size (2, 3, 4, 5, 8); n from "smaller than the world" to a few thousand, never chosen to divide evenly
(padded last slab, slabs that are ALL padding, slabs shorter than a SELL slice); pattern: diagonal only, banded with a reach
below / at / above the slab length (halo <-> all-gather decision), scattered far couplings, symmetric empty rows; library
driver (callback communicator, or its RCCL branch over the stand-in tests/fake_rccl: "rccl*") or Python step driver.

What is checked per case:
  * slab mat-vec, one-sided sampled outer product, and the mat-vec after an in-place update of the non-zeros: BIT FOR BIT
    equal to the one-GPU operator (the slab kernel is the same SELL kernel with redirected gathers); symmetric sampled
    outer product to last-bit rounding (two one-sided launches against one);
  * every rank took the same halo / all-gather decision, and it is the one the pattern implies;
  * the padding of the last slab stays zero;
  * SPD cases of >= 400 rows: E0 / d(E0 + psi.t)/d vals behind the reference API against the dense eigh factors (1e-10 of
    the largest gradient entry), E0 identical on all ranks.

    python tools/fuzz_partitioned_csr.py [--cases 40] [--seed 0] [--cpu] [--only I]      (one set of processes per case)
``--cpu``: the CPU test double (tests/cpu_backend.py over gloo) in place of the HIP slab kernels -- exercises the host logic
(partition, padding, mode decision) in the build container; the bit-for-bit checks become 1e-13 checks against scipy."""
import argparse
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

WORLDS = (2, 3, 4, 5, 8)
KINDS = ("diagonal", "banded", "banded-wide", "scattered", "holes")
CASE_TIMEOUT = 90
FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def draw(rng):
    world = int(rng.choice(WORLDS))
    kind = str(rng.choice(KINDS))
    size = rng.rand()
    if size < 0.2:
        n = int(rng.randint(1, 4 * world + 2))                 # fewer rows than ranks .. a handful per rank
    elif size < 0.5:
        n = int(rng.randint(4 * world, 64 * world + 70))       # slabs around one SELL slice
    else:
        n = int(rng.randint(400, 2600))
    case = dict(world=world, kind=kind, n=n, mseed=int(rng.randint(1, 1 << 30)), python_driver=bool(rng.rand() < 0.25))
    case["rccl"] = bool(rng.rand() < 0.35) and not case["python_driver"]
    return case


def matrix(case):
    """symmetric sparse matrix of the case (scipy CSR, sorted indices), and whether it is SPD with a gapped bottom"""
    import scipy.sparse as sp
    from helpers import banded_spd
    n, kind, world = case["n"], case["kind"], case["world"]
    rng = np.random.RandomState(case["mseed"])
    nloc = -(-n // world)
    if kind == "diagonal" or n < 3:
        M = sp.diags(1.0 + np.linspace(0.0, 3.0, n) + rng.rand(n), 0, shape=(n, n), format="csr")
        return M, True
    if kind == "banded-wide":
        hb = int(min(n - 1, max(1, nloc + rng.randint(-2, 3))))      # reach just below / at / just above the slab length
    else:
        hb = int(min(n - 1, rng.randint(1, 24)))
    M = banded_spd(n, hb, case["mseed"])
    spd = True
    if kind == "scattered":
        extra = sp.random(n, n, density=min(0.3, 2.0 / n), random_state=rng, format="csr") * 0.05
        M = (M + extra + extra.T).tocsr()
    if kind == "holes":
        keep = (rng.rand(n) > 0.15).astype(np.float64)
        D = sp.diags(keep, 0, format="csr")
        M = (D @ M @ D).tocsr()                                       # some rows AND their columns removed: empty rows
        M.eliminate_zeros()
        spd = False
    M.sort_indices()
    return M.tocsr(), spd


def expected_mode(M, world):
    n = M.shape[0]
    nloc = -(-n // world)
    reach = 0
    for r in range(world):
        off = r * nloc
        sub = M[off:min(n, off + nloc)]
        if sub.nnz:
            reach = max(reach, off - int(sub.indices.min()), int(sub.indices.max()) - (off + nloc - 1))
    return ("halo" if reach <= nloc else "gather"), reach


def run_case(rank, world, dev, case, cpu):
    from helpers import PatchRandn, unit
    import dominantsparseeigenad_amd.symeig as symeig
    import dominantsparseeigenad_amd.CG as CG
    from dominantsparseeigenad_amd.partitioned import PartitionedCSROperator, csr_partition
    from dominantsparseeigenad_amd.synthetic import normal_vector
    os.environ.pop("DSEA_DRIVER", None)
    if case["python_driver"]:
        os.environ["DSEA_DRIVER"] = "python"
    CG.EPS_DEFAULT = 1e-12
    M, spd = matrix(case)
    n = M.shape[0]
    nloc, off, real = csr_partition(n, world, rank)
    sub = M[off:off + real] if real else M[0:0]
    vals = torch.from_numpy(np.ascontiguousarray(sub.data, dtype=np.float64).copy()).to(dev).requires_grad_(True)
    rowptr = torch.from_numpy(sub.indptr.astype("int64")).to(dev)
    cols = torch.from_numpy(sub.indices.astype("int64")).to(dev)
    if cpu:
        from cpu_backend import CpuBackend
        op = PartitionedCSROperator(rowptr, cols, vals, n, "cpu", backend=CpuBackend(nloc))
    else:
        from dominantsparseeigenad_amd.partitioned import NativeComm, RankOrderedHostStagedComm
        comm = RankOrderedHostStagedComm()
        if case.get("rccl"):              # the library's RCCL branch (group Send/Recv halos, all-gather, all-reduce) over the stand-in
            os.environ["DSEA_RCCL_LIB"] = FAKE_RCCL
            comm.native_comm = NativeComm.own(None, dev)
        op = PartitionedCSROperator(rowptr, cols, vals, n, dev, comm=comm)
    op.force_driver = True
    pad = nloc * world - n

    def slab_of(v):
        return op.slab(torch.cat([v, torch.zeros(pad, dtype=torch.float64)])).to(dev)

    x = slab_of(torch.from_numpy(normal_vector(n, 8300)))
    v1 = slab_of(torch.from_numpy(normal_vector(n, 8301)))
    out = dict(mode=op.mode, hb=op.hb, driver=getattr(op, "driver", "?"))
    y = op.H(x.clone())
    out["y"] = y.detach().cpu().numpy()[:real].copy()
    out["ypad"] = float(y.detach()[real:].abs().sum())
    out["g_plain"] = op.Aadjoint_to_valsadjoint(v1, x).cpu().numpy().copy()
    out["g_sym"] = op.Aadjoint_to_valsadjoint_symmetric(v1, x).cpu().numpy().copy()
    if spd and n >= 400:
        k = 150 if n < 1500 else 200
        t = slab_of(unit(n, 8100))
        symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint_symmetric)
        with PatchRandn(8200, offset=off):
            E0, psi = symeig.DominantSparseSymeig.apply(vals, k, op.dim, dev)
            loss = E0 + op.dot(psi, t)
            (gv,) = torch.autograd.grad(loss, vals)
        out.update(E=E0.item(), psi=psi.detach().cpu().numpy()[:real].copy(), psipad=float(psi.detach()[real:].abs().sum()),
                   grad=gv.cpu().numpy().copy(), driver=op.driver)
    with torch.no_grad():
        vals.mul_(1.5)
    out["y_upd"] = op.H(x.clone()).detach().cpu().numpy()[:real].copy()
    if not cpu:
        torch.cuda.synchronize()
    return out


def worker(rank, world, port, case, cpu, ret):
    """one case per set of processes: a configuration that raises on one rank leaves its peers inside a collective, so the
    processes are not reused; a rank that is still running after CASE_TIMEOUT seconds is killed by its own alarm"""
    import signal
    import torch.distributed as dist
    signal.alarm(CASE_TIMEOUT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if cpu:
        torch.set_num_threads(1)
        dev = torch.device("cpu")
    else:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = run_case(rank, world, dev, case, cpu)
    finally:
        dist.destroy_process_group()


def one_gpu(case, cpu):
    from dominantsparseeigenad_amd.synthetic import normal_vector
    M, spd = matrix(case)
    n = M.shape[0]
    x = normal_vector(n, 8300)
    v1 = normal_vector(n, 8301)
    rows = np.repeat(np.arange(n), np.diff(M.indptr))
    if cpu or M.nnz == 0:
        gp = v1[rows] * x[M.indices]
        gs = 0.5 * (gp + v1[M.indices] * x[rows])
        return M, spd, M @ x, gp, gs, (M * 1.5) @ x
    from dominantsparseeigenad_amd.operators import CSROperator
    dev = torch.device("cuda:0")
    op = CSROperator.from_scipy(M, dev)
    upd = CSROperator.from_scipy(M * 1.5, dev)
    xd, vd = torch.from_numpy(x).to(dev), torch.from_numpy(v1).to(dev)
    return (M, spd, op(xd).cpu().numpy(), op.sddmm(vd, xd).cpu().numpy(), op.sddmm(vd, xd, symmetric=True).cpu().numpy(),
            upd(xd).cpu().numpy())


def judge(case, ret, cpu):
    """list of findings (empty = ok) and a one-line summary"""
    from helpers import eigh_reference, unit
    world = case["world"]
    M, spd, y1, gp1, gs1, yu1 = one_gpu(case, cpu)
    n = M.shape[0]
    bad = []
    mode, reach = expected_mode(M, world)
    if any(r["mode"] != mode for r in ret):
        bad.append("mode %s, expected %s (reach %d)" % ([r["mode"] for r in ret], mode, reach))

    def same(name, got, want, exact):
        got = np.concatenate(got) if len(got) else np.zeros(0)
        if got.shape != want.shape:
            bad.append("%s: shape %s vs %s" % (name, got.shape, want.shape))
        elif exact and not cpu:
            if not np.array_equal(got, want):
                bad.append("%s: not bit-identical (max dev %.1e)" % (name, float(np.max(np.abs(got - want)))))
        elif want.size and float(np.max(np.abs(got - want))) > (4e-16 if not cpu else 1e-13) * max(1.0, float(np.max(np.abs(want)))):
            bad.append("%s: max dev %.1e" % (name, float(np.max(np.abs(got - want)))))

    same("mat-vec", [r["y"] for r in ret], np.asarray(y1), True)
    same("mat-vec after update", [r["y_upd"] for r in ret], np.asarray(yu1), True)
    same("sddmm", [r["g_plain"] for r in ret], np.asarray(gp1), True)
    same("sddmm symmetric", [r["g_sym"] for r in ret], np.asarray(gs1), False)
    if any(r["ypad"] != 0.0 for r in ret):
        bad.append("padding of y not zero")
    note = ""
    if "E" in ret[0]:
        if any(r["E"] != ret[0]["E"] for r in ret):
            bad.append("E0 differs between ranks")
        if any(r["psipad"] != 0.0 for r in ret):
            bad.append("padding of psi not zero")
        psi = torch.from_numpy(np.concatenate([r["psi"] for r in ret]))
        grad = torch.from_numpy(np.concatenate([r["grad"] for r in ret]))
        E_ref, psi_ref, g_ref = eigh_reference(torch.from_numpy(M.indptr.astype("int64")), torch.from_numpy(M.indices.astype("int64")),
                                               torch.from_numpy(M.data.copy()), n, unit(n, 8100), 1.0, 1.0, psi_like=psi, autograd=False)
        dE = abs(ret[0]["E"] - E_ref.item()) / abs(E_ref.item())
        dg = float((grad - g_ref).abs().max()) / float(g_ref.abs().max())
        dpsi = float((psi - psi_ref).abs().max())
        if dE > 1e-12 or dg > 1e-10 or dpsi > 1e-9:
            bad.append("eigh: E0 %.1e psi %.1e grad %.1e" % (dE, dpsi, dg))
        note = "E0 %.1e psi %.1e grad %.1e vs eigh" % (dE, dpsi, dg)
    return bad, "%s hb=%d %s  %s" % (ret[0]["mode"], ret[0]["hb"], ret[0]["driver"], note)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--only", type=int, default=None, help="run case number ONLY of the drawn sequence")
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    cases = [draw(rng) for _ in range(args.cases)]
    from helpers import spawn_collect
    print("# %s --cases %d --seed %d%s" % (os.path.basename(__file__), args.cases, args.seed, " --cpu" if args.cpu else ""), flush=True)
    failures = eig = 0
    for i, case in enumerate(cases):
        if args.only is not None and i != args.only:
            continue
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        M, _ = matrix(case)
        try:
            ret = spawn_collect(worker, (case["world"], port, case, args.cpu), case["world"], port_index=1)
            bad, line = judge(case, [ret[r] for r in range(case["world"])], args.cpu)
        except Exception as exc:  # noqa: BLE001 -- a configuration that raises (or hangs: CASE_TIMEOUT) is a finding to print
            text = " ".join(str(exc).split())
            bad, line = ["raised: %s: %s" % (type(exc).__name__, text[-400:])], ""
        failures += bool(bad)
        eig += "vs eigh" in line
        print("%-4s #%d world=%d n=%4d (slab %4d) %-11s nnz=%6d %s | %s%s" % (
            "FAIL" if bad else "ok", i, case["world"], case["n"], -(-case["n"] // case["world"]), case["kind"], M.nnz,
            "python" if case["python_driver"] else ("rccl*" if case.get("rccl") and not args.cpu else "library"), line, ("  <-- " + "; ".join(bad)) if bad else ""), flush=True)
    print("cases %d  failures %d  (%d with the eigen-solve + adjoint against dense eigh)" % (len(cases), failures, eig))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
