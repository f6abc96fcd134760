#!/usr/bin/env python3
"""Random configurations of the row-partitioned path on ONE GPU (ranks share the device, collectives staged through
the host over gloo): world size (2, 4, 8), chain length, Krylov dimension (k = 40 INCLUDED), coupling, overlapped exchange,
replicated CG -- against the single-device path on the same synthetic vectors, JUDGED AGAINST THE SINGLE-DEVICE PATH'S
OWN SPREAD, at the reference's CG tolerance (eps = 1e-7, CG.py:25) and at the tight one (1e-12).

Why a spread: with k too small the Ritz pair (theta, psi) is not converged; the adjoint system (A - theta) x = b of
reference symeig.py:81 / CG.py:120 is then INDEFINITE (theta lies inside the spectrum), CG on it is not a convergent
process and its result is set by rounding.  Two evaluations of the same single-GPU path that differ only in summation
order then disagree with each other by as much as the partitioned path disagrees with either.  So for every case the
single-GPU path is run in several rounding-different but equally valid geometries

    auto | rows-per-lane 4 | rows-per-lane 16 | split 4 | bf16 shadow off

(include/dsea.h: dsea_ws_set_rows_per_lane / dsea_ws_set_split; engine.USE_SHADOW) and the partitioned result must lie
within  max(tol(eps), 10 x self-spread)  of the default single-GPU result (E0: 1e-10 relative, always).  The
eigen-residual ||H psi - theta psi|| is printed: a self-spread above 1e-8 must come with a pair that is unconverged AT
THAT eps (residual > eps: b = t - (psi.t) psi then keeps a component of size ~residual along the true eigenvector, which
(A - theta) amplifies by 1 / (theta - lambda_min) ~ gap / residual^2 once CG resolves it).

    python tools/fuzz_partitioned.py [--cases 30] [--seed 0]          (worker processes are started once per world size)
"""
import argparse
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

VARIANTS = ("auto", "rpl4", "rpl16", "split4", "noshadow")
EPS_LIST = (1e-7, 1e-12)      # the reference's hard-coded CG tolerance (CG.py:25) and the tight one of the 1e-10 parity tests
TOL = {1e-7: 2e-8, 1e-12: 1e-8}


def single(L, k, g0, variant="auto", eps=1e-12):
    import dominantsparseeigenad_amd.CG as CG
    import dominantsparseeigenad_amd.symeig as symeig
    from dominantsparseeigenad_amd import engine
    from dominantsparseeigenad_amd.operators import TFIMOperator
    from dominantsparseeigenad_amd.synthetic import normal_vector
    from helpers import PatchRandn
    dev = torch.device("cuda:0")
    n = 1 << L
    g = torch.tensor([g0], dtype=torch.float64, device=dev, requires_grad=True)
    op = TFIMOperator(L, dev)
    op.g = g
    symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)
    t = torch.from_numpy(normal_vector(n, 77)).to(dev)
    t = t / t.norm()
    ws = engine.Workspace.get(n, k, dev)
    old_eps, old_shadow = CG.EPS_DEFAULT, engine.USE_SHADOW
    CG.EPS_DEFAULT = eps
    try:
        if variant == "rpl4":
            ws.set_rows_per_lane(4)
        elif variant == "rpl16":
            ws.set_rows_per_lane(16)
        elif variant == "split4":
            ws.set_split(4)
        elif variant == "noshadow":
            engine.USE_SHADOW = False
        with PatchRandn(4242):
            E0, psi = symeig.DominantSparseSymeig.apply(g, k, n, dev)
            sgn = 1.0 if float(psi.detach() @ t) > 0 else -1.0
            (gl,) = torch.autograd.grad(E0 + sgn * (psi @ t), g)
        p = psi.detach()
        resid = float((op.H(p) - E0.detach() * p).norm())
    finally:
        ws.set_rows_per_lane(0)
        ws.set_split(-1)
        CG.EPS_DEFAULT, engine.USE_SHADOW = old_eps, old_shadow
    return E0.item(), gl.item(), resid, engine.last_cg.iters, engine.last_cg.converged


def worker(rank, world, port, cases, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dominantsparseeigenad_amd.CG as CG
        import dominantsparseeigenad_amd.symeig as symeig
        from dominantsparseeigenad_amd.partitioned import PartitionedTFIMOperator
        from dominantsparseeigenad_amd.synthetic import normal_vector
        from helpers import PatchRandn
        from test_gpu_partitioned import _host_staged_comm
        out = []
        for (L, k, g0, overlap, replicate) in cases:
            n = 1 << L
            nloc = n // world
            off = rank * nloc
            g = torch.tensor([g0], dtype=torch.float64, device=dev, requires_grad=True)
            op = PartitionedTFIMOperator(L, g, dev, comm=_host_staged_comm(), overlap=overlap)
            op.force_driver = True
            op.replicate_cg = replicate
            symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)
            t_full = torch.from_numpy(normal_vector(n, 77))
            t_full = t_full / t_full.norm()
            t = t_full[off:off + nloc].to(dev)
            per_eps = []
            for eps in EPS_LIST:
                CG.EPS_DEFAULT = eps
                with PatchRandn(4242, offset=off):
                    E0, psi = symeig.DominantSparseSymeig.apply(g, k, op.dim, dev)
                    sgn = 1.0 if op.dot(psi.detach(), t).item() > 0 else -1.0
                    (gl,) = torch.autograd.grad(E0 + sgn * op.dot(psi, t), g)
                torch.cuda.synchronize()
                per_eps.append((E0.item(), gl.item(), op.overlap_fallbacks, op.last_cg_iters))
            out.append(per_eps)
            drv = op.driver
        ret[rank] = (out, drv)
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    cases = []
    for _ in range(args.cases):
        world = int(rng.choice([2, 4, 8]))
        L = int(rng.randint(8, 14))
        if world == 8:
            L = max(L, 12)            # the p = 3 geometry of BASELINE configs[4] on slabs of at least 512 rows
        k = int(rng.choice([40, 90, 120, 150]))
        k = min(k, (1 << L) // 2)
        g0 = float(rng.choice([0.8, 1.0, 1.4]))
        overlap = bool(rng.rand() < 0.6)
        replicate = [("auto"), True, False][int(rng.randint(0, 3))]
        cases.append((world, L, k, g0, overlap, replicate))
    results, drivers = {}, {}
    for world in (2, 4, 8):
        sub = [(i, c) for i, c in enumerate(cases) if c[0] == world]
        if not sub:
            continue
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        from helpers import spawn_collect          # (no Manager: it would fork this process)
        ret = spawn_collect(worker, (world, port, [c[1:] for _, c in sub]), world)
        for j, (i, _) in enumerate(sub):
            results[i] = [ret[r][0][j] for r in range(world)]
        drivers[world] = ret[0][1]
    bad = unexplained = illposed = 0
    worst_ok = 0.0
    print("# %s --cases %d --seed %d   partitioned driver: %s" % (os.path.basename(__file__), args.cases, args.seed, drivers))
    print("# per case and CG tolerance eps: partitioned vs default single-GPU gradient | single-GPU SELF-SPREAD over %d "
          "rounding-different geometries | verdict.  Tolerance max(%g (eps 1e-7) / %g (eps 1e-12), 10 x self-spread); a case "
          "whose single-GPU path disagrees with ITSELF by more than 1e-3 is ILL-POSED (not judged); any self-spread above "
          "1e-8 must come with a Ritz residual above eps (the projected system (A - theta) x = b is then inconsistent at "
          "that tolerance), else it is flagged UNEXPLAINED." % (len(VARIANTS), TOL[1e-7], TOL[1e-12]))
    for i, (world, L, k, g0, overlap, replicate) in enumerate(cases):
        head = "world=%d L=%2d k=%3d g=%.1f overlap=%-5s replicate=%-5s" % (world, L, k, g0, overlap, replicate)
        for e, eps in enumerate(EPS_LIST):
            runs = {v: single(L, k, g0, v, eps=eps) for v in VARIANTS}
            E_s, g_s, resid, it_s, conv_s = runs["auto"]
            spread = max(abs(runs[v][1] - g_s) / abs(g_s) for v in VARIANTS)
            E_spread = max(abs(runs[v][0] - E_s) / abs(E_s) for v in VARIANTS)
            E_p, g_p, fb, it_p = results[i][0][e]
            same = all(tuple(results[i][r][e][:2]) == tuple(results[i][0][e][:2]) for r in range(world))
            dE, dg = abs(E_p - E_s) / abs(E_s), abs(g_p - g_s) / abs(g_s)
            ill = spread > 1e-3
            ok = dE <= 1e-10 and same and E_spread <= 1e-10 and (ill or dg <= max(TOL[eps], 10.0 * spread))
            explained = spread <= 1e-8 or resid > eps
            bad += not ok
            unexplained += not explained
            illposed += ill
            if ok and spread <= 1e-8:
                worst_ok = max(worst_ok, dg)
            verdict = "FAIL" if not ok else ("ILL-POSED" if ill else "ok")
            print("%-9s %s eps=%.0e  E0 dev %.1e  grad dev %.1e | self-spread E0 %.1e grad %.1e | ||H psi - theta psi|| "
                  "%.1e  CG its %d (single) / %d (partitioned)%s  premise fallbacks %d%s"
                  % (verdict, head, eps, dE, dg, E_spread, spread, resid, it_s, it_p,
                     "" if conv_s else " not converged", fb, "" if explained else "  UNEXPLAINED SPREAD"), flush=True)
    print("cases %d x %d tolerances  failures %d  unexplained self-spreads %d  ill-posed (single-GPU path disagrees with "
          "itself by > 1e-3) %d  worst gradient deviation among cases with self-spread <= 1e-8: %.1e"
          % (args.cases, len(EPS_LIST), bad, unexplained, illposed, worst_ok))
    return 1 if (bad or unexplained) else 0


if __name__ == "__main__":
    sys.exit(main())
