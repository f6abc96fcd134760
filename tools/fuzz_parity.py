#!/usr/bin/env python3
"""Random-configuration parity campaign (not a test: run it on the GPU box to look for edge cases):
native operators of random kinds and ragged sizes, Lanczos extreme pairs and CG solves against the CPU oracle on the
same start vectors.    python tools/fuzz_parity.py [--cases 150] [--seed 0]"""
import argparse, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
import oracle
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.Lanczos import symeigLanczos
from dominantsparseeigenad_amd.CG import CG_torch
from dominantsparseeigenad_amd.operators import CSROperator, Stencil3Operator, TFIMOperator, dense_symmetric_operand

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=150); ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0"); F64 = torch.float64
rng = np.random.RandomState(args.seed)
bad = 0
ill_posed = 0


def report(tag, ok, msg):
    global bad
    if not ok:
        bad += 1
        print("MISMATCH", tag, msg, flush=True)


def make_case():
    kind = rng.choice(["stencil", "csr", "csr-plain", "tfim", "dense"])
    if kind == "stencil":
        n = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 127, 129, 511, 513, 1000, 4095, 4097, int(rng.randint(1, 70000))]))
        h = 2.0 / max(n, 2)
        V = torch.from_numpy(rng.rand(n) * 3.0)
        ref = oracle.Stencil3(n, h, V)
        return kind, n, Stencil3Operator(n, h, V.to(dev)), ref.H, 4.0 * 0.5 / h ** 2 + 3.0
    if kind in ("csr", "csr-plain"):
        n = int(rng.choice([1, 2, 7, 64, 65, 200, 1000, int(rng.randint(1, 6000))]))
        dens = min(1.0, rng.choice([0.0, 2.0, 8.0, 30.0]) / max(n, 1))
        M = sp.random(n, n, density=dens, random_state=rng, format="csr")
        M = (M + M.T) * 0.5 + sp.diags(rng.rand(n) * 2.0 + (1.0 if rng.rand() < 0.5 else 0.0))
        if rng.rand() < 0.3 and n > 3:      # some completely empty rows / columns
            M = M.tolil(); idx = rng.randint(0, n, size=max(1, n // 10)); M[idx, :] = 0; M[:, idx] = 0; M = M.tocsr()
        M = M.tocsr(); M.eliminate_zeros()
        if kind == "csr" and rng.rand() < 0.35 and M.nnz:      # few distinct values: the host layer value-codes such an operand
            levels = rng.rand(int(rng.randint(2, 40))) + 0.25
            M.data = levels[rng.randint(0, len(levels), size=M.nnz)]
            M = ((M + M.T) * 0.5).tocsr()                      # (still few: at most len(levels)^2 averages)
        Md = torch.from_numpy(M.toarray())
        return kind, n, CSROperator.from_scipy(M, dev, layout="sell" if kind == "csr" else "csr"), (lambda v, Md=Md: Md @ v), float(Md.abs().sum(1).max())
    if kind == "tfim":
        L = int(rng.randint(1, 14)); g = float(rng.choice([0.5, 1.0, 1.7]))
        n = 1 << L
        op = TFIMOperator(L, dev); op.g = torch.tensor([g], dtype=F64, device=dev)
        tab = oracle.TFIMTables(L); tab.g = torch.tensor([g], dtype=F64)
        return kind, n, op, tab.H, L * (1.0 + g)
    n = int(rng.choice([2, 5, 64, 100, 257, int(rng.randint(2, 900))]))
    A = torch.from_numpy(rng.randn(n, n)); A = A + A.T
    return "dense", n, A.to(dev), A, float(A.abs().sum(1).max())


t0 = time.time()
for case in range(args.cases):
    kind, n, op, refmap, bound = make_case()
    k = int(min(rng.choice([1, 2, 3, 5, 17, 64, 120]), max(n, 1)))
    if rng.rand() < 0.1:
        k = min(n + 3, 150)            # beyond the Krylov dimension: breakdown handling
    engine.USE_SHADOW = bool(rng.rand() < 0.7)
    tag = "%s n=%d k=%d shadow=%s" % (kind, n, k, engine.USE_SHADOW)
    q0 = torch.from_numpy(rng.randn(n))
    draws = iter([q0.clone(), torch.zeros(n, dtype=F64)])
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if kind == "dense":
                lo, vlo, hi, vhi = symeigLanczos(op, k, dev, extreme="both", q0=q0.to(dev))
            else:
                lo, vlo, hi, vhi = symeigLanczos(op, k, dev, extreme="both", sparse=True, dim=n, q0=q0.to(dev))
        kk = min(k, n)
        if kind == "dense":
            rlo, rvlo, rhi, rvhi = oracle.symeig_lanczos(refmap, kk, "both", draw=lambda m, dt: next(draws))
            apply_ref = lambda v: refmap @ v
        else:
            rlo, rvlo, rhi, rvhi = oracle.symeig_lanczos(refmap, kk, "both", sparse=True, dim=n, draw=lambda m, dt: next(draws))
            apply_ref = refmap
        scale = max(abs(float(rlo)), abs(float(rhi)), 1e-300)
        # A run that met an invariant subspace (beta ~ 0 before step k) is NOT compared with the oracle's Ritz values: the
        # reference has no breakdown test (Lanczos.py:69-70 divides by beta whatever it is -- SURVEY Q8), so the oracle, which
        # restates it, goes on normalising rounding noise and may return spurious Ritz values; the device loop records the
        # step and takes the pair from the leading block.  Such cases are held to the TRUE extreme eigenvalues below.
        broke = engine.last_break > 0 or getattr(engine, "last_truncated", 0) > 0
        if k <= n and torch.isfinite(rvlo).all() and not broke:
            okr = abs(float(lo) - float(rlo)) <= 1e-9 * scale and abs(float(hi) - float(rhi)) <= 1e-9 * scale
            if not okr and kind != "dense":
                # is the CASE ill-posed at this tolerance?  The oracle against itself with q0 perturbed in the last bit: an
                # unconverged extreme Ritz value of a matrix with empty rows (exact zero eigenvalues) can move by 1e-8 on that
                # (round 6, seed 1: oracle 2.99920648929 / perturbed 2.99920647965 / three device kernels within the same spread)
                q0b = q0 * (1.0 + 1e-15 * torch.from_numpy(np.random.RandomState(case).randn(n)))
                drp = iter([q0b.clone(), torch.zeros(n, dtype=F64)])
                pl, _, ph, _ = oracle.symeig_lanczos(refmap, kk, "both", sparse=True, dim=n, draw=lambda m, dt: next(drp))
                spread = max(abs(float(pl) - float(rlo)), abs(float(ph) - float(rhi)))
                if spread > 0.3e-9 * scale:
                    ill_posed += 1
                    print("   ill-posed at 1e-9 (%s): the oracle moves by %.1e relative when q0 is perturbed by 1e-15; device %.15g %.15g, "
                          "oracle %.15g %.15g" % (tag, spread / scale, float(lo), float(hi), float(rlo), float(rhi)), flush=True)
                    okr = max(abs(float(lo) - float(rlo)), abs(float(hi) - float(rhi))) <= 30.0 * spread
            report(tag, okr, "ritz values %.15g %.15g vs %.15g %.15g" % (float(lo), float(hi), float(rlo), float(rhi)))
            if not okr and kind in ("csr", "csr-plain") and os.environ.get("DSEA_FUZZ_DIAG"):
                # which ingredient moves the value: the operand's layout / kernel variant, the shadow, or the oracle's own sensitivity
                from dominantsparseeigenad_amd import _lib as _l
                Mfull = torch.stack([apply_ref(e) for e in torch.eye(n, dtype=F64)], 1)
                w = torch.linalg.eigvalsh(0.5 * (Mfull + Mfull.T))
                print("   true extremes %.15g %.15g" % (float(w[0]), float(w[-1])))
                for lay, c16, unroll, xcd, sh in (("sell", "auto", 0, 1, True), ("sell", "auto", 0, 1, False), ("sell", False, 0, 1, False),
                                                  ("sell", False, 1, 0, False), ("csr", False, 0, 0, False)):
                    o2 = CSROperator(op.rowptr, op.colidx, op.vals.clone(), n, layout=lay, col16=c16)
                    if lay == "sell":
                        _l.load().dsea_op_set_tuning(o2._H.handle, 3, unroll)
                        _l.load().dsea_op_set_tuning(o2._H.handle, 4, xcd)
                    engine.USE_SHADOW = sh
                    l2, _, h2, _ = symeigLanczos(o2, k, dev, extreme="both", sparse=True, dim=n, q0=q0.to(dev))
                    print("   %s col16=%s unroll=%d xcd=%d shadow=%s: %.15g %.15g" % (lay, c16, unroll, xcd, sh, float(l2), float(h2)))
                q0b = q0 * (1.0 + 1e-15 * torch.from_numpy(rng.randn(n)))
                dr = iter([q0b.clone(), torch.zeros(n, dtype=F64)])
                pl, _, ph, _ = oracle.symeig_lanczos(refmap, kk, "both", sparse=True, dim=n, draw=lambda m, dt: next(dr))
                print("   oracle with q0 perturbed by 1e-15 relative: %.15g %.15g" % (float(pl), float(ph)))
        elif k <= n and broke and n <= 400:
            Mfull = torch.stack([apply_ref(e) for e in torch.eye(n, dtype=F64)], 1)
            w = torch.linalg.eigvalsh(0.5 * (Mfull + Mfull.T))
            dl, dh = float((w - float(lo)).abs().min()), float((w - float(hi)).abs().min())
            print("   breakdown at step %d (k = %d, n = %d): device Ritz values are eigenvalues of A to %.1e / %.1e; oracle's lowest "
                  "is off the spectrum by %.1e" % (engine.last_break, k, n, dl, dh, float((w - float(rlo)).abs().min())), flush=True)
            report(tag, dl <= 1e-8 * scale + 1e-12 and dh <= 1e-8 * scale + 1e-12, "Ritz values after a breakdown are not eigenvalues: %.3e %.3e" % (dl, dh))
        # property that does not need the oracle's conditioning: Ritz residual equals the oracle's
        for lam, v in ((lo, vlo), (hi, vhi)):
            vc = v.detach().cpu().to(F64)
            report(tag, bool(torch.isfinite(vc).all()) and abs(float(vc.norm()) - 1.0) < 1e-8, "Ritz vector norm %r" % float(vc.norm()))
        if k >= min(n, 120) and n <= 120:      # exact Krylov space: compare with the true extreme eigenvalues
            Mfull = torch.stack([apply_ref(e) for e in torch.eye(n, dtype=F64)], 1)
            w = torch.linalg.eigvalsh(0.5 * (Mfull + Mfull.T))
            # (q0 may miss an eigenvector only for special operators: TFIM sectors are all hit by a random q0)
            report(tag, abs(float(lo) - float(w[0])) <= 1e-8 * scale and abs(float(hi) - float(w[-1])) <= 1e-8 * scale,
                   "exact-space eigenvalues %.15g %.15g vs %.15g %.15g" % (float(lo), float(hi), float(w[0]), float(w[-1])))
    except Exception as exc:  # noqa: BLE001
        report(tag, False, "Lanczos raised %s: %s" % (type(exc).__name__, str(exc)[:200]))
        continue
    # CG on the shifted SPD system (A + s) x = b, s chosen from the spectrum found above
    if n >= 2 and kind != "dense":
        s = bound + 1.0                     # Gershgorin: A + s is SPD with condition number <= 2 bound + 1
        b = torch.from_numpy(rng.randn(n)); x0 = torch.from_numpy(rng.randn(n))
        iters = int(rng.choice([1, 3, 20]))
        try:
            st = {}
            eps = 1e-9 * float(b.norm())     # (iterating past convergence divides 0 by 0 in any CG: stop like the reference does)
            xo = oracle.cg_solve(lambda v: apply_ref(v) + s * v, b, x0, sparse=True, eps=eps, maxiter=iters, stats=st)
            xd = engine.cg(b.to(dev), x0.to(dev), native=op, shift=torch.tensor(-s, dtype=F64, device=dev), eps=eps, maxiter=iters)
            report(tag + " cg%d" % iters, engine.last_cg.iters == st["iters"], "CG iterations %d vs %d" % (engine.last_cg.iters, st["iters"]))
            report(tag + " cg%d" % iters, float((xd.cpu() - xo).abs().max()) <= 1e-9 * max(float(xo.abs().max()), 1e-6 * float(x0.abs().max())),
                   "CG iterate dev %.3e" % float((xd.cpu() - xo).abs().max()))
        except Exception as exc:  # noqa: BLE001
            report(tag, False, "CG raised %s: %s" % (type(exc).__name__, str(exc)[:200]))
    # round 6: the explicit matrix as a parameter -- the SDDMM hook (plain, symmetric, accumulating) and the in-place refresh
    # against torch index arithmetic, bit for bit, on the same ragged patterns (16-bit and 32-bit columns, both layouts)
    if kind in ("csr", "csr-plain") and op.nnz > 0:
        try:
            rows = torch.repeat_interleave(torch.arange(n), (op.rowptr[1:] - op.rowptr[:-1]).cpu())
            cols = op.colidx.cpu().long()
            v1, v2 = torch.from_numpy(rng.randn(n)), torch.from_numpy(rng.randn(n))
            plain = v1[rows] * v2[cols]
            report(tag + " sddmm", torch.equal(op.sddmm(v1.to(dev), v2.to(dev)).cpu(), plain), "plain")
            symm = 0.5 * (v1[rows] * v2[cols] + v1[cols] * v2[rows])
            report(tag + " sddmm", torch.equal(op.sddmm(v1.to(dev), v2.to(dev), symmetric=True).cpu(), symm), "symmetric")
            acc0 = torch.from_numpy(rng.randn(op.nnz))
            out = acc0.clone().to(dev)
            op.sddmm(v1.to(dev), v2.to(dev), out=out, alpha=0.375, accumulate=True)
            report(tag + " sddmm", torch.equal(out.cpu(), acc0 + 0.375 * plain), "accumulate")
            newv = torch.from_numpy(rng.randn(op.nnz)).to(dev)
            op.vals.copy_(newv)
            fresh = CSROperator(op.rowptr, op.colidx, newv.clone(), n, layout=op.layout, col16=("auto" if op.col16 else False))
            x = torch.from_numpy(rng.randn(n)).to(dev)
            report(tag + " update_vals", torch.equal(op(x), fresh(x)), "in-place refresh differs from a rebuild")
        except Exception as exc:  # noqa: BLE001
            report(tag, False, "parameter kernels raised %s: %s" % (type(exc).__name__, str(exc)[:200]))
engine.USE_SHADOW = True
print("cases %d  mismatches %d  ill-posed at the tolerance (oracle moves under a last-bit change of q0) %d  %.1f s"
      % (args.cases, bad, ill_posed, time.time() - t0))
