#!/usr/bin/env python3
"""Random-configuration campaign for the partial re-orthogonalisation option (not a test: run it on the GPU box).
Operators of random kinds, sizes and spectra -- including matrices with planted well-separated extreme eigenvalues, where
Ritz values converge early and orthogonality is lost fastest -- with reorth="partial" against the reference's schedule on the
same start vector:  extreme Ritz pair (value 1e-11 ||A||; vector 1e-9 when the pair has converged), semi-orthogonality
of the basis (||Q^T Q - I||_max <= 10 x the threshold 1e-10), no spurious copies of converged Ritz values in T.
    python tools/fuzz_partial_reorth.py [--cases 200] [--seed 0]"""
import argparse, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.Lanczos import symeigLanczos, Lanczos
from dominantsparseeigenad_amd.operators import CSROperator, Stencil3Operator, TFIMOperator

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=200); ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0"); F64 = torch.float64
rng = np.random.RandomState(args.seed)
DELTA = 1e-10          # the option's default threshold
bad = 0
stats = {"cases": 0, "steps": 0, "reorth": 0, "worst_orth": 0.0, "worst_dE": 0.0, "worst_dpsi": 0.0}


def planted(n):
    """dense symmetric matrix with a few well-separated extreme eigenvalues on top of a bulk"""
    U, _ = np.linalg.qr(rng.randn(n, n))
    m = int(rng.randint(1, 6))
    bulk = np.sort(rng.rand(n - 2 * m) * 10.0)
    ev = np.concatenate([-20.0 - 10.0 * np.arange(m)[::-1], bulk, 30.0 + 10.0 * np.arange(m)])
    A = (U * ev) @ U.T
    return torch.from_numpy(0.5 * (A + A.T))


def make_case():
    kind = rng.choice(["tfim", "stencil", "sell", "csr-plain", "planted-dense", "planted-callable"])
    if kind == "tfim":
        L = int(rng.randint(4, 18)); g = float(rng.choice([0.6, 0.9, 1.0, 1.3, 2.0]))     # (from L = 15 the bf16 shadow is in play)
        op = TFIMOperator(L, dev); op.g = torch.tensor([g], dtype=F64, device=dev)
        return kind + " L=%d g=%.1f" % (L, g), 1 << L, op, dict(sparse=True, dim=1 << L), L * (1.0 + g)
    if kind == "stencil":
        n = int(rng.choice([64, 129, 300, 1000, int(rng.randint(65, 30000)), int(rng.randint(32768, 140000))]))
        h = float(rng.choice([2.0 / n, 0.1, 1.0]))          # h = O(1): a well-conditioned operator whose Ritz values converge
        V = torch.from_numpy(rng.rand(n) * 3.0).to(dev)
        return kind + " n=%d h=%.3g" % (n, h), n, Stencil3Operator(n, h, V), dict(sparse=True, dim=n), 2.0 / h ** 2 + 3.0
    if kind in ("sell", "csr-plain"):
        n = int(rng.choice([100, 400, 1000, int(rng.randint(64, 5000))]))
        M = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng, format="csr")
        d = rng.rand(n) * 2.0
        d[: int(rng.randint(1, 5))] -= 15.0                  # a few isolated low eigenvalues
        M = ((M + M.T) * 0.5 + sp.diags(d)).tocsr()
        Md = torch.from_numpy(M.toarray())
        return kind + " n=%d" % n, n, CSROperator.from_scipy(M, dev, layout="sell" if kind == "sell" else "csr"), dict(sparse=True, dim=n), float(Md.abs().sum(1).max())
    n = int(rng.choice([128, 300, 600, int(rng.randint(64, 900))]))
    A = planted(n).to(dev)
    if kind == "planted-dense":
        return kind + " n=%d" % n, n, A, dict(), float(A.abs().sum(1).max())
    return kind + " n=%d" % n, n, (lambda v, A=A: A @ v), dict(sparse=True, dim=n), float(A.abs().sum(1).max())


t0 = time.time()
for case in range(args.cases):
    tag, n, op, kw, anorm = make_case()
    k = int(min(rng.choice([20, 50, 100, 150, 250]), n - 1))
    tag = "%s k=%d" % (tag, k)
    q0 = torch.from_numpy(rng.randn(n)).to(dev)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lo_f, v_f, hi_f, w_f = symeigLanczos(op, k, dev, extreme="both", q0=q0, **kw)
            lo_p, v_p, hi_p, w_p = symeigLanczos(op, k, dev, extreme="both", q0=q0, reorth="partial", **kw)
            steps = engine.last_reorth_steps
            brk = engine.last_break
            engine.PARTIAL_REORTH = 0.0
            try:
                Qp, Tp = Lanczos(op, k, dev, q0=q0, **kw)
            finally:
                engine.PARTIAL_REORTH = None
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print("ERROR", tag, repr(exc), flush=True)
        continue
    m = brk if brk else k                                    # a breakdown truncates the run (both schedules)
    Qm = Qp[:, :m]
    orth = float((Qm.T @ Qm - torch.eye(m, dtype=F64, device=dev)).abs().max())
    dE = max(abs(lo_f.item() - lo_p.item()), abs(hi_f.item() - hi_p.item())) / anorm
    apply = op if callable(op) else (lambda v: op @ v)
    msgs = []
    for (lf, vf, lp, vp) in ((lo_f, v_f, lo_p, v_p), (hi_f, w_f, hi_p, w_p)):
        rf = float((apply(vf) - lf * vf).norm()); rp = float((apply(vp) - lp * vp).norm())
        if rp > 4.0 * rf + 1e-10 * anorm:
            msgs.append("residual %.2e vs %.2e" % (rp, rf))
        if rf < 1e-10 * anorm:
            s = 1.0 if float(vf @ vp) > 0 else -1.0
            dpsi = float((vf - s * vp).abs().max())
            stats["worst_dpsi"] = max(stats["worst_dpsi"], dpsi)
            if dpsi > 1e-9:
                msgs.append("dpsi %.2e" % dpsi)
    if orth > 10 * DELTA:
        msgs.append("orthogonality %.2e" % orth)
    if dE > 1e-11:
        msgs.append("dE/||A|| %.2e" % dE)
    # spurious copies: the extreme eigenvalue of T must be simple to the tolerance of a converged pair
    evT = torch.linalg.eigvalsh(Tp[:m, :m])
    if m >= 3 and float(evT[1] - evT[0]) < 1e-9 * anorm and float((apply(v_p) - lo_p * v_p).norm()) < 1e-9 * anorm:
        evF = torch.linalg.eigvalsh(Lanczos(op, k, dev, q0=q0, **kw)[1][:m, :m])
        if float(evF[1] - evF[0]) > 1e-7 * anorm:
            msgs.append("spurious copy of the lowest Ritz value")
    stats["cases"] += 1; stats["steps"] += m - 1; stats["reorth"] += steps or 0
    stats["worst_orth"] = max(stats["worst_orth"], orth); stats["worst_dE"] = max(stats["worst_dE"], dE)
    if msgs:
        bad += 1
        print("MISMATCH", tag, "; ".join(msgs), "(steps re-orthogonalised %s of %d)" % (steps, m - 1), flush=True)
print("cases %d  mismatches %d  steps %d of which re-orthogonalised %d (%.0f%%)  worst ||QtQ-I|| %.1e  worst |dE|/||A|| %.1e  worst |dpsi| %.1e  (%.0f s)"
      % (stats["cases"], bad, stats["steps"], stats["reorth"], 100.0 * stats["reorth"] / max(stats["steps"], 1), stats["worst_orth"],
         stats["worst_dE"], stats["worst_dpsi"], time.time() - t0))
