"""Re-compute the reference's stored curves (examples/TFIM/datas/E0_N_{N}.npz and chiF_N_{N}.npz: E0, dE0/dg,
d2E0/dg2 and chi_F on 100 couplings) on the device and report the deviation and the wall time.

    python examples/TFIM/sweep.py --N 20 --k 200 --data tests/golden/ref_datas [--points 100]

At N = 20 one point is a forward pass plus a second-order backward (three CG solves) plus the chi_F graph; the
reference's CPU path needs ~43 s for forward + first-order backward alone at that size.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from TFIM import TFIM  # noqa: E402
from E0 import E0_sparseAD  # noqa: E402
from chiF import chiF_sparseAD  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=20)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--points", type=int, default=100)
    ap.add_argument("--data", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "datas"))
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--warm", type=int, default=0,
                    help="Lanczos vectors per forward pass from the second coupling on, with the previous coupling's "
                         "eigenvector as start vector (Lanczos.WARM_START; an extension the reference lacks)")
    ap.add_argument("--skip-zero-rhs", type=int, default=0,
                    help="1: symeig.SKIP_ZERO_RHS -- the first backward of every point (a loss that ignores the "
                         "eigenvector) skips its CG solve of (A - E0) x = 0: two solves per E0 point instead of three")
    ap.add_argument("--reorth", choices=["full", "partial", "twice", "none"], default="full",
                    help="Lanczos.REORTH_DEFAULT for the primitives: 'full' = the reference's schedule (Lanczos.py:66); "
                         "'partial' = re-orthogonalise only when the omega recurrence asks for it (an option the reference lacks)")
    args = ap.parse_args()
    import DominantSparseEigenAD.Lanczos as LZ
    import DominantSparseEigenAD.symeig as SE
    SE.SKIP_ZERO_RHS = bool(args.skip_zero_rhs)
    LZ.REORTH_DEFAULT = args.reorth
    curE = np.load(os.path.join(args.data, "E0_N_%d.npz" % args.N))
    curC = np.load(os.path.join(args.data, "chiF_N_%d.npz" % args.N))
    dev = torch.device(args.device)
    model = TFIM(args.N, dev)
    idxs = np.linspace(0, len(curE["gs"]) - 1, args.points).round().astype(int)
    dev_E = dev_d = dev_d2 = dev_c = 0.0
    torch.manual_seed(0)
    prev = None
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.time()
    for idx in idxs:
        g = float(curE["gs"][idx])
        model.g = torch.tensor([g], dtype=torch.float64, device=dev, requires_grad=True)
        k = args.k
        if args.warm and prev is not None:
            k = args.warm
            LZ.WARM_START = prev
        e, de, d2e = E0_sparseAD(model, k)
        if args.warm and prev is not None:
            LZ.WARM_START = prev
        _, psi, c = chiF_sparseAD(model, k)
        prev = psi.detach()
        dev_E = max(dev_E, abs(e - curE["E0s"][idx]) / abs(curE["E0s"][idx]))
        dev_d = max(dev_d, abs(de - curE["dE0s"][idx]) / abs(curE["dE0s"][idx]))
        dev_d2 = max(dev_d2, abs(d2e - curE["d2E0s"][idx]) / abs(curE["d2E0s"][idx]))
        dev_c = max(dev_c, abs(c - curC["chiFs"][idx]) / abs(curC["chiFs"][idx]))
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dt = time.time() - t0
    LZ.REORTH_DEFAULT = "full"
    print("N=%d k=%d reorth=%s: %d couplings in %.2f s (%.1f ms per point: E0, dE0, d2E0 and chi_F)" % (
        args.N, args.k, args.reorth, len(idxs), dt, dt / len(idxs) * 1e3))
    print("max relative deviation from the reference's stored curves:  E0 %.1e  dE0 %.1e  d2E0 %.1e  chiF %.1e" % (
        dev_E, dev_d, dev_d2, dev_c))
    return dev_E, dev_d, dev_d2, dev_c, dt


if __name__ == "__main__":
    main()
