"""config 4 forward only, N times (for rocprofv3 --kernel-trace --stats: device time per forward against its wall time)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import dominantsparseeigenad_amd.eig as eig  # noqa: E402
from dominantsparseeigenad_amd import krylov  # noqa: E402
from dominantsparseeigenad_amd.operators import TransferOperator  # noqa: E402
from dominantsparseeigenad_amd.synthetic import normal_vector  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
D, d, k = 512, 2, 200
n = D * D
A = (torch.from_numpy(normal_vector(d * n, 11)).reshape(d, D, D) / D ** 0.5).to(dev)
op, opT = TransferOperator(A), TransferOperator(A, transpose=True)
eig.setDominantSparseEig(op, opT, lambda pieces: torch.zeros_like(A))
if os.environ.get("DSEA_C4_SERIAL") == "1":
    eig.CONCURRENT_SIDES = False
Ar = A.clone().requires_grad_(True)
for _ in range(2):
    eig.DominantSparseEig.apply(Ar, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    lam, l, r = eig.DominantSparseEig.apply(Ar, k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print("forward wall %.2f ms per call (%d calls + 2 warm-up), lambda %.12f" % (dt * 1e3, reps, float(lam)))
diag = getattr(krylov, "DIAG", None)
print("diag:", {kk: getattr(diag, kk) for kk in dir(diag) if not kk.startswith("_")} if diag is not None else None)
