#!/bin/bash
# A/B of library variants on config 3 (stencil N = 1e5, k = 300 Lanczos) and mid-size TFIM (L = 14, 16), same box
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in $1; do
  export DSEA_LIB=$PWD/dominantsparseeigenad_amd/csrc/libdsea_$v.so
  python - "$v" <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from dominantsparseeigenad_amd.operators import Stencil3Operator, TFIMOperator
from dominantsparseeigenad_amd.Lanczos import symeigLanczos
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0")
out = []
def run(op, n, k):
    q0 = torch.from_numpy(normal_vector(n, 1)).to(dev)
    best = 1e30
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lam, _ = symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=n, q0=q0)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / k * 1e6, lam.item()
for N in (20000, 100000):
    x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
    out.append("stencil N=%d: %.2f us/step" % ((N,) + run(Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2), N, 300)[:1]))
for L in (14, 16):
    op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=torch.float64, device=dev))
    out.append("TFIM L=%d: %.2f us/step" % ((L,) + run(op, 1 << L, 300)[:1]))
print(sys.argv[1], " | ".join(out))
PY
done; done
