"""us per CG iteration of the native TFIM solve at L = 20 (fixed iteration count): the figure docs/design/04-kernels.md quotes for the streaming form."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator
dev = torch.device("cuda:0")
L = 20; n = 1 << L
op = TFIMOperator(L, dev); op.g = torch.tensor([1.0], dtype=torch.float64, device=dev)
b = torch.randn(n, dtype=torch.float64, device=dev); x0 = torch.randn(n, dtype=torch.float64, device=dev)
shift = torch.tensor(-30.0, dtype=torch.float64, device=dev)
best = 1e9
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    engine.cg(b, x0, native=op, shift=shift, eps=0.0, maxiter=400, poll_every=400)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("TFIM L=20 CG: %.2f us per iteration (%d iterations)" % (best / engine.last_cg.iters * 1e6, engine.last_cg.iters))
