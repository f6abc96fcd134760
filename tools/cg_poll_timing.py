"""Host polling of the streaming CG: pipelined state copies against a drain per poll (profiles/r03_cg_pipelined_polling.txt)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.operators import TFIMOperator
from dominantsparseeigenad_amd.synthetic import normal_vector
dev = torch.device("cuda:0")
for L in (18, 20, 21, 22):
    n = 1 << L
    op = TFIMOperator(L, dev, g=torch.tensor([1.0], dtype=torch.float64, device=dev))
    A = op.to_csr(layout="sell") if L <= 20 else None
    b = torch.from_numpy(normal_vector(n, 2)).to(dev); x0 = torch.from_numpy(normal_vector(n, 3)).to(dev)
    shift = torch.tensor(-30.0, dtype=torch.float64, device=dev)
    for name, nat in (("matrix-free streaming", op), ("SELL", A)):
        if nat is None: continue
        ws = engine.Workspace.get(n, 8, dev); ws.set_persist(0)
        best = 1e30
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            engine.cg(b, x0, native=nat, shift=shift, eps=0.0, maxiter=200)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        ws.set_persist(-1)
        print("L=%d %-22s %.2f us/iteration (poll every 8)" % (L, name, best / 200 * 1e6))
