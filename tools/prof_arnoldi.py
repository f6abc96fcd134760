import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dominantsparseeigenad_amd import krylov
dev = torch.device("cuda:0"); D = 512; n = D * D
torch.manual_seed(0)
Ad = torch.randn(2, D, D, dtype=torch.float64, device=dev) / D ** 0.5; AdT = Ad.transpose(1, 2).contiguous()
fr = lambda v: torch.matmul(torch.matmul(Ad, v.reshape(D, D)), AdT).sum(0).reshape(-1)
krylov.arnoldi_dominant(fr, n, 200, dev)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
lam, x = krylov.arnoldi_dominant(fr, n, 200, dev)
torch.cuda.synchronize(); t1 = time.perf_counter(); pr.disable()
print("arnoldi: %.1f ms lambda %.10f" % ((t1 - t0) * 1e3, lam))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:3000])
