import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from dominantsparseeigenad_amd import engine, krylov
dev = torch.device("cuda:0"); D = 512; n = D * D
torch.manual_seed(0)
Ad = torch.randn(2, D, D, dtype=torch.float64, device=dev) / D ** 0.5; AdT = Ad.transpose(1, 2).contiguous()
fr = lambda v: torch.matmul(torch.matmul(Ad, v.reshape(D, D)), AdT).sum(0).reshape(-1)
v = torch.randn(n, dtype=torch.float64, device=dev)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("matvec (2x2 GEMM 512^3 fp64 + sum): %.1f us" % timeit(lambda: fr(v)))
ph = engine.Phases(n, dev, kmax=32); ldq = engine.round_up(n, 32)
V = torch.randn(22, ldq, dtype=torch.float64, device=dev); zero = ph.zeros(1)
bufs = (ph.empty(n), ph.empty(n), ph.zeros(24), ph.zeros(24), ph.zeros(1))
for j in (2, 10, 19):
    print("cgs2 j=%d: %.1f us" % (j, timeit(lambda: krylov._cgs2(ph, V, ldq, n, j, v, zero, bufs))))
Vb = torch.randn(202, ldq, dtype=torch.float64, device=dev)
ph2 = engine.Phases(n, dev, kmax=204); bufs2 = (ph2.empty(n), ph2.empty(n), ph2.zeros(204), ph2.zeros(204), ph2.zeros(1))
for j in (50, 150, 199):
    print("cgs2 j=%d: %.1f us" % (j, timeit(lambda: krylov._cgs2(ph2, Vb, ldq, n, j, v, zero, bufs2))))
# ---- where does a GMRES solve spend its time?
import cProfile, pstats, io
b = torch.randn(n, dtype=torch.float64, device=dev)
lam = 5.0   # well outside the spectrum (radius ~2): GMRES converges in a few cycles
mv = lambda x: fr(x) - lam * x
krylov.gmres(mv, b, maxiter=50)   # warm
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
x = krylov.gmres(mv, b, maxiter=50)
torch.cuda.synchronize(); t1 = time.perf_counter(); pr.disable()
print("gmres solve: %.1f ms, residual %.2e" % ((t1 - t0) * 1e3, float((mv(x) - b).norm())))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2500])
