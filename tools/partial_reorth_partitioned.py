#!/usr/bin/env python3
"""Row-partitioned library driver (dsea_pop_lanczos_run / dsea_pop_cg_run, one rank over RCCL) at a slab of 2^L rows:
forward + backward with the reference's schedule and with the partial re-orthogonalisation option.
    python tools/partial_reorth_partitioned.py [--L 25] [--k 200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from dominantsparseeigenad_amd import engine
from dominantsparseeigenad_amd.partitioned import PartitionedTFIM
from dominantsparseeigenad_amd.synthetic import normal_vector
L = int(sys.argv[sys.argv.index("--L") + 1]) if "--L" in sys.argv else 25
k = int(sys.argv[sys.argv.index("--k") + 1]) if "--k" in sys.argv else 200
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 1 << L
g = torch.tensor([1.0], dtype=torch.float64, device=dev)
solver = PartitionedTFIM(L, g, dev, eps=1e-7, comm=None)
solver.op.force_driver = True
q0 = torch.from_numpy(normal_vector(n, 5100)).to(dev); x0 = torch.from_numpy(normal_vector(n, 5102)).to(dev)
t = torch.from_numpy(normal_vector(n, 5103)).to(dev)
res = {}
for mode in ("full", "partial"):
    engine.PARTIAL_REORTH = 0.0 if mode == "partial" else None
    best = 1e30
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        E0, psi, grad = solver.forward_backward(k, q0, x0, t)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    res[mode] = (E0.item(), grad.item(), best, engine.last_reorth_steps, solver.last_cg_iters)
engine.PARTIAL_REORTH = None
(Ef, gf, tf, _, mf), (Ep, gp, tp, steps, mp_) = res["full"], res["partial"]
print("row-partitioned library driver (%s), 2^%d rows on one rank, k=%d: full %.1f ms (CG %d its)  partial %.1f ms (CG %d its, %d of %d steps "
      "re-orthogonalised)  E0 rel. dev %.1e  dloss/dg rel. dev %.1e" % (solver.op.driver, L, k, tf * 1e3, mf, tp * 1e3, mp_, steps, k - 1,
                                                                        abs(Ef - Ep) / abs(Ef), abs(gf - gp) / abs(gf)))
dist.destroy_process_group()
