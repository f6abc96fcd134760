#!/usr/bin/env python3
"""Host-side cost per Lanczos step of the row-partitioned driver (tiny slab: device time negligible)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from dominantsparseeigenad_amd.partitioned import PartitionedTFIM
from dominantsparseeigenad_amd.synthetic import normal_vector
L, k = 8, 200
g = torch.tensor([1.0], dtype=torch.float64, device=dev)
solver = PartitionedTFIM(L, g, dev)
q0 = torch.from_numpy(normal_vector(1 << L, 1)).to(dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    E0, psi = solver.forward(k, q0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("partitioned forward, n=256, k=%d: %.2f ms  -> %.1f us per step (host-dominated)" % (k, (t1 - t0) * 1e3, (t1 - t0) / k * 1e6))
x = torch.zeros(8, dtype=torch.float64, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): dist.all_reduce(x)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("dist.all_reduce (world 1) host+device: %.1f us per call" % ((t1 - t0) / 200 * 1e6))
src = torch.zeros(1 << 20, dtype=torch.float64, device=dev); dst = torch.zeros_like(src)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): dist.all_to_all_single(dst, src)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("dist.all_to_all_single 8 MiB (world 1): %.1f us per call" % ((t1 - t0) / 200 * 1e6))
dist.destroy_process_group()
