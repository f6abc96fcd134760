"""world_size 2, 4 and 8 over gloo on CPU (2: pairwise slab exchange; 4, 8: transposed all-to-all form): the row-partitioned driver (partition, hypercube exchange, all-reduce
placement, identical branch on all ranks) against the single-process CPU oracle.  The slab-local numerics
come from a torch-CPU test double (tests/cpu_backend.py); on the GPU box the same driver runs on HipBackend."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from dominantsparseeigenad_amd.synthetic import normal_vector
from helpers import SeedDraws

L, K, G = 8, 120, 1.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from cpu_backend import CpuBackend
        from dominantsparseeigenad_amd.partitioned import PartitionedTFIM
        p = world.bit_length() - 1
        nloc = 1 << (L - p)
        off = rank * nloc
        g = torch.tensor([G], dtype=torch.float64)
        be = CpuBackend(L, L - p, off, g)
        solver = PartitionedTFIM(L, g, "cpu", backend=be, eps=1e-12)
        q0 = torch.from_numpy(normal_vector(nloc, 5000, offset=off))
        x0 = torch.from_numpy(normal_vector(nloc, 5002, offset=off))
        t = torch.from_numpy(normal_vector(nloc, 5003, offset=off))
        E0, psi, grad = solver.forward_backward(K, q0, x0, t)
        ret[rank] = (E0.item(), psi.numpy().copy(), grad.item(), solver.last_cg_iters)  # by value, not shm
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partitioned_matches_single_process_oracle(world):
    n = 1 << L
    model = oracle.TFIMTables(L)
    model.g = torch.tensor([G], dtype=torch.float64, requires_grad=True)
    # the oracle consumes draws in the order q0, (unused), x0 -> seeds 5000, 5001, 5002
    f = oracle.make_sparse_dominant_symeig(model.H, model.adjoint_hook, draw=SeedDraws(5000), eps=1e-12).apply
    t = torch.from_numpy(normal_vector(n, 5003))
    E_o, psi_o = f(model.g, K, n)
    (g_o,) = torch.autograd.grad(E_o + psi_o.matmul(t), model.g)

    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    psi = torch.cat([torch.from_numpy(ret[r][1]) for r in range(world)])
    sgn = 1.0 if float(psi @ psi_o.detach()) > 0 else -1.0
    for r in range(world):
        assert ret[r][0] == ret[0][0] and ret[r][2] == ret[0][2]      # replicated scalars are bit-identical
        assert ret[r][3] == ret[0][3]                                    # same CG branch everywhere
    assert abs(ret[0][0] - E_o.item()) < 1e-12 * abs(E_o.item())
    assert float((psi * sgn - psi_o.detach()).abs().max()) < 1e-10
    # the loss used psi with this run's sign; compare with the oracle's gradient for the same sign
    if sgn < 0:
        (g_o,) = torch.autograd.grad(E_o - psi_o.matmul(t), model.g)
    assert abs(ret[0][2] - g_o.item()) < 1e-9 * abs(g_o.item()), (ret[0][2], g_o.item())
