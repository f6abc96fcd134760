"""Row-partitioned driver on the real HIP backend.  The GPU box has one MI355X, so:
  * world_size 1 over "nccl" (= RCCL): the whole distributed code path with real kernels, against the
    single-GPU primitive;
  * world_size 2 and 4, all ranks on cuda:0, collectives over gloo (RCCL refuses two ranks on one device): real
    slab operators (L_local < L, row_offset != 0), real phase kernels; 2 ranks = pairwise slab exchange,
    4 ranks = transposed all-to-all form with the HIP flip-sum kernel.
The 8-GPU run itself is the driver's; its logic is also covered by tests/test_partitioned_gloo.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from dominantsparseeigenad_amd.synthetic import normal_vector  # noqa: E402
from helpers import SeedDraws  # noqa: E402

L, K, G = 12, 150, 1.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _host_staged_comm():
    """gloo is not stream-ordered for device tensors: stage through the host around each collective."""
    from dominantsparseeigenad_amd.partitioned import TorchDistComm

    class HostStagedComm(TorchDistComm):
        def allreduce(self, t):
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

        def exchange(self, x, recv, peers):
            hx = x.cpu()
            hr = [torch.empty_like(hx) for _ in peers]
            super().exchange(hx, hr, peers)
            for dst, src in zip(recv, hr):
                dst.copy_(src)

        def all_to_all(self, src, dst):
            hs = src.cpu()
            hd = torch.empty_like(hs)
            super().all_to_all(hs, hd)
            dst.copy_(hd)

    return HostStagedComm()


def _worker(rank, world, port, backend, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dominantsparseeigenad_amd.partitioned import PartitionedTFIM
        p = world.bit_length() - 1
        nloc = 1 << (L - p)
        off = rank * nloc
        g = torch.tensor([G], dtype=torch.float64, device=dev)
        solver = PartitionedTFIM(L, g, dev, eps=1e-12, comm=None if backend == "nccl" else _host_staged_comm())
        q0 = torch.from_numpy(normal_vector(nloc, 5100, offset=off)).to(dev)
        x0 = torch.from_numpy(normal_vector(nloc, 5102, offset=off)).to(dev)
        t = torch.from_numpy(normal_vector(nloc, 5103, offset=off)).to(dev)
        E0, psi, grad = solver.forward_backward(K, q0, x0, t)
        if backend == "nccl":   # the RCCL all-to-all call itself (degenerate at world size 1: a copy)
            src = torch.arange(1024, dtype=torch.float64, device=dev)
            dst = torch.zeros_like(src)
            solver.comm.all_to_all(src, dst)
            assert torch.equal(src, dst)
        torch.cuda.synchronize()
        ret[rank] = (E0.item(), psi.cpu().numpy().copy(), grad.item(), solver.last_cg_iters)  # by value, not shm
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo"), (4, "gloo")])
def test_partitioned_hip_backend(world, backend):
    assert torch.cuda.is_available()
    n = 1 << L
    model = oracle.TFIMTables(L)
    model.g = torch.tensor([G], dtype=torch.float64, requires_grad=True)
    f = oracle.make_sparse_dominant_symeig(model.H, model.adjoint_hook, draw=SeedDraws(5100), eps=1e-12).apply
    t = torch.from_numpy(normal_vector(n, 5103))
    E_o, psi_o = f(model.g, K, n)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), backend, ret), nprocs=world, join=True)
    psi = torch.cat([torch.from_numpy(ret[r][1]) for r in range(world)])
    sgn = 1.0 if float(psi @ psi_o.detach()) > 0 else -1.0
    (g_o,) = torch.autograd.grad(E_o + sgn * psi_o.matmul(t), model.g)
    info = {r: (ret[r][0], ret[r][2], ret[r][3]) for r in range(world)}
    assert abs(ret[0][0] - E_o.item()) < 1e-10 * abs(E_o.item()), (info, E_o.item())
    assert float((psi * sgn - psi_o.detach()).abs().max()) < 1e-10, (info, float((psi * sgn - psi_o.detach()).abs().max()))
    assert abs(ret[0][2] - g_o.item()) < 1e-9 * abs(g_o.item()), (info, g_o.item())
    for r in range(world):
        assert ret[r][0] == ret[0][0] and ret[r][3] == ret[0][3], info
