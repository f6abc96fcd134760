"""fixed cost of a multi-rank test: spawn (fresh interpreter per worker) vs forkserver with torch preloaded"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))


def work(rank, world, port, q):
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.ones(1024, device=dev).sum().item()
    from dominantsparseeigenad_amd import _lib
    _lib.load()
    dist.destroy_process_group()
    q.put((rank, x))


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import multiprocessing
    import socket
    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
    for method in ("spawn", "forkserver", "forkserver", "spawn", "forkserver"):
        if method == "forkserver":
            multiprocessing.set_forkserver_preload(["torch", "torch.distributed", "numpy", "scipy.sparse"])
        for world in (2, 4):
            ctx = mp.get_context(method)
            q = ctx.SimpleQueue()
            t0 = time.perf_counter()
            pc = mp.start_processes(work, args=(world, port(), q), nprocs=world, join=False, start_method=method)
            while not pc.join(timeout=0.05):
                pass
            print("%-10s world %d: %.2f s" % (method, world, time.perf_counter() - t0), flush=True)
