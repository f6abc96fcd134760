"""config.sweep_N20 alone (bench.sweep_figures' cold leg): one untimed coupling, then `reps` timed ones, each split into the
E0.py and chiF.py callers -- for kernel traces of the second-order workload (row f-2).
    python tools/sweep_probe.py [reps]"""
import importlib.util
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "TFIM"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def load(fname, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", "TFIM", fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
E0m, chim = load("E0.py", "probe_E0"), load("chiF.py", "probe_chiF")
model = E0m.TFIM(20, dev)
torch.manual_seed(1)
for rep in range(reps + 1):
    g = [0.753, 1.005, 1.258][rep % 3]
    model.g = torch.tensor([g], dtype=torch.float64, device=dev, requires_grad=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = E0m.E0_sparseAD(model, 200)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    c = chim.chiF_sparseAD(model, 200)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("g = %.3f   E0_sparseAD %.2f ms   chiF_sparseAD %.2f ms   total %.2f ms%s" % (g, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3,
                                                                                   "   (untimed warm-up)" if rep == 0 else ""), flush=True)
