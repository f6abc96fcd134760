#!/usr/bin/env python3
"""Headline benchmark: DominantSparseSymeig forward + backward on a TFIM-shaped sparse operator.

    python bench.py --gpus 1 --steps K --warmup W         (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over synthetic inputs already resident in HBM:
    forward   symeig.DominantSparseSymeig.apply(g, k, dim, device)      (Lanczos, k vectors, full reorth)
    backward  torch.autograd.grad(E0 + psi.t, g)                         (projected CG adjoint solve + hook)
N = 1 workload: BASELINE.json configs[1] -- TFIM L=20 (n = 2^20), k = 200, fp64, g = 1.0.
N > 1 workload: the same rows per GPU (2^20), L = 20 + log2(N), vectors row-partitioned, inner products
closed by all-reduce, top-bit flips by pairwise slab exchange (weak scaling).

value = algorithmic GB/s of the whole job:  8 n (k^2 + 12k + 11m + 17) bytes / step time  (SURVEY.md 8d;
m = CG iterations actually run).  Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from ctypes import c_double, c_int64

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
SEED = 12345


def algorithmic_bytes(n, k, m):
    """forward 8n(k^2+12k-7), backward 8n(11m+24)."""
    return 8.0 * n * (k * k + 12 * k + 11 * m + 17)


def analytic_E0_per_site(L, g):
    """Closed-form ground-state energy per site of the periodic chain (sanity value printed next to the measured
    one): free fermions with the momenta of the even-parity sector, k = (2m+1) pi / L.  For even L this is the set
    used by reference examples/TFIM/E0.py:15-18; for odd L (the N = 2 and N = 8 weak-scaling points) the
    reference's linspace would pick integer momenta, which is not the ground-state sector."""
    ks = (2 * np.arange(L) + 1) * np.pi / L
    return float(-0.5 * (2 * np.sqrt(g * g - 2 * g * np.cos(ks) + 1)).sum() / L)


def reorth_bytes_per_launch(n, k):
    """average algorithmic bytes of one launch of each reorth kernel over steps i = 1..k-1:
    dots kernel: 3-term (read u,q,q' + write r = 4) + (i+1)-1 basis reads -> (i + 4) vectors... we use the
    SURVEY split: pass 1 = 4 + (i+1) vectors, pass 2 = (i+2) vectors of 8n bytes."""
    steps = k - 1
    dots = sum(4 + (i + 1) for i in range(1, k)) / steps * 8.0 * n
    axpy = sum(i + 2 for i in range(1, k)) / steps * 8.0 * n
    return dots, axpy


class PinnedRandn:
    """torch.randn replacement handing out pre-generated DEVICE vectors in call order (q0, dummy, x0, ...),
    so the timed region contains no host RNG / H2D traffic and every step solves the identical problem."""

    def __init__(self, vectors):
        self.vectors, self.i, self._orig = vectors, 0, None

    def __call__(self, *size, dtype=None, device=None, **kw):
        v = self.vectors[self.i % len(self.vectors)]
        self.i += 1
        assert v.numel() == size[0]
        return v

    def __enter__(self):
        self._orig, torch.randn, self.i = torch.randn, self, 0
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig


def cpu_baseline(L, k_sample, cg_cap):
    """The oracle (CPU port of the reference path, torch CPU ops incl. the gather-table mat-vec) on a
    bounded sample of the same workload: same L, fewer Lanczos vectors, capped CG."""
    import oracle
    from dominantsparseeigenad_amd.synthetic import normal_vector

    n = 1 << L
    t0 = time.time()
    model = oracle.TFIMTables(L)
    t_init = time.time() - t0
    model.g = torch.tensor([1.0], dtype=torch.float64, requires_grad=True)
    seeds = iter(range(SEED + 10, SEED + 20))
    draws = lambda m, dtype=torch.float64: torch.from_numpy(normal_vector(m, next(seeds))).to(dtype)  # noqa: E731
    t = torch.from_numpy(normal_vector(n, SEED + 1))
    t = t / t.norm()
    stats = []
    f = oracle.make_sparse_dominant_symeig(model.H, model.adjoint_hook, draw=draws, maxiter=cg_cap, stats=stats).apply
    t0 = time.time()
    E0, psi = f(model.g, k_sample, n)
    loss = E0 + psi.matmul(t)
    (gl,) = torch.autograd.grad(loss, model.g)
    dt = time.time() - t0
    m = stats[0]["iters"]
    gbs = algorithmic_bytes(n, k_sample, m) / dt / 1e9
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": round(gbs, 3), "unit": "GB/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": "oracle (torch-CPU port of reference Lanczos.py/CG.py/TFIM.H), TFIM L=%d, k=%d Lanczos vectors, "
                  "CG capped at %d iterations (ran %d), fwd+bwd %.1f s, table build %.1f s not timed; host: %s, "
                  "os.cpu_count()=%d, torch threads=%d"
                  % (L, k_sample, cg_cap, m, dt, t_init, cpu_model, os.cpu_count() or 0, torch.get_num_threads()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--L-local", type=int, default=20, help="log2 rows per GPU")
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-k", type=int, default=64)
    ap.add_argument("--cpu-cg-cap", type=int, default=60)
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--rpl", type=int, default=0)
    ap.add_argument("--operator", choices=["matrix-free", "sell", "csr"], default="matrix-free",
                    help="operand form of the TFIM operator at N=1: native matrix-free kernel (headline) or the "
                         "explicit 21-nnz/row matrix in SELL-64 / CSR layout")
    ap.add_argument("--force-partitioned", action="store_true",
                    help="run the row-partitioned driver even with one rank (measures its host overhead)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    assert torch.cuda.is_available(), "bench.py needs the MI355X; there is no CPU fallback for the product path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from dominantsparseeigenad_amd import _lib, engine
    from dominantsparseeigenad_amd.synthetic import normal_vector
    lib = _lib.load()

    p = int(np.log2(world))
    assert (1 << p) == world, "world size must be a power of two"
    Lloc, k = args.L_local, args.k
    L = Lloc + p
    nloc, n = 1 << Lloc, 1 << L
    off = rank * nloc

    def slab(seed, normalise=False):
        v = torch.from_numpy(normal_vector(nloc, seed, offset=off)).to(dev)
        return v

    g = torch.tensor([1.0], dtype=torch.float64, device=dev, requires_grad=True)
    draws = [slab(SEED + 10 + c) for c in range(3)]  # q0, unused second draw, CG start vector
    tvec = slab(SEED + 1)

    partitioned_path = world > 1 or args.force_partitioned
    if not partitioned_path:
        import dominantsparseeigenad_amd.symeig as symeig
        from dominantsparseeigenad_amd.operators import TFIMOperator
        tvec = tvec / tvec.norm()
        op = TFIMOperator(L, dev)
        op.g = g
        A_operand = op.H
        if args.operator != "matrix-free":
            A_operand = op.to_csr(layout=args.operator)      # explicit matrix (values fixed at the current g)
        symeig.setDominantSparseSymeig(A_operand, op.Hadjoint_to_gadjoint)
        f = symeig.DominantSparseSymeig.apply

        def step():
            with PinnedRandn(draws):
                E0, psi = f(g, k, n, dev)
                loss = E0 + psi.matmul(tvec)
                (gl,) = torch.autograd.grad(loss, g)
            return E0, gl

        def barrier():
            torch.cuda.synchronize()
    else:
        import torch.distributed as dist
        from dominantsparseeigenad_amd import partitioned
        if not dist.is_initialized():
            if world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        nrm = tvec.dot(tvec).reshape(1)
        dist.all_reduce(nrm)
        tvec = tvec / nrm.sqrt()
        solver = partitioned.PartitionedTFIM(L, g, dev)
        last = {}

        def step():
            E0, psi, gl = solver.forward_backward(k, draws[0], draws[2], tvec)
            last["psi"] = psi
            return E0, gl

        def eigen_residual(E0, psi):
            """||H psi - E0 psi|| over all ranks: the self-check of the distributed run (outside the timed region)"""
            w = torch.empty_like(psi)
            solver.matvec(psi, w)
            res = w - E0 * psi
            nrm = res.dot(res).reshape(1)
            dist.all_reduce(nrm)
            return float(nrm.sqrt())

        def barrier():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    ws = engine.Workspace.get(nloc, k, dev)
    if args.rpl:
        ws.set_rows_per_lane(args.rpl)

    fallback_note = None
    if partitioned_path and solver.transposed:
        # first contact with the collectives of this stack: if the transposed all-to-all form raises, every rank
        # raises alike, and all switch to the pairwise slab exchange
        try:
            E0, gl = step()
        except Exception as exc:  # noqa: BLE001
            fallback_note = "transposed exchange unavailable (%s: %s): pairwise slab exchange used" % (
                type(exc).__name__, str(exc)[:120])
            solver.use_pairwise_exchange()
    for _ in range(args.warmup):
        E0, gl = step()
    barrier()
    overlap_note = None
    if partitioned_path:
        # the overlapped exchange is verified before anything is timed; if the eigen-residual is not at the
        # level Lanczos with k vectors reaches on one GPU, the run falls back to the sequential exchange
        resid = eigen_residual(E0, last["psi"])
        if solver.transposed and solver.overlap:
            solver.overlap = False
            E0s, _ = step()
            resid_seq = eigen_residual(E0s, last["psi"])
            solver.overlap = True
            if not (resid <= 10.0 * resid_seq + 1e-9):
                solver.overlap = False
                overlap_note = "overlapped exchange failed its self-check (residual %.2e vs %.2e sequential): " \
                               "sequential exchange timed instead" % (resid, resid_seq)
            else:
                overlap_note = "overlapped exchange verified: eigen-residual %.2e (sequential %.2e)" % (resid, resid_seq)
        else:
            overlap_note = "eigen-residual %.2e" % resid
        barrier()
    # ---- timed region: exactly K steps, no instrumentation inside
    t0 = time.perf_counter()
    for _ in range(args.steps):
        E0, gl = step()
    barrier()
    dt = time.perf_counter() - t0
    # ---- per-launch durations of the dominant kernels: the same K steps again, this time with a HIP event
    # pair recorded on the launch stream around every reorth / mat-vec launch (the event records cost ~4 %
    # of a step, which is why they are kept out of the timed region above)
    use_events = not args.no_kernel_events   # rank 0 reports its local kernels also in the partitioned run
    launches = (c_int64 * 3)()
    total_ms = (c_double * 3)()
    dt_instr = None
    if use_events:
        _lib.check(lib.dsea_profile_begin(ws.handle, 3 * k * args.steps + 8), "dsea_profile_begin")
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_instr = time.perf_counter() - t1
        _lib.check(lib.dsea_profile_end(ws.handle, launches, total_ms), "dsea_profile_end")
    m = engine.last_cg.iters
    if partitioned_path:
        import torch.distributed as dist
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
        m = solver.last_cg_iters
    ms_per_step = dt / args.steps * 1e3
    total_bytes = algorithmic_bytes(n, k, m)
    value = total_bytes / (ms_per_step * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "DominantSparseSymeig fwd+bwd algorithmic HBM GB/s (TFIM, fp64)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "TFIM L=%d (n=2^%d, %d rows/GPU) DominantSparseSymeig k=%d fwd+bwd, g=1.0, "
                                   "loss=E0+psi.t, operand=%s" % (L, L, nloc, k, args.operator),
                       "cg_iterations": int(m), "algorithmic_bytes_per_step": total_bytes,
                       "frac_of_hbm_peak_whole_step": round(value / (HBM_PEAK_GBS * world), 4),
                       "E0_per_site": E0.item() / L, "E0_per_site_closed_form": analytic_E0_per_site(L, 1.0),
                       "dloss_dg": float(gl.reshape(-1)[0].item())},
        }
        if overlap_note:
            out["config"]["distributed_self_check"] = overlap_note
        if fallback_note:
            out["config"]["distributed_fallback"] = fallback_note
        if use_events and launches[0] > 0 and launches[1] > 0:
            dots_b, axpy_b = reorth_bytes_per_launch(nloc, k)
            per = {
                "k_rdots": (dots_b, total_ms[0] / launches[0], launches[0]),
                "k_axpy_norm": (axpy_b, total_ms[1] / launches[1], launches[1]),
            }
            name = max(per, key=lambda kk: per[kk][1] * per[kk][2])
            b, ms, cnt = per[name]
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            lp, fb = engine.lanczos_lp_stats(nloc, dev) if not partitioned_path else (launches[1], 0)
            out["roofline"] = {
                "kernel": name, "bound": "hbm", "achieved": round(b / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(ms, 5), "launches": int(cnt), "algorithmic_bytes_per_launch": b,
                "other": {kk: {"avg_launch_ms": round(v[1], 5), "achieved_GBs": round(v[0] / (v[1] * 1e-3) / 1e9, 1)}
                          for kk, v in per.items() if kk != name},
                "spmv_avg_launch_ms": round(total_ms[2] / max(launches[2], 1), 5),
                "measured": "HIP events on the launch stream, %d instrumented steps run right after the timed "
                            "region (%.3f ms/step with events)" % (args.steps, dt_instr / args.steps * 1e3),
                "note": ("k_axpy_norm streams the bf16 shadow of the basis on %d of %d steps (fp64 fallback %d): "
                         "its real traffic is ~1/4 of its algorithmic bytes" % (lp, lp + fb, fb)) if not partitioned_path
                        else "rank 0's local kernels in the row-partitioned run (correction pass on the bf16 shadow)",
            }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(L, args.cpu_k, args.cpu_cg_cap)
        # RCCL / HIP runtime banners go through C stdio: flush them first so the JSON is the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        final_line = json.dumps(out)
    else:
        final_line = None
    if partitioned_path:
        import torch.distributed as dist
        dist.destroy_process_group()
    if final_line is not None:
        print(final_line, flush=True)


if __name__ == "__main__":
    main()
