#!/usr/bin/env python3
"""Headline benchmark: DominantSparseSymeig forward + backward on a TFIM-shaped sparse operator.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]

One "step" = one pass of the hot path over synthetic inputs already resident in HBM, through the reference API:
    forward   symeig.DominantSparseSymeig.apply(g, k, dim, device)      (Lanczos, k vectors, full reorth)
    backward  torch.autograd.grad(E0 + psi.t, g)                         (projected CG adjoint solve + hook)

Workloads (what the K timed steps run, i.e. what `value` / `ms_per_step` are quoted on)
    N = 1 (default)   BASELINE.json configs[1]: TFIM L=20 (n = 2^20), k = 200, fp64, g = 1.0.  The same run then times,
                      OUTSIDE the K steps, the two one-GPU anchors of the multi-GPU curves (config.one_gpu_anchors):
                      L = 28, k = 100 (strong) and 2^25 rows, k = 200 (weak).
    N > 1 (default)   STRONG scaling, north_star's curve: TFIM L=28 (n = 2^28) row-partitioned over the N GPUs with
                      k = 100 at every N (the largest k whose basis, 215 GB, fits the one 288 GB GPU of the anchor).
                      After the timed steps the WEAK point (2^25 rows per GPU, k = 200; N = 8 is BASELINE configs[4]:
                      L = 28 with 53.7 GB of basis + 13.4 GB bf16 shadow per GPU) is timed too and reported as
                      config.weak_scaling_point.
    --scaling weak|strong   time only that one point.
    --gpus N without a launcher starts its own N worker processes (python -m torch.distributed.run) BEFORE any GPU
    call and relays rank 0's JSON line; under torchrun (WORLD_SIZE set) it is a worker.

value = HBM GB/s of the whole job: bytes the step's kernels move (traffic model below; at the headline size checked
against rocprofv3 PMC counters, profiles/pmc_traffic.json) / step time.  It is bounded by N x 8 TB/s.  The figure of
SURVEY.md 8d -- ALGORITHMIC bytes 8 n (k^2 + 12k + 11m + 17) of the reference's algorithm / step time (m = CG
iterations actually run) -- is config.algorithmic_GBs; it may exceed the peak because the correction pass streams a
bf16 storage shadow of the basis.  Prints ONE JSON line (rank 0) as the last line of stdout.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from ctypes import c_double, c_int64

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
SEED = 12345
# GPU-vs-reference deviation of the adjoint at the reference's hard-coded CG tolerance (eps = 1e-7 absolute,
# CG.py:25), observed over the golden cases of tests/test_gpu_parity.py (asserted there at 2e-8): see DESIGN.md 5
ADJOINT_DEV_EPS7 = "3e-10..1.6e-9 relative (reference's own seed-to-seed spread: 5e-10)"


def algorithmic_bytes(n, k, m):
    """forward 8n(k^2+12k-7), backward 8n(11m+24)."""
    return 8.0 * n * (k * k + 12 * k + 11 * m + 17)


def traffic_model_bytes(n, k, m, shadow_steps, cg_vectors_per_iteration=11.0):
    """HBM bytes one step of THIS implementation moves, as a model: SURVEY 8d's per-phase count with the correction
    pass of ``shadow_steps`` Lanczos steps reading the bf16 shadow (2 instead of 8 bytes per basis element: step i
    saves 6 n i bytes) plus the k shadow rows written once (2 n each).  At L = 20, k = 200, m = 90: 239.3 GB against
    249.9 GB counted by the PMC (the difference is the TFIM mat-vec's cross-XCD re-reads).
    ``cg_vectors_per_iteration``: 11 for the streaming CG; the persistent single-launch CG keeps x, r, d in
    registers and moves only d (one write + the out-of-tile reads, 4 vectors' worth through the fabric): 5."""
    saved = 6.0 * n * sum(range(1, int(shadow_steps) + 1))
    saved += 8.0 * n * m * (11.0 - float(cg_vectors_per_iteration))
    return algorithmic_bytes(n, k, m) - saved + (2.0 * n * k if shadow_steps else 0.0)


def analytic_E0_per_site(L, g):
    """Closed-form ground-state energy per site of the periodic chain (sanity value printed next to the measured
    one): free fermions with the momenta of the even-parity sector, k = (2m+1) pi / L.  For even L this is the set
    used by reference examples/TFIM/E0.py:15-18; for odd L the reference's linspace would pick integer momenta,
    which is not the ground-state sector."""
    ks = (2 * np.arange(L) + 1) * np.pi / L
    return float(-0.5 * (2 * np.sqrt(g * g - 2 * g * np.cos(ks) + 1)).sum() / L)


def reorth_bytes_per_launch(n, k):
    """average algorithmic bytes of one launch of each reorth kernel over steps i = 1..k-1 (SURVEY 8d split):
    pass 1 = 4 + (i+1) vectors, pass 2 = (i+2) vectors of 8n bytes."""
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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(L, k, cg_cap, threads, model=None):
    """The oracle (CPU port of the reference path, torch CPU ops incl. the gather-table mat-vec) on the same
    workload; ``cg_cap`` None = the reference's cap (n iterations, i.e. runs to ||r|| < 1e-7)."""
    import oracle
    from dominantsparseeigenad_amd.synthetic import normal_vector

    n = 1 << L
    torch.set_num_threads(int(threads))
    t_init = 0.0
    if model is None:
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
    E0, psi = f(model.g, k, n)
    t_fwd = time.time() - t0
    loss = E0 + psi.matmul(t)
    (gl,) = torch.autograd.grad(loss, model.g)
    dt = time.time() - t0
    m = stats[0]["iters"]
    return model, {"threads": int(threads), "k": k, "cg_iterations": m, "fwd_s": round(t_fwd, 2),
                   "bwd_s": round(dt - t_fwd, 2), "fwd_bwd_s": round(dt, 2), "table_build_s": round(t_init, 2),
                   "GBs": round(algorithmic_bytes(n, k, m) / dt / 1e9, 3), "E0": E0.item(), "dloss_dg": gl.item()}


def measured_ceilings(dev):
    """the box's own streaming ceilings next to the 8 TB/s spec figure, measured with the library's probe
    kernels (dsea_probe_stream: 8 non-temporal 16-byte loads in flight per lane, nothing else to do) on 1 GiB buffers
    (beyond the 256 MiB Infinity Cache): a read-only pass and a copy (bytes read + written); torch's own copy kernel
    beside them.  Best of 5 after 2 warm-ups, HIP events on the current stream."""
    from dominantsparseeigenad_amd import engine, _lib
    lib = _lib.load()
    nel = 1 << 27                                   # 1 GiB of doubles
    a = torch.ones(nel, dtype=torch.float64, device=dev)
    b = torch.empty(nel, dtype=torch.float64, device=dev)
    ws = engine.Workspace.get(4096, 8, dev)         # the probe only borrows its partial-sum buffer
    st = engine._stream(dev)

    def best(fn):
        for _ in range(2):
            fn()
        t = 1e30
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            t = min(t, e0.elapsed_time(e1) * 1e-3)
        return t
    t_read = best(lambda: engine.check(lib.dsea_probe_stream(ws.handle, engine._ptr(a), None, nel, st), "dsea_probe_stream"))
    t_copy = best(lambda: engine.check(lib.dsea_probe_stream(ws.handle, engine._ptr(a), engine._ptr(b), nel, st), "dsea_probe_stream"))
    t_torch = best(lambda: b.copy_(a))
    res = {"read_GBs": round(8.0 * nel / t_read / 1e9, 1), "copy_GBs": round(2 * 8.0 * nel / t_copy / 1e9, 1),
           "torch_copy_GBs": round(2 * 8.0 * nel / t_torch / 1e9, 1),
           "what": "1 GiB buffers: read-only reduction / copy (bytes read + written) by dsea_probe_stream, torch's "
                   "copy_ beside them; best of 5, HIP events"}
    del a, b
    return res


def c3_figures(dev):
    """BASELINE configs[2] in its well-posed restatement: 3-point stencil N = 100000
    (schrodinger1D.py:11-27 semantics), Lanczos k = 300 forward and CG over a FIXED 1000 iterations."""
    from dominantsparseeigenad_amd import engine
    from dominantsparseeigenad_amd.Lanczos import symeigLanczos
    from dominantsparseeigenad_amd.operators import Stencil3Operator
    from dominantsparseeigenad_amd.synthetic import normal_vector
    N, k, iters = 100000, 300, 1000
    x = torch.from_numpy(np.linspace(-1.0, 1.0, num=N, endpoint=False)).to(dev)
    op = Stencil3Operator(N, 2.0 / N, 0.5 * x ** 2)
    q0 = torch.from_numpy(normal_vector(N, 1)).to(dev)
    best_l = best_c = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        symeigLanczos(op, k, dev, extreme="min", sparse=True, dim=N, q0=q0)
        torch.cuda.synchronize()
        best_l = min(best_l, time.perf_counter() - t0)
    b = torch.from_numpy(normal_vector(N, 2)).to(dev)
    x0 = torch.from_numpy(normal_vector(N, 3)).to(dev)
    shift = torch.tensor(-1.0, dtype=torch.float64, device=dev)
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.cg(b, x0, native=op, shift=shift, eps=0.0, maxiter=iters, poll_every=iters)
        torch.cuda.synchronize()
        best_c = min(best_c, time.perf_counter() - t0)
    ran = engine.last_cg.iters
    best_m = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.cg(b, x0, native=op, shift=shift, eps=0.0, maxiter=iters, poll_every=iters, merged_reductions=True)
        torch.cuda.synchronize()
        best_m = min(best_m, time.perf_counter() - t0)
    return {"workload": "3-point stencil N=100000 (schrodinger1D.py:18-27), Lanczos k=300 forward; CG on (A+1)x=b over "
                        "a fixed %d iterations" % ran,
            "lanczos_k300_ms": round(best_l * 1e3, 3), "lanczos_us_per_step": round(best_l / k * 1e6, 2),
            "lanczos_algorithmic_GBs": round(8.0 * N * (k * k + 12 * k) / best_l / 1e9, 1),
            "cg_us_per_iteration": round(best_c / ran * 1e6, 3),
            "cg_algorithmic_GBs": round(11 * 8.0 * N * ran / best_c / 1e9, 1), "cg_iterations": ran,
            "cg_us_per_iteration_merged_reductions": round(best_m / ran * 1e6, 3),
            "cg_note": "cg_us_per_iteration: persistent single-launch CG with the reference's recurrences (iterates "
                       "bit-identical to CG.py:31-40 evaluated in fp64 on the device, two grid-wide exchanges per "
                       "iteration); merged_reductions: the optional one-exchange form (Chronopoulos-Gear; same iteration "
                       "in exact arithmetic, 6e-14 relative deviation after 40 iterations), off by default"}


def live_pmc_traffic(timeout_s=240):
    """HBM bytes per launch of every kernel of ONE step of the headline workload, from the PMC counters, measured NOW on
    this box: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only, as
    MI355X_MICROARCH.md prescribes) over `bench.py --steps 1 --warmup 1` as CHILD processes.  Must be called before
    this process touches the GPU (a process that has initialised the GPU must not start other programs on this pool).
    Units: KiB -> x 1024; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> x 2.
    Returns (dict in the format of profiles/pmc_traffic.json, None) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    acc = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        outdir = tempfile.mkdtemp(prefix="dsea_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
               "--no-extras", "--no-anchors", "--no-kernel-events", "--no-live-pmc"]
        env = dict(os.environ, TMPDIR="/tmp", DSEA_BENCH_CHILD="1")
        try:
            proc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                  text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            shutil.rmtree(outdir, ignore_errors=True)
            return None, "rocprofv3 --pmc %s pass timed out after %d s" % (counter, timeout_s)
        except OSError as exc:
            shutil.rmtree(outdir, ignore_errors=True)
            return None, "rocprofv3 could not be started: %s" % exc
        files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
        if proc.returncode != 0 or not files:
            shutil.rmtree(outdir, ignore_errors=True)
            return None, "rocprofv3 --pmc %s pass failed (rc %s): %s" % (counter, proc.returncode, proc.stdout[-200:].replace("\n", " "))
        per = collections.defaultdict(lambda: [0, 0.0])
        with open(files[0]) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                short = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dsea::", "").split("<")[0]
                per[short][0] += 1
                per[short][1] += float(row["Counter_Value"])
        acc[counter] = per
        shutil.rmtree(outdir, ignore_errors=True)
    out = {"_method": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over one warm-up + one timed "
                      "step of this script on this box, child processes started before the timed run; bytes = "
                      "FETCH_SIZE[KiB]*1024*2 (gfx950 wide-read correction) + WRITE_SIZE[KiB]*1024"}
    steps = 2
    for name in sorted(acc["FETCH_SIZE"]):
        if not name.startswith("k_"):
            continue
        nf, fs = acc["FETCH_SIZE"][name]
        nw, wsz = acc["WRITE_SIZE"].get(name, [0, 0.0])
        rd = fs / nf * 1024 * 2
        wr = (wsz / nw * 1024) if nw else 0.0
        out[name] = {"launches": nf, "fetch_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                     "hbm_bytes_per_launch": rd + wr}
    out["_steps_profiled"] = steps
    out["_total_hbm_bytes_per_step"] = sum(v["launches"] * v["hbm_bytes_per_launch"] for kk, v in out.items()
                                           if kk.startswith("k_")) / steps
    out["_commit"] = "live"
    return out, None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_workers(n):
    """--gpus N without a launcher: start N fresh worker processes (one per GPU) BEFORE this process has touched
    the GPU, relay their output and finish with rank 0's JSON as the last line of stdout."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = proc.stdout.splitlines()
    final = None
    for line in lines:
        if line.startswith("{") and '"metric"' in line:
            final = line
        else:
            print(line)
    sys.stdout.flush()
    if final is not None:
        print(final, flush=True)
    raise SystemExit(proc.returncode if proc.returncode else (0 if final is not None else 1))


# one-GPU anchors of the two multi-GPU curves as last measured on an MI355X by `bench.py --gpus 1` of this repository
# (profiles/, with the commit of the run); the N = 1 line re-measures them live (config.one_gpu_anchors)
STORED_ANCHORS = {
    "strong_L28_k100_ms": 5128.10, "strong_source": "profiles/r03_bench.json config.one_gpu_anchors.strong_L28_k100 (commit 7c75263; other boxes of the round: 5086 ... 5317)",
    "weak_2p25_rows_k200_ms": 1250.42, "weak_source": "profiles/r03_bench.json config.one_gpu_anchors.weak_2p25_rows_k200 (commit 7c75263; other boxes of the round: 1246 ... 1318)",
}


class Problem:
    """One TFIM workload behind the reference API: operator, pinned draws, step() = forward + backward."""

    def __init__(self, args, L, k, world, rank, dev, dry, partitioned_path, reorth="full"):
        from dominantsparseeigenad_amd import engine
        import dominantsparseeigenad_amd.symeig as symeig
        from dominantsparseeigenad_amd.synthetic import normal_vector
        self.engine, self.symeig = engine, symeig
        self.L, self.k, self.world, self.rank, self.dev, self.dry = L, k, world, rank, dev, dry
        self.partitioned = partitioned_path
        self.reorth = reorth
        self.notes = {}
        p = int(np.log2(world))
        self.p, self.Lloc = p, L - p
        self.nloc, self.n = 1 << (L - p), 1 << L
        off = rank * self.nloc
        # one GPU cannot hold the fp64 basis AND its bf16 shadow at L = 28, k = 100 (215 + 54 GB of 288 GB)
        free_b, total_b = (0, 1 << 62) if dry else torch.cuda.mem_get_info(dev)
        need_shadow = 10.0 * self.nloc * k + 16 * 8.0 * self.nloc
        self.use_shadow = not (need_shadow > 0.92 * total_b or reorth in ("none", "partial"))

        def slab(seed):
            return torch.from_numpy(normal_vector(self.nloc, seed, offset=off)).to(dev)

        self.g = torch.tensor([1.0], dtype=torch.float64, device=dev, requires_grad=True)
        self.draws = [slab(SEED + 10 + c) for c in range(3)]  # q0, unused second draw, CG start vector
        tvec = slab(SEED + 1)
        if not partitioned_path:
            from dominantsparseeigenad_amd.operators import TFIMOperator
            self.tvec = tvec / tvec.norm()
            self.op = TFIMOperator(L, dev)
            self.op.g = self.g
            self.A_operand = self.op.H
            if args.operator != "matrix-free":
                self.A_operand = self.op.to_csr(layout=args.operator)      # explicit matrix (values fixed at the current g)
            self.dot = torch.matmul
        else:
            from dominantsparseeigenad_amd import partitioned
            backend = None
            if dry:   # the torch test double of the slab kernels (test infrastructure; never on the product path)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from cpu_backend import CpuBackend
                backend = CpuBackend(self.nloc)
            # The library-side driver (RCCL calls issued by libdsea) needs its communicators; if creating them fails on
            # ANY rank the decision to use the Python driver instead is taken collectively (an all-reduced flag)
            import torch.distributed as dist
            failed = torch.zeros(1, dtype=torch.float64, device=dev)
            try:
                self.op = partitioned.PartitionedTFIMOperator(L, self.g, dev, backend=backend, overlap=True if dry else "auto")
            except Exception as exc:  # noqa: BLE001
                failed[0] = 1.0
                self.notes["partitioned_driver_fallback_reason"] = "%s: %s" % (type(exc).__name__, str(exc)[:160])
            dist.all_reduce(failed)
            if failed.item() > 0:
                os.environ["DSEA_DRIVER"] = "python"
                self.op = partitioned.PartitionedTFIMOperator(L, self.g, dev, backend=backend, overlap=True if dry else "auto")
                self.notes["partitioned_driver_fallback"] = "library driver unavailable on %d rank(s): Python driver used" \
                                                            % int(failed.item())
            self.op.force_driver = True
            self.A_operand = self.op.H
            self.dot = self.op.dot
            self.tvec = tvec / self.op.dot(tvec, tvec).sqrt()
        self.last = {}

    def barrier(self):
        if not self.dry:
            torch.cuda.synchronize()
        if self.partitioned:
            import torch.distributed as dist
            dist.barrier()
            if not self.dry:
                torch.cuda.synchronize()

    def activate(self):
        """(re-)bind the module-global primitive to this problem's operator (reference symeig.py:66,87: last set wins)"""
        from dominantsparseeigenad_amd import Lanczos as _LZ
        _LZ.REORTH_DEFAULT = self.reorth
        self.engine.USE_SHADOW = self.use_shadow
        self.symeig.setDominantSparseSymeig(self.A_operand, self.op.Hadjoint_to_gadjoint)
        self.f = self.symeig.DominantSparseSymeig.apply

    def step(self):
        with PinnedRandn(self.draws):
            E0, psi = self.f(self.g, self.k, self.n, self.dev)
            loss = E0 + self.dot(psi, self.tvec)
            (gl,) = torch.autograd.grad(loss, self.g)
        self.last["psi"] = psi.detach()
        return E0, gl

    def eigen_residual(self, E0, psi):
        """||H psi - E0 psi|| over all ranks: the self-check of the distributed run (outside the timed region)"""
        res = self.op.H(psi) - E0.detach() * psi
        return float(self.dot(res, res).sqrt())

    def cg_iterations(self):
        return self.op.last_cg_iters if self.partitioned else self.engine.last_cg.iters

    def first_contact(self):
        """first contact with the collectives of this stack.  The decision to leave the transposed all-to-all form
        is COLLECTIVE (an all-reduced failure flag): a rank-local fallback would leave the others in a collective"""
        if not (self.partitioned and self.op.transposed):
            return
        import torch.distributed as dist
        failed = torch.zeros(1, dtype=torch.float64, device=self.dev)
        try:
            probe = torch.zeros(self.world * 8, dtype=torch.float64, device=self.dev)
            self.op.comm.all_to_all(probe, torch.empty_like(probe))
            if not self.dry:
                torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            failed[0] = 1.0
            self.notes["distributed_fallback_reason"] = "%s: %s" % (type(exc).__name__, str(exc)[:120])
        dist.all_reduce(failed)
        if failed.item() > 0:
            self.op.use_pairwise_exchange()
            self.notes["distributed_fallback"] = "transposed exchange unavailable on %d rank(s): pairwise slab " \
                                                 "exchange used" % int(failed.item())

    def measure(self, steps, warmup):
        """W untimed steps, [distributed self-check], barrier, EXACTLY K timed steps, barrier; returns seconds
        (max over ranks), E0, dloss/dg"""
        self.activate()
        self.first_contact()
        E0 = gl = None
        for _ in range(warmup):
            E0, gl = self.step()
        self.barrier()
        if self.partitioned:
            op, notes = self.op, self.notes
            # the overlapped exchange is verified before anything is timed; if the eigen-residual is not at the
            # level the sequential exchange reaches, the run falls back to the sequential exchange
            if warmup == 0:
                E0, gl = self.step()
            resid = self.eigen_residual(E0, self.last["psi"])
            notes["partitioned_driver"] = getattr(op, "driver", "python")
            notes["slab_exchange"] = ("none (one rank)" if op.p == 0 else
                                      ("transposed all-to-all form" if op.transposed else "pairwise hypercube partners")
                                      + (", overlapped with the dots / correction passes" if op.overlap else ""))
            if op.p > 0 and op.overlap:
                op.overlap = False
                E0s, _ = self.step()
                resid_seq = self.eigen_residual(E0s, self.last["psi"])
                op.overlap = True
                if not (resid <= 10.0 * resid_seq + 1e-9):
                    op.overlap = False
                    notes["distributed_self_check"] = "overlapped exchange failed its self-check (residual %.2e vs " \
                                                      "%.2e sequential): sequential exchange timed instead" % (resid, resid_seq)
                else:
                    notes["distributed_self_check"] = "overlapped exchange verified: eigen-residual %.2e (sequential " \
                                                      "%.2e), %d premise fallbacks" % (resid, resid_seq, op.overlap_fallbacks)
            else:
                notes["distributed_self_check"] = "eigen-residual %.2e" % resid
            self.barrier()
        # ---- timed region: exactly K steps, no instrumentation inside
        t0 = time.perf_counter()
        for _ in range(steps):
            E0, gl = self.step()
        self.barrier()
        dt = time.perf_counter() - t0
        if self.partitioned:
            import torch.distributed as dist
            tmax = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = tmax.item()
        return dt, E0, gl

    def describe(self, operator="matrix-free", scaling=None):
        mode = "row-partitioned over %d GPUs%s" % (self.world, ", %s scaling" % scaling if scaling else "") \
            if self.partitioned else "one GPU"
        return "TFIM L=%d (n=2^%d, %d rows/GPU) DominantSparseSymeig k=%d fwd+bwd, g=1.0, loss=E0+psi.t, " \
               "operand=%s, %s" % (self.L, self.L, self.nloc, self.k, operator, mode)

    def release(self):
        """drop the operator and the arena basis (the next problem of this process may need the memory)"""
        self.op = self.A_operand = self.f = None
        self.draws = self.tvec = None
        self.last = {}
        self.engine.BasisArena.release()
        self.engine.Workspace.clear_cache()
        if not self.dry:
            torch.cuda.empty_cache()


def rank_evidence(world, rank, local_rank, dev, dry):
    """what proves the collectives span N distinct GPUs: every rank's device, gathered to rank 0"""
    import torch.distributed as dist
    mine = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid()}
    if not dry:
        pr = torch.cuda.get_device_properties(dev)
        mine.update(device=pr.name, pci="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                                             getattr(pr, "pci_device_id", 0)),
                    uuid=str(getattr(pr, "uuid", "")), hbm_GB=round(pr.total_memory / 1e9, 1))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": gathered}
    if not dry:
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            pass
        info["distinct_devices"] = len({(r["host"], r.get("pci"), r.get("uuid")) for r in gathered})
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: time only this point (default: strong timed, weak reported beside it)")
    ap.add_argument("--L", type=int, default=None, help="chain length (default: by --gpus / --scaling)")
    ap.add_argument("--L-local", type=int, default=None, help="log2 rows per GPU (weak scaling)")
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", action="store_true",
                    help="CPU baseline on a bounded sample (k = --cpu-k, CG capped at --cpu-cg-cap) instead of the FULL "
                         "configuration of SURVEY 8d (k as on the GPU, CG to the reference's tolerance: ~30 s of host "
                         "time at L = 20, k = 200 with 8 threads)")
    ap.add_argument("--cpu-k", type=int, default=64)
    ap.add_argument("--cpu-cg-cap", type=int, default=60)
    ap.add_argument("--cpu-threads", type=str, default="8",
                    help="comma-separated torch thread counts for the CPU baseline; the best run is reported.  Default "
                         "8: on the GPU box's 2 x 64-core host the reference's torch-CPU gather mat-vec runs the full "
                         "configuration in 31.5 s with 8 threads, 33.8 s with 64 and 554 s with all 256")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp64-basis batch and the config-3 figures")
    ap.add_argument("--no-anchors", action="store_true",
                    help="N = 1: skip the live one-GPU anchors of the multi-GPU curves (L = 28, k = 100 and 2^25 rows, "
                         "k = 200: ~40 s and 230 GB of HBM)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1 headline run: do not measure roofline.traffic / pmc_* live with two rocprofv3 --pmc child "
                         "passes (~20 s each) before the timed run; the committed profiles/pmc_traffic.json is quoted instead")
    ap.add_argument("--rpl", type=int, default=0)
    ap.add_argument("--operator", choices=["matrix-free", "sell", "csr"], default="matrix-free",
                    help="operand form of the TFIM operator at N=1: native matrix-free kernel (headline) or the "
                         "explicit 21-nnz/row matrix in SELL-64 / CSR layout")
    ap.add_argument("--reorth", choices=["full", "none", "partial"], default="full",
                    help="'none': basis-free two-pass Lanczos (no stored basis, no re-orthogonalisation) -- NOT the "
                         "reference's algorithm, reported for what it is; lets k = 200 at L = 28 fit one GPU.  'partial': "
                         "stored basis, re-orthogonalised only on the steps the omega recurrence selects (Simon) -- NOT the "
                         "reference's schedule either, priced with its own bytes")
    ap.add_argument("--force-partitioned", action="store_true",
                    help="run the row-partitioned driver even with one rank (measures its host overhead)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="CONTROL-FLOW CHECK ONLY, no measurement: the multi-rank path of this script (self-launch, "
                         "row-partitioned operator behind the reference API, collective fallback decision, exchange "
                         "self-check, max-over-ranks timing, rank-0 JSON) on CPU processes over gloo with the torch test "
                         "double of the slab kernels (tests/cpu_backend.py).  Without --L / --k the default two-point "
                         "schedule of N > 1 runs at toy sizes.  The line it prints is labelled as a dry run and carries "
                         "no roofline / cpu_baseline.")
    args = ap.parse_args()
    dry = args.dry_run_cpu

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        launch_workers(args.gpus)          # does not return
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    live_pmc, live_pmc_note = None, None
    headline_defaults = (world == 1 and not dry and not args.force_partitioned and args.L is None and args.L_local is None
                         and args.k is None and args.operator == "matrix-free" and args.reorth == "full")
    # never from under a profiler: its preloaded library may already have initialised the GPU in THIS process, and a
    # process that has done so must not start other programs on this pool
    under_profiler = any(kk.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "RPD_")) for kk in os.environ) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or "roctracer" in os.environ.get("LD_PRELOAD", "").lower()
    if under_profiler and headline_defaults and not args.no_live_pmc:
        live_pmc_note = "skipped: running under a profiler (committed profiles/pmc_traffic.json quoted)"
    if headline_defaults and not args.no_live_pmc and not under_profiler and not torch.cuda.is_initialized() and \
            os.environ.get("DSEA_BENCH_CHILD", "") != "1":
        # child processes, BEFORE this process initialises the GPU
        t_pmc = time.time()
        live_pmc, live_pmc_note = live_pmc_traffic()
        live_pmc_note = live_pmc_note or "two rocprofv3 --pmc passes took %.0f s" % (time.time() - t_pmc)
    if dry:
        dev = torch.device("cpu")
        torch.set_num_threads(1)
    else:
        assert torch.cuda.is_available(), "bench.py needs the MI355X; there is no CPU fallback for the product path"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    from dominantsparseeigenad_amd import _lib, engine
    lib = None if dry else _lib.load()

    p = int(np.log2(world))
    assert (1 << p) == world, "world size must be a power of two"
    partitioned_path = world > 1 or args.force_partitioned or dry
    # ---- which point is timed
    explicit = args.L is not None or args.L_local is not None
    if world == 1:
        scaling = args.scaling or "weak"
    else:
        scaling = args.scaling or ("weak" if explicit else "strong")
    strong = scaling == "strong"
    toy = dry and not explicit            # dry run of the default schedule: toy sizes
    if args.L is not None:
        L = args.L
    elif strong:
        L = 10 if toy else 28
    elif args.L_local is not None:
        L = args.L_local + p
    elif toy:
        L = 7 + p
    else:
        L = 20 if world == 1 else 25 + p
    k = args.k if args.k is not None else ((80 if strong else 60) if toy else (100 if strong else 200))
    # the weak point reported beside the timed strong one (default schedule of N > 1 only)
    weak_extra = world > 1 and args.scaling is None and not explicit and args.reorth == "full"
    nloc, n = 1 << (L - p), 1 << L
    big = nloc >= (1 << 24)
    steps = args.steps if args.steps is not None else (3 if big else 10)
    warmup = args.warmup if args.warmup is not None else (1 if big else 2)

    evidence = None
    if partitioned_path:
        import torch.distributed as dist
        if not dist.is_initialized():
            if world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(_free_port()))
            if dry:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        evidence = rank_evidence(world, rank, local_rank, dev, dry)

    if os.environ.get("DSEA_PLACEMENT_TRIES"):
        engine.BasisArena.PLACEMENT_TRIES = int(os.environ["DSEA_PLACEMENT_TRIES"])

    prob = Problem(args, L, k, world, rank, dev, dry, partitioned_path, reorth=args.reorth)
    ws = None if dry else engine.Workspace.get(nloc, k, dev)
    if args.rpl and ws is not None:
        ws.set_rows_per_lane(args.rpl)
    dt, E0, gl = prob.measure(steps, warmup)
    notes = prob.notes
    m = prob.cg_iterations()
    step, barrier = prob.step, prob.barrier
    # ---- per-launch durations of the dominant kernels: the same K steps again, this time with a HIP event
    # pair recorded on the launch stream around every reorth / mat-vec launch (the event records cost ~4 %
    # of a step, which is why they are kept out of the timed region above)
    use_events = not args.no_kernel_events and args.reorth == "full" and not dry   # rank 0's local kernels, also when partitioned
    launches = (c_int64 * 3)()
    total_ms = (c_double * 3)()
    dt_instr = None
    ev_steps = min(steps, 5) if big else steps
    if use_events:
        ws = engine.Workspace.get(nloc, k, dev)
        _lib.check(lib.dsea_profile_begin(ws.handle, 3 * k * ev_steps + 8), "dsea_profile_begin")
        t1 = time.perf_counter()
        for _ in range(ev_steps):
            step()
        barrier()
        dt_instr = time.perf_counter() - t1
        _lib.check(lib.dsea_profile_end(ws.handle, launches, total_ms), "dsea_profile_end")
    lp_stats = engine.lanczos_lp_stats(nloc, dev) if not partitioned_path else None
    pr_timed_steps = engine.last_reorth_steps       # (partial option: steps the timed runs re-orthogonalised)
    # ---- the same step with the all-fp64 correction pass (no bf16 shadow of the basis): the figure to hold
    # against real HBM traffic
    ms_fp64 = ms_basisfree = ms_partial = None
    from dominantsparseeigenad_amd import Lanczos as _LZ
    if not args.no_extras and not partitioned_path and args.reorth == "full" and not big:
        # the same workload with the basis-free two-pass Lanczos (reorth='none'): a DIFFERENT forward algorithm (no
        # full re-orthogonalisation, Lanczos.py:66), same eigenpair and gradient to rounding -- reported beside the
        # headline, never as the headline
        _LZ.REORTH_DEFAULT = "none"
        step()
        barrier()
        t3 = time.perf_counter()
        nb2 = min(steps, 5)
        for _ in range(nb2):
            E0bf, glbf = step()
        barrier()
        ms_basisfree = (time.perf_counter() - t3) / nb2 * 1e3
        bf_dev = (abs(E0bf.item() - E0.item()) / abs(E0.item()), abs(float(glbf.reshape(-1)[0]) - float(gl.reshape(-1)[0])) / abs(float(gl.reshape(-1)[0])))
        _LZ.REORTH_DEFAULT = "full"
        # ... and the PARTIAL re-orthogonalisation option (SURVEY 8 f-4 "selective reorth"): the same Krylov process and
        # stored basis, re-orthogonalised only when the omega recurrence asks for it -- also beside the headline only
        _LZ.REORTH_DEFAULT = "partial"
        try:
            step()
            barrier()
            t3 = time.perf_counter()
            for _ in range(nb2):
                E0pr, glpr = step()
            barrier()
            ms_partial = (time.perf_counter() - t3) / nb2 * 1e3
            pr_steps = engine.last_reorth_steps
            pr_dev = (abs(E0pr.item() - E0.item()) / abs(E0.item()),
                      abs(float(glpr.reshape(-1)[0]) - float(gl.reshape(-1)[0])) / abs(float(gl.reshape(-1)[0])))
        except Exception as exc:  # noqa: BLE001
            ms_partial, pr_steps, pr_dev = None, None, str(exc)
        _LZ.REORTH_DEFAULT = "full"
    if not args.no_extras and partitioned_path and args.reorth == "full" and not dry and \
            (world == 1 or os.environ.get("DSEA_BENCH_PARTIAL_EXTRA", "") == "1"):
        # row-partitioned run: the partial re-orthogonalisation option on the library driver (same collective sequence on
        # every rank, one more scalar all-reduce per step) -- beside the timed figure, never in its place.  On more than
        # one rank only on request (DSEA_BENCH_PARTIAL_EXTRA=1): an extra must not be able to cost the line its headline.
        _LZ.REORTH_DEFAULT = "partial"
        try:
            step()
            barrier()
            t3 = time.perf_counter()
            nb3 = 2
            for _ in range(nb3):
                E0pr, glpr = step()
            barrier()
            ms_partial = (time.perf_counter() - t3) / nb3 * 1e3
            pr_steps = engine.last_reorth_steps
            pr_dev = (abs(E0pr.item() - E0.item()) / abs(E0.item()),
                      abs(float(glpr.reshape(-1)[0]) - float(gl.reshape(-1)[0])) / abs(float(gl.reshape(-1)[0])))
        except Exception as exc:  # noqa: BLE001  (e.g. the Python step driver: the option needs the library driver)
            ms_partial, pr_steps, pr_dev = None, None, str(exc)
            if rank == 0:
                print("[bench] partial re-orthogonalisation extra skipped: %s" % exc, file=sys.stderr)
        _LZ.REORTH_DEFAULT = "full"
    if not args.no_extras and engine.USE_SHADOW and not partitioned_path:
        engine.USE_SHADOW = False
        step()
        barrier()
        t2 = time.perf_counter()
        nb = min(steps, 5)
        for _ in range(nb):
            step()
        barrier()
        ms_fp64 = (time.perf_counter() - t2) / nb * 1e3
        engine.USE_SHADOW = True
    ms_per_step = dt / steps * 1e3
    alg_bytes = algorithmic_bytes(n, k, m)
    if lp_stats is not None:
        shadow_steps = int(lp_stats[0])
    else:
        shadow_steps = (k - 1) if (prob.use_shadow and k > 1) else 0
    # the adjoint solve runs as one persistent launch for the full-space TFIM operator up to 2^20 rows
    cg_persistent = (not partitioned_path) and args.operator == "matrix-free" and 14 <= L <= 20 and \
        os.environ.get("DSEA_NO_PERSIST", "") != "1"
    total_bytes = traffic_model_bytes(n, k, m, shadow_steps, 5.0 if cg_persistent else 11.0)
    if args.reorth == "none":
        # the basis-free two-pass option is a DIFFERENT algorithm: it is priced with ITS OWN algorithmic bytes, not with
        # SURVEY 8d's full-reorthogonalisation figure (which it does not move).  Per Lanczos step and pass: mat-vec 2 +
        # three-term 4 + scale/store 2 vectors; the second pass also updates psi (2): 18 k vectors in all.
        total_bytes = alg_bytes = 8.0 * n * (18 * k + 11 * m + 24)
    pr_run_steps = None
    if args.reorth == "partial":
        # the partial re-orthogonalisation option, priced with ITS OWN bytes: every step mat-vec 2 + three-term 4 +
        # scale/store 2 vectors; a re-orthogonalised step i adds the two passes over the basis, (i + 1) + (i + 2) vectors --
        # R such steps, taken as spread evenly (mean i = k / 2); Ritz vector k + 1; backward as SURVEY 8d
        pr_run_steps = int(pr_timed_steps or 0)
        total_bytes = alg_bytes = 8.0 * n * (8 * k + pr_run_steps * (k + 3) + (k + 1) + 11 * m + 24)
    value = total_bytes / (ms_per_step * 1e-3) / 1e9
    workload = prob.describe(args.operator, scaling if world > 1 else None)
    E0_site, gl0 = E0.item() / L, float(gl.reshape(-1)[0].item())
    overlap_fb = prob.op.overlap_fallbacks if partitioned_path else None

    # ---- N > 1, default schedule: the weak point beside the timed strong one
    weak_point = None
    if weak_extra:
        prob.release()
        Lw, kw = ((7 + p, 60) if toy else (25 + p, 200))
        try:
            pw = Problem(args, Lw, kw, world, rank, dev, dry, True)
            sw, ww = (2, 1) if toy else (3, 1)
            dtw, E0w, _ = pw.measure(sw, ww)
            mw = pw.cg_iterations()
            msw = dtw / sw * 1e3
            bw = traffic_model_bytes(1 << Lw, kw, mw, (kw - 1) if pw.use_shadow else 0)
            weak_point = {"workload": pw.describe(args.operator, "weak"), "ms_per_step": round(msw, 4), "steps": sw,
                          "warmup": ww, "GBs": round(bw / (msw * 1e-3) / 1e9, 2),
                          "algorithmic_GBs": round(algorithmic_bytes(1 << Lw, kw, mw) / (msw * 1e-3) / 1e9, 2),
                          "cg_iterations": int(mw), "E0_per_site": E0w.item() / Lw,
                          "E0_per_site_closed_form": analytic_E0_per_site(Lw, 1.0),
                          "one_gpu_anchor_ms": STORED_ANCHORS["weak_2p25_rows_k200_ms"],
                          "weak_efficiency_vs_anchor": None if toy else round(STORED_ANCHORS["weak_2p25_rows_k200_ms"] / msw, 4)}
            weak_point.update(pw.notes)
            pw.release()
        except Exception as exc:  # noqa: BLE001  (every rank runs the same code: the exception is collective)
            weak_point = "failed: %s: %s" % (type(exc).__name__, str(exc)[:200])

    # ---- N = 1, default workload: the live one-GPU anchors of the multi-GPU curves (outside the timed steps)
    anchors = None
    default_headline = world == 1 and not partitioned_path and not explicit and args.k is None and \
        args.operator == "matrix-free" and args.reorth == "full"
    if default_headline and not args.no_anchors and not dry:
        prob_keep = prob
        anchors = {}
        free_b, total_b = torch.cuda.mem_get_info(dev)
        for tag, La, ka, sa in (("weak_2p25_rows_k200", 25, 200, 3), ("strong_L28_k100", 28, 100, 2)):
            need = 8.0 * (1 << La) * (ka + 8) * (1.25 if La == 25 else 1.0)
            if need > 0.9 * total_b:
                anchors[tag] = "skipped: needs %.0f GB of %.0f GB" % (need / 1e9, total_b / 1e9)
                continue
            try:
                engine.BasisArena.release()
                engine.Workspace.clear_cache()
                torch.cuda.empty_cache()
                pa = Problem(args, La, ka, 1, 0, dev, False, False)
                dta, E0a, _ = pa.measure(sa, 1)
                ma = pa.cg_iterations()
                msa = dta / sa * 1e3
                anchors[tag] = {"workload": pa.describe(), "ms_per_step": round(msa, 3), "steps": sa, "warmup": 1,
                                "cg_iterations": int(ma), "bf16_shadow_of_basis": bool(pa.use_shadow),
                                "GBs": round(traffic_model_bytes(1 << La, ka, ma, (ka - 1) if pa.use_shadow else 0) / (msa * 1e-3) / 1e9, 1),
                                "algorithmic_GBs": round(algorithmic_bytes(1 << La, ka, ma) / (msa * 1e-3) / 1e9, 1),
                                "E0_per_site_minus_closed_form": E0a.item() / La - analytic_E0_per_site(La, 1.0)}
                pa.release()
            except Exception as exc:  # noqa: BLE001
                anchors[tag] = "failed: %s: %s" % (type(exc).__name__, str(exc)[:200])
                engine.BasisArena.release()
                torch.cuda.empty_cache()
        prob = prob_keep
        prob.activate()          # module-global primitive, shadow flag and reorth default back to the headline problem

    final_line = None
    if rank == 0:
        out = {
            "metric": "DRY RUN on CPU processes (gloo, torch test double of the slab kernels): control flow of the "
                      "multi-rank bench only, not a measurement" if dry else
                      "DominantSparseSymeig fwd+bwd ms & HBM GB/s (TFIM, fp64)" if args.reorth == "full" else
                      "DominantSparseSymeig fwd+bwd GB/s, partial re-orthogonalisation option (TFIM, fp64; not the "
                      "reference's schedule: it re-orthogonalises on every step)" if args.reorth == "partial" else
                      "DominantSparseSymeig fwd+bwd GB/s, basis-free two-pass Lanczos option (TFIM, fp64; not the "
                      "reference's full-reorthogonalisation algorithm)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "value_is": ("the OPTION's own bytes (every Lanczos step mat-vec 2 + three-term 4 + scale/store 2 "
                                    "vectors; each of the %d re-orthogonalised steps the two passes over the basis, taken at "
                                    "the mean step index; Ritz vector; backward as SURVEY 8d) / step time, all ranks"
                                    % pr_run_steps) if args.reorth == "partial" else
                                   ("the OPTION's own bytes (18 k + 11 m + 24 vectors) / step time, all ranks"
                                    if args.reorth == "none" else
                                    "HBM bytes the step's kernels move (traffic model: SURVEY 8d per-phase count with "
                                    "the correction pass of %d Lanczos steps reading the bf16 shadow of the basis) / "
                                    "step time, all ranks" % shadow_steps),
                       "cg_iterations": int(m), "cg_form": "persistent single launch (x, r, d in registers)" if cg_persistent
                       else "streaming (mat-vec, update, direction launches)",
                       "traffic_model_bytes_per_step": total_bytes,
                       "frac_of_hbm_peak": round(value / (HBM_PEAK_GBS * world), 4),
                       "algorithmic_bytes_per_step": alg_bytes,
                       "algorithmic_GBs": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                       "algorithmic_GBs_note": "SURVEY 8d figure: bytes of the REFERENCE's algorithm (all-fp64 basis) / "
                                               "step time; it may exceed the HBM peak because the implementation "
                                               "moves fewer bytes -- not a roofline fraction",
                       "ms_at_hbm_peak_for_algorithmic_bytes": round(alg_bytes / (HBM_PEAK_GBS * world * 1e9) * 1e3, 3),
                       "bf16_shadow_of_basis": bool(prob.use_shadow), "lanczos_reorthogonalisation": args.reorth,
                       **({"steps_reorthogonalised": pr_run_steps, "of": k - 1} if pr_run_steps is not None else {}),
                       "E0_per_site": E0_site, "E0_per_site_closed_form": analytic_E0_per_site(L, 1.0),
                       "dloss_dg": gl0,
                       "adjoint_vs_reference_at_eps1e-7": ADJOINT_DEV_EPS7,
                       "basis_placement_probe_us": [round(t, 1) for t in (engine.BasisArena.last_placement or [])]},
        }
        out["config"].update(notes)
        if evidence is not None:
            out["config"]["collectives"] = evidence
        if world > 1:
            anchor_ms = STORED_ANCHORS["strong_L28_k100_ms"] if strong else STORED_ANCHORS["weak_2p25_rows_k200_ms"]
            canonical = (not explicit and args.k is None and not dry)
            out["config"]["one_gpu_anchor"] = {
                "ms_per_step": anchor_ms, "source": STORED_ANCHORS["strong_source" if strong else "weak_source"],
                "live": "the N = 1 line of the same sequence re-measures it (config.one_gpu_anchors)",
                ("speedup_vs_one_gpu" if strong else "weak_efficiency"):
                    round(anchor_ms / ms_per_step, 4) if canonical else None}
            if weak_point is not None:
                out["config"]["weak_scaling_point"] = weak_point
            if overlap_fb is not None:
                out["config"]["overlap_premise_fallbacks"] = int(overlap_fb)
        if world == 1 and not partitioned_path:
            out["config"]["multi_gpu_schedule"] = (
                "bench.py --gpus N (N > 1) times the STRONG point TFIM L=28, k=100 over N GPUs as value/ms_per_step and "
                "reports the WEAK point (2^25 rows/GPU, k=200) as config.weak_scaling_point; their one-GPU anchors are "
                "config.one_gpu_anchors of this line (speed-up at N = anchor ms / ms_per_step of the N-GPU line)")
        if anchors is not None:
            out["config"]["one_gpu_anchors"] = anchors
            out["config"]["one_gpu_anchors_stored"] = STORED_ANCHORS
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pmc = live_pmc
        if live_pmc_note:
            out["config"]["pmc_live"] = live_pmc_note
        if pmc is None and os.path.exists(tpath):
            try:
                pmc = json.load(open(tpath))
            except Exception:
                pmc = None
        if not partitioned_path and L == 20 and k == 200 and args.operator == "matrix-free":
            # the traffic model against the counters: bytes that crossed the HBM interface in a profiled run of this step
            if pmc and pmc.get("_total_hbm_bytes_per_step"):
                real = float(pmc["_total_hbm_bytes_per_step"])
                out["config"]["pmc_hbm_bytes_per_step"] = real
                out["config"]["pmc_source"] = ("rocprofv3 PMC FETCH_SIZE/WRITE_SIZE measured in THIS run (child processes, same box)"
                                               if pmc.get("_commit") == "live" else
                                               "rocprofv3 PMC FETCH_SIZE/WRITE_SIZE at commit %s (committed file)" % pmc.get("_commit", "?"))
                out["config"]["pmc_GBs"] = round(real / (ms_per_step * 1e-3) / 1e9, 2)
                out["config"]["frac_of_hbm_peak_pmc_traffic"] = round(real / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if ms_basisfree is not None:
            out["config"]["basisfree_two_pass_lanczos"] = {
                "ms_per_step": round(ms_basisfree, 4), "E0_rel_dev_vs_full_reorth": bf_dev[0],
                "dloss_dg_rel_dev_vs_full_reorth": bf_dev[1],
                "note": "reorth='none' option: no stored basis, no re-orthogonalisation; not the reference's algorithm"}
        if ms_partial is not None:
            out["config"]["partial_reorth_lanczos"] = {
                "ms_per_step": round(ms_partial, 4), "steps_reorthogonalised": pr_steps, "of": k - 1,
                "E0_rel_dev_vs_full_reorth": pr_dev[0], "dloss_dg_rel_dev_vs_full_reorth": pr_dev[1],
                "note": "reorth='partial' option (Simon's partial re-orthogonalisation, threshold 1e-10): same stored "
                        "basis, re-orthogonalised only on the steps the omega recurrence selects; not the reference's "
                        "schedule (Lanczos.py:66 re-orthogonalises on every step), never the headline"}
        if ms_fp64 is not None:
            out["config"]["ms_per_step_fp64_basis"] = round(ms_fp64, 4)
            out["config"]["GBs_fp64_basis"] = round(alg_bytes / (ms_fp64 * 1e-3) / 1e9, 2)
        if use_events and launches[0] > 0 and launches[1] > 0:
            dots_b, axpy_b = reorth_bytes_per_launch(nloc, k)
            # the correction pass is priced with the bytes IT reads: bf16 shadow (2 bytes/element) when it is on
            axpy_real = axpy_b if not prob.use_shadow else \
                sum(2.0 * i + 16.0 for i in range(1, k)) / (k - 1) * nloc
            per = {
                "k_rdots": (dots_b, total_ms[0] / launches[0], launches[0]),
                "k_axpy_norm": (axpy_real, total_ms[1] / launches[1], launches[1]),
            }
            name = max(per, key=lambda kk: per[kk][1] * per[kk][2])
            b, ms, cnt = per[name]
            traffic = None
            if pmc and L == 20 and k == 200:
                traffic = pmc.get(name, {}).get("hbm_bytes_per_launch")
            lp, fb = lp_stats if lp_stats is not None else (launches[1] if prob.use_shadow else 0, 0)
            out["roofline"] = {
                "kernel": name, "bound": "hbm", "achieved": round(b / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_commit": pmc.get("_commit") if (pmc and traffic) else None,
                "avg_launch_ms": round(ms, 5), "launches": int(cnt), "launches_per_step": int(cnt) // max(ev_steps, 1),
                "algorithmic_bytes_per_launch": b,
                "other": {kk: {"avg_launch_ms": round(v[1], 5), "achieved_GBs": round(v[0] / (v[1] * 1e-3) / 1e9, 1),
                               "bytes_per_launch": v[0]}
                          for kk, v in per.items() if kk != name},
                "spmv_avg_launch_ms": round(total_ms[2] / max(launches[2], 1), 5),
                "measured": "HIP events on the launch stream, %d instrumented steps run right after the timed "
                            "region (%.3f ms/step with events)" % (ev_steps, dt_instr / ev_steps * 1e3),
                "note": ("k_axpy_norm streams the bf16 shadow of the basis on %d of %d steps (fp64 fallback %d) and is "
                         "priced with the bytes it reads (2 per basis element + r in and out), not with SURVEY 8d's "
                         "8 per element" % (lp, lp + fb, fb)) if not partitioned_path
                        else "rank 0's local kernels in the row-partitioned run",
            }
        if not args.no_extras and world == 1 and not partitioned_path and not big:
            try:
                ceil = measured_ceilings(dev)
                out["config"]["measured_ceilings"] = ceil
                if "roofline" in out:
                    out["roofline"]["frac_of_measured_read_ceiling"] = round(out["roofline"]["achieved"] / ceil["read_GBs"], 4)
            except Exception as exc:  # noqa: BLE001
                out["config"]["measured_ceilings"] = "failed: %s" % exc
            try:
                out["config"]["config3"] = c3_figures(dev)
            except Exception as exc:  # noqa: BLE001
                out["config"]["config3"] = "failed: %s" % exc
        if not args.no_cpu_baseline and world == 1 and not big:
            ncpu = os.cpu_count() or 1
            host = "%s, os.cpu_count()=%d" % (_cpu_model(), ncpu)
            want = [min(int(t), ncpu) for t in args.cpu_threads.split(",") if t] or [min(8, ncpu)]
            if args.cpu_sample:
                # bounded sample of the same workload: same L, fewer Lanczos vectors, capped CG
                _, r = cpu_baseline(L, args.cpu_k, args.cpu_cg_cap, want[0])
                out["cpu_baseline"] = {
                    "value": r["GBs"], "unit": "GB/s", "cores": r["threads"], "kind": "port",
                    "sample": "oracle (torch-CPU port of reference Lanczos.py/CG.py/TFIM.H), TFIM L=%d, k=%d Lanczos "
                              "vectors, CG capped at %d iterations (ran %d), fwd+bwd %.1f s, table build %.1f s not "
                              "timed; GB/s of the algorithmic bytes (SURVEY 8d), which is what a CPU run moves; host: %s"
                              % (L, r["k"], args.cpu_cg_cap, r["cg_iterations"], r["fwd_bwd_s"], r["table_build_s"], host)}
            else:
                # the FULL configuration (k as on the GPU, CG to the reference's tolerance)
                model, runs = None, []
                for th in want:
                    model, r = cpu_baseline(L, k, None, th, model=model)
                    runs.append(r)
                best = max(runs, key=lambda r: r["GBs"])
                out["cpu_baseline"] = {
                    "value": best["GBs"], "unit": "GB/s", "cores": best["threads"], "kind": "port",
                    "sample": "oracle (torch-CPU port of reference Lanczos.py/CG.py/TFIM.H incl. the gather-table "
                              "mat-vec) on the FULL workload: TFIM L=%d, k=%d, CG to ||r||<1e-7 (%d iterations): fwd "
                              "%.1f s + bwd %.1f s; GB/s of the algorithmic bytes (SURVEY 8d); host: %s"
                              % (L, k, best["cg_iterations"], best["fwd_s"], best["bwd_s"], host),
                    "ms_per_step": round(best["fwd_bwd_s"] * 1e3, 1), "runs": runs}
        # RCCL / HIP runtime banners go through C stdio: flush them first so the JSON is the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        final_line = json.dumps(out)
    if partitioned_path:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if final_line is not None:
        print(final_line, flush=True)


if __name__ == "__main__":
    main()
