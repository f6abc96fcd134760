"""What bounds an N > 1 line of bench.py: the collectives measured on the run's own communicators, the exchange time a
Lanczos step / CG iteration still shows, and SURVEY 8e's scaling model evaluated with those numbers beside the measured
speed-up.  Called by bench.py after the timed point (outside the timed region); every rank runs it (collective)."""
import time

import torch

F64 = torch.float64


class _Clock:
    """elapsed ms of what is enqueued between start() and stop(): HIP events on the given stream, wall clock on CPU"""

    def __init__(self, dev, stream=None):
        self.cuda = dev.type == "cuda"
        self.dev, self.stream = dev, stream

    def start(self):
        if self.cuda:
            torch.cuda.synchronize(self.dev)
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record(self.stream or torch.cuda.current_stream(self.dev))
        else:
            self.t0 = time.perf_counter()

    def stop(self):
        if self.cuda:
            self.e1.record(self.stream or torch.cuda.current_stream(self.dev))
            torch.cuda.synchronize(self.dev)
            return self.e0.elapsed_time(self.e1)
        return (time.perf_counter() - self.t0) * 1e3


def _max_over_ranks(x, ctx):
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=F64, device=ctx.ctrl_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def _allreduce_us(ctx, op, count, reps):
    """the all-reduce the solvers issue (library communicator on the solver stream; Python-level comm otherwise)"""
    from ctypes import c_void_p
    buf = torch.ones(max(count, 1), dtype=F64, device=ctx.dev)
    nc = getattr(op, "_ncomm", None) if getattr(op, "_pop", None) else None
    if nc is not None:
        from dominantsparseeigenad_amd import _lib
        lib = _lib.load()
        st = c_void_p(torch.cuda.current_stream(ctx.dev).cuda_stream)

        def once():
            _lib.check(lib.dsea_comm_allreduce(nc.handle, c_void_p(buf.data_ptr()), count, st), "dsea_comm_allreduce")
    else:
        def once():
            op.comm.allreduce(buf[:count])
    for _ in range(3):
        once()
        buf.fill_(1.0)
    clk = _Clock(ctx.dev)
    clk.start()
    for _ in range(reps):
        once()
    return _max_over_ranks(clk.stop() / reps * 1e3, ctx)


def _exchange_figures(ctx, op, reps):
    """one mat-vec's slab exchange, alone, on the stream it runs on in the step (the operator's side stream for the
    library driver): ms and GB/s SENT per GPU"""
    if op.p == 0:
        return None
    n, P = op.nloc, op.world
    x = torch.ones(n, dtype=F64, device=ctx.dev)
    side = getattr(op, "_side_stream", None) if getattr(op, "_pop", None) else None

    def once():
        op._exchange(x)

    lib_path = False
    if getattr(op, "_pop", None) and op.transposed:
        # transposed form through the library's own communicator: two all-to-alls of chunk = nloc / P (+ the flip sum)
        from ctypes import c_void_p
        from dominantsparseeigenad_amd import _lib
        lib = _lib.load()
        nc = op._ncomm
        a, b = torch.empty_like(x), torch.empty_like(x)
        stream = side if side is not None else torch.cuda.current_stream(ctx.dev)
        st = c_void_p(stream.cuda_stream)
        chunk = n // P

        def once():  # noqa: F811
            _lib.check(lib.dsea_comm_alltoall(nc.handle, c_void_p(x.data_ptr()), c_void_p(a.data_ptr()), chunk, st), "alltoall")
            _lib.check(lib.dsea_comm_alltoall(nc.handle, c_void_p(a.data_ptr()), c_void_p(b.data_ptr()), chunk, st), "alltoall")
        lib_path = True
        clk = _Clock(ctx.dev, stream)
    else:
        clk = _Clock(ctx.dev)
    once()
    clk.start()
    for _ in range(reps):
        once()
    ms = _max_over_ranks(clk.stop() / reps, ctx)
    sent = 8.0 * n * (2.0 * (P - 1) / P if op.transposed else op.p)
    return {"form": "transposed: two all-to-alls of 1/P slab per peer" if op.transposed else "pairwise: one slab per hypercube partner",
            "through": "library communicator (exchange communicator, side stream)" if lib_path else "the operator's Python-level exchange",
            "bytes_sent_per_gpu_per_matvec": sent, "ms_per_matvec": round(ms, 4),
            "GBs_sent_per_gpu": round(sent / (ms * 1e-3) / 1e9, 2) if ms > 0 else None}


def _forward_ms(ctx, prob, reps):
    """the k-step Lanczos of the timed point alone, ms (max over ranks)"""
    op = prob.op
    q0 = prob.draws[0]
    op.lanczos(prob.k, q0, arena=True)
    clk = _Clock(ctx.dev)
    clk.start()
    for _ in range(reps):
        op.lanczos(prob.k, q0, arena=True)
    return _max_over_ranks(clk.stop() / reps, ctx)


def _cg_ms_per_iteration(ctx, prob, E0, iters):
    """`iters` iterations of the adjoint solve's CG on this operator (tolerance 0: it cannot stop early), ms each"""
    op = prob.op
    b = prob.draws[1] / 1e3
    x0 = prob.draws[2].clone()
    t0 = _Clock(ctx.dev)
    t0.start()
    op.solve_shifted(E0.detach(), b, x0, eps=0.0, maxiter=iters)
    return _max_over_ranks(t0.stop() / iters, ctx)


def scaling_decomposition(ctx, prob, pt, ms_per_step, m, E0, one_gpu_ms):
    """dict for config.scaling_decomposition (see the module docstring); ``one_gpu_ms``: the like-by-like one-GPU anchor
    of the timed point or None"""
    op, k, P = prob.op, pt.k, ctx.world
    out = {"note": "measured after the timed region on the run's own communicators; every figure is the max over ranks"}
    reps = 20 if ctx.dry else 200
    out["allreduce_us"] = {"8_bytes": round(_allreduce_us(ctx, op, 1, reps), 2),
                           "1600_bytes": round(_allreduce_us(ctx, op, 200, reps), 2),
                           "through": "library communicator, solver stream" if getattr(op, "_pop", None) else "Python-level communicator"}
    out["exchange"] = _exchange_figures(ctx, op, 3 if pt.big else 10)
    reps_f = 1 if pt.big else 2
    saved_overlap, saved_fb = op.overlap, op.overlap_fallbacks
    try:
        t_timed = _forward_ms(ctx, prob, reps_f)
        t_off = None
        if op.p > 0 and saved_overlap:
            op.overlap = False
            t_off = _forward_ms(ctx, prob, reps_f)
            op.overlap = saved_overlap
        op.measure_without_exchange = True
        t_none = _forward_ms(ctx, prob, reps_f) if op.p > 0 else t_timed
        op.measure_without_exchange = False
        out["lanczos_forward_ms"] = {"as_timed": round(t_timed, 3), "exchange_after_correction": round(t_off, 3) if t_off else None,
                                     "without_exchange": round(t_none, 3)}
        exp_lz = max(t_timed - t_none, 0.0) / k
        out["exposed_exchange_ms_per_lanczos_step"] = round(exp_lz, 5)
        out["hidden_exchange_ms_per_lanczos_step"] = round(max(t_off - t_timed, 0.0) / k, 5) if t_off else 0.0
        replicated = bool(getattr(op, "_replicated", lambda: False)())
        iters = max(min(int(m), 20 if pt.big else 60), 4)
        c_timed = _cg_ms_per_iteration(ctx, prob, E0, iters)
        if op.p > 0 and not replicated:
            op.measure_without_exchange = True
            c_none = _cg_ms_per_iteration(ctx, prob, E0, iters)
            op.measure_without_exchange = False
        else:
            c_none = c_timed
        exp_cg = max(c_timed - c_none, 0.0)
        out["cg_ms_per_iteration"] = {"as_timed": round(c_timed, 5), "without_exchange": round(c_none, 5), "iterations_timed": iters,
                                      "solve": "replicated on every rank (no exchange, no all-reduce)" if replicated else "row-partitioned"}
        out["exposed_exchange_ms_per_cg_iteration"] = round(exp_cg, 5)
    finally:
        op.measure_without_exchange = False
        op.overlap, op.overlap_fallbacks = saved_overlap, saved_fb
    # SURVEY 8e's model: local work divides by P; each Lanczos step adds two latency-bound all-reduces (coefficients; the
    # pair of scalars) and whatever of its exchange is not hidden; each CG iteration two scalar all-reduces and its exchange
    ar8, ar1600 = out["allreduce_us"]["8_bytes"] * 1e-3, out["allreduce_us"]["1600_bytes"] * 1e-3
    # all-reduces per CG iteration of the form ACTUALLY run (round-5 advisor): the library driver's default for the TFIM operand
    # is the one-reduction form (one 16-byte all-reduce), the Python step driver and the reference recurrences issue two scalars
    from dominantsparseeigenad_amd import engine as _engine
    form = str(getattr(_engine.last_cg, "form", ""))
    one_red = "one all-reduce" in form
    out["allreduce_us"]["16_bytes"] = round(_allreduce_us(ctx, op, 2, reps), 2)
    ar16 = out["allreduce_us"]["16_bytes"] * 1e-3
    cg_ar = ar16 if one_red else 2.0 * ar8
    out["cg_allreduces_per_iteration"] = {"count": 1 if one_red else 2, "form": form}
    comm_ms = k * (ar1600 + ar8 + exp_lz) + (0.0 if replicated else m * (cg_ar + exp_cg))
    out["communication_ms_per_step"] = round(comm_ms, 3)
    out["model"] = {"formula": "T_P = T_1 / P + k (allreduce_1600B + allreduce_8B + exposed_exchange_lanczos) + m (%s + "
                               "exposed_exchange_cg)   [m-term dropped when the solve is replicated]; SURVEY 8e"
                               % ("allreduce_16B" if one_red else "2 allreduce_8B"),
                    "timed_point": model_entry(one_gpu_ms, ms_per_step, P, comm_ms)}
    return out


def model_entry(one_gpu_ms, measured_ms, P, comm_ms):
    """SURVEY 8e's model for one (one-GPU anchor, N-GPU measurement) pair that divides like by like"""
    if not one_gpu_ms or not measured_ms:
        return {"one_gpu_ms": one_gpu_ms, "measured_ms": measured_ms, "predicted_ms": None, "predicted_speedup": None,
                "measured_speedup": None, "why_null": "no like-by-like one-GPU anchor for this point (toy sizes, explicit --L / "
                                                      "--k, shadow on one side only, or no N = 1 line found)"}
    pred = one_gpu_ms / P + comm_ms
    return {"one_gpu_ms": one_gpu_ms, "measured_ms": round(measured_ms, 3), "predicted_ms": round(pred, 3),
            "predicted_speedup": round(one_gpu_ms / pred, 4), "measured_speedup": round(one_gpu_ms / measured_ms, 4),
            "unexplained_ms": round(measured_ms - pred, 3)}
