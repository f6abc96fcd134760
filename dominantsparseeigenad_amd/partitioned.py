"""Row-partitioned operators: the dominant-eigenpair primitives across P GPUs (one process per GPU) BEHIND THE
REFERENCE API.

New capability (the reference is single-device).  A row-partitioned operator is handed to
the unchanged entry points

    op = PartitionedTFIMOperator(L, g, device)            # or PartitionedStencil3Operator(...)
    symeig.setDominantSparseSymeig(op.H, op.Hadjoint_to_gadjoint)        reference symeig.py:33
    E0, psi = symeig.DominantSparseSymeig.apply(g, k, op.dim, device)    reference symeig.py:71
    loss = E0 + op.dot(psi, t);  torch.autograd.grad(loss, g, create_graph=True) ...

and every n-vector (``psi``, ``t``, the Krylov basis, the CG vectors) is THIS RANK'S SLAB of n/P contiguous rows:

  * all vector algebra is slab-local and runs in the same HIP phase kernels as the single-GPU path
    (include/dsea.h "vector phases" / "row-partitioned macro phases"); each phase leaves its LOCAL partial
    sum in a device scalar;
  * inner products are closed by ``torch.distributed.all_reduce`` on those device scalars (backend "nccl" =
    RCCL over xGMI on the GPU box, "gloo" in the CPU tests): per Lanczos step two small all-reduces, per CG
    iteration two scalars; all ranks hold bit-identical scalars and take the same branch in the CG stopping
    test (reference CG.py:28,35);
  * the mat-vec exchanges what the operator's coupling needs: TFIM -- the low L-p bit flips are slab-local,
    the top p bits come from the partner slabs rank ^ (1<<b) (pairwise hypercube exchange for P = 2, transposed
    all-to-all form from P = 4 on); 3-point stencil (reference examples/schrodinger1D.py:18-27) -- one halo
    element per neighbour;
  * the autograd glue of symeig.py / CG.py is shared with the single-GPU path; its inner products and
    scalar-vector products go through ``PartitionedSpace`` (differentiable global dot, see _space.py), so the
    re-entrant backward -- d2E0/dg2, the fidelity susceptibility -- works distributed exactly as reference
    symeig.py:77-86 / CG.py:128-138 do on one device.

The numerical kernels are reached through a small backend object; the product backend is ``HipBackend``
(libdsea.so, no fallback).  The CPU tests inject a torch-CPU test double to exercise the partition /
exchange / all-reduce logic with gloo on machines without GPUs.
"""
from __future__ import annotations

from ctypes import byref, c_void_p

import numpy as np
import torch
import torch.distributed as dist

from . import engine as _engine_mod

F64 = torch.float64
CG_RR, CG_DAD, CG_RRNEW, CG_ALPHA, CG_BETA, CG_RESNORM, CG_DONE, CG_ITERS = range(8)


# =========================================================================== product backend (HIP)
class HipBackend(_engine_mod.Phases):
    """Slab-local numerics through the C ABI (the vector phases come from ``engine.Phases``)."""

    def __init__(self, n_local, device):
        from . import _lib
        super().__init__(int(n_local), device, kmax=8)
        self._lib_mod = _lib
        self.engine = _engine_mod
        self.op = None          # slab-local operator object (exposes .handle), set by attach_*
        self._shadow = None

    def spawn(self, n_local):
        """a backend of the same kind for slabs of another size (the replicated CG of two ranks)"""
        return HipBackend(n_local, self.device)

    # -- helpers
    @staticmethod
    def _p(t):
        return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)

    def _ck(self, rc, what):
        self._lib_mod.check(rc, what)

    # -- slab-local operators
    def attach_tfim(self, L, L_local, row_offset, g):
        from .operators import TFIMOperator
        self.op = TFIMOperator(L, self.device, g=g, L_local=L_local, row_offset=row_offset)
        return self.op

    def tfim_local(self, x, y, which="H"):
        handle = self.op.handle if which == "H" else self.op._dHdg.handle
        self._ck(self.lib.dsea_spmv(handle, None, self._p(x), self._p(y), None, None, None, self._st()), "dsea_spmv")

    def tfim_local_shift_dot(self, x, y, shift, out, skip):
        """one rank: y = (H - shift) x with the local x.y from the mat-vec's own epilogue (nothing to exchange)"""
        self._ck(self.lib.dsea_spmv(self.op.handle, self.ws.handle, self._p(x), self._p(y), self._p(shift),
                                    self._p(out), self._p(skip), self._st()), "dsea_spmv")

    def attach_stencil(self, n_local, coef, V, halo, has_lo, has_hi):
        """3-point stencil on this slab; halo[0] / halo[1] are the neighbours' edge elements (device doubles
        filled by the halo exchange); a missing neighbour is the Dirichlet zero of schrodinger1D.py:20-21"""
        from .operators import _Handle, _NativeView
        raw = c_void_p()
        lo = c_void_p(halo.data_ptr()) if has_lo else c_void_p(None)
        hi = c_void_p(halo.data_ptr() + 8) if has_hi else c_void_p(None)
        self._ck(self.lib.dsea_op_create_stencil3(int(n_local), float(coef), self._p(V), lo, hi, byref(raw)),
                 "dsea_op_create_stencil3")
        self.op = _NativeView(_Handle(raw, n_local, (V, halo)))
        return self.op

    def stencil_local(self, x, y, shift, out, skip):
        """y = A_slab x - shift x (halo already exchanged) ; out = local x.y"""
        self._ck(self.lib.dsea_spmv(self.op.handle, self.ws.handle, self._p(x), self._p(y), self._p(shift),
                                    self._p(out), self._p(skip), self._st()), "dsea_spmv")

    def attach_csr(self, rowptr, cols, vals, n_local, hb, halo, xg):
        """explicit-matrix slab: SELL-64 operator on n_local rows whose columns are LOCAL in [-hb, n_local + hb) with the
        two neighbour halos in ``halo`` (2 hb doubles: [from rank-1 | from rank+1]), or GLOBAL with the all-gathered
        vector ``xg`` (hb = -1)                                                   include/dsea.h: dsea_op_set_slab"""
        from .operators import CSROperator
        self.op = CSROperator(rowptr, cols, vals, int(n_local), values="plain")   # (a slab is never value-coded: dsea_op_set_slab)
        lo = c_void_p(halo.data_ptr()) if hb > 0 else c_void_p(None)
        hi = c_void_p(halo.data_ptr() + 8 * hb) if hb > 0 else c_void_p(None)
        self._ck(self.lib.dsea_op_set_slab(self.op._H.handle, int(hb), lo, hi, self._p(xg) if hb < 0 else c_void_p(None)),
                 "dsea_op_set_slab")
        self._csr_keep = (halo, xg)
        return self.op

    def csr_local(self, x, y, shift, out, skip):
        """y = A_slab x - shift x (halo / gathered vector already exchanged) ; out = local x.y"""
        self._ck(self.lib.dsea_spmv(self.op.handle, self.ws.handle, self._p(x), self._p(y), self._p(shift),
                                    self._p(out), self._p(skip), self._st()), "dsea_spmv")

    def csr_sddmm_local(self, v1, v2, alpha, accumulate, out):
        """out[e] (+)= alpha v1[row e] v2[col e] on the slab (v2's halo / gathered copy already exchanged)"""
        from . import _lib
        self._ck(self.lib.dsea_op_sddmm(self.op._H.handle, self._p(self.op.rowptr), self._p(v1), self._p(v2),
                                        float(alpha), _lib.SDDMM_ACCUMULATE if accumulate else 0, self._p(out),
                                        self._st()), "dsea_op_sddmm")

    # -- macro phases of the partitioned Lanczos step (include/dsea.h "row-partitioned macro phases")
    def basis(self, k, ldq, arena):
        if arena:
            return self.engine.BasisArena.matrix(self.device, "Q", k, ldq, F64, n_hint=self.n)
        return self.empty(k, ldq)

    def set_shadow(self, k, ldq, arena=False):
        self._shadow = self.engine.BasisArena.matrix(self.device, "Qs", k, ldq, torch.bfloat16) if arena else \
            torch.empty((k, ldq), dtype=torch.bfloat16, device=self.device)
        self._ck(self.lib.dsea_ws_set_shadow(self.ws.handle, self._p(self._shadow), ldq, int(k),
                                             float(self.engine.SHADOW_TAU)), "dsea_ws_set_shadow")

    def clear_shadow(self):
        self._ck(self.lib.dsea_ws_set_shadow(self.ws.handle, None, 0, 0, 0.0), "dsea_ws_set_shadow")
        self._shadow = None

    def form_r(self, Q, ldq, n, i, u, alpha, beta, r, r_copy):
        self._ck(self.lib.dsea_lanczos_form_r(self.ws.handle, self._p(Q), ldq, n, i, self._p(u), self._p(alpha),
                                              self._p(beta), self._p(r), self._p(r_copy), self._st()),
                 "dsea_lanczos_form_r")

    def flipsum(self, xT, zT, P):
        self._ck(self.lib.dsea_hypercube_flipsum(self._p(xT), self._p(zT), int(P), xT.numel() // int(P), self._st()),
                 "dsea_hypercube_flipsum")

    def plz_dots(self, Q, ldq, n, i, u, alpha, beta, r, c):
        self._ck(self.lib.dsea_plz_dots(self.ws.handle, self._p(Q), ldq, n, i, self._p(u), self._p(alpha),
                                        self._p(beta), self._p(r), self._p(c), self._st()), "dsea_plz_dots")

    def plz_correct(self, Q, ldq, n, row, c, r, pair):
        self._ck(self.lib.dsea_plz_correct(self.ws.handle, self._p(Q), ldq, n, int(row), self._p(c), self._p(r),
                                           self._p(pair), self._st()), "dsea_plz_correct")

    def plz_correct_matvec(self, Q, ldq, row, c, r, y, pair):
        self._ck(self.lib.dsea_plz_correct_matvec(self.op.handle, self.ws.handle, self._p(Q), ldq, int(row),
                                                  self._p(c), self._p(r), self._p(y), self._p(pair), self._st()),
                 "dsea_plz_correct_matvec")

    def axpy_multi_dot(self, a_host, a_dev, xs, shift, skip, x, y, out):
        arr = (c_void_p * max(len(xs), 1))(*[t.data_ptr() for t in xs])
        self._ck(self.lib.dsea_axpy_multi_dot(self.ws.handle, float(a_host), self._p(a_dev), arr, len(xs),
                                              self._p(shift), self._p(skip), self._p(x), self._p(y), x.numel(),
                                              self._p(out), self._st()), "dsea_axpy_multi_dot")

    def plz_finish(self, r, y, pair, q_out, row, u_out, alpha_out, beta_out):
        self._ck(self.lib.dsea_plz_finish(self.ws.handle, self._p(r), self._p(y), self._p(pair), self._p(q_out),
                                          int(row), self._p(u_out), self._p(alpha_out), self._p(beta_out),
                                          r.numel(), self._st()), "dsea_plz_finish")

    def cg_init(self, b, Ax0, r, d, state):
        self._ck(self.lib.dsea_cg_init(self.ws.handle, self._p(b), self._p(Ax0), self._p(r), self._p(d),
                                       self._p(state), b.numel(), self._st()), "dsea_cg_init")

    def cg_init_check(self, state, eps):
        self._ck(self.lib.dsea_cg_init_check(self.ws.handle, self._p(state), float(eps), self._st()), "dsea_cg_init_check")

    def cg_update(self, x, r, d, Ad, state):
        self._ck(self.lib.dsea_cg_update(self.ws.handle, self._p(x), self._p(r), self._p(d), self._p(Ad),
                                         self._p(state), x.numel(), self._st()), "dsea_cg_update")

    def cg_check(self, state, eps):
        self._ck(self.lib.dsea_cg_check(self.ws.handle, self._p(state), float(eps), self._st()), "dsea_cg_check")

    def cg_direction(self, r, d, state):
        self._ck(self.lib.dsea_cg_direction(self.ws.handle, self._p(r), self._p(d), self._p(state), r.numel(),
                                            self._st()), "dsea_cg_direction")


# =========================================================================== collectives
class TorchDistComm:
    """Stream-ordered collectives of ``torch.distributed``: backend "nccl" (= RCCL over xGMI) for device
    tensors, "gloo" for host tensors.  (gloo is NOT stream-ordered for device tensors; tests that drive two
    ranks on one GPU pass a host-staging subclass instead.)"""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._side = None

    def allreduce(self, t):
        if self.world > 1:
            dist.all_reduce(t, group=self.group)

    def sendrecv(self, items):
        """items = [(send tensor, receive buffer, peer rank), ...]: all posted at once, then waited for"""
        ops = []
        for snd, rcv, peer in items:
            ops.append(dist.P2POp(dist.isend, snd, peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, rcv, peer, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def exchange(self, x, recv, peers):
        """send slab x to every peer, receive theirs into recv[b] (pairwise, all at once)"""
        self.sendrecv([(x, buf, peer) for buf, peer in zip(recv, peers)])

    def start_flip_exchange(self, be, x_send, xT, zT, z):
        """z = sum over the top-bit partner slabs of x_send, in the transposed form (all-to-all, local flip sum,
        all-to-all back), started NOW and left running: on a device the three stages are enqueued on a side
        stream that waits for the producer of x_send, so they overlap whatever the caller enqueues next on its
        own stream.  Returns a token for ``finish_flip_exchange``.  (Host tensors / gloo: done synchronously.)"""
        if not x_send.is_cuda:
            self.all_to_all(x_send, xT)
            be.flipsum(xT, zT, self.world)
            self.all_to_all(zT, z)
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=x_send.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(x_send.device))
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            self.all_to_all(x_send, xT)
            be.flipsum(xT, zT, self.world)
            self.all_to_all(zT, z)
            done = torch.cuda.Event()
            done.record(self._side)
        return done

    def finish_flip_exchange(self, token, device):
        if token is not None:
            torch.cuda.current_stream(device).wait_event(token)

    def start_pair_exchange(self, x_send, recv, peers):
        """pairwise form of the same: the slab x_send goes to every hypercube partner, theirs arrive in recv[b];
        started NOW on the side stream (device tensors) and joined by ``finish_flip_exchange``.  With P = 2 a
        mat-vec moves one whole slab over the ONE xGMI link between the two GPUs (268 MB at 2^25 rows: about as long
        as the dots pass of a Lanczos step), so hiding it matters most exactly there."""
        if not x_send.is_cuda:
            self.exchange(x_send, recv, peers)
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=x_send.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(x_send.device))
        with torch.cuda.stream(self._side):
            self._side.wait_event(ready)
            self.exchange(x_send, recv, peers)
            done = torch.cuda.Event()
            done.record(self._side)
        return done

    def all_gather(self, slab, full):
        """full = the slabs of all ranks in rank order (equal slabs)"""
        if self.world == 1:
            full.copy_(slab)
            return
        dist.all_gather_into_tensor(full, slab.contiguous(), group=self.group)

    def all_to_all(self, src, dst):
        """chunk j of src goes to rank j, chunk j of dst comes from rank j (equal chunks; own chunk copied).
        Over RCCL this is ``all_to_all_single``; elsewhere (gloo in the CPU tests) the same exchange is written
        as one group of point-to-point operations, which is what an all-to-all is."""
        P, me = self.world, self.rank
        if src.is_cuda and dist.get_backend(self.group) == "nccl":
            # RCCL: one call (a Python-level group of 2(P-1) point-to-point ops costs more host time per step
            # than the whole step takes on the device)
            dist.all_to_all_single(dst, src, group=self.group)
            return
        chunk = src.numel() // P
        dst[me * chunk:(me + 1) * chunk].copy_(src[me * chunk:(me + 1) * chunk])
        self.sendrecv([(src[j * chunk:(j + 1) * chunk], dst[j * chunk:(j + 1) * chunk], j) for j in range(P) if j != me])


class HostStagedComm(TorchDistComm):
    """The same collectives staged through the host around each call, for a process group whose backend is not
    stream-ordered on device tensors (gloo): several ranks SHARING ONE GPU -- RCCL refuses two ranks on a device --
    which is how the multi-rank paths (P = 2, 4, 8 geometry, library driver through the callback communicator) run on
    real slab kernels on a one-GPU box: the tests of tests/test_gpu_partitioned.py and ``bench.py --host-staged``.
    A transport for rehearsals, not for measurements."""

    def allreduce(self, t):
        h = t.cpu()
        if self.world > 1:
            dist.all_reduce(h, group=self.group)
        t.copy_(h)

    def sendrecv(self, items):
        hs = [(s.cpu(), torch.empty(r.shape, dtype=r.dtype), peer) for s, r, peer in items]
        super().sendrecv(hs)
        for (_, dst, _), (_, src, _) in zip(items, hs):
            dst.copy_(src)

    def all_to_all(self, src, dst):
        hs = src.cpu()
        hd = torch.empty_like(hs)
        super().all_to_all(hs, hd)
        dst.copy_(hd)

    def all_gather(self, slab, full):
        hf = torch.empty(full.shape, dtype=full.dtype)
        super().all_gather(slab.cpu(), hf)
        full.copy_(hf)


class RankOrderedHostStagedComm(HostStagedComm):
    """HostStagedComm whose all-reduce adds the contributions in RANK ORDER (all-gather, then a left-to-right sum):
    the order tests/fake_rccl uses, so that a run over the stand-in RCCL communicators and a run over the callback
    communicator can be compared bit for bit at any world size (gloo's ring all-reduce associates differently from
    P = 3 on)."""

    def allreduce(self, t):
        h = t.cpu()
        if self.world > 1:
            parts = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(parts, h, group=self.group)
            acc = parts[0].clone()
            for q in parts[1:]:
                acc += q
            h = acc
        t.copy_(h)


# =========================================================================== library-side communicator
def _device_view(ptr, count, device):
    """zero-copy fp64 torch tensor over ``count`` doubles of device memory at ``ptr`` (used by the callback
    communicator to hand the library's buffers to Python-level collectives)"""
    class _Arr:
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(_Arr(), device=device)


class NativeComm:
    """A communicator handle of include/dsea.h (``dsea_comm_t``): what the library-side row-partitioned solvers
    (``dsea_pop_lanczos_run`` / ``dsea_pop_cg_run``) issue their collectives on.  Three ways to get one:

      * ``adopt_torch(group)``  -- the ncclComm_t values PyTorch's ProcessGroupNCCL already holds (``_comm_ptr()``):
        the group's own communicator for the all-reduces and the communicator of a second process group for the slab
        exchange (RCCL orders the operations of ONE communicator across streams; the exchange must not queue behind
        the all-reduces it is meant to overlap);
      * ``own(group)``          -- the library creates both RCCL communicators itself from two unique ids produced on
        rank 0 and broadcast over ``group``;
      * ``from_python(comm)``   -- any object with ``allreduce / all_to_all / sendrecv`` on torch tensors (the
        host-staged gloo communicator of the one-GPU tests, MPI wrappers ...) as blocking callbacks.
    """

    # One communicator pair per (process group, device) for the life of the process: operators are built freely
    # (replica operators, bench problems, sweeps) and must not each cost a process group and an RCCL communicator.
    _cache = {}

    def __init__(self, handle, rank, world, kind, keep=(), xgroup=None):
        self.handle, self.rank, self.world, self.kind = handle, rank, world, kind
        self._keep = keep            # ctypes callbacks / process groups that must outlive the handle
        self._xgroup = xgroup        # exchange process group created HERE (close() destroys it), else None

    def close(self):
        """destroy the library handle and the exchange process group this object created (idempotent)"""
        try:
            if getattr(self, "handle", None):
                from . import _lib
                _lib.load().dsea_comm_destroy(self.handle)
        except Exception:
            pass
        self.handle = None
        xg, self._xgroup = getattr(self, "_xgroup", None), None
        if xg is not None:
            try:
                if dist.is_initialized():
                    dist.destroy_process_group(xg)
            except Exception:
                pass

    def __del__(self):
        # the handle only: tearing a process group down from a finaliser (interpreter exit, arbitrary order across
        # ranks) is not safe -- ``close()`` / ``release_all()`` do that explicitly
        try:
            if getattr(self, "handle", None):
                from . import _lib
                _lib.load().dsea_comm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @classmethod
    def release_all(cls):
        """close every cached communicator pair (call before ``dist.destroy_process_group()``; operators built on
        them must be gone)"""
        for nc in list(cls._cache.values()):
            nc.close()
        cls._cache.clear()

    @staticmethod
    def _is_world(group):
        return group is None or group is dist.group.WORLD

    @staticmethod
    def _torch_comm_ptr(group, device):
        """ncclComm_t of ``group`` on ``device`` as an integer (the communicator exists after the group's first
        collective, or at once with eager initialisation)"""
        t = torch.zeros(1, dtype=F64, device=device)
        dist.all_reduce(t, group=group)
        torch.cuda.synchronize(device)
        backend = (group if group is not None else dist.group.WORLD)._get_backend(torch.device(device))
        return int(backend._comm_ptr())

    @classmethod
    def adopt_torch(cls, group, device, exchange_group=None):
        """``exchange_group``: a second process group over the same ranks for the slab exchange.  When omitted it is
        created here -- ``dist.new_group`` is collective over the WHOLE world, so that is only done when ``group`` IS
        the world; for a sub-group the caller passes one (or ``for_torch_group`` lets the library create its own pair,
        which needs a broadcast inside the group only)."""
        from . import _lib
        lib = _lib.load()
        import os
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        made = None
        if world > 1 and os.environ.get("DSEA_COMM_SINGLE", "") == "1":
            xgroup = group          # one communicator for the all-reduces and the exchange (see ``own``)
        elif world > 1:
            if exchange_group is None:
                if not cls._is_world(group):
                    raise ValueError("adopt_torch on a sub-group needs exchange_group= (dist.new_group is collective "
                                     "over the whole world)")
                ranks = dist.get_process_group_ranks(dist.group.WORLD)
                exchange_group = made = dist.new_group(ranks=ranks, backend="nccl")
            xgroup = exchange_group
        else:
            xgroup = group
        coll = cls._torch_comm_ptr(group, device)
        xchg = cls._torch_comm_ptr(xgroup, device) if world > 1 else coll
        h = c_void_p()
        _lib.check(lib.dsea_comm_adopt(c_void_p(coll), c_void_p(xchg), rank, world, byref(h)), "dsea_comm_adopt")
        return cls(h, rank, world, "rccl (adopted from torch.distributed: %s)" %
                   ("two communicators" if xchg != coll else "one communicator"), keep=(group, xgroup), xgroup=made)

    @classmethod
    def own(cls, group, device, single=None):
        """``group`` only carries the two unique ids from its rank 0 to the others (any backend: the ids travel as
        Python objects).  ``single`` (default: ``DSEA_COMM_SINGLE=1``): ONE communicator for the all-reduces and the
        slab exchange -- the exchange then queues behind the all-reduces of its own step (RCCL orders the operations of
        one communicator), which costs the overlap and removes every cross-communicator ordering question: the second
        stage of bench.py's N > 1 fallback ladder."""
        from . import _lib
        import ctypes
        import os
        lib = _lib.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if single is None:
            single = os.environ.get("DSEA_COMM_SINGLE", "") == "1"
        ids = [None, None]
        if rank == 0:
            for j in range(2):
                buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
                _lib.check(lib.dsea_comm_unique_id(buf), "dsea_comm_unique_id")
                ids[j] = bytes(buf.raw)
        dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        two = world > 1 and not single
        h = c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.dsea_comm_init_rank(ids[0], ids[1] if two else None, rank, world, byref(h)),
                       "dsea_comm_init_rank")
        return cls(h, rank, world, "rccl (library-owned, %s)" % ("two communicators" if two else "one communicator"))

    @classmethod
    def for_torch_group(cls, group, device, exchange_group=None):
        """RCCL communicator pair for ``group`` on ``device``, CACHED per (group, device): adopted from torch where torch
        exposes its ncclComm_t and the exchange group is available (``exchange_group=``, or ``group`` is the world so
        that it can be created collectively), otherwise created by the library from ids broadcast inside ``group``
        (the choice depends only on the software stack and on the arguments, so every rank takes the same branch).
        ``DSEA_COMM=own|adopt`` forces one."""
        import os
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        mode = os.environ.get("DSEA_COMM", "")
        key = ("WORLD" if cls._is_world(group) else id(group), str(device), mode,
               id(exchange_group) if exchange_group is not None else None, os.environ.get("DSEA_COMM_SINGLE", ""))
        live = group if group is not None else dist.group.WORLD
        hit = cls._cache.get(key)
        if hit is not None and hit.handle and getattr(hit, "_group_ref", lambda: None)() is live:
            return hit
        if hit is not None:
            # the process group this pair was made for is gone (dist.destroy_process_group() and a new init in the same
            # process give a NEW group object): an adopted handle points at a destroyed ncclComm_t -- drop it
            hit.handle = None if "adopted" in hit.kind else hit.handle
            hit.close()
            del cls._cache[key]
        backend_cls = getattr(torch._C._distributed_c10d, "ProcessGroupNCCL", None)
        can_adopt = backend_cls is not None and hasattr(backend_cls, "_comm_ptr")
        adoptable = can_adopt and (exchange_group is not None or cls._is_world(group) or dist.get_world_size(group) == 1)
        if mode == "own" or not adoptable:
            if mode == "adopt":
                raise RuntimeError("DSEA_COMM=adopt: torch's communicators cannot be adopted here (%s)" %
                                   ("no ProcessGroupNCCL._comm_ptr in this torch build" if not can_adopt else
                                    "sub-group without exchange_group="))
            nc = cls.own(group, device)
        else:
            nc = cls.adopt_torch(group, device, exchange_group)
        import weakref
        nc._group_ref = weakref.ref(live)
        cls._cache[key] = nc
        return nc

    @classmethod
    def from_python(cls, comm, device):
        from . import _lib
        lib = _lib.load()
        device = torch.device(device)

        def on(stream):
            return torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=device) if stream else
                                     torch.cuda.default_stream(device))

        def allreduce(user, buf, count, stream):
            try:
                with on(stream):
                    comm.allreduce(_device_view(buf, count, device))
                return 0
            except Exception:       # noqa: BLE001 -- reported to the library as DSEA_ERR_COMM
                import traceback
                traceback.print_exc()
                return 1

        def alltoall(user, send, recv, chunk, stream):
            try:
                with on(stream):
                    comm.all_to_all(_device_view(send, chunk * comm.world, device), _device_view(recv, chunk * comm.world, device))
                return 0
            except Exception:       # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        def sendrecv(user, send, recv, count, peer, stream):
            try:
                with on(stream):
                    comm.sendrecv([(_device_view(send, count, device), _device_view(recv, count, device), int(peer))])
                return 0
            except Exception:       # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        if comm.world == 1 and not hasattr(comm, "all_to_all"):
            cbs = (None, None, None)
            args = (None, None, None)
        else:
            cbs = (_lib.ALLREDUCE_FN(allreduce), _lib.ALLTOALL_FN(alltoall), _lib.SENDRECV_FN(sendrecv))
            import ctypes
            args = tuple(ctypes.cast(cb, c_void_p) for cb in cbs)
        h = c_void_p()
        _lib.check(lib.dsea_comm_create_callbacks(comm.rank, comm.world, args[0], args[1], args[2], None, byref(h)),
                   "dsea_comm_create_callbacks")
        return cls(h, comm.rank, comm.world, "callbacks (%s)" % type(comm).__name__, keep=cbs + (comm,))


class _SelfComm:
    """the communicator of ONE rank: nothing to reduce, nothing to exchange (replicated solves)"""
    rank, world = 0, 1

    def allreduce(self, t):
        pass


def _vec(v):
    """detached, fp64, contiguous, 16-byte aligned (what the phase kernels take); copies only when it has to"""
    v = v.detach()
    if v.dtype != F64:
        v = v.to(F64)
    if not v.is_contiguous():
        v = v.contiguous()
    if v.is_cuda and v.data_ptr() % 16:
        v = v.clone()
    return v


# =========================================================================== the vector space of slabs
class _GlobalDot(torch.autograd.Function):
    """s = sum over ranks of a_slab . b_slab (replicated 0-dim tensor).  Gradients of replicated quantities are
    kept FULL on every rank, so the backward needs no communication: a-bar = s-bar b, b-bar = s-bar a."""

    @staticmethod
    def forward(ctx, a, b, space):
        ctx.space = space
        ctx.save_for_backward(a, b)
        out = torch.zeros(1, dtype=F64, device=a.device)
        space.be.dot(_vec(a), _vec(b), out)
        space.comm.allreduce(out)
        return out.reshape(())

    @staticmethod
    def backward(ctx, gs):
        a, b = ctx.saved_tensors
        sp = ctx.space
        return sp.scale(gs, b), sp.scale(gs, a), None


class _Scale(torch.autograd.Function):
    """v_out = s * v_slab for a replicated scalar s.  s-bar = GLOBAL dot(v_out-bar, v) -- the reason this is
    not a plain multiplication: autograd's own rule would leave only this rank's part of the sum."""

    @staticmethod
    def forward(ctx, s, v, space):
        ctx.space = space
        ctx.save_for_backward(s, v)
        return s.detach().reshape(()) * v.detach()

    @staticmethod
    def backward(ctx, gv):
        s, v = ctx.saved_tensors
        sp = ctx.space
        gs = sp.dot(gv, v).reshape(s.shape) if ctx.needs_input_grad[0] else None
        gvv = sp.scale(s, gv) if ctx.needs_input_grad[1] else None
        return gs, gvv, None


class PartitionedSpace:
    """Inner products / scalar products of slab vectors (see _space.py): ``dot`` is closed by an all-reduce and
    both operations are mutually re-entrant autograd Functions, so derivatives of any order are consistent."""

    partitioned = True

    def __init__(self, comm, be):
        self.comm, self.be = comm, be

    def dot(self, a, b):
        return _GlobalDot.apply(a, b, self)

    def scale(self, s, v):
        if not torch.is_tensor(s):
            s = torch.tensor(float(s), dtype=v.dtype, device=v.device)
        return _Scale.apply(s, v, self)


# =========================================================================== operators
class _LanczosState:
    pass


class _NoOwner:
    """stand-in for Workspace.owned_by when the backend has no library workspace (the CPU test double)"""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class PartitionedOperator:
    """Base of the row-partitioned operators: owns the communicator, the slab backend and the two distributed
    loops (Lanczos forward, projected-CG adjoint); subclasses provide the mat-vec with its exchange."""

    partitioned = True
    _native_methods = ("H", "__call__", "Hsparse")

    def __init__(self, n_global, n_local, row_offset, device, comm, be):
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.dim, self.n = int(n_global), int(n_global)
        self.nloc, self.row_offset = int(n_local), int(row_offset)
        self.device = torch.device(device)
        self.be = be
        self.space = PartitionedSpace(comm, be)
        self.use_shadow = True
        # CG: the host looks at the device-side stop flag every ``poll_every`` iterations.  After convergence the rest of
        # a chunk still runs its collectives (a slab exchange cannot be skipped by a device flag), so with large slabs --
        # an iteration of ~2 ms at 2^25 rows against a ~60 us poll -- short chunks waste less than they cost.
        self.poll_every = 2 if (int(n_local) >= (1 << 24) and comm.world > 1) else 8
        # world size 1: the slab is the whole operator, so the in-library single-GPU loops apply; set this to run
        # the distributed driver anyway (tests, host-overhead measurements)
        self.force_driver = False
        self.last_cg_iters = 0
        self.last_cg_resnorm = float("nan")
        # library-side driver (include/dsea.h "row-partitioned solvers"): communicator + partitioned-operator handles;
        # None = the Python driver below (CPU test double, DSEA_DRIVER=python, transports without a library binding)
        self._ncomm = None
        self._pop = None
        self.overlap_fallbacks = 0

    def __del__(self):
        self._destroy_pops()

    def _destroy_pops(self):
        try:
            lib = self.be.lib
            for name in ("_pop", "_pop_dHdg"):
                h = getattr(self, name, None)
                if h:
                    lib.dsea_pop_destroy(h)
                    setattr(self, name, None)
        except Exception:
            pass

    def _make_native_comm(self):
        """the library-side communicator for this operator's collectives, or None"""
        import os
        if not isinstance(self.be, HipBackend) or os.environ.get("DSEA_DRIVER", "") == "python":
            return None
        comm = self.comm
        if getattr(comm, "native_comm", None) is not None:
            # the Python-level communicator carries the library-side one to use (e.g. library-owned RCCL communicators
            # beside a gloo group that only serves the host-side collectives of the autograd wrappers)
            return comm.native_comm
        if isinstance(comm, _SelfComm):
            return NativeComm.from_python(comm, self.device)
        if type(comm) is TorchDistComm:
            if dist.get_backend(comm.group) == "nccl":
                return NativeComm.for_torch_group(comm.group, self.device, getattr(self, "_exchange_group", None))
            return None      # gloo on device tensors is not stream-ordered: Python driver with explicit staging
        if all(hasattr(comm, a) for a in ("allreduce", "all_to_all", "sendrecv")):
            return NativeComm.from_python(comm, self.device)
        return None

    @property
    def driver(self):
        return "library (%s)" % self._ncomm.kind if self._pop else "python"

    # library driver's CG: None = the library's default per operator (TFIM: one all-reduce per iteration, Chronopoulos-Gear
    # recurrences; stencil: the reference's recurrences, two all-reduces), True / False force the reference's recurrences
    # on / off.  DSEA_CG_REFERENCE_RECURRENCES=1 (read when the solve starts) = True everywhere, as for the one-GPU solver.
    cg_reference_recurrences = None

    def _cg_flags(self):
        import os
        from . import _lib
        ref = self.cg_reference_recurrences
        if os.environ.get("DSEA_CG_REFERENCE_RECURRENCES", "") == "1" or _engine_mod.CG_TFIM_REFERENCE_RECURRENCES:
            ref = True
        return 0 if ref is None else (_lib.POP_CG_REFERENCE if ref else _lib.POP_CG_ONE_REDUCTION)

    def _pop_flags(self):
        return self._cg_flags()

    def _lanczos_library(self, k, q0_slab, arena):
        """the whole k-step loop inside libdsea (dsea_pop_lanczos_run): slab kernels and collectives issued back to
        back by the library, no Python and no host synchronisation per step"""
        from . import _lib
        from ctypes import c_int
        be, n, lib = self.be, self.nloc, self.be.lib
        be.reserve(k)
        ws = be.ws
        ldq = (n + 31) // 32 * 32
        Q = be.basis(k, ldq, arena)
        alphas, betas = be.zeros(k), be.zeros(max(k - 1, 1))
        q0 = _vec(q0_slab)
        # partial re-orthogonalisation option (engine.PARTIAL_REORTH, reorth="partial"): fp64 basis, no overlapped exchange
        pro = _engine_mod.partial_reorth()               # (this thread's view of the option)
        partial = pro is not None
        _engine_mod.last_reorth_steps = None
        use_shadow = self.use_shadow and _engine_mod.USE_SHADOW and k > 1 and not partial and \
            _engine_mod.shadow_fits(self.device, k, ldq, n, arena)
        flags = self._pop_flags()
        with ws.owned_by("row-partitioned Lanczos (library driver)"):
            if use_shadow:
                be.set_shadow(k, ldq, arena)
            if getattr(ws, "partial_reorth", None) != pro:
                _lib.check(lib.dsea_ws_set_partial_reorth(ws.handle, 1 if partial else 0, float(pro or 0.0)),
                           "dsea_ws_set_partial_reorth")
                ws.partial_reorth = pro
            try:
                for attempt in (0, 1):
                    _lib.check(lib.dsea_pop_set_flags(self._pop, flags), "dsea_pop_set_flags")
                    _lib.check(lib.dsea_pop_lanczos_run(self._pop, ws.handle, int(k), be._p(q0), be._p(Q), ldq,
                                                        be._p(alphas), be._p(betas), be._st()), "dsea_pop_lanczos_run")
                    step = c_int(0)
                    rc = lib.dsea_pop_lanczos_status(self._pop, ws.handle, byref(step), be._st())
                    if rc == _lib.ERR_PREMISE and attempt == 0:
                        # some step's coefficients were not at rounding level (the same step on every rank: c is
                        # replicated): the overlapped exchange sent an r that differs from the corrected one by more than
                        # the mat-vec's own rounding -- discard the run and repeat it with the exchange after the correction
                        self.overlap_fallbacks += 1
                        flags &= ~_lib.POP_OVERLAP
                        continue
                    _lib.check(rc, "dsea_pop_lanczos_status")
                    break
                if partial:
                    from ctypes import c_double, c_int64
                    cnt, an = c_int64(0), c_double(0.0)
                    _lib.check(lib.dsea_lanczos_reorth_stats(ws.handle, byref(cnt), byref(an), be._st()), "dsea_lanczos_reorth_stats")
                    _engine_mod.last_reorth_steps = int(cnt.value)
            finally:
                lib.dsea_pop_set_flags(self._pop, self._pop_flags())
                if use_shadow:
                    be.clear_shadow()
        return Q, ldq, alphas, betas[:k - 1]

    def _solve_library(self, E0, b, x0, eps, maxiter):
        from . import _lib
        from ctypes import c_double, c_int64
        be, lib = self.be, self.be.lib
        ws = be.ws
        x, b = _vec(x0), _vec(b)
        shift = E0.detach().reshape(-1)[:1].to(F64).contiguous() if E0 is not None else None
        iters, res = c_int64(0), c_double(0.0)
        cap = self.dim if maxiter is None else int(maxiter)
        with ws.owned_by("row-partitioned CG (library driver)"):
            lib.dsea_pop_set_flags(self._pop, self._pop_flags())
            rc = lib.dsea_pop_cg_run(self._pop, ws.handle, be._p(shift), be._p(b), be._p(x), be._p(ws.state), float(eps),
                                     cap, int(self.poll_every), byref(iters), byref(res), be._st())
        _lib.check(rc, "dsea_pop_cg_run", allow=(_lib.ERR_NOT_CONVERGED,))
        self.last_cg_iters, self.last_cg_resnorm = int(iters.value), float(res.value)
        info = _engine_mod.last_cg
        info.iters, info.resnorm, info.converged = self.last_cg_iters, self.last_cg_resnorm, rc == 0
        flags = self._pop_flags()
        one = bool(flags & _lib.POP_CG_ONE_REDUCTION) or (isinstance(self, PartitionedTFIMOperator) and
                                                         not flags & _lib.POP_CG_REFERENCE)
        info.form = "row-partitioned, one all-reduce per iteration" if one else \
            "row-partitioned, reference recurrences (two all-reduces per iteration)"
        if x.data_ptr() != x0.data_ptr():
            x0.copy_(x)
        return x0

    # ---- to be provided by subclasses
    def apply_shift_dot(self, x, y, shift, out, skip):
        """y = (A - shift) x over all ranks ; out = LOCAL x.y (1-element tensor or None)"""
        raise NotImplementedError

    def _local_native(self):
        """the slab operator as a native single-GPU operand (world size 1) or None"""
        return None

    # ---- helpers for users
    def dot(self, a, b):
        """global inner product of two slab vectors (differentiable, replicated result)"""
        return self.space.dot(a, b)

    def slab(self, full):
        """this rank's rows of a replicated full-length vector"""
        return full[self.row_offset:self.row_offset + self.nloc]

    def global_dot_(self, x, y, out):
        self.be.dot(x, y, out)
        self.comm.allreduce(out)

    # ---- forward: Lanczos (reference Lanczos.py:49-77)
    def lanczos_step(self, i, S):
        """[dots] -> all-reduce(c, ||r||^2) -> [correction] -> mat-vec with its exchange (+ local r.Ar) ->
        all-reduce(||r||^2, r.Ar) -> [normalise, store].  The mat-vec acts on the un-normalised r (linearity),
        which is what lets the two scalar reductions of the tail travel together."""
        be = self.be
        if i >= 1:
            be.plz_dots(S.Q, S.ldq, S.n, i, S.u, S.alphas[i - 1:i], S.betas[i - 2:i - 1] if i >= 2 else None, S.r, S.c)
            self.comm.allreduce(S.c[:i + 1])
        be.plz_correct(S.Q, S.ldq, S.n, i, S.c, S.r, S.pair)
        self.apply_shift_dot(S.r, S.y, None, S.pair[1:2], None)
        self.comm.allreduce(S.pair)
        be.plz_finish(S.r, S.y, S.pair, S.Q[i], i, S.u, S.alphas[i:i + 1], S.betas[i - 1:i] if i >= 1 else None)

    def lanczos(self, k, q0_slab, arena=False):
        """k-step Lanczos with full re-orthogonalisation on slabs.  Returns (Q (k, ldq) slab basis, ldq,
        alphas (k,), betas (k-1,)) -- the scalars replicated bit-identically on every rank.  ``arena``: the caller
        does not keep the basis, so it may live in the persistent arena (engine.BasisArena)."""
        native = self._local_native()
        if self.world == 1 and native is not None and not self.force_driver:
            return _engine_mod.lanczos(native, k, self.nloc, self.device, q0_slab, native=native, arena=arena)
        if self._pop:
            return self._lanczos_library(k, q0_slab, arena)
        if _engine_mod.partial_reorth() is not None:
            raise NotImplementedError("reorth='partial' on a row-partitioned operator needs the library driver "
                                      "(dsea_pop_lanczos_run); the Python step driver re-orthogonalises on every step")
        be, n = self.be, self.nloc
        if hasattr(be, "reserve"):
            be.reserve(k)
        S = _LanczosState()
        S.n, S.k = n, k
        S.ldq = (n + 31) // 32 * 32
        S.Q = be.basis(k, S.ldq, arena) if hasattr(be, "basis") else be.empty(k, S.ldq)
        S.alphas, S.betas = be.zeros(k), be.zeros(max(k - 1, 1))
        S.c, S.pair = be.zeros(k + 2), be.zeros(2)
        S.r, S.u, S.y = q0_slab.detach().to(F64).clone(), be.empty(n), be.empty(n)
        use_shadow = self.use_shadow and _engine_mod.USE_SHADOW and k > 1 and hasattr(be, "set_shadow") and \
            _engine_mod.shadow_fits(self.device, k, S.ldq, n, arena)
        ws = getattr(be, "ws", None)
        with (ws.owned_by("row-partitioned Lanczos") if ws is not None else _NoOwner()):
            if use_shadow:
                be.set_shadow(k, S.ldq, arena)
            try:
                for i in range(k):
                    self.lanczos_step(i, S)
            finally:
                if use_shadow:
                    be.clear_shadow()
        return S.Q, S.ldq, S.alphas, S.betas[:k - 1]

    def ritz_vector(self, Q, ldq, k, s_host):
        out = self.be.empty(self.nloc)
        s = torch.from_numpy(np.ascontiguousarray(s_host, dtype=np.float64)).to(self.device)
        self.be.ritz(Q, ldq, self.nloc, int(s.numel()), s, out)
        return out

    # ---- backward: CG on (A - E0) x = b (reference CG.py:24-41 with the closure of :120)
    def solve_shifted(self, E0, b, x0, eps=1e-7, maxiter=None):
        """x0 is overwritten and returned.  The stopping test runs on the device on replicated scalars; the host
        polls the flag every ``poll_every`` iterations."""
        native = self._local_native()
        if self.world == 1 and native is not None and not self.force_driver:
            x = _engine_mod.cg(b, x0, native=native, shift=E0, eps=eps, maxiter=self.dim if maxiter is None else maxiter)
            self.last_cg_iters, self.last_cg_resnorm = _engine_mod.last_cg.iters, _engine_mod.last_cg.resnorm
            return x
        if self._pop:
            return self._solve_library(E0, b, x0, eps, maxiter)
        be, n = self.be, self.nloc
        state = be.zeros(8)
        r, d, Ad = be.empty(n), be.empty(n), be.empty(n)
        x = x0
        shift = E0.detach().reshape(-1)[:1].to(F64).contiguous() if E0 is not None else None
        self.apply_shift_dot(x, Ad, shift, state[CG_DAD:CG_DAD + 1], None)
        be.cg_init(b, Ad, r, d, state)
        self.comm.allreduce(state[CG_RR:CG_RR + 1])
        be.cg_init_check(state, eps)
        done = state[CG_DONE:CG_DONE + 1]
        issued, cap = 0, (self.dim if maxiter is None else int(maxiter))
        host = state.cpu()
        while host[CG_DONE].item() == 0.0 and issued < cap:
            for _ in range(min(self.poll_every, cap - issued)):
                self.apply_shift_dot(d, Ad, shift, state[CG_DAD:CG_DAD + 1], done)
                self.comm.allreduce(state[CG_DAD:CG_DAD + 1])
                be.cg_update(x, r, d, Ad, state)
                self.comm.allreduce(state[CG_RRNEW:CG_RRNEW + 1])
                be.cg_check(state, eps)
                be.cg_direction(r, d, state)
                issued += 1
            host = state.cpu()
        self.last_cg_iters = int(host[CG_ITERS].item())
        self.last_cg_resnorm = float(host[CG_RESNORM].item())
        info = _engine_mod.last_cg
        info.iters, info.resnorm, info.converged = self.last_cg_iters, self.last_cg_resnorm, host[CG_DONE].item() != 0.0
        info.form = "row-partitioned, reference recurrences (two all-reduces per iteration; Python driver)"
        return x

    # ---- plain (non-differentiable) mat-vec on buffers
    def matvec(self, x, y, which="H"):
        if which != "H":
            raise ValueError("this operator has no '%s' map" % which)
        self.apply_shift_dot(x, y, None, None, None)


class _PartApply(torch.autograd.Function):
    """y = M v over all ranks for a symmetric M that does not depend on differentiable parameters
    (``which`` selects the map); dy/dv^T g = M g, re-entrant."""

    @staticmethod
    def forward(ctx, v, op, which):
        ctx.op, ctx.which = op, which
        vv = _as_slab(v, op)
        y = op.be.empty(op.nloc)
        op.matvec(vv, y, which)
        return y

    @staticmethod
    def backward(ctx, gy):
        return _PartApply.apply(gy, ctx.op, ctx.which), None, None


def _as_slab(v, op):
    if v.numel() != op.nloc:
        raise ValueError("expected this rank's slab of %d rows, got %d" % (op.nloc, v.numel()))
    return _vec(v)


# ------------------------------------------------------------------------------------------ TFIM
class _PartTFIMApply(torch.autograd.Function):
    """H(g) v on slabs (reference TFIM.py:91-98), differentiable in v and in the replicated parameter g."""

    @staticmethod
    def forward(ctx, v, g, op):
        ctx.op = op
        ctx.save_for_backward(v, g)
        vv = _as_slab(v, op)
        y = op.be.empty(op.nloc)
        op.matvec(vv, y, "H")
        return y

    @staticmethod
    def backward(ctx, gy):
        v, g = ctx.saved_tensors
        op = ctx.op
        gv = _PartTFIMApply.apply(gy, g, op) if ctx.needs_input_grad[0] else None
        gg = op.space.dot(op.pHpg(v), gy).reshape(g.shape) if ctx.needs_input_grad[1] else None
        return gv, gg, None


class PartitionedTFIMOperator(PartitionedOperator):
    """Transverse-field Ising chain of L sites (reference examples/TFIM/TFIM.py:39-101) with every vector
    row-partitioned over the P = 2^p ranks of ``group``; ``g`` is the (1,) parameter tensor on this rank's
    device (same value on all ranks).  Attribute names follow the reference model class / ``TFIMOperator``."""

    def __init__(self, L, g, device, backend=None, group=None, comm=None, overlap="auto", exchange_group=None):
        """``exchange_group``: a second NCCL process group over the ranks of ``group`` for the slab exchange of the
        library driver (needed only when ``group`` is a sub-group of the world and torch's communicators are to be
        adopted; see ``NativeComm.for_torch_group``)."""
        comm = comm if comm is not None else TorchDistComm(group)
        self._exchange_group = exchange_group
        p = int(round(np.log2(comm.world)))
        if (1 << p) != comm.world:
            raise ValueError("the TFIM row partition needs a power-of-two world size, got %d" % comm.world)
        if p > L:
            raise ValueError("more ranks than rows")
        self.L = self.N = int(L)
        self.p, self.Lloc = p, int(L) - p
        nloc = 1 << self.Lloc
        device = torch.device(device)
        if backend is None:
            backend = HipBackend(nloc, device)
        backend.attach_tfim(self.L, self.Lloc, comm.rank * nloc, g)
        super().__init__(1 << self.L, nloc, comm.rank * nloc, device, comm, backend)
        self._g = g
        # Overlap of the slab exchange with the dots / correction passes (both exchange forms): the remote part
        # of u = A r' is then taken from the UN-corrected r (its exchange starts before the coefficients c are
        # known).  r - r' = Q c lies at the 1e-14 relative level per element while |c_j| <= ~1e-15 ||r||, the
        # level of the mat-vec's own rounding error.  That premise is CHECKED every step (max|c_j| <= tau ||r||,
        # the same bound the bf16-shadow pass uses); a step that violates it redoes the exchange with the
        # corrected r.  The check costs one small D2H copy per step, so "auto" enables the overlap only where a
        # step is long compared with a host round trip (>= 2^22 rows per rank).
        self.overlap = (self.nloc >= (1 << 22)) if overlap == "auto" else bool(overlap)
        self.overlap_fallbacks = 0
        self._pairwise_forced = False
        # top-bit flips: pairwise slab exchange for P = 2; from P = 4 on the transposed form -- all-to-all,
        # local flip sum, all-to-all back -- which puts 1/P of a slab on each of the P-1 links per phase instead
        # of a whole slab on log2(P) links (P = 8: a quarter of the transfer time)
        self.transposed = self.world >= 4 and self.nloc >= self.world and hasattr(self.be, "flipsum")
        if self.transposed:
            self._xT, self._zT, self._z = self.be.empty(nloc), self.be.empty(nloc), self.be.empty(nloc)
            self._recv = []
        else:
            self._recv = [self.be.empty(nloc) for _ in range(self.p)]
        self._pop_dHdg = None
        self._ncomm = self._make_native_comm()
        if self._ncomm is not None:
            self._create_pops()

    # the parameter tensor (reference ``model.g``, TFIM.py; E0.py:95-96).  The kernels read g through its device
    # pointer, so in-place updates are seen at once; REBINDING ``op.g = new_tensor`` (the reference's ``model.g = ...``
    # pattern) rebuilds the slab operator and the library-side partitioned operators around the new tensor.
    @property
    def g(self):
        return self._g

    @g.setter
    def g(self, value):
        if value is self._g:
            return
        if not torch.is_tensor(value) or value.dtype != F64 or value.device != self._g.device:
            raise ValueError("g must be a float64 tensor on %s" % self._g.device)
        self._g = value
        local = getattr(self.be, "op", None)
        if local is not None and hasattr(type(local), "g"):
            local.g = value                                  # operators.TFIMOperator: new handle on the new pointer
        elif hasattr(self.be, "attach_tfim"):
            self.be.attach_tfim(self.L, self.Lloc, self.rank * self.nloc, value)
        if getattr(self, "_replica", None) is not None:
            self._replica = None
        if self._pop:
            self._destroy_pops()
            self._create_pops()

    def _create_pops(self):
        """library-side partitioned operators for H and dH/dg (they share the exchange scratch and the side stream)"""
        from . import _lib
        lib = self.be.lib
        nd = int(lib.dsea_pop_tfim_scratch_doubles(self.L, self.world)) if self.p > 0 else 0
        if getattr(self, "_scratch", None) is None:
            self._scratch = self.be.empty(max(nd, 2))
            self._side_stream = torch.cuda.Stream(device=self.device) if self.p > 0 else None
        side = c_void_p(self._side_stream.cuda_stream) if self._side_stream is not None else c_void_p(None)
        gdev = self.g.detach()
        self._g_keep = gdev
        tau = float(_engine_mod.SHADOW_TAU)
        for name, gptr, diag in (("_pop", c_void_p(gdev.data_ptr()), 1.0), ("_pop_dHdg", c_void_p(None), 0.0)):
            h = c_void_p()
            _lib.check(lib.dsea_pop_create_tfim(self.L, self._ncomm.handle, gptr, 1.0, diag,
                                                c_void_p(self._scratch.data_ptr()) if self.p > 0 else c_void_p(None),
                                                side, self._pop_flags(), tau, byref(h)), "dsea_pop_create_tfim")
            setattr(self, name, h)

    def _pop_flags(self):
        from . import _lib
        return (_lib.POP_OVERLAP if (self.overlap and self.p > 0) else 0) | \
            (_lib.POP_PAIRWISE if self._pairwise_forced else 0) | \
            (_lib.POP_NO_EXCHANGE if getattr(self, "measure_without_exchange", False) else 0) | self._cg_flags()

    # With TWO ranks every mat-vec moves one whole slab over the single xGMI link between the two GPUs (268 MB at 2^25
    # rows: 3.5-5 ms) -- in the Lanczos step that hides behind the dots pass, in a CG iteration it does not: ~0.9 ms of
    # slab-local work against the exchange.  The CG vectors are small next to the basis, so at P = 2 the solve is
    # REPLICATED instead: b and the start vector are gathered once (two all-gathers), both ranks run the same
    # single-device solve on the full vectors -- identical kernels on identical data, hence bit-identical results --
    # and keep their slab.  Twice the local work per iteration (1.8 ms) and no exchange at all.  From four ranks on the
    # transposed exchange is cheaper than replication and the partitioned solve is kept.  "auto" | True | False.
    replicate_cg = "auto"

    def _replicated(self):
        rep = self.replicate_cg
        return (self.world == 2) if rep == "auto" else (bool(rep) and self.world > 1)

    def _replica_operator(self):
        if getattr(self, "_replica", None) is None:
            be = self.be.spawn(1 << self.L)
            self._replica = PartitionedTFIMOperator(self.L, self.g, self.device, backend=be, comm=_SelfComm(), overlap=False)
            self._replica.force_driver = self.force_driver and self._replica._local_native() is None
            self._replica.poll_every = self.poll_every
        return self._replica

    def solve_shifted(self, E0, b, x0, eps=1e-7, maxiter=None):
        if not self._replicated() or not hasattr(self.be, "spawn") or not hasattr(self.comm, "all_gather"):
            return super().solve_shifted(E0, b, x0, eps=eps, maxiter=maxiter)
        n = 1 << self.L
        b_full, x_full = self.be.empty(n), self.be.empty(n)
        self.comm.all_gather(_vec(b), b_full)
        self.comm.all_gather(_vec(x0), x_full)
        rep = self._replica_operator()
        x_full = rep.solve_shifted(E0, b_full, x_full, eps=eps, maxiter=self.dim if maxiter is None else maxiter)
        self.last_cg_iters, self.last_cg_resnorm = rep.last_cg_iters, rep.last_cg_resnorm
        x0.copy_(x_full[self.row_offset:self.row_offset + self.nloc])
        return x0

    def use_pairwise_exchange(self):
        """switch to the pairwise hypercube exchange (one full slab per partner), e.g. if the transposed form is
        unavailable on some stack"""
        self.transposed = False
        self._pairwise_forced = True
        self._recv = [self.be.empty(self.nloc) for _ in range(self.p)]
        if self._pop:
            for h in (self._pop, self._pop_dHdg):
                self.be.lib.dsea_pop_set_flags(h, self._pop_flags())

    def _local_native(self):
        return getattr(self.be, "op", None)

    # ---- exchange
    def _exchange(self, x):
        """receive the slabs of the p hypercube partners (rank ^ (1<<b)); returns the list of buffers"""
        if self.p == 0:
            return []
        if getattr(self, "measure_without_exchange", False):     # bench.py only: the step without its exchange (stale buffers)
            return [self._z] if self.transposed else self._recv
        if self.transposed:   # returns ONE buffer holding the sum over all partner slabs
            self.comm.all_to_all(x, self._xT)
            self.be.flipsum(self._xT, self._zT, self.world)
            self.comm.all_to_all(self._zT, self._z)
            return [self._z]
        self.comm.exchange(x, self._recv, [self.rank ^ (1 << b) for b in range(self.p)])
        return self._recv

    def _start_exchange(self, x):
        """the same exchange started on the communicator's side stream; ``_finish_exchange`` joins it and returns the
        buffers.  x must stay untouched until then."""
        if getattr(self, "measure_without_exchange", False):
            return None
        if self.transposed:
            return self.comm.start_flip_exchange(self.be, x, self._xT, self._zT, self._z)
        return self.comm.start_pair_exchange(x, self._recv, [self.rank ^ (1 << b) for b in range(self.p)])

    def _finish_exchange(self, token):
        self.comm.finish_flip_exchange(token, self.device)
        return [self._z] if self.transposed else self._recv

    # ---- mat-vec on buffers
    def matvec(self, x, y, which="H"):
        """y = H x (or dH/dg x) on this slab: local low-bit part in HIP, top-bit flips from the partners"""
        if self._pop:
            from . import _lib
            be = self.be
            _lib.check(be.lib.dsea_pop_matvec(self._pop if which == "H" else self._pop_dHdg, be.ws.handle, be._p(x),
                                              be._p(y), None, None, None, be._st()), "dsea_pop_matvec")
            return
        self.be.tfim_local(x, y, which)
        for buf in self._exchange(x):
            if which == "H":
                self.be.axpy(-1.0, self.g.detach(), buf, y)     # y -= g * x_partner   (TFIM.py:97)
            else:
                self.be.axpy(-1.0, None, buf, y)                 # dH/dg: y -= x_partner (TFIM.py:64)

    def apply_shift_dot(self, x, y, shift, out, skip):
        """y = (H - shift) x with the remote part, the shift and the local x.y in ONE kernel after the exchange.
        The exchange is started first (side stream) and runs behind the slab-local part of the mat-vec -- x is final
        here, so unlike the Lanczos overlap nothing is approximated."""
        if self.p == 0 and hasattr(self.be, "tfim_local_shift_dot"):
            self.be.tfim_local_shift_dot(x, y, shift, out if out is not None else self.be.zeros(1), skip)
            return
        if self.p > 0 and hasattr(self.comm, "start_pair_exchange"):
            token = self._start_exchange(x)
            self.be.tfim_local(x, y, "H")
            recv = self._finish_exchange(token)
        else:
            self.be.tfim_local(x, y, "H")
            recv = self._exchange(x)
        if out is None:
            out = self.be.zeros(1)
        self.be.axpy_multi_dot(-1.0, self.g.detach(), recv, shift, skip, x, y, out)

    # ---- Lanczos step with the fused correction + local mat-vec and the overlapped exchange
    def lanczos_step(self, i, S):
        be, n = self.be, S.n
        overlap = self.p > 0 and self.overlap and hasattr(be, "form_r") and hasattr(self.comm, "start_pair_exchange")
        prev_a = S.alphas[i - 1:i] if i >= 1 else None
        prev_b = S.betas[i - 2:i - 1] if i >= 2 else None
        if overlap:
            if not hasattr(S, "r_send"):
                S.r_send, S.zero = be.empty(n), be.zeros(1)
            # the exchange of the (un-corrected) r runs behind the dots and correction passes
            if i >= 1:
                be.form_r(S.Q, S.ldq, n, i, S.u, prev_a, prev_b, S.r, S.r_send)
            else:
                S.r_send.copy_(S.r)
            token = self._start_exchange(S.r_send)
            premise_ok = True
            if i >= 1:
                # alpha = 0: r is rewritten with its own values (read from the snapshot copy, so that input and
                # output of the kernel do not alias), c = Q^T r, c[i] = r.r
                be.plz_dots(S.Q, S.ldq, n, i, S.r_send, S.zero, None, S.r, S.c)
                self.comm.allreduce(S.c[:i + 1])
                tau = _engine_mod.SHADOW_TAU
                premise_ok = bool((S.c[:i].abs().max() <= tau * S.c[i].sqrt()).item())
            be.plz_correct_matvec(S.Q, S.ldq, i, S.c, S.r, S.y, S.pair)
            recv = self._finish_exchange(token)
            if not premise_ok:   # identical decision on every rank (c is replicated): redo the exchange with the corrected r
                self.overlap_fallbacks += 1
                recv = self._exchange(S.r)
        else:
            if i >= 1:
                be.plz_dots(S.Q, S.ldq, n, i, S.u, prev_a, prev_b, S.r, S.c)
                self.comm.allreduce(S.c[:i + 1])
            if self.p == 0 and hasattr(be, "tfim_local_shift_dot"):
                be.plz_correct(S.Q, S.ldq, n, i, S.c, S.r, S.pair)
                be.tfim_local_shift_dot(S.r, S.y, None, S.pair[1:2], None)
                recv = None
            else:
                be.plz_correct_matvec(S.Q, S.ldq, i, S.c, S.r, S.y, S.pair)
                recv = self._exchange(S.r)
        if recv is not None:
            be.axpy_multi_dot(-1.0, self.g.detach(), recv, None, None, S.r, S.y, S.pair[1:2])
        self.comm.allreduce(S.pair)
        be.plz_finish(S.r, S.y, S.pair, S.Q[i], i, S.u, S.alphas[i:i + 1], S.betas[i - 1:i] if i >= 1 else None)

    # ---- differentiable user surface (names of the reference model class, TFIM.py:58-65,91-101)
    def H(self, v):
        return _PartTFIMApply.apply(v, self.g, self)

    __call__ = H

    def pHpg(self, v):
        """dH/dg v = -sum_j v[i xor (1<<j)] over all L sites"""
        return _PartApply.apply(v, self, "dHdg")

    def Hadjoint_to_gadjoint(self, v1, v2):
        """adjoint hook of TFIM.py:100-101:  g-bar = v1^T (dH/dg) v2 (global), shape (1,)"""
        return self.space.dot(self.pHpg(v2), v1)[None]


# ------------------------------------------------------------------------------------------ 3-point stencil
class _PartStencilApply(torch.autograd.Function):
    """H(V) v on slabs (reference schrodinger1D.py:18-27); dV = gy o v is slab-local (:29-34)."""

    @staticmethod
    def forward(ctx, v, V, op):
        ctx.op = op
        ctx.save_for_backward(v, V)
        vv = _as_slab(v, op)
        y = op.be.empty(op.nloc)
        op.matvec(vv, y, "H")
        return y

    @staticmethod
    def backward(ctx, gy):
        v, V = ctx.saved_tensors
        gv = _PartStencilApply.apply(gy, V, ctx.op) if ctx.needs_input_grad[0] else None
        gV = gy * v if ctx.needs_input_grad[1] else None
        return gv, gV, None


def stencil_partition(n, world, rank):
    """(rows, first row) of rank's contiguous slab: the first n % world ranks hold one row more"""
    base, extra = divmod(int(n), int(world))
    rows = base + (1 if rank < extra else 0)
    off = rank * base + min(rank, extra)
    return rows, off


class PartitionedStencil3Operator(PartitionedOperator):
    """H v = -0.5/h^2 (-2 v + v_{+1} + v_{-1}) + V o v with Dirichlet ends (reference
    examples/schrodinger1D.py:18-27) on a grid of n points cut into contiguous slabs; ``potential_slab`` is this
    rank's part of the parameter tensor.  One halo element travels to each neighbour per mat-vec."""

    def __init__(self, n, h, potential_slab, device=None, backend=None, group=None, comm=None, exchange_group=None):
        comm = comm if comm is not None else TorchDistComm(group)
        self._exchange_group = exchange_group
        rows, off = stencil_partition(n, comm.world, comm.rank)
        if rows < 1:
            raise ValueError("more ranks than grid points")
        if potential_slab.numel() != rows:
            raise ValueError("rank %d holds rows %d..%d: expected a potential slab of %d elements, got %d"
                             % (comm.rank, off, off + rows, rows, potential_slab.numel()))
        device = torch.device(device) if device is not None else potential_slab.device
        self.h = float(h)
        self.coef = -0.5 / self.h ** 2
        self._V = potential_slab
        self._Vdata = _vec(potential_slab)          # contiguous, 16-byte aligned on the device (pair loads)
        self.has_lo, self.has_hi = comm.rank > 0, comm.rank < comm.world - 1
        if backend is None:
            backend = HipBackend(rows, device)
        self._halo = backend.zeros(2)
        backend.attach_stencil(rows, self.coef, self._Vdata, self._halo, self.has_lo, self.has_hi)
        super().__init__(n, rows, off, device, comm, backend)
        self._ncomm = self._make_native_comm()
        if self._ncomm is not None:
            from . import _lib
            h = c_void_p()
            _lib.check(self.be.lib.dsea_pop_create_stencil3(rows, float(self.coef), c_void_p(self._Vdata.data_ptr()),
                                                            c_void_p(self._halo.data_ptr()), self._ncomm.handle, byref(h)),
                       "dsea_pop_create_stencil3")
            self._pop = h

    @property
    def potential(self):
        return self._V

    def _local_native(self):
        return getattr(self.be, "op", None)

    def _halo_exchange(self, x):
        items = []
        if self.has_lo:
            items.append((x[0:1], self._halo[0:1], self.rank - 1))
        if self.has_hi:
            items.append((x[self.nloc - 1:self.nloc], self._halo[1:2], self.rank + 1))
        if items:
            self.comm.sendrecv(items)

    def apply_shift_dot(self, x, y, shift, out, skip):
        self._halo_exchange(x)
        self.be.stencil_local(x, y, shift, out, skip)

    def matvec(self, x, y, which="H"):
        if which != "H":
            raise ValueError("this operator has no '%s' map" % which)
        if self._pop:
            from . import _lib
            be = self.be
            _lib.check(be.lib.dsea_pop_matvec(self._pop, be.ws.handle, be._p(x), be._p(y), None, None, None, be._st()),
                       "dsea_pop_matvec")
            return
        self.apply_shift_dot(x, y, None, None, None)

    def H(self, v):
        return _PartStencilApply.apply(v, self._V, self)

    __call__ = H
    Hsparse = H

    @staticmethod
    def Hadjoint_to_padjoint(v1, v2):
        """adjoint hook of schrodinger1D.py:29-34: potential-bar = v1 o v2 (this rank's slab)"""
        return v1 * v2


# ------------------------------------------------------------------------------------------ explicit sparse matrix
def csr_partition(n, world, rank):
    """(slab rows incl. padding, first row, real rows) of rank's slab: EQUAL slabs of ceil(n / world) rows -- the last
    one padded with empty rows (include/dsea.h: dsea_pop_create_csr wants the same n_local everywhere)"""
    nloc = -(-int(n) // int(world))
    off = rank * nloc
    return nloc, off, max(0, min(nloc, int(n) - off))


class _PartSampledOuter(torch.autograd.Function):
    """vals-bar of this rank's rows: out[e] = v1[row e] v2[col e] (sym: averaged with v1 <-> v2); the column operand is
    exchanged like the x of a mat-vec.  First order (the hook of reference symeig.py:84 on slabs)."""

    @staticmethod
    def forward(ctx, v1, v2, op, sym):
        out = op._sddmm(_as_slab(v1.detach(), op), _as_slab(v2.detach(), op), sym)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("second order through the non-zeros of a ROW-PARTITIONED matrix is not implemented "
                                  "(one GPU: operators.CSROperator)")


class PartitionedCSROperator(PartitionedOperator):
    """General sparse symmetric matrix in contiguous row slabs: this rank passes ITS rows in CSR with GLOBAL column
    indices (``rowptr`` of its real rows, local element offsets).  Slabs are equal -- ceil(n / world) rows, the last one
    padded with empty rows; vectors are slabs of that length and the padding stays zero (start vectors are masked).

    SpMV locality: if every column a slab touches lies within ``hb <= n_local`` rows of the slab (a banded
    matrix), a mat-vec exchanges hb elements with each neighbour; otherwise it all-gathers x first.  The decision is made
    once here from the global pattern (one all-reduce).  The slab operator is the SELL-64 kernel of operators.CSROperator
    with its gathers redirected to the halo / gathered copy (include/dsea.h: dsea_op_set_slab, dsea_pop_create_csr).

    ``vals`` may be a leaf with requires_grad: ``Aadjoint_to_valsadjoint`` is the hook (this rank's non-zeros)."""

    def __init__(self, rowptr, colidx, vals, n, device=None, backend=None, group=None, comm=None, exchange_group=None,
                 mode="auto"):
        comm = comm if comm is not None else TorchDistComm(group)
        self._exchange_group = exchange_group
        nloc, off, real = csr_partition(n, comm.world, comm.rank)
        if rowptr.numel() != real + 1:
            raise ValueError("rank %d holds rows %d..%d: expected rowptr of %d entries, got %d"
                             % (comm.rank, off, off + real, real + 1, rowptr.numel()))
        device = torch.device(device) if device is not None else vals.device
        rp = rowptr.to(torch.int64)
        rowptr_pad = torch.cat([rp, rp[-1:].expand(nloc - real)]) if real < nloc else rp
        gcols = colidx.to(torch.int64)
        # how far outside the slab do the columns reach?  (global maximum: every rank takes the same decision)
        if gcols.numel():
            reach = max(0, off - int(gcols.min().item()), int(gcols.max().item()) - (off + nloc - 1))
        else:
            reach = 0
        r = torch.tensor([float(reach)], dtype=F64)
        if comm.world > 1:
            rr = r.to(device) if (isinstance(comm, TorchDistComm) and type(comm) is TorchDistComm and
                                  dist.get_backend(comm.group) == "nccl") else r
            rr = rr.clone()
            dist.all_reduce(rr, op=dist.ReduceOp.MAX, group=getattr(comm, "group", None))
            reach = int(rr.item())
        if mode not in ("auto", "halo", "gather"):
            raise ValueError("mode must be 'auto', 'halo' or 'gather'")
        if mode == "halo" and reach > nloc:
            raise ValueError("mode='halo': columns reach %d rows beyond the slab of %d rows" % (reach, nloc))
        self.mode = "halo" if (mode == "halo" or (mode == "auto" and reach <= nloc)) else "gather"
        self.hb = reach if self.mode == "halo" else -1
        self.real_rows = real
        self.vals = vals
        if backend is None:
            backend = HipBackend(nloc, device)
        self._halo = backend.zeros(max(2 * max(self.hb, 0), 2))
        self._xg = backend.zeros(nloc * comm.world) if self.mode == "gather" else None
        cols = (gcols - off) if self.mode == "halo" else gcols
        self._local = backend.attach_csr(rowptr_pad, cols.to(torch.int32), vals, nloc, self.hb, self._halo, self._xg)
        self.has_lo, self.has_hi = comm.rank > 0, comm.rank < comm.world - 1
        super().__init__(nloc * comm.world, nloc, off, device, comm, backend)
        self.n_unpadded = int(n)
        self._mask = None
        if real < nloc:
            self._mask = backend.zeros(nloc)
            self._mask[:real] = 1.0
        self._ncomm = self._make_native_comm()
        if self._ncomm is not None:
            from . import _lib
            h = c_void_p()
            _lib.check(self.be.lib.dsea_pop_create_csr(self._local._H.handle, self._ncomm.handle, byref(h)),
                       "dsea_pop_create_csr")
            self._pop = h

    def _local_native(self):
        return getattr(self.be, "op", None)

    # the padding rows of the last slab are empty: a vector that is zero there stays zero there under A
    def _masked(self, v):
        return v if self._mask is None else v * self._mask

    def _sync_vals(self):
        """an optimiser may have stepped this rank's ``vals`` in place since the last solve: the slab operator's SELL copy
        follows (CSROperator tracks the tensor's version counter; the CPU test double reads the tensor itself)"""
        cur = getattr(self._local, "_current", None)
        if cur is not None:
            cur()

    def lanczos(self, k, q0_slab, arena=False):
        self._sync_vals()
        return super().lanczos(k, self._masked(q0_slab.detach().to(F64)), arena=arena)

    def solve_shifted(self, E0, b, x0, eps=1e-7, maxiter=None):
        self._sync_vals()
        if self._mask is not None:
            x0.mul_(self._mask)
            b = b * self._mask
        return super().solve_shifted(E0, b, x0, eps=eps, maxiter=maxiter)

    def _exchange(self, x):
        if self.mode == "gather":
            self.comm.all_gather(x, self._xg)
            return
        hb = self.hb
        if hb == 0:
            return
        items = []
        if self.has_lo:
            items.append((x[0:hb], self._halo[0:hb], self.rank - 1))
        if self.has_hi:
            items.append((x[self.nloc - hb:self.nloc], self._halo[hb:2 * hb], self.rank + 1))
        if items:
            self.comm.sendrecv(items)

    def apply_shift_dot(self, x, y, shift, out, skip):
        self._exchange(x)
        self.be.csr_local(x, y, shift, out, skip)

    def matvec(self, x, y, which="H"):
        if which != "H":
            raise ValueError("this operator has no '%s' map" % which)
        if self._pop:
            from . import _lib
            be = self.be
            _lib.check(be.lib.dsea_pop_matvec(self._pop, be.ws.handle, be._p(x), be._p(y), None, None, None, be._st()),
                       "dsea_pop_matvec")
            return
        self.apply_shift_dot(x, y, None, None, None)

    def H(self, v):
        self._sync_vals()
        return _PartApply.apply(v, self, "H")

    __call__ = H
    Hsparse = H

    def refresh(self):
        """after an in-place update of this rank's ``vals`` (optimiser step): rewrite the slab operator's SELL copy"""
        if hasattr(self._local, "refresh"):
            self._local.refresh()

    def _sddmm(self, v1, v2, sym):
        be = self.be
        nnz = int(self.vals.numel())
        out = be.zeros(max(nnz, 1))      # (a slab without a stored entry still takes part in the exchanges: one dummy element)
        if self._pop:
            from . import _lib
            _lib.check(be.lib.dsea_pop_sddmm(self._pop, be._p(self._local.rowptr), be._p(v1), be._p(v2), 1.0,
                                             _lib.SDDMM_SYMMETRIC if sym else 0, be._p(out), be._st()), "dsea_pop_sddmm")
            return out[:nnz]
        self._exchange(v2)
        be.csr_sddmm_local(v1, v2, 0.5 if sym else 1.0, False, out)
        if sym:
            self._exchange(v1)
            be.csr_sddmm_local(v2, v1, 0.5, True, out)
        return out[:nnz]

    def Aadjoint_to_valsadjoint(self, v1, v2):
        """vals-bar[e] = v1[row e] v2[col e] for this rank's non-zeros (CSR order)"""
        return _PartSampledOuter.apply(v1, v2, self, False)

    def Aadjoint_to_valsadjoint_symmetric(self, v1, v2):
        return _PartSampledOuter.apply(v1, v2, self, True)


# =========================================================================== convenience driver
class PartitionedTFIM:
    """Hand-written forward + first-order backward on a ``PartitionedTFIMOperator`` (loss = E0 + psi.t) WITHOUT
    autograd: an independent statement of reference symeig.py:77-86 used by the tests to cross-check the autograd
    path, and by the host-overhead tool.  The product surface is the operator + the reference API."""

    def __init__(self, L, g, device, backend=None, group=None, eps=1e-7, poll_every=8, comm=None):
        self.op = PartitionedTFIMOperator(L, g, device, backend=backend, group=group, comm=comm)
        self.op.force_driver = True
        self.op.poll_every = int(poll_every)
        self.eps = float(eps)
        self.comm, self.rank, self.world = self.op.comm, self.op.rank, self.op.world
        self.be, self.nloc, self.n, self.device, self.g = self.op.be, self.op.nloc, self.op.dim, self.op.device, g

    transposed = property(lambda self: self.op.transposed)
    last_cg_iters = property(lambda self: self.op.last_cg_iters)
    last_cg_resnorm = property(lambda self: self.op.last_cg_resnorm)

    @property
    def overlap(self):
        return self.op.overlap

    @overlap.setter
    def overlap(self, v):
        self.op.overlap = bool(v)

    def use_pairwise_exchange(self):
        self.op.use_pairwise_exchange()

    def matvec(self, x, y, which="H"):
        self.op.matvec(x, y, which)

    def forward(self, k, q0_slab):
        Q, ldq, alphas, betas = self.op.lanczos(k, q0_slab)
        (lam, s), = _engine_mod.tridiag_extreme(alphas, betas, "min")
        psi = self.op.ritz_vector(Q, ldq, k, s)
        return torch.tensor(lam, dtype=F64, device=self.device), psi

    def _project_out(self, v, unit, scratch):
        """v <- v - (unit.v) unit  in place (CG.py:122, symeig.py:80)"""
        self.op.global_dot_(unit, v, scratch)
        self.be.axpy(-1.0, scratch, unit, v)

    def backward(self, E0, psi, grad_E0, grad_psi, x0_slab):
        """d(loss)/dg for loss with dloss/dE0 = grad_E0 (float) and dloss/dpsi = grad_psi (slab)."""
        be, n = self.be, self.nloc
        scratch = be.zeros(1)
        b = grad_psi.clone()
        self._project_out(b, psi, scratch)                       # symeig.py:80
        x0 = x0_slab.clone()
        self._project_out(x0, psi, scratch)                      # CG.py:122
        lam0 = self.op.solve_shifted(E0, b, x0, eps=self.eps)    # symeig.py:81
        v1 = psi * float(grad_E0) - lam0                         # symeig.py:82
        w = be.empty(n)
        self.op.matvec(psi, w, which="dHdg")                     # hook: (dH/dg v2).v1  (TFIM.py:100-101)
        out = be.zeros(1)
        self.op.global_dot_(w, v1, out)
        return out

    def forward_backward(self, k, q0_slab, x0_slab, t_slab):
        """loss = E0 + psi.t ; returns (E0, psi slab, dloss/dg (1,))."""
        E0, psi = self.forward(k, q0_slab)
        grad = self.backward(E0, psi, 1.0, t_slab, x0_slab)
        return E0, psi, grad
