"""Row-partitioned dominant eigenpair + adjoint across P = 2^p GPUs (one process per GPU).

New capability (the reference is single-device).  Every n-vector -- Krylov basis
vectors, CG vectors -- is cut into P contiguous slabs of n/P rows, one per rank:

  * all vector algebra is slab-local and runs in the same HIP phase kernels as the single-GPU generic
    path (include/dsea.h "vector phases"); each phase leaves its LOCAL partial sum in a device scalar;
  * inner products are closed by ``torch.distributed.all_reduce`` on those device scalars (backend
    "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests): per Lanczos step one all-reduce of
    the i re-orthogonalisation coefficients, one of ||r||^2, one of alpha; per CG iteration two scalars;
    all ranks therefore hold bit-identical scalars and take the same branch in the CG stopping test;
  * the TFIM mat-vec flips the low L-p bits inside the slab (HIP kernel) and obtains the top p bits from the
    partner slabs rank ^ (1<<b) by a pairwise exchange (hypercube), then  y -= g * x_partner.

The numerical kernels are reached through a small backend object; the product backend is ``HipBackend``
(libdsea.so, no fallback).  The CPU tests inject a torch-CPU test double to exercise the partition /
exchange / all-reduce logic with gloo on machines without GPUs.
"""
from __future__ import annotations

from ctypes import c_void_p

import numpy as np
import torch
import torch.distributed as dist

from . import engine as _engine_mod

F64 = torch.float64


# =========================================================================== product backend (HIP)
class HipBackend(_engine_mod.Phases):
    """Slab-local numerics through the C ABI (the vector phases come from ``engine.Phases``).
    ``L``/``L_local``/``row_offset`` describe this rank's slab."""

    def __init__(self, L, L_local, row_offset, g, device):
        from . import _lib
        from .operators import TFIMOperator
        super().__init__(1 << L_local, device, kmax=8)
        self._lib_mod = _lib
        self.engine = _engine_mod
        self.op = TFIMOperator(L, self.device, g=g, L_local=L_local, row_offset=row_offset)
        self._shadow = None

    # -- helpers
    @staticmethod
    def _p(t):
        return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)

    def _ck(self, rc, what):
        self._lib_mod.check(rc, what)

    # -- operator (slab-local part)
    def tfim_local(self, x, y, which="H"):
        handle = self.op.handle if which == "H" else self.op._dHdg.handle
        self._ck(self.lib.dsea_spmv(handle, None, self._p(x), self._p(y), None, None, None, self._st()), "dsea_spmv")

    # -- macro phases of the partitioned Lanczos step (include/dsea.h "row-partitioned macro phases")
    def set_shadow(self, k, ldq):
        self._shadow = torch.empty((k, ldq), dtype=torch.bfloat16, device=self.device)
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

    def shift_dot(self, x, y, shift, out, skip):
        self._ck(self.lib.dsea_shift_dot(self.ws.handle, self._p(x), self._p(y), self._p(shift), self._p(out),
                                         self._p(skip), x.numel(), self._st()), "dsea_shift_dot")

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

    def exchange(self, x, recv, peers):
        """send slab x to every peer, receive theirs into recv[b] (pairwise, all at once)"""
        ops = []
        for buf, peer in zip(recv, peers):
            ops.append(dist.P2POp(dist.isend, x, peer, group=self.group))
            ops.append(dist.P2POp(dist.irecv, buf, peer, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

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
        ops = []
        for j in range(P):
            if j != me:
                ops.append(dist.P2POp(dist.isend, src[j * chunk:(j + 1) * chunk], j, group=self.group))
                ops.append(dist.P2POp(dist.irecv, dst[j * chunk:(j + 1) * chunk], j, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


# =========================================================================== the partitioned solver
CG_RR, CG_DAD, CG_RRNEW, CG_ALPHA, CG_BETA, CG_RESNORM, CG_DONE, CG_ITERS = range(8)


class PartitionedTFIM:
    """Ground state of the TFIM chain of L sites and d(loss)/dg with vectors row-partitioned over the
    default process group.  ``g`` is the (1,) parameter tensor on this rank's device (same value on all
    ranks)."""

    def __init__(self, L, g, device, backend=None, group=None, eps=1e-7, poll_every=8, comm=None):
        self.comm = comm if comm is not None else TorchDistComm(group)
        self.rank, self.world = self.comm.rank, self.comm.world
        self.p = int(round(np.log2(self.world)))
        if (1 << self.p) != self.world:
            raise ValueError("the row partition needs a power-of-two world size, got %d" % self.world)
        if self.p > L:
            raise ValueError("more ranks than rows")
        self.L, self.Lloc = int(L), int(L) - self.p
        self.nloc = 1 << self.Lloc
        self.n = 1 << self.L
        self.row_offset = self.rank * self.nloc
        self.g = g
        self.device = torch.device(device)
        self.be = backend if backend is not None else HipBackend(self.L, self.Lloc, self.row_offset, g, self.device)
        self.eps = float(eps)
        self.poll_every = int(poll_every)
        self.use_shadow = True
        # Overlap of the slab exchange with the dots / correction passes (transposed form only): the remote part
        # of u = A r' is then taken from the UN-corrected r (its exchange starts before the coefficients c are
        # known).  r - r' = Q c lies at the 1e-14 relative level per element (|c_j| <= ~1e-15 ||r||), i.e. at the
        # level of the mat-vec's own rounding error; the local part uses the corrected r'.
        self.overlap = True
        self.last_cg_iters = 0
        self.last_cg_resnorm = float("nan")
        # top-bit flips: pairwise slab exchange for P = 2; from P = 4 on the transposed form -- all-to-all,
        # local flip sum, all-to-all back -- which puts 1/P of a slab on each of the P-1 links per phase instead
        # of a whole slab on log2(P) links (P = 8: a quarter of the transfer time)
        self.transposed = self.world >= 4 and self.nloc >= self.world and hasattr(self.be, "flipsum")
        if self.transposed:
            self._xT, self._zT, self._z = self.be.empty(self.nloc), self.be.empty(self.nloc), self.be.empty(self.nloc)
            self._recv = []
        else:
            self._recv = [self.be.empty(self.nloc) for _ in range(self.p)]

    def use_pairwise_exchange(self):
        """switch to the pairwise hypercube exchange (one full slab per partner), e.g. if the transposed form is
        unavailable on some stack"""
        self.transposed = False
        self._recv = [self.be.empty(self.nloc) for _ in range(self.p)]

    # ---------------------------------------------------------------- collectives
    def _allreduce(self, t):
        self.comm.allreduce(t)

    def _exchange(self, x):
        """receive the slabs of the p hypercube partners (rank ^ (1<<b)); returns the list of buffers"""
        if self.p == 0:
            return []
        if self.transposed:   # returns ONE buffer holding the sum over all partner slabs
            self.comm.all_to_all(x, self._xT)
            self.be.flipsum(self._xT, self._zT, self.world)
            self.comm.all_to_all(self._zT, self._z)
            return [self._z]
        self.comm.exchange(x, self._recv, [self.rank ^ (1 << b) for b in range(self.p)])
        return self._recv

    # ---------------------------------------------------------------- operator
    def matvec(self, x, y, which="H"):
        """y = H x (or dH/dg x) on this slab: local low-bit part in HIP, top-bit flips from the partners"""
        self.be.tfim_local(x, y, which)
        for buf in self._exchange(x):
            if which == "H":
                self.be.axpy(-1.0, self.g.detach(), buf, y)     # y -= g * x_partner   (TFIM.py:97)
            else:
                self.be.axpy(-1.0, None, buf, y)                 # dH/dg: y -= x_partner (TFIM.py:64)

    def matvec_shift_dot(self, x, y, shift, out, skip):
        """y = (H - shift) x with the remote part, the shift and the local x.y in ONE kernel after the exchange"""
        self.be.tfim_local(x, y, "H")
        recv = self._exchange(x)
        self.be.axpy_multi_dot(-1.0, self.g.detach(), recv, shift, skip, x, y, out)

    def global_dot(self, x, y, out):
        self.be.dot(x, y, out)
        self._allreduce(out)

    # ---------------------------------------------------------------- forward: Lanczos (Lanczos.py:49-105)
    def forward(self, k, q0_slab):
        """Per step: [dots] -> all-reduce(c, ||r||^2) -> [correction + local mat-vec] -> slab exchange ->
        [remote part + r.Ar] -> all-reduce(||r||^2, r.Ar) -> [normalise, store].  The mat-vec acts on the
        un-normalised r (linearity), which is what lets the two scalar reductions travel together."""
        be, n = self.be, self.nloc
        if hasattr(be, "reserve"):
            be.reserve(k)
        ldq = (n + 31) // 32 * 32
        Q = be.empty(k, ldq)
        alphas, betas = be.zeros(k), be.zeros(max(k - 1, 1))
        c, pair = be.zeros(k + 2), be.zeros(2)
        r, u, y = q0_slab.clone(), be.empty(n), be.empty(n)
        use_shadow = self.use_shadow and k > 1 and hasattr(be, "set_shadow")
        if use_shadow:
            be.set_shadow(k, ldq)
        try:
            overlap = self.transposed and self.overlap and hasattr(be, "form_r")
            zero = be.zeros(1)
            r_send = be.empty(n) if overlap else None
            for i in range(k):
                token = None
                if overlap:
                    # the exchange of the (un-corrected) r runs behind the dots and correction passes
                    if i >= 1:
                        be.form_r(Q, ldq, n, i, u, alphas[i - 1:i], betas[i - 2:i - 1] if i >= 2 else None, r, r_send)
                    else:
                        r_send.copy_(r)
                    token = self.comm.start_flip_exchange(be, r_send, self._xT, self._zT, self._z)
                    if i >= 1:
                        be.plz_dots(Q, ldq, n, i, r, zero, None, r, c)        # alpha = 0: r stays, c = Q^T r, c[i] = r.r
                        self._allreduce(c[:i + 1])
                    be.plz_correct_matvec(Q, ldq, i, c, r, y, pair)
                    self.comm.finish_flip_exchange(token, self.device)
                    recv = [self._z]
                else:
                    if i >= 1:
                        be.plz_dots(Q, ldq, n, i, u, alphas[i - 1:i], betas[i - 2:i - 1] if i >= 2 else None, r, c)
                        self._allreduce(c[:i + 1])
                    be.plz_correct_matvec(Q, ldq, i, c, r, y, pair)
                    recv = self._exchange(r)
                be.axpy_multi_dot(-1.0, self.g.detach(), recv, None, None, r, y, pair[1:2])
                self._allreduce(pair)
                be.plz_finish(r, y, pair, Q[i], i, u, alphas[i:i + 1], betas[i - 1:i] if i >= 1 else None)
        finally:
            if use_shadow:
                be.clear_shadow()
        # Ritz pair: T is replicated (identical scalars on all ranks), solved on the host (Lanczos.py:98)
        from scipy.linalg import eigh_tridiagonal
        d, e = alphas.cpu().numpy(), betas[:k - 1].cpu().numpy()
        if k == 1:
            lam, s = float(d[0]), np.ones(1)
        else:
            w, v = eigh_tridiagonal(d, e, select="i", select_range=(0, 0))
            lam, s = float(w[0]), np.ascontiguousarray(v[:, 0])
        psi = be.empty(n)
        be.ritz(Q, ldq, n, k, torch.from_numpy(s).to(self.device), psi)
        return torch.tensor(lam, dtype=F64, device=self.device), psi

    # ---------------------------------------------------------------- projected CG (CG.py:24-41,119-123)
    def _project_out(self, v, unit, scratch):
        """v <- v - (unit.v) unit  in place (CG.py:122, symeig.py:80)"""
        self.global_dot(unit, v, scratch)
        self.be.axpy(-1.0, scratch, unit, v)

    def solve_shifted(self, E0, b, x0):
        """(H - E0) x = b by CG from x0 (both already orthogonal to psi); returns x (overwrites x0)."""
        be, n = self.be, self.nloc
        state = be.zeros(8)
        r, d, Ad = be.empty(n), be.empty(n), be.empty(n)
        x = x0
        shift = E0.reshape(1)
        self.matvec_shift_dot(x, Ad, shift, state[CG_DAD:CG_DAD + 1], None)
        be.cg_init(b, Ad, r, d, state)
        self._allreduce(state[CG_RR:CG_RR + 1])
        be.cg_init_check(state, self.eps)
        done = state[CG_DONE:CG_DONE + 1]
        issued, cap = 0, self.n
        host = state.cpu()
        while host[CG_DONE].item() == 0.0 and issued < cap:
            for _ in range(min(self.poll_every, cap - issued)):
                self.matvec_shift_dot(d, Ad, shift, state[CG_DAD:CG_DAD + 1], done)
                self._allreduce(state[CG_DAD:CG_DAD + 1])
                be.cg_update(x, r, d, Ad, state)
                self._allreduce(state[CG_RRNEW:CG_RRNEW + 1])
                be.cg_check(state, self.eps)
                be.cg_direction(r, d, state)
                issued += 1
            host = state.cpu()
        self.last_cg_iters = int(host[CG_ITERS].item())
        self.last_cg_resnorm = float(host[CG_RESNORM].item())
        return x

    # ---------------------------------------------------------------- backward (symeig.py:77-86)
    def backward(self, E0, psi, grad_E0, grad_psi, x0_slab):
        """d(loss)/dg for loss with dloss/dE0 = grad_E0 (float) and dloss/dpsi = grad_psi (slab)."""
        be, n = self.be, self.nloc
        scratch = be.zeros(1)
        b = grad_psi.clone()
        self._project_out(b, psi, scratch)                       # symeig.py:80
        x0 = x0_slab.clone()
        self._project_out(x0, psi, scratch)                      # CG.py:122
        lam0 = self.solve_shifted(E0, b, x0)                     # symeig.py:81
        v1 = psi * float(grad_E0) - lam0                         # symeig.py:82
        w = be.empty(n)
        self.matvec(psi, w, which="dHdg")                        # hook: (dH/dg v2).v1  (TFIM.py:100-101)
        out = be.zeros(1)
        self.global_dot(w, v1, out)
        return out

    def forward_backward(self, k, q0_slab, x0_slab, t_slab):
        """loss = E0 + psi.t ; returns (E0, psi slab, dloss/dg (1,))."""
        E0, psi = self.forward(k, q0_slab)
        grad = self.backward(E0, psi, 1.0, t_slab, x0_slab)
        return E0, psi, grad
