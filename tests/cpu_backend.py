"""TEST DOUBLE (lives under tests/, never imported by the package): a torch-CPU stand-in for
``partitioned.HipBackend`` with the same phase semantics as include/dsea.h, used to exercise the
partition / exchange / all-reduce logic of ``PartitionedTFIM`` with gloo on machines without GPUs."""
import torch

F64 = torch.float64
RR, DAD, RRNEW, ALPHA, BETA, RESNORM, DONE, ITERS = range(8)


class CpuBackend:
    def __init__(self, n_local):
        self.device = torch.device("cpu")
        self.n = int(n_local)
        self.op = None      # no native single-GPU operand: the distributed driver always runs

    def spawn(self, n_local):
        return CpuBackend(n_local)

    def empty(self, *shape):
        return torch.zeros(*shape, dtype=F64)

    zeros = empty

    # ---- slab-local operators
    def attach_tfim(self, L, L_local, row_offset, g):
        assert (1 << L_local) == self.n
        self.L, self.Lloc, self.off, self.g = L, L_local, row_offset, g
        idx = torch.arange(self.n, dtype=torch.int64)
        gi = idx + row_offset
        rot = ((gi << 1) | (gi >> (L - 1))) & ((1 << L) - 1)
        x = gi ^ rot
        pop = torch.zeros_like(x)
        for b in range(L):
            pop += (x >> b) & 1
        self.diag = (-(L - 2 * pop)).to(F64)
        self.idx = idx

    def tfim_local(self, x, y, which="H"):
        s = torch.zeros(self.n, dtype=F64)
        for j in range(self.Lloc):
            s += x[self.idx ^ (1 << j)]
        if which == "H":
            y.copy_(x * self.diag - self.g.detach() * s)
        else:
            y.copy_(-s)

    def attach_stencil(self, n_local, coef, V, halo, has_lo, has_hi):
        assert n_local == self.n
        self.coef, self.V, self.halo, self.has_lo, self.has_hi = coef, V, halo, has_lo, has_hi

    def stencil_local(self, x, y, shift, out, skip):
        if skip is not None and skip[0] != 0:
            return
        lo = self.halo[0:1] if self.has_lo else torch.zeros(1, dtype=F64)
        hi = self.halo[1:2] if self.has_hi else torch.zeros(1, dtype=F64)
        xp = torch.cat([lo, x, hi])
        res = self.coef * ((-2.0 * x + xp[2:]) + xp[:-2]) + self.V * x
        if shift is not None:
            res = res - shift[0] * x
        y.copy_(res)
        if out is not None:
            out[0] = torch.dot(x, y)

    def attach_csr(self, rowptr, cols, vals, n_local, hb, halo, xg):
        """explicit-matrix slab (same semantics as include/dsea.h dsea_op_set_slab): columns LOCAL in [-hb, n + hb) with the
        neighbour halos in ``halo`` = [from rank-1 | from rank+1], or GLOBAL with the gathered vector ``xg`` (hb = -1)"""
        assert n_local == self.n and rowptr.numel() == n_local + 1
        rp = rowptr.to(torch.int64)
        self.csr_rowptr = rp
        self.csr_rows = torch.repeat_interleave(torch.arange(n_local, dtype=torch.int64), rp[1:] - rp[:-1])
        self.csr_cols, self.csr_vals = cols.to(torch.int64), vals
        self.csr_hb, self.csr_halo, self.csr_xg = int(hb), halo, xg

        class _Local:            # what PartitionedCSROperator keeps as the slab operator
            pass
        loc = _Local()
        loc.rowptr = rp
        return loc

    def _csr_source(self, x):
        """(vector the columns index into, offset to add to the stored column)"""
        hb = self.csr_hb
        if hb < 0:
            return self.csr_xg, 0
        if hb == 0:
            return x, 0
        return torch.cat([self.csr_halo[:hb], x, self.csr_halo[hb:2 * hb]]), hb

    def csr_local(self, x, y, shift, out, skip):
        if skip is not None and skip[0] != 0:
            return
        src, o = self._csr_source(x)
        res = torch.zeros(self.n, dtype=F64).index_add_(0, self.csr_rows, self.csr_vals.detach() * src[self.csr_cols + o])
        if shift is not None:
            res = res - shift[0] * x
        y.copy_(res)
        if out is not None:
            out[0] = torch.dot(x, y)

    def csr_sddmm_local(self, v1, v2, alpha, accumulate, out):
        src, o = self._csr_source(v2)
        g = alpha * (v1[self.csr_rows] * src[self.csr_cols + o])
        o_ = out[:g.numel()]              # (a slab without a stored entry hands over one dummy element)
        if accumulate:
            o_.add_(g)
        else:
            o_.copy_(g)

    def form_r(self, Q, ldq, n, i, u, alpha, beta, r, r_copy):
        r.copy_(u - alpha[0] * Q[i - 1, :n] - (beta[0] * Q[i - 2, :n] if (beta is not None and i >= 2) else 0.0))
        if r_copy is not None:
            r_copy.copy_(r)

    def flipsum(self, xT, zT, P):
        chunk = xT.numel() // P
        X = xT.view(P, chunk)
        Z = torch.zeros_like(X)
        b = 1
        while b < P:
            Z += X[[s ^ b for s in range(P)]]
            b <<= 1
        zT.copy_(Z.reshape(-1))

    def axpy(self, a_host, a_dev, x, y):
        a = a_host * (a_dev.reshape(-1)[0] if a_dev is not None else 1.0)
        y.add_(a * x)

    def dot(self, x, y, out):
        out[0] = torch.dot(x, y)

    def scale_store(self, r, nrm2, q_out, beta_out):
        beta = nrm2[0].sqrt()
        q_out[: r.numel()] = r / beta
        if beta_out is not None:
            beta_out[0] = beta

    def rdots(self, Q, ldq, n, i, u, alpha, beta, r, c):
        r.copy_(u - alpha[0] * Q[i - 1, :n] - (beta[0] * Q[i - 2, :n] if beta is not None else 0.0))
        c[:i] = Q[:i, :n] @ r

    def axpy_norm(self, Q, ldq, n, i, c, r, nrm2):
        r.sub_(Q[:i, :n].T @ c[:i])
        nrm2[0] = torch.dot(r, r)

    def ritz(self, Q, ldq, n, k, s, out):
        out.copy_(Q[:k, :n].T @ s)

    # macro phases (same semantics as include/dsea.h "row-partitioned macro phases"; no shadow here)
    def plz_dots(self, Q, ldq, n, i, u, alpha, beta, r, c):
        self.rdots(Q, ldq, n, i, u, alpha, beta, r, c)
        c[i] = torch.dot(r, r)

    def plz_correct(self, Q, ldq, n, row, c, r, pair):
        if row >= 1:
            r.sub_(Q[:row, :n].T @ c[:row])
        pair[0] = torch.dot(r, r)

    def plz_correct_matvec(self, Q, ldq, row, c, r, y, pair):
        self.plz_correct(Q, ldq, r.numel(), row, c, r, pair)
        self.tfim_local(r, y, "H")

    def axpy_multi_dot(self, a_host, a_dev, xs, shift, skip, x, y, out):
        if skip is not None and skip[0] != 0:
            return
        a = a_host * (a_dev.reshape(-1)[0] if a_dev is not None else 1.0)
        for t in xs:
            y.add_(a * t)
        if shift is not None:
            y.sub_(shift[0] * x)
        out[0] = torch.dot(x, y)

    def plz_finish(self, r, y, pair, q_out, row, u_out, alpha_out, beta_out):
        beta = pair[0].sqrt()
        q_out[: r.numel()] = r / beta
        u_out.copy_(y / beta)
        alpha_out[0] = pair[1] / pair[0]
        if beta_out is not None:
            beta_out[0] = beta

    def shift_dot(self, x, y, shift, out, skip):
        if skip is not None and skip[0] != 0:
            return
        if shift is not None:
            y.sub_(shift[0] * x)
        out[0] = torch.dot(x, y)

    def cg_init(self, b, Ax0, r, d, state):
        state.zero_()
        r.copy_(b - Ax0)
        d.copy_(r)
        state[RR] = torch.dot(r, r)

    def cg_init_check(self, state, eps):
        rn = state[RR].sqrt()
        state[RESNORM] = rn
        state[DONE] = 1.0 if rn < eps else 0.0

    def cg_update(self, x, r, d, Ad, state):
        if state[DONE] != 0:
            return
        a = state[RR] / state[DAD]
        x.add_(a * d)
        r.sub_(a * Ad)
        state[RRNEW] = torch.dot(r, r)

    def cg_check(self, state, eps):
        if state[DONE] != 0:
            return
        rn = state[RRNEW].sqrt()
        state[ITERS] += 1
        state[RESNORM] = rn
        if rn < eps:
            state[DONE] = 1.0
        else:
            state[BETA] = state[RRNEW] / state[RR]
            state[RR] = state[RRNEW].clone()

    def cg_direction(self, r, d, state):
        if state[DONE] != 0:
            return
        d.copy_(r + state[BETA] * d)
