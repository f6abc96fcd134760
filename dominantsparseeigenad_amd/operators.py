"""Native (HIP) operators: the user mat-vec ``A(v)`` of the reference as a device kernel.

Each operator is
  * a plain callable ``op(v) -> A v`` on torch tensors, so it can be handed to
    ``setDominantSparseSymeig(A, Aadjoint_to_gadjoint)`` exactly like the Python functions of the
    reference examples (reference examples/TFIM/TFIM.py:91-98, examples/schrodinger1D.py:18-27); the call
    is a differentiable ``torch.autograd.Function`` (backward = the same symmetric kernel), so the
    adjoint hooks built from it stay differentiable for second-order AD;
  * a holder of a C-ABI operator handle (``.handle``), which lets ``Lanczos`` / ``CG_torch`` run the
    whole loop inside libdsea.so without returning to Python per iteration.
"""
from __future__ import annotations

from ctypes import byref, c_void_p

import torch

from . import _lib, engine
from ._lib import check

F64 = torch.float64


class _Handle:
    """RAII wrapper of a dsea_op_t."""

    def __init__(self, raw, n, keepalive):
        self.raw, self.n, self.keepalive = raw, int(n), keepalive

    def __del__(self):
        try:
            _lib.load().dsea_op_destroy(self.raw)
        except Exception:
            pass


class _NativeView:
    """minimal (handle, n) pair accepted by engine.* as ``native=``"""

    def __init__(self, h):
        self.handle, self.n = h.raw, h.n
        self._h = h


class _SymmetricApply(torch.autograd.Function):
    """y = M v for a fixed symmetric M given by a handle; dy/dv^T g = M g (re-entrant)."""

    @staticmethod
    def forward(ctx, v, view):
        ctx.view = view
        return engine.spmv(view, v.detach())

    @staticmethod
    def backward(ctx, gy):
        return _SymmetricApply.apply(gy, ctx.view), None


# ------------------------------------------------------------------------------------------ TFIM
class TFIMOperator:
    """H = -sum_i (g sx_i + sz_i sz_{i+1}) on a periodic chain of L sites, dimension 2^L, matrix-free.

    Replaces the index tables of reference examples/TFIM/TFIM.py:39-51 (an (n, L) int64 gather table:
    168 MB and 11.8 s to build at L = 20) by index arithmetic inside the kernel:
        diag_i = -(L - 2 popcount(i xor rotl_L(i,1))),   neighbours  i xor (1 << j).
    Attribute names follow the reference model class: ``N``, ``dim``, ``g``, ``H``, ``pHpg``,
    ``Hadjoint_to_gadjoint``.
    ``L_local`` / ``row_offset`` describe a row slab for the multi-GPU partition (default: all rows).
    """

    _native_methods = ("H", "__call__")

    def __init__(self, N, device=None, g=None, L_local=None, row_offset=0):
        self.N = int(N)
        self.L_local = self.N if L_local is None else int(L_local)
        self.row_offset = int(row_offset)
        self.dim = 1 << self.N
        self.n = 1 << self.L_local
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise ValueError("TFIMOperator is a device operator; use device='cuda'")
        if self.device.index is None:   # "cuda" -> "cuda:<current>", so that tensors on the device compare equal
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._g = None
        self._H = None
        lib = _lib.load()
        raw = c_void_p()
        check(lib.dsea_op_create_tfim(self.N, self.L_local, self.row_offset, None, 1.0, 0.0, byref(raw)),
              "dsea_op_create_tfim")
        self._dHdg = _NativeView(_Handle(raw, self.n, None))
        if g is not None:
            self.g = g

    # -- the parameter tensor (reference: ``model.g``, shape (1,), requires_grad; E0.py:95-96)
    @property
    def g(self):
        return self._g

    @g.setter
    def g(self, value):
        if not torch.is_tensor(value):
            value = torch.tensor([float(value)], dtype=F64)
        if value.device != self.device or value.dtype != F64:
            raise ValueError("g must be a float64 tensor on %s" % self.device)
        self._g = value
        lib = _lib.load()
        raw = c_void_p()
        # the kernel reads g through this device pointer: no host sync, and in-place updates of g are seen
        check(lib.dsea_op_create_tfim(self.N, self.L_local, self.row_offset, c_void_p(value.data_ptr()), 0.0, 1.0,
                                      byref(raw)), "dsea_op_create_tfim")
        self._H = _NativeView(_Handle(raw, self.n, value))

    # C-ABI handle of H (what Lanczos / CG use for the in-library loops)
    @property
    def handle(self):
        if self._H is None:
            raise RuntimeError("set the parameter g before using the operator")
        return self._H.handle

    def to_csr(self, layout="sell", col16="auto", values="auto"):
        """The same operator as an explicit device CSR matrix (L + 1 entries per row, int32 columns): the
        "generic sparse operand" form of BASELINE config 2.  Built on the device with index arithmetic."""
        n, L = self.n, self.L_local
        idx = torch.arange(n, dtype=torch.int64, device=self.device)
        gi = idx + self.row_offset
        rot = ((gi << 1) | (gi >> (self.N - 1))) & (self.dim - 1)
        x = gi ^ rot
        pop = torch.zeros_like(x)
        for b in range(self.N):
            pop += (x >> b) & 1
        diag = (-(self.N - 2 * pop)).to(F64)
        cols = torch.empty((n, L + 1), dtype=torch.int64, device=self.device)
        vals = torch.empty((n, L + 1), dtype=F64, device=self.device)
        cols[:, 0], vals[:, 0] = idx, diag
        for j in range(L):
            cols[:, j + 1] = idx ^ (1 << j)
            vals[:, j + 1] = -self._g.detach()
        order = torch.argsort(cols, dim=1)
        cols = torch.gather(cols, 1, order)
        vals = torch.gather(vals, 1, order)
        rowptr = torch.arange(n + 1, dtype=torch.int64, device=self.device) * (L + 1)
        return CSROperator(rowptr, cols.reshape(-1), vals.reshape(-1), n, layout=layout, col16=col16, values=values)

    def pHpg(self, v):
        """dH/dg v = -sum_j v[i xor (1<<j)]   (TFIM.py:58-65)"""
        return _SymmetricApply.apply(v, self._dHdg)

    def H(self, v):
        """H v, differentiable in v and in g  (TFIM.py:91-98)"""
        return _TFIMApply.apply(v, self._g, self)

    __call__ = H

    def Hadjoint_to_gadjoint(self, v1, v2):
        """adjoint hook of TFIM.py:100-101:  g-bar = v1^T (dH/dg) v2, shape (1,)"""
        return self.pHpg(v2).matmul(v1)[None]


class _TFIMApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g, op):
        ctx.op = op
        ctx.save_for_backward(v, g)
        return engine.spmv(op._H, v.detach())

    @staticmethod
    def backward(ctx, gy):
        v, g = ctx.saved_tensors
        op = ctx.op
        gv = _TFIMApply.apply(gy, g, op) if ctx.needs_input_grad[0] else None
        gg = None
        if ctx.needs_input_grad[1]:
            gg = op.pHpg(v).matmul(gy).reshape(g.shape)
        return gv, gg, None


# ------------------------------------------------------------------------------------------ stencil
class Stencil3Operator:
    """H v = coef (-2 v + v_{+1} + v_{-1}) + V o v, Dirichlet ends (reference examples/schrodinger1D.py:18-27
    with coef = -0.5/h^2).  ``potential`` is the parameter tensor (n,), the hook is v1 o v2 (:29-34)."""

    _native_methods = ("H", "__call__", "Hsparse")

    def __init__(self, n, h, potential):
        self.n = int(n)
        self.h = float(h)
        self.coef = -0.5 / self.h ** 2
        self.device = potential.device
        self._H = None
        self.potential = potential

    @property
    def potential(self):
        return self._V

    @potential.setter
    def potential(self, value):
        if value.device.type != "cuda" or value.dtype != F64 or value.numel() != self.n:
            raise ValueError("potential must be a float64 CUDA tensor of %d elements" % self.n)
        self._V = value
        self._Vdata = engine.as_vector(value.detach(), self.n)     # contiguous, 16-byte aligned (pair loads)
        raw = c_void_p()
        check(_lib.load().dsea_op_create_stencil3(self.n, self.coef, c_void_p(self._Vdata.data_ptr()), None, None,
                                                  byref(raw)), "dsea_op_create_stencil3")
        self._H = _NativeView(_Handle(raw, self.n, self._Vdata))

    @property
    def handle(self):
        return self._H.handle

    def H(self, v):
        return _Stencil3Apply.apply(v, self._V, self)

    __call__ = H
    Hsparse = H

    @staticmethod
    def Hadjoint_to_padjoint(v1, v2):
        return v1 * v2


class _Stencil3Apply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, V, op):
        ctx.op = op
        ctx.save_for_backward(v, V)
        return engine.spmv(op._H, v.detach())

    @staticmethod
    def backward(ctx, gy):
        v, V = ctx.saved_tensors
        gv = _Stencil3Apply.apply(gy, V, ctx.op) if ctx.needs_input_grad[0] else None
        gV = gy * v if ctx.needs_input_grad[1] else None
        return gv, gV, None


# ------------------------------------------------------------------------------------------ CSR
def sell_layout(rowptr, colidx, n, col16="auto", pad_cols=1):
    """SELL-64 structure of a CSR pattern (device tensors, torch index ops -- built once per operator):
    returns (slice_ptr int64 [nslices+1], columns, total) with ``columns`` = (int32 [total],) or, when every
    64-element slice column spans fewer than 65536 columns (``col16`` "auto" / True), (colbase int32 [total/64],
    coldelta uint16-as-int16 [total]) -- include/dsea.h dsea_op_create_sell / dsea_op_create_sell16.  Entry k of
    row i sits at slice_ptr[i // 64] + 64 k + i % 64; padding entries carry the smallest real column of their
    slice column (value 0), so gathers stay in range and the 16-bit deltas are not widened by them."""
    dev = rowptr.device
    nsl = (n + 63) // 64
    lens = rowptr[1:] - rowptr[:-1]
    padded = torch.zeros(nsl * 64, dtype=torch.int64, device=dev)
    padded[:n] = lens
    width = padded.view(nsl, 64).max(dim=1).values
    if pad_cols > 1:      # (the value-coded layout packs four slice columns to a lane: widths in multiples of pad_cols)
        width = (width + (pad_cols - 1)) // pad_cols * pad_cols
    slice_ptr = torch.zeros(nsl + 1, dtype=torch.int64, device=dev)
    slice_ptr[1:] = torch.cumsum(width * 64, 0)
    total = int(slice_ptr[-1].item())
    rows = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=dev), lens)
    k_in_row = torch.arange(rows.numel(), dtype=torch.int64, device=dev) - rowptr[rows]
    dest = slice_ptr[rows // 64] + k_in_row * 64 + (rows % 64)
    big = torch.iinfo(torch.int32).max
    s_cols = torch.full((max(total, 64),), big, dtype=torch.int32, device=dev)[:total]
    s_cols[dest] = colidx.to(torch.int32)
    blocks = s_cols.view(-1, 64)
    cmin = blocks.min(dim=1, keepdim=True).values
    cmin = torch.where(cmin == big, torch.zeros_like(cmin), cmin)     # (a slice column that is ALL padding: column 0)
    blocks.copy_(torch.where(blocks == big, cmin.expand_as(blocks), blocks))
    span_ok = total > 0 and int((blocks.max(dim=1).values - cmin[:, 0]).max().item()) < 65536
    if col16 is True and not span_ok and total > 0:
        raise ValueError("col16=True: a slice column of this pattern spans 65536 columns or more")
    if total == 0:        # no stored entry at all: keep the arrays addressable (the C ABI refuses null pointers, and torch
        return slice_ptr, (torch.zeros(64, dtype=torch.int32, device=dev),), 0      # reports one for every EMPTY tensor)
    if col16 in ("auto", True) and span_ok:
        delta = blocks - cmin                                        # 0 .. 65535
        delta = torch.where(delta >= 32768, delta - 65536, delta).to(torch.int16)   # the same 16 bits, as torch can hold them
        return slice_ptr, (cmin[:, 0].contiguous(), delta.reshape(-1).contiguous()), total
    return slice_ptr, (s_cols.contiguous(),), total


def sell_positions(rowptr, n, slice_ptr, packed=False):
    """position of every CSR element in the SELL arrays of ``sell_layout`` (int64 [nnz]); ``packed``: in the per-element
    arrays of the value-coded layout (four slice columns to a lane: 256 G + 4 l + j, include/dsea.h dsea_op_create_sell16v8)"""
    lens = rowptr[1:] - rowptr[:-1]
    rows = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=rowptr.device), lens)
    k = torch.arange(rows.numel(), dtype=torch.int64, device=rowptr.device) - rowptr[rows]
    if packed:
        return slice_ptr[rows // 64] + (k // 4) * 256 + (rows % 64) * 4 + (k % 4)
    return slice_ptr[rows // 64] + k * 64 + (rows % 64)


def pack2(a):
    """... -> two slice columns to a lane (dsea_op_create_sell16p2; slices hold multiples of 128)"""
    return a.view(-1, 2, 64).permute(0, 2, 1).contiguous().view(-1)


def pack4(a):
    """a per-element SELL array in slice-column-major order -> four slice columns to a lane (slices hold multiples of 256)"""
    return a.view(-1, 4, 64).permute(0, 2, 1).contiguous().view(-1)


def value_codes(data, limit=255):
    """(table float64 [256], codes uint8 [nnz]) with data == table[codes] BIT FOR BIT and table[0] = 0.0 (the padding's
    code), or None when ``data`` takes more than ``limit`` distinct bit patterns.  A strided sample is looked at first, so
    that a matrix of arbitrary values costs one small sort."""
    bits = data.detach().contiguous().view(torch.int64)
    if bits.numel() > (1 << 16):
        step = bits.numel() >> 16
        if torch.unique(bits[::step]).numel() > limit:
            return None
    uniq, inv = torch.unique(bits, return_inverse=True)
    if uniq.numel() > limit:
        return None
    table = torch.zeros(256, dtype=torch.int64, device=data.device)
    table[1:1 + uniq.numel()] = uniq
    return table.view(F64), (inv + 1).to(torch.uint8)


class CSROperator:
    """General sparse symmetric matrix given in CSR (rowptr int64, colidx int32, vals fp64) on the device.

    The mat-vec kernel reads a SELL-64 copy (sliced ELLPACK, slices of one wave = 64 rows, column-major inside a
    slice) built once here on the device with torch index ops: matrix loads become perfectly coalesced and, for
    banded / structured operators, so does the gather of x; where the pattern allows it the columns are stored as
    16-bit deltas (10 instead of 12 bytes per non-zero: the kernel is bound by the bytes it moves).
    ``layout="csr"`` keeps the plain CSR kernel.  ``from_scipy`` / ``from_dense`` build the CSR arrays on the host once.

    THE MATRIX AS A PARAMETER (reference README.md:88-126, symeig.py:56-64,82-84: the adjoint A-bar = v1 v2^T is pushed to
    the parameters that produced A).  ``vals`` may be a leaf tensor with ``requires_grad``:

        op = CSROperator(rowptr, colidx, vals, n)                       # vals: nn.Parameter-like, fp64, on the device
        symeig.setDominantSparseSymeig(op, op.Aadjoint_to_valsadjoint)  # hook: vals-bar[e] = v1[row e] v2[col e]
        E0, psi = symeig.DominantSparseSymeig.apply(op.vals, k, n)
        loss.backward(); optimiser.step()                               # in-place update of vals ...
        E0, psi = symeig.DominantSparseSymeig.apply(op.vals, k, n)      # ... is picked up: the SELL copy is refreshed in
                                                                        #     place (dsea_op_update_vals), no rebuild

    The hooks are differentiable (second order: examples/TFIM/E0.py:63-64 pattern) -- their backward is a mat-vec with
    the incoming gradient as the values of the same pattern.  ``Aadjoint_to_valsadjoint_symmetric`` is the adjoint for a
    matrix whose entry pairs (i,j), (j,i) are tied, i.e. of  vals -> eigh((M + M^T)/2)."""

    _native_methods = ("__call__",)

    def __init__(self, rowptr, colidx, vals, n, layout="sell", col16="auto", values="auto"):
        self.n = int(n)
        self.rowptr = rowptr.to(torch.int64).contiguous()
        self.colidx = colidx.to(torch.int32).contiguous()
        data = vals.detach()
        if data.dtype != F64 or not data.is_contiguous():
            data = data.to(F64).contiguous()
            self._vals = data         # a converted copy: not the caller's storage (no parameter semantics)
        else:
            self._vals = vals         # the caller's tensor: its in-place updates are seen by refresh()
        self._vals_data = data
        self._seen_version = None
        self.nnz = int(data.numel())
        self.device = data.device
        self.layout = layout
        if self.device.type != "cuda":
            raise ValueError("CSROperator is a device operator")
        self._T = None
        raw = c_void_p()
        lib = _lib.load()
        if layout == "sell":
            nsl = (self.n + 63) // 64
            import os as _os
            if values == "auto":
                values = _os.environ.get("DSEA_SELL_VALUES", "auto")      # A/B switch: "plain" | "auto" | "coded"
            if values not in ("auto", "plain", "coded"):
                raise ValueError("values must be 'auto', 'plain' or 'coded'")
            # VALUE CODES (include/dsea.h dsea_op_create_sell16v8): few distinct stored values -> a uint8 per element into a
            # table of 256 doubles, 3 instead of 10 bytes per non-zero, bit-identical products.  Not for a parameter (its
            # values spread with the first optimiser step) unless asked for.
            coded = None
            if values != "plain" and col16 is not False and self.nnz > 0 and (values == "coded" or not vals.requires_grad):
                coded = value_codes(data)
            if coded is not None:
                slice_ptr, cols, total = sell_layout(self.rowptr, self.colidx, self.n, col16, pad_cols=4)
                if len(cols) != 2:
                    coded = None          # (a slice column spans 65536 columns or more: 32-bit columns, fp64 values)
            if values == "coded" and coded is None:
                raise ValueError("values='coded' needs 16-bit columns and at most 255 distinct stored values")
            self._coded = coded is not None
            if self._coded:
                self._vtab = coded[0]
                self._codes = torch.zeros(total, dtype=torch.uint8, device=self.device)
                self._codes[sell_positions(self.rowptr, self.n, slice_ptr, packed=True)] = coded[1]
                cols = (cols[0], pack4(cols[1]))
                check(lib.dsea_op_create_sell16v8(self.n, nsl, c_void_p(slice_ptr.data_ptr()), c_void_p(cols[0].data_ptr()),
                                                  c_void_p(cols[1].data_ptr()), c_void_p(self._codes.data_ptr()),
                                                  c_void_p(self._vtab.data_ptr()), byref(raw)), "dsea_op_create_sell16v8")
                self._sell = (slice_ptr,) + tuple(cols) + (None,)
                keep = self._sell + (self._codes, self._vtab)
            else:
                slice_ptr, cols, total = self._plain_layout(col16)
                s_vals = torch.zeros(max(total, 1), dtype=F64, device=self.device)
                raw = self._create_plain(slice_ptr, cols, s_vals)
                self._sell = (slice_ptr,) + tuple(cols) + (s_vals,)
                keep = self._sell
            self._sell_total = total
            self.col16 = len(cols) == 2
        elif layout == "csr":
            # (a matrix without a stored entry -- e.g. a slab that is all padding: one addressable dummy element per array)
            c_arr = self.colidx if self.nnz else torch.zeros(1, dtype=torch.int32, device=self.device)
            v_arr = data if self.nnz else torch.zeros(1, dtype=F64, device=self.device)
            check(lib.dsea_op_create_csr(self.n, self.nnz, c_void_p(self.rowptr.data_ptr()),
                                         c_void_p(c_arr.data_ptr()), c_void_p(v_arr.data_ptr()),
                                         byref(raw)), "dsea_op_create_csr")
            self.col16 = False
            keep = (self.rowptr, c_arr, v_arr)
        else:
            raise ValueError("layout must be 'sell' or 'csr'")
        self._H = _NativeView(_Handle(raw, self.n, keep))
        import os as _os
        if layout == "sell":
            self._hint_width(raw)
        if layout == "sell" and _os.environ.get("DSEA_SELL_NT", "") == "1" and (getattr(self, "_pack2", False) or self._coded):
            raise ValueError("DSEA_SELL_NT=1 is an A/B switch of the unpacked layout: set DSEA_SELL_PACK2=0 and DSEA_SELL_VALUES=plain")
        if layout == "sell" and _os.environ.get("DSEA_SELL_NT", "") in ("0", "1"):     # A/B switch (tools/gpu_evidence.sh abenv)
            check(lib.dsea_op_set_tuning(raw, _lib.TUNE_SELL_NT, int(_os.environ["DSEA_SELL_NT"])), "dsea_op_set_tuning")
        self._seen_version = None
        if getattr(self, "_coded", False):
            self._seen_version = self.vals._version        # (the codes were just taken from these values)
        else:
            self.refresh()

    def _hint_width(self, raw):
        """tell the library the widest slice (LDS reservation of the parameter kernels: dsea_op_set_tuning DSEA_TUNE_SELL_MAX_WIDTH)"""
        sp = self._sell[0]
        width = int(((sp[1:] - sp[:-1]).max().item()) // 64) if sp.numel() > 1 else 0
        check(_lib.load().dsea_op_set_tuning(raw, _lib.TUNE_SELL_MAX_WIDTH, width), "dsea_op_set_tuning")

    def _plain_layout(self, col16):
        """(slice_ptr, columns, total) of the fp64-value SELL layout: with 16-bit deltas the per-element arrays are packed two
        slice columns to a lane (include/dsea.h dsea_op_create_sell16p2; DSEA_SELL_PACK2=0: the unpacked layout, for A/B)"""
        import os as _os
        self._pack2 = False
        if col16 is not False and _os.environ.get("DSEA_SELL_PACK2", "1") != "0":
            slice_ptr, cols, total = sell_layout(self.rowptr, self.colidx, self.n, col16, pad_cols=2)
            if len(cols) == 2:
                self._pack2 = True
                return slice_ptr, (cols[0], pack2(cols[1])), total
        return sell_layout(self.rowptr, self.colidx, self.n, col16)

    def _create_plain(self, slice_ptr, cols, s_vals):
        """SELL operand with fp64 values on this pattern (16- or 32-bit columns as ``cols`` has them)"""
        raw, lib, nsl = c_void_p(), _lib.load(), (self.n + 63) // 64
        if len(cols) == 2:
            create = lib.dsea_op_create_sell16p2 if getattr(self, "_pack2", False) else lib.dsea_op_create_sell16
            check(create(self.n, nsl, c_void_p(slice_ptr.data_ptr()), c_void_p(cols[0].data_ptr()),
                         c_void_p(cols[1].data_ptr()), c_void_p(s_vals.data_ptr()), byref(raw)),
                  "dsea_op_create_sell16(p2)")
        else:
            check(lib.dsea_op_create_sell(self.n, nsl, c_void_p(slice_ptr.data_ptr()), c_void_p(cols[0].data_ptr()),
                                          c_void_p(s_vals.data_ptr()), byref(raw)), "dsea_op_create_sell")
        return raw

    # -- the parameter tensor (reference: ``model.g`` / ``model.potential``): in-place updates are followed through its version
    #    counter; binding ANOTHER tensor of the same shape (``op.vals = new``) marks the device copy stale as well
    @property
    def vals(self):
        return self._vals

    @vals.setter
    def vals(self, value):
        if value.numel() != self.nnz or value.device != self.device or value.dtype != F64 or not value.is_contiguous():
            raise ValueError("vals must be a contiguous float64 tensor of %d elements on %s" % (self.nnz, self.device))
        self._vals = value
        self._seen_version = None

    # -- structure shared, values replaced (the mat-vec M(G) x of the hooks' backward)
    def with_vals(self, vals):
        """an operator on the SAME pattern with other values (shares the SELL structure; one value array is allocated)"""
        if getattr(self, "_coded", False):      # (the value-coded layout pads and packs differently: a fresh fp64-value operator)
            return CSROperator(self.rowptr, self.colidx, vals.detach().to(F64).contiguous(), self.n, layout=self.layout, values="plain")
        twin = object.__new__(CSROperator)
        twin.__dict__.update({k: v for k, v in self.__dict__.items()
                              if k not in ("_H", "_sell", "_vals", "_vals_data", "_T", "_codes", "_vtab", "_coded")})
        data = vals.detach().to(F64).contiguous()
        twin._vals, twin._vals_data, twin._T = data, data, None
        raw = c_void_p()
        lib = _lib.load()
        if self.layout == "sell":
            s_vals = torch.zeros(max(self._sell_total, 1), dtype=F64, device=self.device)
            st = self._sell
            raw = self._create_plain(st[0], st[1:-1], s_vals)
            self._hint_width(raw)
            twin._sell = st[:-1] + (s_vals,)
            twin._coded = False
            keep = twin._sell
        else:
            check(lib.dsea_op_create_csr(self.n, self.nnz, c_void_p(self.rowptr.data_ptr()), c_void_p(self.colidx.data_ptr()),
                                         c_void_p(data.data_ptr()), byref(raw)), "dsea_op_create_csr")
            keep = (self.rowptr, self.colidx, data)
        twin._H = _NativeView(_Handle(raw, self.n, keep))
        twin._seen_version = None
        twin.refresh()
        return twin

    def refresh(self):
        """make the device operator's values equal ``vals`` as it is now (dsea_op_update_vals: the SELL copy is rewritten
        in place through rowptr, the layout is not rebuilt).  Called automatically when ``vals`` has been modified in
        place since the last look (optimiser step); call it yourself after writing through a view torch cannot see."""
        data = self.vals.detach()
        if self.nnz == 0:                   # nothing stored: nothing to rewrite
            self._seen_version = self.vals._version
            return
        if data.data_ptr() != self._vals_data.data_ptr():
            self._vals_data = data          # (vals re-bound to another tensor of the same shape)
        if getattr(self, "_coded", False):
            self._recode()
            if self._coded:
                self._seen_version = self.vals._version
                return
        check(_lib.load().dsea_op_update_vals(self._H.handle, c_void_p(self.rowptr.data_ptr()),
                                              c_void_p(self._vals_data.data_ptr()), engine._stream(self.device)),
              "dsea_op_update_vals")
        self._seen_version = self.vals._version

    def _recode(self):
        """value-coded operand whose values have changed: new table and codes IN PLACE (the handle stays), or -- once they no
        longer fit 255 codes -- the fp64 layout on the same pattern, built once (the handle changes: ``handle`` is read
        per call; a row-partitioned slab is never value-coded)"""
        coded = value_codes(self._vals_data)
        if coded is not None:
            self._vtab.copy_(coded[0])
            self._codes[sell_positions(self.rowptr, self.n, self._sell[0], packed=True)] = coded[1]
            return
        slice_ptr, cols, total = self._plain_layout(True)
        s_vals = torch.zeros(max(total, 1), dtype=F64, device=self.device)
        raw = self._create_plain(slice_ptr, cols, s_vals)
        self._sell = (slice_ptr,) + tuple(cols) + (s_vals,)
        self._hint_width(raw)
        self._sell_total = total
        self._coded = False
        del self._codes, self._vtab
        self._H = _NativeView(_Handle(raw, self.n, self._sell))

    def _current(self):
        # (plain CSR layout: the kernel reads the registered tensor itself, so only a re-bound ``vals`` needs the copy)
        if self._seen_version is None or (self.layout == "sell" and self._vals._version != self._seen_version):
            self.refresh()
        return self._H

    @classmethod
    def from_scipy(cls, M, device="cuda", layout="sell", col16="auto", values="auto"):
        M = M.tocsr()
        M.sort_indices()
        return cls(torch.from_numpy(M.indptr.astype("int64")).to(device),
                   torch.from_numpy(M.indices.astype("int32")).to(device),
                   torch.from_numpy(M.data.astype("float64")).to(device), M.shape[0], layout=layout, col16=col16, values=values)

    @classmethod
    def from_npz(cls, path, device="cuda", layout="sell"):
        """on-disk sparse symmetric operand: a file written by ``scipy.sparse.save_npz`` (any scipy sparse format)"""
        import scipy.sparse as sp
        return cls.from_scipy(sp.load_npz(path), device, layout=layout)

    @classmethod
    def from_dense(cls, A, device="cuda", layout="sell"):
        import scipy.sparse as sp
        return cls.from_scipy(sp.csr_matrix(A.detach().cpu().numpy()), device, layout=layout)

    @property
    def handle(self):
        return self._current().handle

    def __call__(self, v):
        return _SymmetricApply.apply(v, self._current())

    # -- the adjoint hooks (reference symeig.py:84 calls Aadjoint_to_padjoint(v1, v2) with A-bar = v1 v2^T)
    def Aadjoint_to_valsadjoint(self, v1, v2):
        """vals-bar[e] = v1[row(e)] * v2[col(e)], in the caller's CSR order (dsea_op_sddmm)"""
        return _SampledOuter.apply(v1, v2, self, False)

    def Aadjoint_to_valsadjoint_symmetric(self, v1, v2):
        """vals-bar[e] = (v1[row] v2[col] + v1[col] v2[row]) / 2"""
        return _SampledOuter.apply(v1, v2, self, True)

    def apply_with_vals(self, vals, v, transpose=False):
        """M(vals) v (or M(vals)^T v) on this operator's pattern, differentiable in ``vals`` and ``v``"""
        return _ValsApply.apply(vals, v, self, bool(transpose))

    def sddmm(self, v1, v2, symmetric=False, out=None, alpha=1.0, accumulate=False):
        """raw kernel call (no autograd): out[e] (+)= alpha * v1[row e] * v2[col e]"""
        v1, v2 = engine.as_vector(v1.detach(), self.n), engine.as_vector(v2.detach(), self.n)
        if out is None:
            out = torch.empty(self.nnz, dtype=F64, device=self.device)
        if self.nnz == 0:
            return out
        flags = (_lib.SDDMM_ACCUMULATE if accumulate else 0) | (_lib.SDDMM_SYMMETRIC if symmetric else 0)
        check(_lib.load().dsea_op_sddmm(self._H.handle, c_void_p(self.rowptr.data_ptr()), c_void_p(v1.data_ptr()),
                                        c_void_p(v2.data_ptr()), float(alpha), flags, c_void_p(out.data_ptr()),
                                        engine._stream(self.device)), "dsea_op_sddmm")
        return out

    def transposed(self):
        """(operator of the transposed PATTERN with this operator's values, perm) with vals_T = vals[perm]; built lazily
        (torch sort, once) -- the backward of the hooks needs M(G)^T x for non-symmetric G"""
        if self._T is None:
            n = self.n
            lens = self.rowptr[1:] - self.rowptr[:-1]
            rows = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=self.device), lens)
            cols = self.colidx.to(torch.int64)
            perm = torch.argsort(cols * n + rows)
            rowptr_t = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
            rowptr_t[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0)
            opT = CSROperator(rowptr_t, rows[perm].to(torch.int32), self._vals_data[perm], n, layout=self.layout)
            self._T = (opT, perm)
        return self._T


class _SampledOuter(torch.autograd.Function):
    """out[e] = v1[row e] v2[col e]  (sym: the average with v1 <-> v2) on a CSROperator's pattern.
    backward (G = gradient w.r.t. out, itself on the pattern):  v1-bar = M(G) v2,  v2-bar = M(G)^T v1  (sym: both averaged
    with their transposes) -- mat-vecs of the same SELL kernel, differentiable again through _ValsApply."""

    @staticmethod
    def forward(ctx, v1, v2, op, sym):
        ctx.op, ctx.sym = op, sym
        ctx.save_for_backward(v1, v2)
        return op.sddmm(v1, v2, symmetric=sym)

    @staticmethod
    def backward(ctx, G):
        v1, v2 = ctx.saved_tensors
        op, sym = ctx.op, ctx.sym
        g1 = g2 = None
        if ctx.needs_input_grad[0]:
            g1 = _ValsApply.apply(G, v2, op, False)
            if sym:
                g1 = 0.5 * (g1 + _ValsApply.apply(G, v2, op, True))
        if ctx.needs_input_grad[1]:
            g2 = _ValsApply.apply(G, v1, op, True)
            if sym:
                g2 = 0.5 * (g2 + _ValsApply.apply(G, v1, op, False))
        return g1, g2, None, None


class _ValsApply(torch.autograd.Function):
    """y = M(vals) x  /  M(vals)^T x  on a CSROperator's pattern, differentiable in vals and x."""

    @staticmethod
    def forward(ctx, vals, x, op, transpose):
        ctx.op, ctx.transpose = op, transpose
        ctx.save_for_backward(vals, x)
        if transpose:
            opT, perm = op.transposed()
            twin = opT.with_vals(vals.detach()[perm])
        else:
            twin = op.with_vals(vals)
        return engine.spmv(twin._H, engine.as_vector(x.detach(), op.n))

    @staticmethod
    def backward(ctx, gy):
        vals, x = ctx.saved_tensors
        op, tr = ctx.op, ctx.transpose
        gvals = gx = None
        if ctx.needs_input_grad[0]:       # y_r = sum_e vals_e x[col e]  ->  vals-bar_e = gy[row e] x[col e]   (transpose: swapped)
            gvals = _SampledOuter.apply(x, gy, op, False) if tr else _SampledOuter.apply(gy, x, op, False)
        if ctx.needs_input_grad[1]:
            gx = _ValsApply.apply(vals, gy, op, not tr)
        return gvals, gx, None, None


# ------------------------------------------------------------------------------------------ GEMM-shaped operands
class DenseOperator:
    """A dense real matrix (row-major device tensor) as a library operand of the NON-symmetric primitives
    (reference eig.py:28-30 hands the matrix to ARPACK); ``transpose=True`` applies A^T.  The mat-vec is libdsea's
    hand-written row-major GEMV (include/dsea.h: dsea_op_create_dense; HBM-bound, deterministic); for ``transpose=True`` the
    transposed matrix is materialised once so that both orientations stream rows."""

    _native_methods = ("__call__", "matvec")

    def __init__(self, A, transpose=False):
        if A.device.type != "cuda" or A.dim() != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("DenseOperator takes a square CUDA matrix")
        self.transpose = bool(transpose)
        A = A.detach().to(F64)
        self.A = (A.T if self.transpose else A).contiguous()       # the matrix whose ROWS the kernel streams
        self.n = int(A.shape[0])
        self.shape = (self.n, self.n)
        self.device = self.A.device
        raw = c_void_p()
        check(_lib.load().dsea_op_create_dense(self.n, c_void_p(self.A.data_ptr()), self.n, 0, byref(raw)),
              "dsea_op_create_dense")
        self._H = _NativeView(_Handle(raw, self.n, self.A))

    @property
    def handle(self):
        return self._H.handle

    def matvec(self, v):
        return engine.spmv(self._H, v)

    __call__ = matvec


class SymmetricDenseOperator:
    """A dense real SYMMETRIC matrix as a native operand of the symmetric primitives (reference symeig.py:15-31,
    CG.py:43-71 apply ``torch.matmul(A, v)``): hand-written HIP mat-vec that reads only the UPPER triangle, each
    64 x 64 tile once for both its row and its column block -- half the bytes of a GEMV -- and lets the dense
    primitive run its Lanczos / CG loops inside libdsea like the sparse ones (no Python per iteration).

    Handed to ``setDominantSparseSymeig(op, hook)`` it is also the way to get the adjoint of a LARGE dense matrix in
    its lazy rank-1 form: the hook receives (v1, v2) with A-bar = v1 v2^T (reference symeig.py:56-64) and contracts
    them with whatever produced A -- the n x n gradient of ``DominantSymeig`` (symeig.py:29) is never formed."""

    _native_methods = ("__call__", "matvec")

    def __init__(self, A):
        if A.device.type != "cuda" or A.dim() != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("SymmetricDenseOperator takes a square CUDA matrix")
        self.n = int(A.shape[0])
        A = A.detach()
        if A.dtype not in (F64, torch.float32):
            A = A.to(F64)
        if self.n % 2:   # rows are read as element pairs: pad the leading dimension to an even number
            Ap = torch.zeros((self.n, self.n + 1), dtype=A.dtype, device=A.device)
            Ap[:, : self.n] = A
            self.A, lda = Ap, self.n + 1
        else:
            self.A, lda = A.contiguous(), self.n
        if self.A.data_ptr() % 16:
            self.A = self.A.clone()
        self.shape = (self.n, self.n)
        self.device = self.A.device
        lib = _lib.load()
        self._work = torch.empty(lib.dsea_op_symdense_work_bytes(self.n) // 8, dtype=F64, device=self.device)
        raw = c_void_p()
        check(lib.dsea_op_create_symdense(self.n, c_void_p(self.A.data_ptr()), self.A.element_size(), lda,
                                          c_void_p(self._work.data_ptr()), byref(raw)), "dsea_op_create_symdense")
        self._H = _NativeView(_Handle(raw, self.n, (self.A, self._work)))

    @property
    def handle(self):
        return self._H.handle

    def matvec(self, v):
        """A v, differentiable in v (dy/dv^T g = A g: the same symmetric kernel, re-entrant)"""
        return _SymmetricApply.apply(v, self._H)

    __call__ = matvec


def dense_symmetric_operand(A):
    """native operand for a dense symmetric tensor: the upper-triangle kernel where it wins (measured on MI355X:
    n = 6144: 50 vs 74 us, n = 16384: 317 vs 439 us for the rocBLAS GEMV), the library's rocBLAS GEMV operand below
    (n = 4096: 29 vs 21 us -- two launches do not pay off there).  Either way the Lanczos / CG
    loops run inside libdsea."""
    if A.shape[0] >= 6144 or (A.dtype == torch.float32 and A.shape[0] >= 2048):
        return SymmetricDenseOperator(A)      # fp32 matrices are read as fp32: no promoted copy (Lanczos.py:47)
    try:
        return DenseOperator(A)
    except _lib.DseaError as exc:
        # no rocBLAS in this process (dsea_op_create_dense -> DSEA_ERR_UNSUPPORTED on another ROCm stack): the
        # hand-written upper-triangle mat-vec has no such dependency -- still inside the library, no torch arithmetic
        if "unsupported" not in str(exc):
            raise
        return SymmetricDenseOperator(A)


class TransferOperator:
    """MPS transfer matrix of a real rank-3 tensor A (d, D, D) acting on D x D matrices stored as D^2-vectors
    (reference examples/TFIM_vumps/general.py:59-66):

        transpose=False:  r -> sum_s A_s r A_s^T      ("Gong",  the reference's ``fr``)
        transpose=True :  l -> sum_s A_s^T l A_s      ("GongT", the reference's ``fl``)

    applied by libdsea as one strided-batched fp64 GEMM + one GEMM of depth d*D (rocBLAS) -- the explicit D^2 x D^2
    matrix of ``matrix_forward`` (general.py:47-49) would be 550 GB at D = 512."""

    _native_methods = ("__call__", "matvec")

    def __init__(self, A, transpose=False):
        if A.device.type != "cuda" or A.dim() != 3 or A.shape[1] != A.shape[2]:
            raise ValueError("TransferOperator takes a CUDA tensor of shape (d, D, D)")
        # a private fp64 COPY: the operator is a snapshot of A at construction, whatever path applies it (the hand-written
        # kernels read a fragment-packed copy made by dsea_op_create_transfer, the rocBLAS path reads this tensor) -- an
        # in-place update of the caller's tensor is never seen; build a new operator for a new A (general.py:59 does)
        self.A = A.detach().to(F64).clone(memory_format=torch.contiguous_format)
        self.d, self.D = int(A.shape[0]), int(A.shape[1])
        self.n = self.D * self.D
        self.shape = (self.n, self.n)
        self.device = self.A.device
        self.transpose = bool(transpose)
        lib = _lib.load()
        nbytes = lib.dsea_op_transfer_work_bytes(self.D, self.d)
        self._work = torch.empty(nbytes // 8, dtype=F64, device=self.device)
        raw = c_void_p()
        check(lib.dsea_op_create_transfer(self.D, self.d, c_void_p(self.A.data_ptr()), int(self.transpose),
                                          c_void_p(self._work.data_ptr()), engine._stream(self.device), byref(raw)),
              "dsea_op_create_transfer")
        self._H = _NativeView(_Handle(raw, self.n, (self.A, self._work)))

    @property
    def handle(self):
        return self._H.handle

    def matvec(self, v):
        return engine.spmv(self._H, v)

    __call__ = matvec
