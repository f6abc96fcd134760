// dsea_cg_persist_tfim.hip -- conjugate gradients (reference CG.py:24-41 with A' = A - shift, CG.py:120) for the
// matrix-free TFIM operator on README-sized problems (n = 2^L <= 8192: reference examples/TFIM/E0.py, chiF.py at
// N = 10 ... 13) as ONE persistent launch.  The streaming form costs three dependent launches per iteration, ~11 us
// whatever the size below 2^16 rows (tools/cg_small_timing.py); the three adjoint solves of a second-order point
// (E0.py:53-67) then take as long as the Lanczos forward.
//
// Same recipe as k_cg_persist_stencil (dsea_kernels.hip), with the hypercube coupling of the TFIM mat-vec in place of
// the halo: G = n / 128 workgroups own 128 rows each and keep x, r, d of their rows in registers for the whole solve.
// Per iteration two grid-wide exchanges through data-tagged granules (8-byte word = 32 bits of data + 32-bit epoch,
// relaxed agent-scope stores / polls, no fences; state zeroed per launch; spins bounded by a wall-clock timeout):
//   (1) the slab partials of d.A'd  ->  alpha                                                CG.py:31
//   (2) the slab partials of r.r of the UPDATED residual + the rows of r                     CG.py:33-38
// The partner slabs' rows of d (bit flips above the slab: workgroup g ^ (1 << b)) are not exchanged at all: every
// workgroup replays d' = r + beta d for its partners' rows from the exchanged r, as it does for its own (CG.py:39).
// Every workgroup sums the same partials in the same order: identical scalars everywhere, the same exit in every
// workgroup.  Expressions follow the streaming kernels (k_spmv_tfim, k_cg_update_fused, k_cg_direction_fused) term by
// term; partial sums are combined per 128-row slab instead of per mat-vec tile / 512-row tile, so the iterates agree with
// the streaming form to rounding, not bit for bit (tests/test_gpu_persistent.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsea_internal.h"
#include "dsea_device.h"

namespace dsea {

namespace {
typedef gran_u64 cgt_gu64;
#define CGT_TIMEOUT_TICKS DSEA_GRANULE_TIMEOUT_TICKS
#define CGT_ROWS 128
#define CGT_MAX_G 64

__device__ __forceinline__ double cgt_tfim_diag(const TfimParams& p, int64_t i, uint64_t maskL) {
  const uint64_t gi = (uint64_t)(p.row_offset + i);
  const uint64_t rot = ((gi << 1) | (gi >> (p.L - 1))) & maskL;
  const int pop = __popcll(gi ^ rot);
  return p.diag_scale * (double)(-(p.L - 2 * pop));
}
}  // namespace

struct CgtArgs {
  TfimParams tf;
  const double* shift;
  const double* b;
  double* x;        // in: start vector, out: solution
  double* state;    // DSEA_CG_* (written by workgroup 0 at the end)
  double eps;
  long long maxiter;
  unsigned long long* comm;   // granules: [G] d.Ad | [G][1 + 128] r.r + rows of r | [G][128] rows of x0 ; zeroed per launch
  int G;
  int lose_peer;   // test hook: the last workgroup exits at once
};

__global__ __launch_bounds__(256) void k_cg_persist_tfim(CgtArgs a) {
  __shared__ double s_w[CGT_ROWS];            // own rows of the vector the mat-vec is applied to
  __shared__ double s_wnb[6][CGT_ROWS];       // the partner slabs' rows of that vector (x0, then d)
  __shared__ double s_rnb[6][CGT_ROWS];       // the partner slabs' rows of the new residual
  __shared__ double s_part[CGT_MAX_G];
  __shared__ double s_b[2];                   // [0] gathered total  [1] fail

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = blockIdx.x, G = a.G;
  if (a.lose_peer && G > 1 && g == G - 1) return;
  const int L = a.tf.L;
  const int64_t n = (int64_t)1 << L, row = (int64_t)g * CGT_ROWS + 2 * lane;
  const int nlocal = L < 7 ? L : 7, nfar = L - nlocal;
  const uint64_t maskL = (L >= 64) ? ~0ull : ((1ull << L) - 1ull);
  const bool v0 = row < n, v1 = row + 1 < n;
  cgt_gu64* SA = (cgt_gu64*)a.comm;
  cgt_gu64* SB = SA + (int64_t)2 * G;
  cgt_gu64* SX = SB + (int64_t)2 * G * (1 + CGT_ROWS);
  const bool has_shift = a.shift != nullptr;
  const double sh = has_shift ? a.shift[0] : 0.0;
  const double gpar = a.tf.g_dev ? a.tf.g_dev[0] : a.tf.g_const;
  double d0 = 0.0, d1 = 0.0;
  if (wv == 0) {
    d0 = cgt_tfim_diag(a.tf, row, maskL);
    d1 = cgt_tfim_diag(a.tf, row + 1, maskL);
  }
  if (tid == 0) s_b[1] = 0.0;
  __syncthreads();

  // y = A' w for the vector staged in s_w (own rows) / s_wnb (partner rows); wave 0 only
  auto apply = [&](double2 w) -> double2 {
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < nlocal; ++b) {
      s0 += s_w[(2 * lane) ^ (1 << b)];
      s1 += s_w[(2 * lane + 1) ^ (1 << b)];
    }
    for (int b = 0; b < nfar; ++b) {
      s0 += s_wnb[b][2 * lane];
      s1 += s_wnb[b][2 * lane + 1];
    }
    double2 y = make_double2(0.0, 0.0);
    if (v0) {
      y.x = __dsub_rn(__dmul_rn(w.x, d0), __dmul_rn(gpar, s0));
      if (has_shift) y.x = __dsub_rn(y.x, __dmul_rn(sh, w.x));
    }
    if (v1) {
      y.y = __dsub_rn(__dmul_rn(w.y, d1), __dmul_rn(gpar, s1));
      if (has_shift) y.y = __dsub_rn(y.y, __dmul_rn(sh, w.y));
    }
    return y;
  };
  // all threads: gather the G slab partials published in `base` (stride `stride` granules) under `epoch` -> total in
  // every thread; with `rows_from` != null also the partner slabs' rows (granule row_off + r of slab g ^ (1 << b), slabs
  // `row_stride` granules apart) -> dst
  auto gather = [&](cgt_gu64* base, int stride, unsigned epoch, cgt_gu64* rows_from, int row_stride, int row_off,
                    double (*dst)[CGT_ROWS], bool& fail) -> double {
    const long long t0 = wall_clock64();
    if (wv == 1 && lane < G) {
      double v = 0.0;
      while (!granule_try_get(base + (int64_t)lane * stride * 2, epoch, v)) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > CGT_TIMEOUT_TICKS) {
          s_b[1] = 1.0;
          break;
        }
      }
      s_part[lane] = v;
    }
    if (rows_from && tid >= 128 && nfar > 0) {
      const int rr = tid - 128;
      double pv[6];
      bool ok;
      do {
        ok = true;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          pv[b] = 0.0;
          if (b < nfar) ok &= granule_try_get(rows_from + ((int64_t)(g ^ (1 << b)) * row_stride + row_off + rr) * 2, epoch, pv[b]);
        }
        if (!ok) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > CGT_TIMEOUT_TICKS) {
            s_b[1] = 1.0;
            break;
          }
        }
      } while (!ok);
#pragma unroll
      for (int b = 0; b < 6; ++b)
        if (b < nfar) dst[b][rr] = pv[b];
    }
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < G; ++w) tot += s_part[w];      // fixed order: identical in every workgroup
      s_b[0] = tot;
    }
    __syncthreads();
    fail = s_b[1] != 0.0;
    const double tot = s_b[0];
    __syncthreads();   // s_part / s_b[0] may be rewritten by the next gather
    return tot;
  };

  double2 xv = make_double2(0.0, 0.0), rv = xv, dv = xv;
  bool fail = false;
  unsigned epoch = 1;
  // ---- r = b - A' x0 ; d = r ; rr = r.r                                          (CG.py:26-30)
  if (wv == 0) {
    xv = ld2<true>(a.x, row, n);
    granule_put(SX + ((int64_t)g * CGT_ROWS + 2 * lane) * 2, epoch, xv.x);
    granule_put(SX + ((int64_t)g * CGT_ROWS + 2 * lane + 1) * 2, epoch, xv.y);
    s_w[2 * lane] = xv.x;
    s_w[2 * lane + 1] = xv.y;
    if (lane == 0) granule_put(SA + (int64_t)g * 2, epoch, 0.0);    // (the gather below reads G slab granules: publish a dummy)
  }
  (void)gather(SA, 1, epoch, SX, CGT_ROWS, 0, s_wnb, fail);   // the partner slabs' rows of x0
  if (fail) {
    if (g == 0 && tid == 0) a.state[DSEA_CG_DONE] = -1.0;
    return;
  }
  if (wv == 0) {
    const double2 Ax = apply(xv);
    const double2 bv = ld2<true>(a.b, row, n);
    rv.x = __dsub_rn(bv.x, Ax.x);
    rv.y = __dsub_rn(bv.y, Ax.y);
    dv = rv;
  }
  epoch = 2;
  auto publish_r = [&]() {
    if (wv == 0) {
      double acc = 0.0;
      acc = fma(rv.x, rv.x, acc);
      acc = fma(rv.y, rv.y, acc);
      acc = wave_sum(acc);
      cgt_gu64* mine = SB + (int64_t)g * (1 + CGT_ROWS) * 2;
      granule_put(mine + (1 + 2 * lane) * 2, epoch, rv.x);
      granule_put(mine + (2 + 2 * lane) * 2, epoch, rv.y);
      if (lane == 0) granule_put(mine, epoch, acc);
    }
  };
  publish_r();
  double rr = gather(SB, 1 + CGT_ROWS, epoch, SB, 1 + CGT_ROWS, 1, s_rnb, fail);
  // d = r: the partner slabs' rows of d are their rows of r
  for (int idx = tid; idx < nfar * CGT_ROWS; idx += 256) s_wnb[idx >> 7][idx & 127] = s_rnb[idx >> 7][idx & 127];
  double rn = sqrt(rr);
  long long iters = 0;
  bool done = rn < a.eps;
  // ---- iterations                                                                  (CG.py:31-40)
  while (!done && !fail && iters < a.maxiter) {
    __syncthreads();
    if (wv == 0) {
      s_w[2 * lane] = dv.x;
      s_w[2 * lane + 1] = dv.y;
    }
    __syncthreads();
    double2 Ad = make_double2(0.0, 0.0);
    ++epoch;
    if (wv == 0) {
      Ad = apply(dv);
      double acc = 0.0;
      acc = fma(dv.x, Ad.x, acc);
      acc = fma(dv.y, Ad.y, acc);
      acc = wave_sum(acc);
      if (lane == 0) granule_put(SA + (int64_t)g * 2, epoch, acc);
    }
    const double dAd = gather(SA, 1, epoch, nullptr, 0, 0, nullptr, fail);
    if (fail) break;
    const double alpha = rr / dAd;
    if (wv == 0) {
      xv.x = __dadd_rn(xv.x, __dmul_rn(alpha, dv.x));
      xv.y = __dadd_rn(xv.y, __dmul_rn(alpha, dv.y));
      rv.x = __dsub_rn(rv.x, __dmul_rn(alpha, Ad.x));
      rv.y = __dsub_rn(rv.y, __dmul_rn(alpha, Ad.y));
    }
    ++epoch;
    publish_r();
    const double rr_new = gather(SB, 1 + CGT_ROWS, epoch, SB, 1 + CGT_ROWS, 1, s_rnb, fail);
    if (fail) break;
    ++iters;
    rn = sqrt(rr_new);
    if (rn < a.eps) {
      done = true;
      break;
    }
    const double beta = rr_new / rr;
    rr = rr_new;
    if (wv == 0) {
      dv.x = __dadd_rn(rv.x, __dmul_rn(beta, dv.x));
      dv.y = __dadd_rn(rv.y, __dmul_rn(beta, dv.y));
    }
    // the partner slabs' rows of d, updated as their owners update them
    for (int idx = tid; idx < nfar * CGT_ROWS; idx += 256) {
      const int b = idx >> 7, r2 = idx & 127;
      s_wnb[b][r2] = __dadd_rn(s_rnb[b][r2], __dmul_rn(beta, s_wnb[b][r2]));
    }
  }
  if (wv == 0) st2<true>(a.x, row, n, xv);
  if (g == 0 && tid == 0) {
    a.state[DSEA_CG_RR] = rr;
    a.state[DSEA_CG_RESNORM] = rn;
    a.state[DSEA_CG_ITERS] = (double)iters;
    a.state[DSEA_CG_DONE] = fail ? -1.0 : (done ? 1.0 : 0.0);
  }
}

bool cg_persist_tfim_applicable(const OpDesc& op) {
  return op.kind == OP_TFIM && op.tfim.L_local == op.tfim.L && op.tfim.row_offset == 0 && op.tfim.L >= 1 &&
         op.tfim.L <= 13;
}
size_t cg_persist_tfim_comm_bytes(int64_t n) {
  const int64_t G = (n + CGT_ROWS - 1) / CGT_ROWS;
  return (size_t)(2 * G * (1 + (1 + CGT_ROWS) + CGT_ROWS)) * sizeof(unsigned long long);
}
// returns 0 if launched, -1 if not applicable, -2 on a HIP error
int launch_cg_persist_tfim(const OpDesc& op, const double* shift, const double* b, double* x, double* state, double eps,
                           int64_t maxiter, void* comm, hipStream_t st, int lose_peer) {
  if (!cg_persist_tfim_applicable(op)) return -1;
  const int64_t n = op.n;
  const int G = (int)((n + CGT_ROWS - 1) / CGT_ROWS);
  {
    static thread_local int cu_dev = -1, cu_count = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (dev != cu_dev) {
      if (hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -2;
      cu_dev = dev;
    }
    if (G > cu_count) return -1;     // all workgroups must be resident together
  }
  if (hipMemsetAsync(comm, 0, cg_persist_tfim_comm_bytes(n), st) != hipSuccess) return -2;
  CgtArgs a;
  a.tf = op.tfim;
  a.shift = shift;
  a.b = b;
  a.x = x;
  a.state = state;
  a.eps = eps;
  a.maxiter = (long long)maxiter;
  a.comm = static_cast<unsigned long long*>(comm);
  a.G = G;
  a.lose_peer = lose_peer;
  hipLaunchKernelGGL(k_cg_persist_tfim, dim3(G), dim3(256), 0, st, a);
  return 0;
}

}  // namespace dsea
