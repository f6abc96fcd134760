// dsea_cg_persist_tfim_big.hip -- conjugate gradients (reference CG.py:24-41 with A' = A - shift, CG.py:120) for the
// full-space matrix-free TFIM operator at 2^11 ... 2^20 rows (BASELINE configs[1] is L = 20) as ONE persistent launch.
// Two forms, one kernel template: MERGED = false keeps the reference's recurrences -- iterates BIT-IDENTICAL to the streaming
// form (mat-vec + update + direction launches); MERGED = true -- THE DEFAULT, see below -- makes one grid-wide exchange per
// iteration with the Chronopoulos-Gear recurrences: the same iteration in exact arithmetic, NOT the same rounding sequence.
// dsea_cg_last_form() / engine.last_cg.form say which one ran.
//
// Streaming form at L = 20: 25.6 us per iteration for 92 MB of algorithmic traffic -- three dependent launches, x / r /
// d / A'd written and re-read through HBM between them.  Here every workgroup keeps x, r, d of its rows in registers
// for the whole solve; per iteration only d crosses the chip (8 MB written once, the out-of-tile bit flips read from
// it) plus two small all-to-all reductions:
//
//   workgroup = NVB "virtual blocks" of 256 threads; a virtual block IS one block of k_spmv_tfim<11>: it owns one tile
//   of 2^11 rows, thread t holds the row pairs t, t+256, t+512, t+768 of the tile -- which are, at the same time, thread
//   t's rows of the four 512-row "canonical tiles" of k_cg_update_fused / k_cg_init.  Every partial sum the streaming
//   kernels leave per block is therefore reproduced by the same threads in the same order:
//     d.A'd  per 2^11-row tile   (k_spmv_tfim's epilogue: fma chain over the thread's pairs, block_sum)
//     r.r    per 512-row tile    (k_cg_update_fused / k_cg_init: fma(x,x) then fma(y,y), block_sum)
//   and every virtual block sums ALL published partials in the order of the consumer kernels (sum_partials_block for
//   d.A'd and r'.r', k_finalize1 for the initial r.r).  Elementwise updates use the same rounded operations.
//
//   exchanges per iteration (data-tagged granules, relaxed agent-scope stores / polls, no fences, epoch = iteration):
//     (F)  pairwise: the L - 11 partner tiles' "d is published" flags, then their rows of d -- bulk data, moved with
//          sc1 (device-coherent, write-through) 16-byte buffer stores / loads instead of granules (cdna_hip_programming.md
//          Guideline 16, sc1 variant: stores -> s_waitcnt vmcnt(0) -> barrier -> flag; readers poll the flag relaxed and
//          read with sc1 loads).  d is double-buffered by iteration parity.
//     (S1) all-to-all: tile partials of d.A'd -> alpha ;  (S2) all-to-all: canonical-tile partials of r'.r' -> beta, stop.
//   State zeroed per launch, spins bounded by a 3 s wall-clock timeout (-> DSEA_ERR_TIMEOUT, the host falls back to
//   the streaming form).  G = 2^(L-11) / NVB <= 256 workgroups: one per compute unit.
//
// MERGED = true (round 4, the default from 2^11 rows = one tile: measured faster than the small-problem kernel of
// dsea_cg_persist_tfim.hip from there on -- L = 12: 6.8 -> 4.9 us, L = 13: 9.1 -> 5.0 us per iteration): ONE all-to-all per iteration instead of two.  Each of the two
// exchanges above costs ~6 us at 256 workgroups (gather + arrival skew) -- more than the mat-vec (6.6 us).  The two
// scalars of CG.py:31-40 are dependent (alpha = rr / d.A'd, then r'.r' of the UPDATED residual), so merging them needs the
// Chronopoulos-Gear recurrences: w = A'r is formed once per iteration, gamma = r.r and delta = r.w are reduced TOGETHER,
// and s = A'p is carried by s <- w + beta s:
//     p <- r + beta p ; s <- w + beta s ; x <- x + alpha p ; r <- r - alpha s ; [publish r] w <- A'r ; [gamma', delta]
//     beta = gamma'/gamma ; alpha = gamma' / (delta - beta gamma'/alpha)
// The same iteration in exact arithmetic, not the rounding sequence of CG.py's recurrences: iterates agree with the
// streaming form / the CPU oracle to ~1e-13 relative after 50 iterations, converged runs take the same number of
// iterations (tests/test_gpu_persistent.py).  Partials are per WORKGROUP (256 x 2 doubles gathered instead of 2048 + 512).
// dsea_ws_set_persist(200) selects the bit-identical two-exchange form.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsea_internal.h"
#include "dsea_device.h"

namespace dsea {

namespace {
typedef gran_u64 cgb_gu64;
typedef unsigned int cgb_v4u __attribute__((ext_vector_type(4)));
#define CGB_TIMEOUT_TICKS DSEA_GRANULE_TIMEOUT_TICKS
#define CGB_T 11
#define CGB_TILE 2048
#define CGB_PER 4          /* row pairs per thread */
#define CGB_SC1 16         /* aux bits of the buffer intrinsics: sc1 */

__device__ __forceinline__ double cgb_tfim_diag(const TfimParams& p, int64_t i, uint64_t maskL) {
  const uint64_t gi = (uint64_t)(p.row_offset + i);
  const uint64_t rot = ((gi << 1) | (gi >> (p.L - 1))) & maskL;
  const int pop = __popcll(gi ^ rot);
  return p.diag_scale * (double)(-(p.L - 2 * pop));
}
__device__ __forceinline__ double2 cgb_as_d2(cgb_v4u v) {
  return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
__device__ __forceinline__ cgb_v4u cgb_as_v4(double2 d) {
  cgb_v4u v;
  v.x = (unsigned)__double2loint(d.x);
  v.y = (unsigned)__double2hiint(d.x);
  v.z = (unsigned)__double2loint(d.y);
  v.w = (unsigned)__double2hiint(d.y);
  return v;
}
}  // namespace

struct CgbArgs {
  TfimParams tf;
  const double* shift;
  const double* b;
  double* x;        // in: start vector, out: solution
  double* state;    // DSEA_CG_* (written by workgroup 0 at the end)
  double eps;
  long long maxiter;
  unsigned long long* comm;   // granules: [ntiles] d.Ad | [4 ntiles] r.r | [ntiles] flags ; zeroed per launch
  double* dbuf[2];            // d of even / odd iterations (n doubles each)
  int ntiles;
  int lose_peer;              // test hook: the last workgroup exits at once
};

struct CgbSm {
  double red[2][4];
  double red4[2][CGB_PER][4];
  double red2[8][2];     // merged form: per-wave partials of (r.r, r.w)
  double tot2[2];
  double bcast;
  double fail;
};

template <int NVB, bool MERGED>
__global__ __launch_bounds__(256 * NVB) void k_cg_persist_tfim_big(CgbArgs a) {
  __shared__ double2 tile2[NVB][CGB_TILE / 2];
  __shared__ CgbSm sm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, vb = tid >> 8, t = tid & 255, wv4 = wave & 3;
  if (a.lose_peer && gridDim.x > 1 && blockIdx.x == gridDim.x - 1) return;
  const int tile = blockIdx.x * NVB + vb;
  const int L = a.tf.L, nfar = L - CGB_T;
  const int64_t n = (int64_t)1 << L, base = (int64_t)tile * CGB_TILE;
  const uint64_t maskL = (L >= 64) ? ~0ull : ((1ull << L) - 1ull);
  const int nctiles = a.ntiles * 4;
  cgb_gu64* PA = (cgb_gu64*)a.comm;
  cgb_gu64* PC = PA + (int64_t)2 * a.ntiles;
  cgb_gu64* FL = PC + (int64_t)2 * nctiles;
  const bool has_shift = a.shift != nullptr;
  const double sh = has_shift ? a.shift[0] : 0.0;
  const double gpar = a.tf.g_dev ? a.tf.g_dev[0] : a.tf.g_const;
  if (tid == 0) sm.fail = 0.0;
  __syncthreads();

  // y = A' w for this virtual block's tile: own rows staged in LDS, the out-of-tile flips read from `src` (a full-length
  // vector in global memory) with device-coherent loads.  Also returns the thread's fma chain of w.y (the tile's x.y
  // partial is its block sum).  Same order of operations as k_spmv_tfim<11, false>.
  auto matvec = [&](const double2* w, const double* src, double2* y) -> double {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(n * 8), 0x00020000);
    // (an opaque copy of the thread index: without it the compiler hoists the ~100 loop-invariant addresses and diagonal
    //  values of this routine out of the iteration loop and spills the solver state instead)
    int tt = t;
    asm volatile("" : "+v"(tt));
    double2 far[CGB_PER];
    // NVB == 2: two batches of 18 loads in flight (all 36 at once need more registers than a 512-thread workgroup has);
    // NVB == 1: one batch
    constexpr int NB = NVB == 2 ? 2 : 1, MB = CGB_PER / NB;
#pragma unroll
    for (int half = 0; half < NB; ++half) {
      cgb_v4u fb[MB][9];
#pragma unroll
      for (int mm = 0; mm < MB; ++mm) {
        const int64_t i0 = base + 2 * (int64_t)(tt + 256 * (MB * half + mm));
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          fb[mm][e] = (cgb_v4u){0u, 0u, 0u, 0u};
          if (e < nfar)
            fb[mm][e] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((i0 ^ ((int64_t)1 << (CGB_T + e))) * 8), 0, CGB_SC1);
        }
      }
#pragma unroll
      for (int mm = 0; mm < MB; ++mm) {
        const int m = MB * half + mm;
        far[m] = make_double2(0.0, 0.0);
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          const double2 f = cgb_as_d2(fb[mm][e]);
          far[m].x += f.x;
          far[m].y += f.y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // previous readers of the LDS tile are done
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) tile2[vb][tt + 256 * m] = w[m];
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) {
      const int lp = tt + 256 * m;
      const int64_t i0 = base + 2 * (int64_t)lp;
      const double2 xv = w[m];
      double2 sum = make_double2(xv.y, xv.x);   // bit 0: the other element of the pair
#pragma unroll
      for (int jb = 1; jb < CGB_T; ++jb) {
        const double2 nbv = tile2[vb][lp ^ (1 << (jb - 1))];
        sum.x += nbv.x;
        sum.y += nbv.y;
      }
      sum.x += far[m].x;
      sum.y += far[m].y;
      double2 v;
      v.x = __dsub_rn(__dmul_rn(xv.x, cgb_tfim_diag(a.tf, i0, maskL)), __dmul_rn(gpar, sum.x));
      v.y = __dsub_rn(__dmul_rn(xv.y, cgb_tfim_diag(a.tf, i0 + 1, maskL)), __dmul_rn(gpar, sum.y));
      if (has_shift) {
        v.x = __dsub_rn(v.x, __dmul_rn(sh, xv.x));
        v.y = __dsub_rn(v.y, __dmul_rn(sh, xv.y));
      }
      y[m] = v;
      acc = fma(xv.x, v.x, acc);
      acc = fma(xv.y, v.y, acc);
      __builtin_amdgcn_sched_barrier(0);   // one pair at a time: the unrolled LDS reads of all four pairs do not fit
    }
    return acc;
  };
  // block_sum of the streaming kernels for this virtual block: total in thread t == 0
  auto vb_sum = [&](double v) -> double {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sm.red[vb][wv4] = v;
    __syncthreads();
    return ((sm.red[vb][0] + sm.red[vb][1]) + sm.red[vb][2]) + sm.red[vb][3];
  };
  // every virtual block gathers ALL `count` partials published under `epoch` and sums them in the order of
  // sum_partials_block (two_acc) or k_finalize1 (!two_acc); the total is returned in every thread
  auto gather = [&](cgb_gu64* src, int count, unsigned epoch, bool two_acc, bool& fail) -> double {
    double v = 0.0;
    if (vb == 0) {
      double pv[8];
      const long long t0 = wall_clock64();
      bool ok;
      do {
        ok = true;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          pv[m] = 0.0;
          const int b = t + 256 * m;
          if (b < count) ok &= granule_try_get(src + 2 * (int64_t)b, epoch, pv[m]);
        }
        if (!ok) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > CGB_TIMEOUT_TICKS) {
            sm.fail = 1.0;
            break;
          }
        }
      } while (!ok);
      if (two_acc) {   // sum_partials_block: a0 takes b = t, t + 512, ... ; a1 takes b = t + 256, t + 768, ...
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int m = 0; m < 8; m += 2) {
          const int b = t + 256 * m;
          if (b + 256 < count) {
            a0 += pv[m];
            a1 += pv[m + 1];
          } else if (b < count) {
            a0 += pv[m];
          }
        }
        v = a0 + a1;
      } else {         // k_finalize1: one accumulator, b = t, t + 256, ...
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m)
          if (t + 256 * m < count) acc += pv[m];
        v = acc;
      }
      v = wave_sum(v);
    }
    __syncthreads();
    if (vb == 0 && lane == 0) sm.red[0][wv4] = v;
    __syncthreads();
    const double tot = ((sm.red[0][0] + sm.red[0][1]) + sm.red[0][2]) + sm.red[0][3];
    fail = sm.fail != 0.0;
    __syncthreads();   // sm.red[0] may be rewritten by the next reduction
    return tot;
  };
  // publish this virtual block's rows of d into dbuf[which] (device-coherent stores), then its flag under `epoch`
  auto publish_d = [&](const double2* dv, int which, unsigned epoch) {
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dbuf[which], 0, (int)(n * 8), 0x00020000);
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m)
      __builtin_amdgcn_raw_buffer_store_b128(cgb_as_v4(dv[m]), rd, (int)((base + 2 * (int64_t)(t + 256 * m)) * 8), 0, CGB_SC1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) granule_put(FL + 2 * (int64_t)tile, epoch, 1.0);
  };
  // wait until the partner tiles of this virtual block have published d under `epoch` (or later)
  auto wait_partners = [&](unsigned epoch, bool& fail) {
    if (t < nfar) {
      const long long t0 = wall_clock64();
      cgb_gu64* f = FL + 2 * (int64_t)(tile ^ (1 << t));
      for (;;) {
        if (granule_epoch(f) >= epoch) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > CGB_TIMEOUT_TICKS) {
          sm.fail = 1.0;
          break;
        }
      }
    }
    __syncthreads();
    fail = sm.fail != 0.0;
  };

  #ifdef DSEA_CGB_TIMING
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = wall_clock64();
#define CGB_TICK(k) { const long long tn = wall_clock64(); tacc[k] += tn - tprev; tprev = tn; }
#else
#define CGB_TICK(k)
#endif
  if constexpr (MERGED) {
    // ---- one exchange per iteration (Chronopoulos-Gear recurrences, see the header)
    // workgroup sums of two values, then all workgroups' pairs gathered and summed in a fixed order (thread t of the first
    // virtual block takes workgroup t): the totals are identical in every workgroup
    // The pair granules are DOUBLE-BUFFERED by epoch parity (second buffer: the r.r region, unused in this form).  With
    // ONE buffer a workgroup could overwrite its epoch-e pair with the epoch-(e+1) pair -- it only waits for its hypercube
    // partners in between -- while a stalled non-partner workgroup had not read epoch e yet; that reader would then never
    // see its tag and run into the time-out.  With two, the slot of epoch e is rewritten at epoch e + 2, which a workgroup
    // reaches only after ALL workgroups have published epoch e + 1, i.e. after every one of them has left the gather of e.
    auto exchange2 = [&](double v0, double v1, unsigned epoch, bool& fail, double& out0, double& out1) {
      cgb_gu64* PE = PA + (int64_t)(epoch & 1u) * 4 * (int64_t)gridDim.x;
      v0 = wave_sum(v0);
      v1 = wave_sum(v1);
      __syncthreads();
      if (lane == 0) {
        sm.red2[wave][0] = v0;
        sm.red2[wave][1] = v1;
      }
      __syncthreads();
      if (tid < 2) {
        double tot = sm.red2[0][tid];
#pragma unroll
        for (int w = 1; w < 4 * NVB; ++w) tot += sm.red2[w][tid];
        granule_put(PE + 2 * ((int64_t)blockIdx.x * 2 + tid), epoch, tot);
      }
      double g0 = 0.0, g1 = 0.0;
      if (vb == 0) {
        const int G = (int)gridDim.x;
        if (t < G) {
          const long long t0 = wall_clock64();
          for (;;) {
            const bool ok0 = granule_try_get(PE + 2 * ((int64_t)t * 2), epoch, g0);
            const bool ok1 = granule_try_get(PE + 2 * ((int64_t)t * 2 + 1), epoch, g1);
            if (ok0 && ok1) break;
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > CGB_TIMEOUT_TICKS) {
              sm.fail = 1.0;
              break;
            }
          }
        }
        g0 = wave_sum(g0);
        g1 = wave_sum(g1);
      }
      __syncthreads();
      if (vb == 0 && lane == 0) {
        sm.red2[wv4][0] = g0;
        sm.red2[wv4][1] = g1;
      }
      __syncthreads();
      out0 = ((sm.red2[0][0] + sm.red2[1][0]) + sm.red2[2][0]) + sm.red2[3][0];
      out1 = ((sm.red2[0][1] + sm.red2[1][1]) + sm.red2[2][1]) + sm.red2[3][1];
      fail = sm.fail != 0.0;
    };
    double2 xv[CGB_PER], rv[CGB_PER], pv[CGB_PER], sv[CGB_PER], wv[CGB_PER];
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) xv[m] = *reinterpret_cast<const double2*>(a.x + base + 2 * (int64_t)(t + 256 * m));
    bool fail = false;
    (void)matvec(xv, a.x, wv);                                                          // r = b - A'x0   (CG.py:26)
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) {
      const double2 bv = *reinterpret_cast<const double2*>(a.b + base + 2 * (int64_t)(t + 256 * m));
      rv[m].x = __dsub_rn(bv.x, wv[m].x);
      rv[m].y = __dsub_rn(bv.y, wv[m].y);
      pv[m] = make_double2(0.0, 0.0);
      sv[m] = make_double2(0.0, 0.0);
    }
    unsigned epoch = 1;
    auto rr_chain = [&]() -> double {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < CGB_PER; ++m) {
        acc = fma(rv[m].x, rv[m].x, acc);
        acc = fma(rv[m].y, rv[m].y, acc);
      }
      return acc;
    };
    publish_d(rv, 0, epoch);
    wait_partners(epoch, fail);
    double gam = 0.0, del = 0.0;
    if (!fail) {
      const double accw = matvec(rv, a.dbuf[0], wv);                                    // w = A'r, local r.w
      exchange2(rr_chain(), accw, epoch, fail, gam, del);
    }
    double rn = sqrt(gam);
    long long iters = 0;
    bool done = !fail && rn < a.eps;
    double alpha = gam / del, beta = 0.0;
    while (!done && !fail && iters < a.maxiter) {
#pragma unroll
      for (int m = 0; m < CGB_PER; ++m) {
        pv[m].x = __dadd_rn(rv[m].x, __dmul_rn(beta, pv[m].x));
        pv[m].y = __dadd_rn(rv[m].y, __dmul_rn(beta, pv[m].y));
        sv[m].x = __dadd_rn(wv[m].x, __dmul_rn(beta, sv[m].x));
        sv[m].y = __dadd_rn(wv[m].y, __dmul_rn(beta, sv[m].y));
        xv[m].x = __dadd_rn(xv[m].x, __dmul_rn(alpha, pv[m].x));
        xv[m].y = __dadd_rn(xv[m].y, __dmul_rn(alpha, pv[m].y));
        rv[m].x = __dsub_rn(rv[m].x, __dmul_rn(alpha, sv[m].x));
        rv[m].y = __dsub_rn(rv[m].y, __dmul_rn(alpha, sv[m].y));
      }
      ++epoch;
      const int which = (int)((iters + 1) & 1);
      CGB_TICK(4)
      publish_d(rv, which, epoch);
      CGB_TICK(0)
      wait_partners(epoch, fail);
      if (fail) break;
      CGB_TICK(1)
      const double accw = matvec(rv, a.dbuf[which], wv);
      CGB_TICK(2)
      double gam2 = 0.0, del2 = 0.0;
      exchange2(rr_chain(), accw, epoch, fail, gam2, del2);
      if (fail) break;
      CGB_TICK(3)
      ++iters;
      rn = sqrt(gam2);
      if (rn < a.eps) {
        gam = gam2;
        done = true;
        break;
      }
      beta = gam2 / gam;
      alpha = gam2 / (del2 - beta * gam2 / alpha);
      gam = gam2;
    }
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) *reinterpret_cast<double2*>(a.x + base + 2 * (int64_t)(t + 256 * m)) = xv[m];
#ifdef DSEA_CGB_TIMING
    if (blockIdx.x == 7 && tid == 0)
      for (int q = 0; q < 6; ++q) a.dbuf[0][q] = (double)tacc[q] * 0.01 / (double)(iters > 0 ? iters : 1);   // us per iteration
#endif
    if (blockIdx.x == 0 && tid == 0) {
      a.state[DSEA_CG_RR] = gam;
      a.state[DSEA_CG_RESNORM] = rn;
      a.state[DSEA_CG_ITERS] = (double)iters;
      a.state[DSEA_CG_DONE] = fail ? -1.0 : (done ? 1.0 : 0.0);
    }
    return;
  }
  double2 xv[CGB_PER], rv[CGB_PER], dv[CGB_PER], Ad[CGB_PER];
#pragma unroll
  for (int m = 0; m < CGB_PER; ++m) xv[m] = *reinterpret_cast<const double2*>(a.x + base + 2 * (int64_t)(t + 256 * m));
  bool fail = false;
  // ---- r = b - A' x0 ; d = r ; rr = r.r                                          (CG.py:26-30)
  (void)matvec(xv, a.x, Ad);
#pragma unroll
  for (int m = 0; m < CGB_PER; ++m) {
    const double2 bv = *reinterpret_cast<const double2*>(a.b + base + 2 * (int64_t)(t + 256 * m));
    rv[m].x = __dsub_rn(bv.x, Ad[m].x);
    rv[m].y = __dsub_rn(bv.y, Ad[m].y);
    dv[m] = rv[m];
  }
  unsigned epoch = 1;
  // canonical-tile partials of r.r: pair m of thread t belongs to the 512-row tile 4 * tile + m, thread t
  auto publish_rr = [&](unsigned ep) {
    double ws4[CGB_PER];
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) {
      double acc = 0.0;
      acc = fma(rv[m].x, rv[m].x, acc);
      acc = fma(rv[m].y, rv[m].y, acc);
      ws4[m] = wave_sum(acc);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < CGB_PER; ++m) sm.red4[vb][m][wv4] = ws4[m];
    }
    __syncthreads();
    if (t < CGB_PER) {     // block_sum's order: ((w0 + w1) + w2) + w3
      const double tot = ((sm.red4[vb][t][0] + sm.red4[vb][t][1]) + sm.red4[vb][t][2]) + sm.red4[vb][t][3];
      granule_put(PC + 2 * ((int64_t)tile * 4 + t), ep, tot);
    }
  };
  publish_rr(epoch);
  publish_d(dv, 0, epoch);
  double rr = gather(PC, nctiles, epoch, false, fail);
  double rn = sqrt(rr);
  long long iters = 0;
  bool done = rn < a.eps;
  // ---- iterations                                                                  (CG.py:31-40)
  while (!done && !fail && iters < a.maxiter) {
    const int which = (int)(iters & 1);
    CGB_TICK(5)
    wait_partners(epoch, fail);
    if (fail) break;
    CGB_TICK(0)
    const double acc = matvec(dv, a.dbuf[which], Ad);
    CGB_TICK(1)
    const double tilesum = vb_sum(acc);
    ++epoch;
    if (t == 0) granule_put(PA + 2 * (int64_t)tile, epoch, tilesum);
    const double dAd = gather(PA, a.ntiles, epoch, true, fail);
    if (fail) break;
    CGB_TICK(2)
    const double alpha = rr / dAd;
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) {
      xv[m].x = __dadd_rn(xv[m].x, __dmul_rn(alpha, dv[m].x));
      xv[m].y = __dadd_rn(xv[m].y, __dmul_rn(alpha, dv[m].y));
      rv[m].x = __dsub_rn(rv[m].x, __dmul_rn(alpha, Ad[m].x));
      rv[m].y = __dsub_rn(rv[m].y, __dmul_rn(alpha, Ad[m].y));
    }
    publish_rr(epoch);
    const double rr_new = gather(PC, nctiles, epoch, true, fail);
    if (fail) break;
    CGB_TICK(3)
    ++iters;
    rn = sqrt(rr_new);
    if (rn < a.eps) {
      done = true;
      break;
    }
    const double beta = rr_new / rr;
    rr = rr_new;
#pragma unroll
    for (int m = 0; m < CGB_PER; ++m) {
      dv[m].x = __dadd_rn(rv[m].x, __dmul_rn(beta, dv[m].x));
      dv[m].y = __dadd_rn(rv[m].y, __dmul_rn(beta, dv[m].y));
    }
    publish_d(dv, (int)(iters & 1), epoch);
    CGB_TICK(4)
  }
#ifdef DSEA_CGB_TIMING
  if (blockIdx.x == 7 && tid == 0)
    for (int q = 0; q < 6; ++q) a.dbuf[0][q] = (double)tacc[q] * 0.01 / (double)(iters > 0 ? iters : 1);   // us per iteration
#endif
#pragma unroll
  for (int m = 0; m < CGB_PER; ++m) *reinterpret_cast<double2*>(a.x + base + 2 * (int64_t)(t + 256 * m)) = xv[m];
  if (blockIdx.x == 0 && tid == 0) {
    a.state[DSEA_CG_RR] = rr;
    a.state[DSEA_CG_RESNORM] = rn;
    a.state[DSEA_CG_ITERS] = (double)iters;
    a.state[DSEA_CG_DONE] = fail ? -1.0 : (done ? 1.0 : 0.0);
  }
}

bool cg_persist_tfim_big_applicable(const OpDesc& op) {
  return op.kind == OP_TFIM && op.tfim.L_local == op.tfim.L && op.tfim.row_offset == 0 && op.tfim.L >= 11 &&
         op.tfim.L <= 20 && op.tune_tile_log2 == CGB_T;
}
size_t cg_persist_tfim_big_comm_bytes(int64_t n) {
  const int64_t ntiles = n / CGB_TILE;
  return (size_t)(2 * (ntiles + 4 * ntiles + ntiles)) * sizeof(unsigned long long);
}
// returns 0 if launched, -1 if not applicable, -2 on a HIP error.  dbuf0 / dbuf1: two scratch vectors of n doubles.
int launch_cg_persist_tfim_big(const OpDesc& op, const double* shift, const double* b, double* x, double* state,
                               double eps, int64_t maxiter, void* comm, double* dbuf0, double* dbuf1, hipStream_t st,
                               int lose_peer, bool merged) {
  if (!cg_persist_tfim_big_applicable(op)) return -1;
  const int64_t n = op.n;
  const int ntiles = (int)(n / CGB_TILE);
  const int nvb = ntiles > 256 ? 2 : 1;
  const int G = ntiles / nvb;
  {
    static thread_local int cu_dev = -1, cu_count = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    if (dev != cu_dev) {
      if (hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -2;
      cu_dev = dev;
    }
    if (G > cu_count) return -1;     // all workgroups must be resident together: one per compute unit
  }
  if (hipMemsetAsync(comm, 0, cg_persist_tfim_big_comm_bytes(n), st) != hipSuccess) return -2;
  CgbArgs a;
  a.tf = op.tfim;
  a.shift = shift;
  a.b = b;
  a.x = x;
  a.state = state;
  a.eps = eps;
  a.maxiter = (long long)maxiter;
  a.comm = static_cast<unsigned long long*>(comm);
  a.dbuf[0] = dbuf0;
  a.dbuf[1] = dbuf1;
  a.ntiles = ntiles;
  a.lose_peer = lose_peer;
  if (nvb == 2) {
    if (merged) hipLaunchKernelGGL((k_cg_persist_tfim_big<2, true>), dim3(G), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((k_cg_persist_tfim_big<2, false>), dim3(G), dim3(512), 0, st, a);
  } else {
    if (merged) hipLaunchKernelGGL((k_cg_persist_tfim_big<1, true>), dim3(G), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_cg_persist_tfim_big<1, false>), dim3(G), dim3(256), 0, st, a);
  }
  return 0;
}

}  // namespace dsea
